import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qaray_amd import hip
from qaray_amd.host import load_scene_blob
mode = sys.argv[1]
c = hip.Context(0)
if mode == "two":
    d = hip.Context(0)
if mode == "maps":
    c.upload_scene(load_scene_blob("trc_mtl_glass.xml", size=(64, 48)))
    c.build_photon_maps((3000, 20, 1.5), (400, 20, 2.5))
import torch
try:
    print(mode, "torch cuda ok:", torch.zeros(1, device="cuda").item() == 0)
except Exception as e:
    print(mode, "torch cuda FAILED:", e)
