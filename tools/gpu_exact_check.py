"""The exact walks of the cooperative kernel alone (option "cs_force_exact") against the normal frame and the counting kernel.
   python tools/gpu_exact_check.py [scene w h spp]"""
import os, subprocess, sys, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
subprocess.run([sys.executable, os.path.join(ROOT, "scenes", "gen_assets.py")], check=True, stdout=subprocess.DEVNULL)
from qaray_amd.host import load_scene_blob
from qaray_amd import hip
scene = sys.argv[1] if len(sys.argv) > 1 else "trc_scene_tower.xml"
w, h, spp = (int(x) for x in sys.argv[2:5]) if len(sys.argv) > 4 else (480, 270, 4)
blob = load_scene_blob(scene, size=(w, h))
ctx = hip.Context(0)
ctx.upload_scene(blob)
ref = ctx.render_region((0, 0, w, h), spp, stats=True)
for force in (0, 1, 2, 3):
    ctx.set_option("cs_force_exact", force)
    out = ctx.render_region((0, 0, w, h), spp)
    same = [bool(np.array_equal(a.view(np.uint32), b.view(np.uint32))) for a, b in zip(out, ref)]
    nbad = int((out[0].view(np.uint32) != ref[0].view(np.uint32)).any(axis=2).sum())
    print(f"{scene} {w}x{h}@{spp} cs_force_exact={force} [{ctx.kernel_name()}]: rgb/depth/ns equal the counting kernel's: {same}, pixels that differ: {nbad}", flush=True)
ctx.close()
