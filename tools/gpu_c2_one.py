"""One Cornell-box frame (BASELINE C2 frame size) for counter passes: python tools/gpu_c2_one.py [spp]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from qaray_amd.host import load_scene_blob
from qaray_amd import hip
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 128
ctx = hip.Context(0)
ctx.upload_scene(load_scene_blob("example_project12_box.xml", size=(1920, 1080)))
ctx.reset_kernel_time(); ctx.reset_counters()
ctx.render_region((0, 0, 1920, 1080), spp)
ms, _ = ctx.kernel_time(); c = ctx.counters()
print(f"C2 1920x1080 @ {spp}: {ms:.1f} ms, {c['samples'] / ms * 1e-3:.1f} Msamples/s [{ctx.kernel_name()}]", flush=True)
