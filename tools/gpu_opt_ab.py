"""One library, one option (qa_set_option) off / on, on the BASELINE scenes: bitwise comparison of the frames and frame times.
   python tools/gpu_opt_ab.py OPTION [spp] [cases]      e.g.  python tools/gpu_opt_ab.py xcd_tiles 16 c3,c5"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
subprocess.run([sys.executable, os.path.join(ROOT, "scenes", "gen_assets.py")], check=True, stdout=subprocess.DEVNULL)
from qaray_amd.host import load_scene_blob
from qaray_amd import hip
opt = sys.argv[1]
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 16
cases = sys.argv[3].split(",") if len(sys.argv) > 3 else ["c3", "c4", "c5"]
CASES = {"c2": ("example_project12_box.xml", (1920, 1080)), "c3": ("example_project7_object.xml", (1920, 1080)),
         "c4": ("example_project12_caustics_glossy.xml", (3840, 2160)), "c5": ("trc_scene_tower.xml", (3840, 2160))}
ctx = hip.Context(0)
for tag in cases:
    scene, size = CASES[tag]
    blob = load_scene_blob(scene, size=size)
    out = {}
    for rep in range(2):
        for v in (0, 1):
            ctx.set_option(opt, v)
            ctx.upload_scene(blob)
            ctx.render_region((0, 0, 64, 64), 1)
            ctx.reset_kernel_time(); ctx.reset_counters()
            out[v] = ctx.render_region((0, 0) + size, spp)
            ms, _ = ctx.kernel_time(); c = ctx.counters()
            print(f"{tag} {opt}={v}: {ms:8.1f} ms  {c['samples'] / ms * 1e-3:8.1f} Msamples/s  [{ctx.kernel_name()}]", flush=True)
    same = all(np.array_equal(a.view(np.uint32), b.view(np.uint32)) for a, b in zip(out[0], out[1]))
    print(f"{tag}: frames with {opt}=1 and {opt}=0 equal bit for bit: {same}", flush=True)
ctx.close()
