"""Section shares of the megakernel's wave time (diagnostic build: make hip LIBDIR=../lib_stamp OBJDIR=../lib_stamp/obj EXTRA=-DQA_STAMPS).
   QA_HIP_LIB=qaray_amd/lib_stamp/libqaray_hip.so python tools/gpu_stamps.py [spp]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
subprocess.run([sys.executable, os.path.join(ROOT, "scenes", "gen_assets.py")], check=True, stdout=subprocess.DEVNULL)
os.environ.setdefault("QA_PIPELINE", "mega")
only = os.environ.get("QA_STAMP_CASES", "c2,c3,c4,c5").split(",")
from qaray_amd.host import load_scene_blob
from qaray_amd import hip
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 32
CASES = [("c2", "example_project12_box.xml", (1920, 1080), 4 * spp), ("c3", "example_project7_object.xml", (1920, 1080), spp),
         ("c4", "example_project12_caustics_glossy.xml", (3840, 2160), spp), ("c5", "trc_scene_tower.xml", (3840, 2160), spp)]
ctx = hip.Context(0)
for tag, scene, size, n in CASES:
    if tag not in only: continue
    ctx.upload_scene(load_scene_blob(scene, size=size))
    ctx.render_region((0, 0, 64, 64), 1)
    ctx.reset_kernel_time(); ctx.reset_counters()
    ctx.render_region((0, 0) + size, n)
    ms, _ = ctx.kernel_time()
    sys.stderr.flush()
    print(f"{tag}: {size[0]}x{size[1]} @ {n} spp {ms:.1f} ms [{ctx.kernel_name()}]", flush=True)
    c = ctx.counters()
    print(f"    {c['samples'] / ms * 1e-3:.1f} Msamples/s, casts/sample {(c['casts_normal'] + c['casts_shadow']) / c['samples']:.2f}", flush=True)
ctx.close()
