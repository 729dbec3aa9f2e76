"""One scene, one frame: python tools/gpu_one.py <scene.xml> <w> <h> <spp> [warm]  (QA_PIPELINE selects the integrator)."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
subprocess.run([sys.executable, os.path.join(ROOT, "scenes", "gen_assets.py")], check=True, stdout=subprocess.DEVNULL)
from qaray_amd.host import load_scene_blob
from qaray_amd import hip
scene, w, h, spp = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
ctx = hip.Context(0)
ctx.upload_scene(load_scene_blob(scene, size=(w, h)))
ctx.render_region((0, 0, 64, 64), 1)
ctx.reset_kernel_time(); ctx.reset_counters()
t0 = time.time(); ctx.render_region((0, 0, w, h), spp); wall = time.time() - t0
ms, _ = ctx.kernel_time(); c = ctx.counters()
casts = c["casts_normal"] + c["casts_shadow"]
print(f"{scene} {w}x{h} @ {spp}: {ms:.1f} ms (wall {wall*1e3:.0f}), {c['samples'] / ms * 1e-3:.1f} Msamples/s, {casts / c['samples']:.2f} casts/sample, "
      f"{casts / ms * 1e-6:.2f} Gcasts/s", flush=True)
if "staged" in ctx.kernel_name():
    print(ctx.staged_stats())
