"""Frame time of one scene for several values of option "sync_samples" (0 = a lane starts its next sample at once, 1 = when the whole
   wave is between samples, n >= 2 = cooperative kernel: finished paths wait until n have gathered).
   python tools/gpu_sync_sweep.py scene.xml W H SPP 0,1,8,16,32"""
import hashlib, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
subprocess.run([sys.executable, os.path.join(ROOT, "scenes", "gen_assets.py")], check=True, stdout=subprocess.DEVNULL)
from qaray_amd.host import load_scene_blob
from qaray_amd import hip
scene, w, h, spp = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
ctx = hip.Context(0)
ctx.upload_scene(load_scene_blob(scene, size=(w, h)))
ctx.render_region((0, 0, 64, 64), 1)
for rep in range(2):
    for v in [int(x) for x in sys.argv[5].split(",")]:
        ctx.set_option("sync_samples", v)
        ctx.reset_kernel_time(); ctx.reset_counters()
        out = ctx.render_region((0, 0, w, h), spp)
        ms, _ = ctx.kernel_time(); c = ctx.counters()
        hsh = hashlib.sha256()
        for a in out: hsh.update(a.tobytes())
        print(f"{scene} sync_samples={v:2d}: {ms:8.2f} ms {c['samples'] / ms * 1e-3:9.1f} Msamples/s sha {hsh.hexdigest()[:12]} [{ctx.kernel_name()}]", flush=True)
ctx.close()
