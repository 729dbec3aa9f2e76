"""Tiles in sample chunks (options "chunk_spp" = a tile's first chunk, "chunk_tail" = every further one; 0 = whole tiles):
   frame time and hash of one scene for several settings.  python tools/gpu_chunk_sweep.py scene.xml W H SPP "0:0,384:128,384:64,..." """
import hashlib, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
subprocess.run([sys.executable, os.path.join(ROOT, "scenes", "gen_assets.py")], check=True, stdout=subprocess.DEVNULL)
from qaray_amd.host import load_scene_blob
from qaray_amd import hip
scene, w, h, spp = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
ctx = hip.Context(0)
ctx.set_option("coop", int(os.environ.get("QA_SWEEP_COOP", "1")))
ctx.upload_scene(load_scene_blob(scene, size=(w, h)))
ctx.render_region((0, 0, 64, 64), 1)
for rep in range(2):
    for spec in sys.argv[5].split(","):
        a, b = [int(x) for x in spec.split(":")]
        ctx.set_option("chunk_spp", a); ctx.set_option("chunk_tail", b)
        ctx.reset_kernel_time(); ctx.reset_counters()
        out = ctx.render_region((0, 0, w, h), spp)
        ms, _ = ctx.kernel_time(); c = ctx.counters()
        hsh = hashlib.sha256()
        for x in out: hsh.update(x.tobytes())
        print(f"{scene} @{spp} chunk_spp={a:4d} chunk_tail={b:4d}: {ms:8.2f} ms {c['samples'] / ms * 1e-3:9.1f} Msamples/s sha {hsh.hexdigest()[:12]} [{ctx.kernel_name()}]", flush=True)
ctx.close()
