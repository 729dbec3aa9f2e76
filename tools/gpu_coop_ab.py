"""Cooperative kernel on / off (option "coop") on arbitrary scenes: python tools/gpu_coop_ab.py spp scene.xml[:WxH] ..."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
subprocess.run([sys.executable, os.path.join(ROOT, "scenes", "gen_assets.py")], check=True, stdout=subprocess.DEVNULL)
from qaray_amd.host import load_scene_blob
from qaray_amd import hip
spp = int(sys.argv[1])
ctx = hip.Context(0)
for arg in sys.argv[2:]:
    scene, _, sz = arg.partition(":")
    size = tuple(int(x) for x in sz.split("x")) if sz else (1920, 1080)
    blob = load_scene_blob(scene, size=size)
    out = {}
    for coop in (0, 1):
        ctx.set_option("coop", coop)
        ctx.upload_scene(blob)
        ctx.render_region((0, 0, 64, 64), 1)
        ctx.reset_kernel_time(); ctx.reset_counters()
        out[coop] = ctx.render_region((0, 0) + size, spp)
        ms, _ = ctx.kernel_time(); c = ctx.counters()
        print(f"{scene} {size[0]}x{size[1]} @ {spp} coop={coop}: {ms:8.1f} ms {c['samples'] / ms * 1e-3:8.1f} Msamples/s {(c['casts_normal'] + c['casts_shadow']) / c['samples']:.2f} casts/sample [{ctx.kernel_name()}]", flush=True)
    same = all(np.array_equal(a.view(np.uint32), b.view(np.uint32)) for a, b in zip(out[0], out[1]))
    print(f"   cooperative frame == per-lane frame, bit for bit: {same}", flush=True)
ctx.close()
