"""A/B of the staged integrator against the megakernel on the non-resident BASELINE scenes (C3 - C5):
bitwise comparison of the two pipelines on a small frame, then throughput at the BASELINE frame size."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
subprocess.run([sys.executable, os.path.join(ROOT, "scenes", "gen_assets.py")], check=True, stdout=subprocess.DEVNULL)
from qaray_amd.host import load_scene_blob
from qaray_amd import hip

CASES = [("C3", "example_project7_object.xml", (1920, 1080), 16), ("C4", "example_project12_caustics_glossy.xml", (3840, 2160), 16),
         ("C5", "trc_scene_tower.xml", (3840, 2160), 8)]
which = sys.argv[1:] or ["mega", "staged"]
res = {}
for mode in which:
    os.environ["QA_PIPELINE"] = mode
    ctx = hip.Context(0)
    for tag, scene, size, spp in CASES:
        small = (size[0] // 8, size[1] // 8)
        ctx.upload_scene(load_scene_blob(scene, size=small))
        ctx.reset_counters()
        out = ctx.render_region((0, 0) + small, 4)
        res[(mode, tag)] = (out, ctx.counters())
        ctx.upload_scene(load_scene_blob(scene, size=size))
        ctx.render_region((0, 0, 64, 64), 1)
        ctx.reset_kernel_time(); ctx.reset_counters()
        t0 = time.time(); ctx.render_region((0, 0) + size, spp); wall = time.time() - t0
        ms, _ = ctx.kernel_time(); c = ctx.counters()
        casts = c["casts_normal"] + c["casts_shadow"]
        print(f"{mode} {tag}: {size[0]}x{size[1]} @ {spp} spp: {ms:.1f} ms (wall {wall*1e3:.0f}), {c['samples'] / ms * 1e-3:.1f} Msamples/s, "
              f"{casts / c['samples']:.2f} casts/sample, {casts / ms * 1e-6:.2f} Gcasts/s", flush=True)
        if "staged" in ctx.kernel_name():
            print("   ", ctx.staged_stats(), flush=True)
    ctx.close()
if len(which) == 2:
    for tag, *_ in CASES:
        (a, ca), (b, cb) = res[(which[0], tag)], res[(which[1], tag)]
        same = [bool(np.array_equal(x.view(np.uint32), y.view(np.uint32))) for x, y in zip(a, b)]
        print(f"{tag}: rgb/depth/ns bit-equal {same}, max |drgb| {np.abs(a[0] - b[0]).max():.3e}, counters equal "
              f"{[ca[k] == cb[k] for k in ('samples', 'casts_normal', 'casts_shadow', 'pixels')]}", flush=True)
