"""Parity + throughput probe on the synthetic-asset scenes (C3-C5 stand-ins)."""
import sys, time, os, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scenes", "gen_assets.py")], check=True)
from qaray_amd.host import load_scene_blob
from qaray_amd import hip
from oracle import binding as ob
ctx = hip.Context(0)
for name in ["example_project7_object.xml", "example_project12_caustics_glossy.xml", "trc_scene_tower.xml"]:
    w, h, spp = 96, 72, 2
    blob = load_scene_blob(name, size=(w, h))
    ctx.upload_scene(blob); ctx.reset_counters()
    rgb, depth, ns = ctx.render_region((0, 0, w, h), spp)
    cnt = ctx.counters()
    orgb, odepth, ons, ocnt = ob.render(blob, (0, 0, w, h), spp)
    rmse = float(np.sqrt(np.mean((rgb.astype(np.float64) - orgb) ** 2)))
    print(f"{name} {w}x{h}x{spp}: rmse={rmse:.3e} max={np.abs(rgb-orgb).max():.3e} depth-equal={np.array_equal(depth, odepth)} "
          f"casts {cnt['casts_normal']}/{cnt['casts_shadow']} oracle {ocnt.casts_normal}/{ocnt.casts_shadow}", flush=True)
for name, (w, h), spp in [("example_project7_object.xml", (1920, 1080), 16), ("example_project12_caustics_glossy.xml", (1920, 1080), 32),
                          ("trc_scene_tower.xml", (1920, 1080), 16)]:
    blob = load_scene_blob(name, size=(w, h))
    ctx.upload_scene(blob); ctx.reset_counters(); ctx.reset_kernel_time()
    t0 = time.time()
    rgb, depth, ns = ctx.render_region((0, 0, w, h), spp)
    dt = time.time() - t0
    ms, n = ctx.kernel_time(); cnt = ctx.counters()
    print(f"{name} {w}x{h}x{spp}: kernel {ms:.1f} ms -> {cnt['samples']/ms*1e-3:.1f} Msamples/s, casts/sample {(cnt['casts_normal']+cnt['casts_shadow'])/cnt['samples']:.2f} "
          f"({(cnt['casts_normal']+cnt['casts_shadow'])/ms*1e-6:.2f} Gcasts/s) finite={np.isfinite(rgb).all()}", flush=True)
