"""Photon-map render timing + determinism: custom_photon.xml, default maps (10000 / 1000 photons)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from qaray_amd.host import load_scene_blob
from qaray_amd import hip
ctx = hip.Context(0)
w, h, spp = 640, 480, 8
blob = load_scene_blob("custom_photon.xml", size=(w, h))
ctx.upload_scene(blob)
t0 = time.time(); ctx.build_photon_maps(); tb = time.time() - t0
ctx.render_region((0, 0, w, h), 1)
outs = []
for rep in range(3):
    ctx.reset_kernel_time(); ctx.reset_counters()
    outs.append(ctx.render_region((0, 0, w, h), spp)[0])
    ms, n = ctx.kernel_time(); cnt = ctx.counters()
print(os.environ.get("QA_HIP_LIB", "default").split("/")[-1], f"build {tb*1e3:.0f} ms, render {ms:.1f} ms -> {cnt['samples']/ms*1e-3:.1f} Msamples/s; deterministic:",
      bool(np.array_equal(outs[0], outs[1]) and np.array_equal(outs[1], outs[2])), "finite:", bool(np.isfinite(outs[0]).all()))
ctx.clear_photon_maps(); ctx.reset_kernel_time()
ctx.render_region((0, 0, w, h), spp); ms, n = ctx.kernel_time()
print(f"   same frame without maps: {ms:.1f} ms")
