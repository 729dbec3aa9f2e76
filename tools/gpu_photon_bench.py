"""Photon-map render timing: frames with and without the maps (kernel ms from HIP events)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from qaray_amd.host import load_scene_blob
from qaray_amd import hip
ctx = hip.Context(0)
w, h, spp = 1920, 1080, 8
for scene, pm, cm in [("trc_mtl_glass.xml", (10000, 20, 0.2), (1000, 20, 1.0)),
                      ("trc_mtl_glass.xml", (100000, 20, 1.0), (10000, 20, 1.0)),
                      ("trc_mtl_glass.xml", (1000000, 20, 0.5), (100000, 20, 0.5)),
                      ("custom_photon.xml", (10000, 20, 0.2), (1000, 20, 1.0))]:
    blob = load_scene_blob(scene, size=(w, h))
    ctx.upload_scene(blob)
    ctx.reset_kernel_time(); ctx.render_region((0, 0, w, h), spp); ms0, _ = ctx.kernel_time()
    t0 = time.time(); ctx.build_photon_maps(pm, cm); tb = time.time() - t0
    emitted, emissions = ctx.photon_maps_info()
    ctx.render_region((0, 0, w, h), 1)
    ctx.reset_kernel_time(); ctx.reset_counters()
    a = ctx.render_region((0, 0, w, h), spp)[0]
    ms, n = ctx.kernel_time(); cnt = ctx.counters()
    b = ctx.render_region((0, 0, w, h), spp)[0]
    print(f"{scene} maps {pm[0]}/{cm[0]} r {pm[2]}/{cm[2]}: build {tb*1e3:.0f} ms ({emissions} emissions), frame {ms:.1f} ms = {cnt['samples']/ms*1e-3:.1f} Msamples/s "
          f"(without maps {ms0:.1f} ms); deterministic {bool(np.array_equal(a, b))} finite {bool(np.isfinite(a).all())}", flush=True)
