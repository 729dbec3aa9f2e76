"""Photon-map parity probe on the GPU: maps and gathered image against the committed goldens
(tests/golden/photon, produced by the reference) and against the oracle."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from conftest import load_photon_golden, photon_golden_names, golden_blob
from qaray_amd import hip
ctx = hip.Context(0)
for name in photon_golden_names():
    g = load_photon_golden(name); meta = g["meta"]
    blob = golden_blob(meta)
    ctx.upload_scene(blob)
    t0 = time.time()
    ctx.build_photon_maps(tuple(meta["photon"]), tuple(meta["caustics"]), seed=meta["seed"])
    dt = time.time() - t0
    emitted, emissions = ctx.photon_maps_info()
    pm, cm = ctx.download_photon_map(0), ctx.download_photon_map(1)
    okp = pm[1:].tobytes() == g["photon"].tobytes(); okc = cm[1:].tobytes() == g["caustics"].tobytes()
    print(f"{name}: build {dt*1e3:.1f} ms emitted {emitted} (ref {meta['emitted']}) emissions {emissions} (ref {meta['emissions']}) "
          f"photon map equal {okp} caustics equal {okc}", flush=True)
    for ok, a, b in ((okp, pm[1:], g["photon"]), (okc, cm[1:], g["caustics"])):
        if ok:
            continue
        for f in hip.PHOTON_DTYPE.names:
            print("   ", f, "differing records:", int((a[f] != b[f]).reshape(len(a), -1).any(axis=1).sum()))
        sa = np.sort(a.view(np.dtype((np.void, 24)))); sb = np.sort(b.view(np.dtype((np.void, 24))))
        print("    as multisets equal:", np.array_equal(sa, sb))
        bad = np.nonzero(a.view(np.dtype((np.void, 24))) != b.view(np.dtype((np.void, 24))))[0][:4]
        for i in bad:
            print("    gpu", a[i], "\n    ref", b[i])
    ctx.reset_counters()
    rgb, depth, ns = ctx.render_region(tuple(meta["crop"]), meta["spp_min"], max_bounce=meta["bounce"], seed=meta["seed"])
    cnt = ctx.counters()
    d = rgb.astype(np.float64) - g["rgb"]
    print(f"   render rmse {np.sqrt((d**2).mean()):.3e} max {np.abs(d).max():.3e} rel-max {np.abs(d).max()/max(1e-9, np.abs(g['rgb']).max()):.3e} "
          f"depth equal {np.array_equal(depth, g['depth'])} casts {cnt['casts_normal']}/{cnt['casts_shadow']} ref {meta['casts_normal']}/{meta['casts_shadow']}", flush=True)
    rgbB, _, _ = ctx.render_region(tuple(meta["crop"]), meta["spp_min"], max_bounce=meta["bounce"], seed=meta["seed"])
    print("   second render identical:", np.array_equal(rgb, rgbB), " nan pixels gpu/ref:", int(np.isnan(rgb).any(axis=2).sum()), int(np.isnan(g["rgb"]).any(axis=2).sum()))
    ad = np.abs(d).max(axis=2); ys, xs = np.nonzero(ad > 1e-3 * np.abs(g["rgb"]).max())
    print("   pixels off by > 1e-3 of max:", len(ys), list(zip(xs[:6].tolist(), ys[:6].tolist())), [float(ad[y, x]) for y, x in zip(ys[:6], xs[:6])])
    ctx.clear_photon_maps()
    rgb2, _, _ = ctx.render_region(tuple(meta["crop"]), meta["spp_min"], max_bounce=meta["bounce"], seed=meta["seed"])
    print("   after clear differs from photon render:", not np.array_equal(rgb, rgb2))
