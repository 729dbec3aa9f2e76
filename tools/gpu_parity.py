"""Quick HIP-vs-oracle parity probe (developer tool; the real checks live in tests/)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from qaray_amd.host import load_scene_blob
from qaray_amd import hip
from oracle import binding as ob

cases = [("example_project12_box.xml", (64, 64), 4), ("example_project3_sphere.xml", (256, 256), 1),
         ("example_project12_box.xml", (96, 80), 16), ("example_project3_sphere.xml", (100, 75), 8),
         ("example_project2_blinn.xml", (96, 72), 8), ("example_project3_box.xml", (96, 72), 8),
         ("example_project4.xml", (96, 72), 8), ("trc_mtl_glass.xml", (96, 72), 8),
         ("trc_mtl_glossy.xml", (96, 72), 8), ("trc_mtl_coffee.xml", (96, 72), 8),
         ("custom_textures.xml", (160, 120), 4), ("custom_textures.xml", (80, 60), 2),
         ("custom_softshadow.xml", (120, 90), 4), ("custom_softshadow.xml", (60, 45), 2)]
ctx = hip.Context(0)
for name, (w, h), spp in cases:
    blob = load_scene_blob(name, size=(w, h))
    t0 = time.time(); ctx.upload_scene(blob); ctx.reset_counters()
    rgb, depth, ns = ctx.render_region((0, 0, w, h), spp, stats=True)
    t1 = time.time()
    cnt = ctx.counters()
    orgb, odepth, ons, ocnt = ob.render(blob, (0, 0, w, h), spp)
    same = np.array_equal(rgb.view(np.uint32), orgb.view(np.uint32))
    npx = int((rgb.view(np.uint32) != orgb.view(np.uint32)).any(axis=2).sum())
    rmse = float(np.sqrt(np.mean((rgb.astype(np.float64) - orgb) ** 2)))
    big = int((np.abs(rgb - orgb).max(axis=2) > 1e-4).sum())
    dsame = np.array_equal(depth.view(np.uint32), odepth.view(np.uint32))
    print(f"{name} {w}x{h}x{spp}: bit-equal={same} pixels!= {npx}/{w*h} (>1e-4: {big}) rmse={rmse:.3e} "
          f"max={np.abs(rgb-orgb).max():.3e} depth-equal={dsame} ns-ok={np.array_equal(ns, ons)} "
          f"gpu casts {cnt['casts_normal']}/{cnt['casts_shadow']} oracle {ocnt.casts_normal}/{ocnt.casts_shadow} "
          f"bvh {cnt['bvh_nodes']} vs {ocnt.bvh_nodes} tri {cnt['tri_tests']} vs {ocnt.tri_tests} ({t1-t0:.2f}s)", flush=True)
