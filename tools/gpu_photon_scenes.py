"""Photon maps on more scenes (non-resident, textured, area lights): HIP vs oracle."""
import sys, os, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scenes", "gen_assets.py")], check=True, stdout=subprocess.DEVNULL)
from qaray_amd.host import load_scene_blob
from qaray_amd import hip
from oracle import binding as ob
ctx = hip.Context(0)
w, h, spp = 64, 48, 2
for name, pm, cm in [("trc_scene_xmas.xml", (3000, 20, 0.5), (300, 20, 1.0)), ("example_project7_object.xml", (3000, 20, 2.0), (200, 20, 4.0)),
                     ("example_project12_caustics_glossy.xml", (3000, 20, 2.0), (300, 20, 4.0)), ("trc_scene_tower.xml", (2000, 20, 1.0), (100, 20, 2.0)),
                     ("example_project11_caustics.xml", (3000, 20, 2.0), (300, 20, 4.0))]:
    blob = load_scene_blob(name, size=(w, h))
    pp = ob.photon_params(pm, cm)
    try:
        opm, ocm, oe, oem = ob.photon_build(blob, pp)
    except RuntimeError as e:
        print(name, "oracle:", e)
        try:
            ctx.upload_scene(blob); ctx.build_photon_maps(pm, cm); print("   HIP built although the oracle refused!")
        except hip.HipError as e2:
            print("   HIP:", str(e2)[:100])
        continue
    ctx.upload_scene(blob)
    ctx.build_photon_maps(pm, cm)
    e, em = ctx.photon_maps_info()
    a, b = ctx.download_photon_map(0), ctx.download_photon_map(1)
    rgb, depth, ns = ctx.render_region((0, 0, w, h), spp)
    orgb, od, ons, oc = ob.render(blob, (0, 0, w, h), spp, photon=(pp, opm, ocm))
    rel = float(np.abs(rgb - orgb).max() / max(1.0, np.abs(orgb).max()))
    print(f"{name}: emitted {e}=={oe} {e == oe}, maps equal {a.tobytes() == opm.tobytes()} {b.tobytes() == ocm.tobytes()}, frame rel-max {rel:.2e} depth {np.array_equal(depth, od)}", flush=True)
