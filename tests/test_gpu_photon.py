"""-use-photon-map on the MI355X (qa_photon_maps_build + the PHOTON kernel variants) against the
reference's goldens (tests/golden/photon, produced by oracle/ref_harness --photon-map) and the oracle.

Contract:
  * the stored photons - positions, packed directions, colours, split planes, AND the float powers -
    the emitted-ray counts and the emission counts: EXACT (every record byte-identical with what the
    reference's cyPhotonMap holds after scaling and balancing);
  * first-hit depth, sample counts, cast counters of the gathered frame: EXACT;
  * radiance: these scenes reach 1e4, so the float tolerance is relative: max |diff| <= 1e-6 of the
    frame's largest value, and north_star's absolute RMSE < 1e-4 on top (front-to-back summation of
    the recursive radiance formula moves results by a few ulp, see test_gpu_parity.py).
"""
import numpy as np
import pytest

from conftest import bits, golden_blob, load_golden, load_photon_golden, photon_golden_names

pytestmark = pytest.mark.gpu

REL_MAX_TOL = 1e-6
RMSE_TOL = 1e-4


@pytest.fixture(scope="module")
def ctx():
    from qaray_amd import hip
    c = hip.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("name", photon_golden_names())
def test_photon_maps_and_gathered_frame_match_reference(ctx, name):
    g = load_photon_golden(name)
    meta = g["meta"]
    ctx.upload_scene(golden_blob(meta))
    ctx.build_photon_maps(tuple(meta["photon"]), tuple(meta["caustics"]), seed=meta["seed"])
    emitted, emissions = ctx.photon_maps_info()
    assert emitted == meta["emitted"] and emissions == meta["emissions"]
    pm, cm = ctx.download_photon_map(0), ctx.download_photon_map(1)
    assert pm[1:].tobytes() == g["photon"].tobytes()
    assert cm[1:].tobytes() == g["caustics"].tobytes()
    assert pm[:1].tobytes() == bytes(24) and cm[:1].tobytes() == bytes(24)

    ctx.reset_counters()
    rgb, depth, ns = ctx.render_region(tuple(meta["crop"]), meta["spp_min"], max_bounce=meta["bounce"], seed=meta["seed"])
    cnt = ctx.counters()
    assert np.array_equal(bits(depth), bits(g["depth"])) and np.array_equal(ns, g["ns"])
    assert (cnt["casts_normal"], cnt["casts_shadow"]) == (meta["casts_normal"], meta["casts_shadow"])
    ref = g["rgb"].astype(np.float64)
    assert float(np.abs(rgb - ref).max()) <= REL_MAX_TOL * float(np.abs(ref).max())
    assert float(np.sqrt(np.mean((rgb - ref) ** 2))) <= RMSE_TOL
    # same bits every time (the gather keeps per-lane heaps in global scratch and spills heavily)
    for _ in range(2):
        again = ctx.render_region(tuple(meta["crop"]), meta["spp_min"], max_bounce=meta["bounce"], seed=meta["seed"])[0]
        assert np.array_equal(bits(again), bits(rgb))
    # a sub-region and the stats variant of the kernel see the same pixels
    x0, y0, x1, y1 = 8, 4, 40, 28
    sub = ctx.render_region((x0, y0, x1, y1), meta["spp_min"], max_bounce=meta["bounce"], seed=meta["seed"], stats=True)[0]
    assert np.array_equal(bits(sub), bits(rgb[y0:y1, x0:x1]))
    # qa_photon_maps_clear brings Scene::usePhotonMap = false back
    ctx.clear_photon_maps()
    plain = ctx.render_region(tuple(meta["crop"]), meta["spp_min"], max_bounce=meta["bounce"], seed=meta["seed"])[0]
    assert not np.array_equal(bits(plain), bits(rgb))
    with pytest.raises(Exception, match="no photon maps"):
        ctx.photon_maps_info()


def test_default_sized_maps_against_oracle(ctx):
    """RendererParam's default maps (10000 / 1000 photons, radii 0.2 / 1.0) on the photon test scene:
    more than one emission batch for the caustics map, textures, an OBJ MultiMtl, three lights."""
    from oracle import binding as oracle
    from qaray_amd.host import load_scene_blob
    w, h, spp = 96, 72, 2
    blob = load_scene_blob("custom_photon.xml", size=(w, h))
    ctx.upload_scene(blob)
    ctx.build_photon_maps()
    pp = oracle.photon_params()
    o_pm, o_cm, o_emitted, o_emissions = oracle.photon_build(blob, pp)
    assert ctx.photon_maps_info() == (o_emitted, o_emissions)
    assert ctx.download_photon_map(0).tobytes() == o_pm.tobytes()
    assert ctx.download_photon_map(1).tobytes() == o_cm.tobytes()
    ctx.reset_counters()
    rgb, depth, ns = ctx.render_region((0, 0, w, h), spp)
    cnt = ctx.counters()
    o_rgb, o_depth, o_ns, o_cnt = oracle.render(blob, (0, 0, w, h), spp, photon=(pp, o_pm, o_cm))
    assert np.array_equal(bits(depth), bits(o_depth)) and np.array_equal(ns, o_ns)
    assert (cnt["casts_normal"], cnt["casts_shadow"]) == (o_cnt.casts_normal, o_cnt.casts_shadow)
    assert float(np.abs(rgb - o_rgb).max()) <= REL_MAX_TOL * float(np.abs(o_rgb).max())
    # a new scene drops the maps
    ctx.upload_scene(blob)
    with pytest.raises(Exception, match="no photon maps"):
        ctx.photon_maps_info()


@pytest.mark.parametrize("scene,pm,cm", [("trc_scene_xmas.xml", (3000, 20, 0.5), (300, 20, 1.0)),     # area + spot lights, 40 nodes
                                         ("trc_scene_tower.xml", (2000, 20, 1.0), (100, 20, 2.0))])   # 361 k triangles in global memory
def test_photon_maps_on_large_scenes_against_oracle(ctx, scene, pm, cm):
    from conftest import ensure_assets
    from oracle import binding as oracle
    from qaray_amd.host import load_scene_blob
    ensure_assets()
    w, h, spp = 64, 48, 2
    blob = load_scene_blob(scene, size=(w, h))
    pp = oracle.photon_params(pm, cm)
    o_pm, o_cm, o_emitted, o_emissions = oracle.photon_build(blob, pp)
    ctx.upload_scene(blob)
    ctx.build_photon_maps(pm, cm)
    assert ctx.photon_maps_info() == (o_emitted, o_emissions)
    assert ctx.download_photon_map(0).tobytes() == o_pm.tobytes()
    assert ctx.download_photon_map(1).tobytes() == o_cm.tobytes()
    rgb, depth, ns = ctx.render_region((0, 0, w, h), spp)
    o_rgb, o_depth, o_ns, _ = oracle.render(blob, (0, 0, w, h), spp, photon=(pp, o_pm, o_cm))
    assert np.array_equal(bits(depth), bits(o_depth)) and np.array_equal(ns, o_ns)
    assert float(np.abs(rgb - o_rgb).max()) <= REL_MAX_TOL * max(1.0, float(np.abs(o_rgb).max()))
    # a tile's samples handed out in chunks (options chunk_spp / chunk_tail, qa_kernel.h section A): the PHOTON variants too
    ctx.set_option("chunk_spp", 1)
    ctx.set_option("chunk_tail", 1)
    again = ctx.render_region((0, 0, w, h), spp)
    ctx.set_option("chunk_spp", -1)
    ctx.set_option("chunk_tail", 0)
    for a, b in zip(again, (rgb, depth, ns)):
        assert np.array_equal(bits(a), bits(b))
    ctx.clear_photon_maps()


def test_strips_with_photon_maps_equal_full_frame(ctx):
    """The multi-GPU partition (round-robin 8-row strips) with maps built independently per 'rank'."""
    import torch
    from qaray_amd import distributed as qd, hip
    from qaray_amd.host import load_scene_blob
    w, h, spp = 80, 60, 2
    blob = load_scene_blob("trc_mtl_glass.xml", size=(w, h))
    ctx.upload_scene(blob)
    ctx.build_photon_maps((3000, 20, 1.5), (400, 20, 2.5))
    full = ctx.render_region((0, 0, w, h), spp)[0]
    other = hip.Context(0)   # a second context = a second rank: its own deterministic build of the same maps
    try:
        other.upload_scene(blob)
        other.build_photon_maps((3000, 20, 1.5), (400, 20, 2.5))
        assert other.download_photon_map(0).tobytes() == ctx.download_photon_map(0).tobytes()
        parts = []
        for rank, c in enumerate((ctx, other)):
            rows = hip.strip_count(0, h, rank, 2) * 8
            rgb = torch.zeros((rows, w, 3), dtype=torch.float32, device="cuda")
            depth = torch.zeros((rows, w), dtype=torch.float32, device="cuda")
            ns = torch.zeros((rows, w), dtype=torch.int32, device="cuda")
            c.render_strips_device((0, 0, w, h), rank, 2, spp, rgb, depth, ns)
            c.synchronize()
            parts.append(rgb.cpu().numpy())
        out = np.zeros((h, w, 3), np.float32)
        for rank, part in enumerate(parts):
            for k, y in enumerate(range(rank * 8, h, 16)):
                n = min(8, h - y)
                out[y:y + n] = part[k * 8:k * 8 + n]
        assert np.array_equal(bits(out), bits(full))
    finally:
        other.close()


def test_photon_build_errors(ctx):
    from qaray_amd import hip
    _, _, _, meta = load_golden("c2_box_64x64_4spp")        # Cornell box: emissive plane, no point light
    ctx.upload_scene(golden_blob(meta))
    with pytest.raises(hip.HipError) as e:
        ctx.build_photon_maps((100, 20, 0.2), (10, 20, 1.0))
    assert e.value.code == -6 and "point light" in str(e.value)
    _, _, _, meta = load_golden("box3_48x36_4spp")          # no specular object: the caustics map can never fill
    ctx.upload_scene(golden_blob(meta))
    with pytest.raises(hip.HipError) as e:
        ctx.build_photon_maps((50, 20, 0.2), (2, 20, 1.0))
    assert e.value.code == -6 and "not full" in str(e.value)
    with pytest.raises(hip.HipError) as e:
        ctx.build_photon_maps((0, 20, 0.2), (2, 20, 1.0))
    assert e.value.code == -1   # QA_EINVAL
    # a failed build leaves the context usable, without maps
    rgb = ctx.render_region((0, 0, 48, 36), 1)[0]
    assert np.isfinite(rgb).all()


def test_cli_use_photon_map_matches_python_path(tmp_path):
    """qaray_hip -use-photon-map (the reference's flag, src/main.cpp:35-40) = build_photon_maps + render."""
    import os
    import subprocess
    from conftest import ROOT
    from qaray_amd import hip
    from qaray_amd.host import FrameBuffer, load_scene_blob, SCENES_DIR
    exe = os.path.join(ROOT, "qaray_amd", "lib", "qaray_hip")
    out = str(tmp_path) + "/"
    r = subprocess.run([exe, "-batch", "-spp", "2", "-size", "64", "48", "-root", SCENES_DIR, "-out", out, "-use-photon-map",
                        "-photon-map-size", "3000", "-caustics-map-size", "400", os.path.join(SCENES_DIR, "custom_photon.xml")],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    assert "Photon Map (3000 photons" in r.stdout and "Caustics Map (400" in r.stdout
    c = hip.Context(0)
    try:
        c.upload_scene(load_scene_blob("custom_photon.xml", size=(64, 48)))
        c.build_photon_maps((3000, 20, 0.2), (400, 20, 1.0))
        rgb, depth, ns = c.render_region((0, 0, 64, 48), 2)
        fb = FrameBuffer(64, 48)
        fb.deposit(0, 0, 64, 48, rgb, depth, ns, 2, use_srgb=True)
        ref = str(tmp_path / "py.png")
        fb.save_image(ref)
        assert open(out + "colorBuffer.png", "rb").read() == open(ref, "rb").read()
    finally:
        c.close()
