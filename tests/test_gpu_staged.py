"""Scenes whose geometry does not fit LDS (BASELINE C3 / C4 / C5) on a real MI355X: the staged integrator
(qa_wf.h: logic / cull / trace / redo stages, queues in HBM) and the 4-wide search tree over the reference tree's
leaves (qa_widebvh.h), each against the reference's goldens, the CPU oracle, the megakernel and the counting
kernel that walks the reference's cy::BVH exactly as the reference does.

Tolerances: as tests/test_gpu_parity.py (sample counts, first-hit depth, cast counters EXACT; radiance RMSE <= 1e-6,
max abs <= 1e-4 against the reference).  Between the library's own integrators everything is compared BIT FOR BIT:
they perform the same arithmetic in the same order and differ only in where the search for the closest hit looks."""
import os

import numpy as np
import pytest

from conftest import bits, ensure_assets, golden_blob, load_golden, reference_input_names

pytestmark = pytest.mark.gpu

RMSE_TOL = 1e-6
MAXABS_TOL = 1e-4
BIG = ["c3_object_1080p_crop_2spp", "c4_caustics_4k_crop_4spp", "c5_tower_4k_crop_2spp"]
BIG_FULLSPP = ["c3_object_1080p_crop_256spp", "c4_caustics_4k_crop_1024spp", "c5_tower_4k_crop_2048spp"]


@pytest.fixture(scope="module")
def ctx():
    from qaray_amd import hip
    c = hip.Context(0)
    yield c
    c.set_pipeline("auto")
    c.close()


def rmse(a, b):
    return float(np.sqrt(np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)))


def _golden_available(name):
    from conftest import GOLDEN
    return os.path.exists(os.path.join(GOLDEN, name + ".npz"))


@pytest.mark.parametrize("pipeline", ["staged", "mega"])
@pytest.mark.parametrize("name", BIG + BIG_FULLSPP)
def test_big_scene_goldens_on_both_integrators(ctx, name, pipeline):
    """The reference's own pixels (oracle/_ref) for the three non-resident BASELINE scenes, at 2 - 4 spp and at the
    spp BASELINE.json quotes them on (256 / 1024 / 2048), through either integrator."""
    if not _golden_available(name):
        pytest.skip("golden not generated")
    rgb, depth, ns, meta = load_golden(name)
    ctx.set_pipeline(pipeline)
    ctx.upload_scene(golden_blob(meta))
    assert ("staged" in ctx.kernel_name()) == (pipeline == "staged")
    ctx.reset_counters()
    g_rgb, g_depth, g_ns = ctx.render_region(tuple(meta["crop"]), meta["spp_min"], max_bounce=meta["bounce"], seed=meta["seed"],
                                             spp_max=meta["spp_max"])
    cnt = ctx.counters()
    assert np.array_equal(g_ns, ns)
    assert np.array_equal(bits(g_depth), bits(depth))
    assert (cnt["samples"], cnt["casts_normal"], cnt["casts_shadow"]) == (meta["samples"], meta["casts_normal"], meta["casts_shadow"])
    assert rmse(g_rgb, rgb) <= RMSE_TOL
    assert float(np.abs(g_rgb - rgb).max()) <= MAXABS_TOL


@pytest.mark.parametrize("name", reference_input_names())
def test_every_staged_eligible_reference_input_vs_oracle(ctx, name):
    """The reference's inputs/*.xml that the staged integrator takes (meshes beyond LDS, no area lights): exact sample
    counts, depth and cast counters, radiance within the module's tolerances of the CPU oracle."""
    from oracle import binding as oracle
    from qaray_amd.host import load_scene_blob
    ensure_assets()
    w, h, spp = 40, 30, 2
    blob = load_scene_blob(name, size=(w, h))
    ctx.set_pipeline("staged")
    ctx.upload_scene(blob)
    if "staged" not in ctx.kernel_name():
        pytest.skip("scene runs on the megakernel (LDS-resident, area lights, ...)")
    ctx.reset_counters()
    rgb, depth, ns = ctx.render_region((0, 0, w, h), spp)
    cnt = ctx.counters()
    o_rgb, o_depth, o_ns, o_cnt = oracle.render(blob, (0, 0, w, h), spp)
    assert np.array_equal(ns, o_ns) and np.array_equal(bits(depth), bits(o_depth))
    assert (cnt["samples"], cnt["casts_normal"], cnt["casts_shadow"]) == (o_cnt.samples, o_cnt.casts_normal, o_cnt.casts_shadow)
    scale = max(1.0, float(np.abs(o_rgb[np.isfinite(o_rgb)]).max()) if np.isfinite(o_rgb).any() else 1.0)
    assert float(np.nanmax(np.abs(rgb - o_rgb))) <= MAXABS_TOL * scale
    assert rmse(np.nan_to_num(rgb), np.nan_to_num(o_rgb)) <= RMSE_TOL * scale


@pytest.mark.parametrize("scene,size,spp", [("example_project7_object.xml", (480, 270), 8), ("example_project12_caustics_glossy.xml", (480, 270), 8),
                                            ("trc_scene_tower.xml", (480, 270), 8), ("trc_scene_xmas.xml", (320, 180), 4)])
def test_staged_megakernel_and_reference_walk_agree_bitwise(ctx, scene, size, spp):
    """10^6..10^7 casts per scene: the staged integrator, the megakernel on the 4-wide tree and the counting kernel
    (the reference's binary tree, walked as the reference walks it) return the same bits and the same cast counts.
    Adaptive sampling included (the sample count of a pixel depends on every radiance bit before it)."""
    from qaray_amd.host import load_scene_blob
    ensure_assets()
    w, h = size
    ctx.upload_scene(load_scene_blob(scene, size=size))
    outs, cnts = {}, {}
    for mode in ("staged", "mega", "stats"):
        ctx.set_pipeline("staged" if mode == "staged" else "mega")
        ctx.reset_counters()
        outs[mode] = ctx.render_region((0, 0, w, h), spp, spp_max=2 * spp, stats=(mode == "stats"))
        cnts[mode] = ctx.counters()
        if mode == "staged":
            if "staged" not in ctx.kernel_name():
                pytest.skip("scene is not staged-eligible")
            st = ctx.staged_stats()
            assert st["jobs_done"] == st["jobs_queued"] > 0
    for mode in ("staged", "mega"):
        for a, b in zip(outs[mode], outs["stats"]):
            assert np.array_equal(bits(a), bits(b)), mode
        assert all(cnts[mode][k] == cnts["stats"][k] for k in ("samples", "casts_normal", "casts_shadow", "pixels")), mode
    assert cnts["stats"]["tri_tests"] > 0


def _write_big_fuzz_scene(tmp, rng, kind):
    """Meshes of a few thousand triangles (beyond LDS, so the 4-wide tree and the staged integrator take them) built to
    hit the corners of the reference's mesh search: exactly coplanar axis-aligned sheets cut into many triangles (flat
    leaf boxes, hits on shared edges, ties between coincident sheets), a closed bumpy shell seen from inside and
    outside, random soup with duplicated triangles, and a second instance of the same mesh touching the first."""
    v, f = [], []

    def tri(a, b, c):
        base = len(v)
        v.extend([a, b, c])
        f.append((base + 1, base + 2, base + 3))

    def grid(origin, du, dv, n, bump=0.0):
        P = [[origin + du * (i / n) + dv * (j / n) + bump * np.sin(3.1 * i) * np.cos(2.3 * j) * np.cross(du, dv) / np.linalg.norm(np.cross(du, dv))
              for j in range(n + 1)] for i in range(n + 1)]
        for i in range(n):
            for j in range(n):
                tri(P[i][j], P[i + 1][j], P[i + 1][j + 1])
                tri(P[i][j], P[i + 1][j + 1], P[i][j + 1])

    X, Y, Z = np.eye(3)
    if kind == "sheets":      # coplanar axis-aligned sheets, two of them coincident, one offset by one ulp-ish amount
        for z in (-2.0, 0.0, 0.0, 0.0 + 1e-6, 3.0):
            grid(np.array([-5.0, -5.0, z]), 10 * X, 10 * Y, 12)
        grid(np.array([2.0, -5.0, -4.0]), 10 * Y, 9 * Z, 12)     # a wall the sheets run into
    elif kind == "shell":     # closed box with bumpy faces around the origin: rays start inside after the first bounce
        for s in (-1.0, 1.0):
            grid(np.array([-4.0, -4.0, 4.0 * s]), 8 * X, 8 * Y, 14, bump=0.05)
            grid(np.array([-4.0, 4.0 * s, -4.0]), 8 * X, 8 * Z, 14, bump=0.05)
            grid(np.array([4.0 * s, -4.0, -4.0]), 8 * Y, 8 * Z, 14, bump=0.05)
    else:                     # "soup": random triangles, every fourth one repeated exactly, a few degenerate
        for k in range(700):
            c = rng.uniform(-5, 5, 3)
            t = c + rng.uniform(-1.2, 1.2, (3, 3))
            tri(*t)
            if k % 4 == 0:
                tri(*t)
            if k % 50 == 0:
                tri(t[0], t[0], t[1])
    with open(os.path.join(tmp, "fuzz.obj"), "w") as o:
        for p in v:
            o.write("v %.9g %.9g %.9g\n" % tuple(p))
        for t in f:
            o.write("f %d %d %d\n" % t)
    glass = kind == "shell"
    xml = """<xml><scene>
      <object type="obj" name="fuzz.obj" material="%s"><rotate angle="%s" x="1" y="0.3" z="0.1"/><translate x="0.5" z="1"/></object>
      <object type="obj" name="fuzz.obj" material="m"><scale value="0.5"/><translate x="0.5" y="3" z="1"/></object>
      <object type="sphere" name="ball" material="g"><scale value="1.5"/><translate x="-3" y="-6" z="0"/></object>
      <object type="plane" name="floor" material="m"><scale value="40"/><translate z="-7"/></object>
      <material type="blinn" name="m"><diffuse r="0.7" g="0.6" b="0.5"/><specular value="0.3"/><glossiness value="20"/><emission value="0.1"/></material>
      <material type="blinn" name="g"><diffuse value="0.05"/><specular value="0.8"/><glossiness value="60"/><refraction value="0.9" index="1.4"/>
        <absorption r="0.05" g="0.1" b="0.2"/></material>
      <light type="point" name="l"><intensity value="60"/><position x="3" y="-9" z="11"/></light>
      <light type="direct" name="d"><intensity value="0.4"/><direction x="-0.3" y="0.5" z="-1"/></light>
    </scene><camera><position x="0" y="-24" z="7"/><target x="0" y="0" z="0"/><up x="0" y="0" z="1"/><fov value="40"/>
      <width value="128"/><height value="96"/></camera></xml>""" % ("g" if glass else "m", "0" if kind == "sheets" else "20")
    path = os.path.join(tmp, "fuzz.xml")
    open(path, "w").write(xml)
    return path


@pytest.mark.parametrize("seed", [0, 1])
@pytest.mark.parametrize("kind", ["sheets", "shell", "soup"])
def test_big_fuzz_meshes_all_searches_agree(ctx, tmp_path, kind, seed):
    """Non-resident fuzz meshes: staged == megakernel (4-wide tree) == counting kernel (reference walk) bit for bit, and
    depth / cast counts == the CPU oracle.  The sheets case makes the searches disagree about WHICH of two coincident
    triangles is hit unless ties and the order check send those rays to the exact walk - the test fails if they do not."""
    from oracle import binding as oracle
    from qaray_amd.host import load_scene_blob
    rng = np.random.default_rng(100 * seed + {"sheets": 1, "shell": 2, "soup": 3}[kind])
    xml = _write_big_fuzz_scene(str(tmp_path), rng, kind)
    w, h, spp = 160 + 32 * seed, 120, 4
    blob = load_scene_blob(xml, size=(w, h), asset_root=str(tmp_path))
    ctx.upload_scene(blob)
    outs, cnts = {}, {}
    for mode in ("staged", "mega", "stats"):
        ctx.set_pipeline("staged" if mode == "staged" else "mega")
        ctx.reset_counters()
        outs[mode] = ctx.render_region((0, 0, w, h), spp, stats=(mode == "stats"))
        cnts[mode] = ctx.counters()
        if mode == "staged":
            assert "staged" in ctx.kernel_name(), "fuzz scene should be staged-eligible: " + ctx.kernel_name()
    for mode in ("staged", "mega"):
        for a, b in zip(outs[mode], outs["stats"]):
            assert np.array_equal(bits(a), bits(b)), mode
        assert (cnts[mode]["casts_normal"], cnts[mode]["casts_shadow"]) == (cnts["stats"]["casts_normal"], cnts["stats"]["casts_shadow"]), mode
    o = oracle.render(blob, (0, 0, w, h), spp)
    assert np.array_equal(bits(o[1]), bits(outs["stats"][1]))
    assert (cnts["stats"]["casts_normal"], cnts["stats"]["casts_shadow"]) == (o[3].casts_normal, o[3].casts_shadow)
    assert rmse(np.nan_to_num(outs["stats"][0]), np.nan_to_num(o[0])) <= RMSE_TOL


@pytest.mark.parametrize("case", ["tower", "object", "sheets"])
def test_cooperative_walks_equal_the_sequential_megakernel(tmp_path, case):
    """qa_integrate_cs (the whole wave walks the closest-hit and shadow queries of a mesh from a pool in LDS, qa_kernel_cs.h)
    against qa_integrate (every lane walks its own ray; option "coop" = 0) and the counting kernel (reference tree, walked
    as the reference walks it): same bits, same cast counts.  The coincident sheets make equal-distance accepts from
    different lanes in one round - the case the 64-bit key's return value has to catch."""
    from qaray_amd import hip
    from qaray_amd.host import load_scene_blob
    ensure_assets()
    if case == "sheets":
        xml = _write_big_fuzz_scene(str(tmp_path), np.random.default_rng(11), "sheets")
        size, spp = (256, 192), 4
        blob = load_scene_blob(xml, size=size, asset_root=str(tmp_path))
    else:
        size, spp = (480, 270), 8
        blob = load_scene_blob("trc_scene_tower.xml" if case == "tower" else "example_project7_object.xml", size=size)
    w, h = size
    outs, cnts, names = {}, {}, {}
    for mode in ("coop", "tiny", "own", "stats"):
        c = hip.Context(0)
        c.set_option("coop", 0 if mode == "own" else 1)
        # "tiny": a pool of 64 items overflows constantly - those rays must come back from the exact repeat unchanged
        c.set_option("cs_pool_limit", 64 if mode == "tiny" else 0)
        c.upload_scene(blob)
        c.set_pipeline("mega")
        c.reset_counters()
        names[mode] = c.kernel_name()
        outs[mode] = c.render_region((0, 0, w, h), spp, stats=(mode == "stats"))
        cnts[mode] = c.counters()
        if mode == "stats":
            assert "counting variant" in c.kernel_name(), c.kernel_name()   # the name of what was launched
        c.close()
    assert "qa_integrate_cs" in names["coop"] and "qa_integrate_cs" in names["tiny"] and "qa_integrate_cs" not in names["own"], names
    for mode in ("coop", "tiny", "own"):
        for a, b in zip(outs[mode], outs["stats"]):
            assert np.array_equal(bits(a), bits(b)), mode
        assert all(cnts[mode][k] == cnts["stats"][k] for k in ("samples", "casts_normal", "casts_shadow", "pixels")), mode


def _write_many_nodes_scene(path, n_side=6, spot=False):
    """A field of n_side^2 objects (spheres, every fifth a rotated teapot instance) on a floor under two lights: 38 nodes."""
    objs = ['<object type="plane" name="floor" material="floor"><scale value="60"/></object>']
    for i in range(n_side):
        for j in range(n_side):
            x, y = (i - (n_side - 1) / 2) * 7.0, (j - (n_side - 1) / 2) * 7.0
            if (i + j) % 5 == 0:
                objs.append('<object type="obj" name="teapot-low.obj" material="m%d"><scale value="0.25"/><rotate angle="%d" z="1"/>'
                            '<translate x="%g" y="%g" z="0"/></object>' % ((i * 7 + j) % 3, (i * 37 + j * 11) % 360, x, y))
            else:
                objs.append('<object type="sphere" name="s%d_%d" material="m%d"><scale value="%g"/><translate x="%g" y="%g" z="2.2"/></object>'
                            % (i, j, (i * 7 + j) % 3, 1.5 + ((i * 3 + j) % 4) * 0.4, x, y))
    mats = ('<material type="blinn" name="floor"><diffuse r="0.8" g="0.8" b="0.8"/><specular value="0"/></material>'
            '<material type="blinn" name="m0"><diffuse r="0.8" g="0.2" b="0.2"/><specular value="0.6"/><glossiness value="30"/></material>'
            '<material type="blinn" name="m1"><diffuse r="0.2" g="0.7" b="0.3"/><specular value="0.5"/><glossiness value="20"/><reflection value="0.4"/></material>'
            '<material type="blinn" name="m2"><diffuse r="0.1" g="0.1" b="0.1"/><specular value="0.8"/><glossiness value="50"/><refraction value="0.9" index="1.5"/></material>')
    lights = ('<light type="ambient" name="amb"><intensity value="0.1"/></light>'
              '<light type="point" name="p1"><intensity value="0.7"/><position x="20" y="-30" z="40"/></light>'
              '<light type="direct" name="d1"><intensity value="0.4"/><direction x="0.3" y="0.2" z="-1"/></light>')
    cam = '<camera><position x="0" y="-70" z="35"/><target x="0" y="0" z="0"/><up x="0" y="0" z="1"/><fov value="35"/><width value="800"/><height value="600"/></camera>'
    open(path, "w").write("<xml><scene>" + "".join(objs) + mats + lights + "</scene>" + cam + "</xml>")


@pytest.mark.parametrize("case", ["many_nodes", "object", "tower", "caustics"])
def test_instance_culling_changes_no_bit(tmp_path, case):
    """The cooperative kernel's sweeps skip a scene-graph node when the rays of a wave all miss its root-space bounds (the
    reference visits every node, src/scene/scene.cpp:35-74).  Same bits and cast counts with the test switched off (option
    "cs_cull" = 0) and as the per-lane kernel; the many-nodes scene (38 nodes: spheres, rotated mesh instances, a point and a
    direct light) also against the oracle."""
    from oracle import binding as oracle
    from qaray_amd import hip
    from qaray_amd.host import load_scene_blob
    ensure_assets()
    if case == "many_nodes":
        xml = str(tmp_path / "many_nodes.xml")
        _write_many_nodes_scene(xml)
        size, spp = (160, 120), 2
        blob = load_scene_blob(xml, size=size, asset_root=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scenes"))
    else:
        size, spp = (480, 270), 4
        blob = load_scene_blob({"object": "example_project7_object.xml", "tower": "trc_scene_tower.xml", "caustics": "example_project12_caustics_glossy.xml"}[case], size=size)
    w, h = size
    outs, cnts = {}, {}
    for mode in ("cull", "nocull", "own"):
        c = hip.Context(0)
        c.set_option("coop", 0 if mode == "own" else 1)
        c.set_option("cs_cull", 1 if mode == "cull" else 0)
        c.upload_scene(blob)
        if mode != "own":
            assert "qa_integrate_cs" in c.kernel_name(), c.kernel_name()
            if case in ("many_nodes", "object"):
                assert "CULL=1" in c.kernel_name(), c.kernel_name()   # (the variant that holds the test; "nocull" switches it off at run time)
        c.reset_counters()
        outs[mode] = c.render_region((0, 0, w, h), spp)
        cnts[mode] = c.counters()
        c.close()
    for mode in ("nocull", "own"):
        for a, b in zip(outs["cull"], outs[mode]):
            assert np.array_equal(bits(a), bits(b)), mode
        assert cnts["cull"] == cnts[mode], mode
    if case == "many_nodes":
        o = oracle.render(blob, (0, 0, w, h), spp)
        assert np.array_equal(bits(outs["cull"][1]), bits(o[1])) and np.array_equal(outs["cull"][2], o[2])
        assert (cnts["cull"]["casts_normal"], cnts["cull"]["casts_shadow"]) == (o[3].casts_normal, o[3].casts_shadow)
        assert rmse(np.nan_to_num(outs["cull"][0]), np.nan_to_num(o[0])) <= RMSE_TOL


@pytest.mark.parametrize("scene", ["trc_scene_tower.xml", "example_project7_object.xml"])
def test_exact_walks_alone_return_the_frame(scene):
    """Option "cs_force_exact": every closest-hit (bit 0) / shadow (bit 1) query of the cooperative kernel takes the path that
    ties, failed order checks and a full pool take - the exact sequential walks on the reference's trees.  Same bits as the
    normal frame and as the counting kernel (the reference's walk)."""
    from qaray_amd import hip
    from qaray_amd.host import load_scene_blob
    ensure_assets()
    w, h, spp = 240, 136, 2
    c = hip.Context(0)
    c.upload_scene(load_scene_blob(scene, size=(w, h)))
    ref = c.render_region((0, 0, w, h), spp, stats=True)
    for force in (0, 1, 2, 3):
        c.set_option("cs_force_exact", force)
        out = c.render_region((0, 0, w, h), spp)
        assert "qa_integrate_cs" in c.kernel_name(), c.kernel_name()
        for a, b in zip(out, ref):
            assert np.array_equal(bits(a), bits(b)), force
    c.close()


@pytest.mark.parametrize("scene", ["example_project3_sphere.xml", "example_project7_object.xml", "trc_scene_tower.xml", "trc_scene_xmas.xml",
                                   "example_project10_test.xml", "custom_textures.xml"])
def test_unwalked_shadow_rays_change_nothing(scene):
    """A light whose term is zero in every component before its shadow factor (the surface faces away from it: cosNL = max(0, N.L)
    = 0) adds that zero whether it is occluded or not: the product counts its shadow ray and does not walk it (directLight of
    qa_kernel.h, csShadowBatch / the AREA replay of qa_kernel_cs.h).  Option "walk_zero_terms" walks them as the reference does:
    same bits, same counters - LDS-resident, cooperative (textured, untextured, many lights, area lights) kernels; and the frame
    is the counting kernel's (the reference's walk, every ray cast)."""
    from qaray_amd import hip
    from qaray_amd.host import load_scene_blob
    ensure_assets()
    w, h, spp = 240, 136, 2
    c = hip.Context(0)
    c.upload_scene(load_scene_blob(scene, size=(w, h)))
    c.reset_counters()
    ref = c.render_region((0, 0, w, h), spp, stats=True)
    ref_cnt = c.counters()
    outs, cnts = {}, {}
    for walk in (0, 1):
        c.set_option("walk_zero_terms", walk)
        c.reset_counters()
        outs[walk] = c.render_region((0, 0, w, h), spp)
        cnts[walk] = c.counters()
    c.close()
    for a, b, r in zip(outs[0], outs[1], ref):
        assert np.array_equal(bits(a), bits(b))
        assert np.array_equal(bits(a), bits(r))
    for k in ("samples", "casts_normal", "casts_shadow", "pixels"):
        assert cnts[0][k] == cnts[1][k] == ref_cnt[k], k


@pytest.mark.parametrize("scene", ["example_project7_object.xml", "trc_scene_tower.xml", "trc_scene_xmas.xml", "example_project3_sphere.xml"])
def test_when_a_wave_starts_its_samples_changes_nothing(scene):
    """Option "sync_samples": a lane starts its next sample at once (0), when the whole wave is between samples (1), or - cooperative
    kernel - finished paths wait until n of the wave's have gathered (n >= 2; the per-scene default is 32 where it used to be 1).
    Every pixel owns its random-number stream and its Halton index, so the frame and the counters are the same whichever it is."""
    from qaray_amd import hip
    from qaray_amd.host import load_scene_blob
    ensure_assets()
    w, h, spp = 250, 130, 3   # (ragged tiles: lanes without a pixel sit beside waiting ones)
    c = hip.Context(0)
    c.upload_scene(load_scene_blob(scene, size=(w, h)))
    outs, cnts = {}, {}
    for v in (-1, 0, 1, 2, 8, 32, 64):
        c.set_option("sync_samples", v)
        c.reset_counters()
        outs[v] = c.render_region((0, 0, w, h), spp)
        cnts[v] = c.counters()
    c.close()
    for v in outs:
        for a, b in zip(outs[v], outs[-1]):
            assert np.array_equal(bits(a), bits(b)), v
        assert all(cnts[v][k] == cnts[-1][k] for k in ("samples", "casts_normal", "casts_shadow", "pixels")), v


@pytest.mark.parametrize("scene,coop", [("example_project12_box.xml", 1), ("example_project3_sphere.xml", 1), ("trc_mtl_glass.xml", 1),
                                        ("custom_photon.xml", 1), ("example_project10_test.xml", 0), ("example_project7_object.xml", 0),
                                        ("example_project7_object.xml", 1), ("example_project10_test.xml", 1), ("example_project9.xml", 1)])
def test_tiles_in_sample_chunks_change_nothing(scene, coop):
    """Options "chunk_spp" / "chunk_tail": the per-lane kernels and the cooperative kernel's textured variants hand a tile's samples out in chunks (a pixel's RNG state, sample count and
    running mean / variance wait in device memory between chunks; the tile's next chunk may be taken by any wave, which waits for the
    previous one to be published).  Same samples in the same order for every pixel: same bits, counters and sample counts - with
    adaptive sampling (pixels finish in different chunks), ragged tiles, far fewer tiles than waves (every hand-over is waited for)
    and frames of many tiles; LDS-resident, global-memory (coop = 0), textured and area-light variants; and the counting
    kernel of a chunked frame equals the unchunked one."""
    from qaray_amd import hip
    from qaray_amd.host import load_scene_blob
    ensure_assets()
    c = hip.Context(0)
    c.set_option("coop", coop)
    for (w, h, spp_min, spp_max) in ((250, 130, 3, 40), (640, 360, 24, 24)):
        c.upload_scene(load_scene_blob(scene, size=(w, h)))
        outs, cnts = {}, {}
        for chunk, tail in ((0, 0), (-1, 0), (16, 8), (1, 1), (8, 16), (23, 5)):
            c.set_option("chunk_spp", chunk)
            c.set_option("chunk_tail", tail)
            c.reset_counters()
            outs[(chunk, tail)] = c.render_region((0, 0, w, h), spp_min, spp_max=spp_max)
            cnts[(chunk, tail)] = c.counters()
            assert ("qa_integrate_cs<" in c.kernel_name()) == (coop == 1 and scene.startswith("example_project") and "sphere" not in scene and "box" not in scene), c.kernel_name()
        c.set_option("chunk_spp", 7)
        c.set_option("chunk_tail", 3)
        stats = c.render_region((0, 0, w, h), spp_min, spp_max=spp_max, stats=True)
        for k in outs:
            for a, b in zip(outs[k], outs[(0, 0)]):
                assert np.array_equal(bits(a), bits(b)), k
            assert all(cnts[k][n] == cnts[(0, 0)][n] for n in ("samples", "casts_normal", "casts_shadow", "pixels")), k
        for a, b in zip(stats, outs[(0, 0)]):
            assert np.array_equal(bits(a), bits(b))
    c.close()


def _write_area_lights_scene(path):
    """Two global-memory mesh instances, a sphere and a floor under an area point light, an area spot light, a plain point light and a
    direct light: every branch of the AREA variants' light replay."""
    open(path, "w").write(
        '<xml><scene>'
        '<object type="plane" name="floor" material="white"><scale value="40"/></object>'
        '<object type="obj" name="teapot-high.obj" material="red"><scale value="0.6"/><rotate angle="-40" z="1"/><translate x="-4" y="0" z="0"/></object>'
        '<object type="obj" name="teapot-high.obj" material="glass"><scale value="0.4"/><rotate angle="70" z="1"/><translate x="6" y="-3" z="0"/></object>'
        '<object type="sphere" name="ball" material="glossy"><scale value="2.5"/><translate x="2" y="6" z="2.5"/></object>'
        '<material type="blinn" name="white"><diffuse r="0.8" g="0.8" b="0.8"/><specular value="0"/></material>'
        '<material type="blinn" name="red"><diffuse r="0.8" g="0.2" b="0.2"/><specular value="0.5"/><glossiness value="30"/></material>'
        '<material type="blinn" name="glass"><diffuse value="0.05"/><specular value="0.8"/><glossiness value="60"/><refraction value="0.9" index="1.5"/></material>'
        '<material type="blinn" name="glossy"><diffuse r="0.2" g="0.3" b="0.8"/><specular value="0.7"/><glossiness value="20"/><reflection value="0.5" glossiness="0.05"/></material>'
        '<light type="ambient" name="amb"><intensity value="0.1"/></light>'
        '<light type="point" name="area"><intensity value="60"/><position x="0" y="-8" z="14"/><size value="2.5"/></light>'
        '<light type="spot" name="areaspot"><intensity value="90"/><position x="-10" y="6" z="12"/><rotation angle="50" x="1" y="1" z="0"/><angle value="55"/><blend value="0.3"/><size value="1.2"/></light>'
        '<light type="point" name="hard"><intensity value="25"/><position x="9" y="9" z="10"/></light>'
        '<light type="direct" name="sun"><intensity value="0.3"/><direction x="0.4" y="0.3" z="-1"/></light>'
        '</scene><camera><position x="0" y="-32" z="14"/><target x="0" y="0" z="2"/><up x="0" y="0" z="1"/><fov value="40"/><width value="800"/><height value="600"/></camera></xml>')


@pytest.mark.parametrize("case", ["project10_test", "every_light_kind"])
def test_cooperative_area_lights_equal_the_per_lane_kernel(tmp_path, case):
    """Area lights over global-memory meshes: qa_integrate_cs<..., AREA=1> logs every lit hit and evaluates all lights when the path
    has ended, the whole wave walking an area light's 16 - 64 sample rays four per lane at a time from the pool.  Same bits, sample
    counts, depth and cast counters as qa_integrate's AREA variant (every lane walks its own rays; option "coop" = 0), and the
    oracle's casts / depth / radiance."""
    from oracle import binding as oracle
    from qaray_amd import hip
    from qaray_amd.host import load_scene_blob
    ensure_assets()
    if case == "project10_test":
        size, spp = (200, 150), 2
        blob = load_scene_blob("example_project10_test.xml", size=size)
    else:
        xml = str(tmp_path / "area_lights.xml")
        _write_area_lights_scene(xml)
        size, spp = (160, 120), 2
        blob = load_scene_blob(xml, size=size, asset_root=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scenes"))
    w, h = size
    outs, cnts = {}, {}
    for mode in ("coop", "own"):
        c = hip.Context(0)
        c.set_option("coop", 1 if mode == "coop" else 0)
        c.upload_scene(blob)
        assert ("qa_integrate_cs" in c.kernel_name() and "AREA=1" in c.kernel_name()) == (mode == "coop"), c.kernel_name()
        c.reset_counters()
        outs[mode] = c.render_region((0, 0, w, h), spp)
        cnts[mode] = c.counters()
        c.close()
    for a, b in zip(outs["coop"], outs["own"]):
        assert np.array_equal(bits(a), bits(b))
    assert cnts["coop"] == cnts["own"]
    o = oracle.render(blob, (0, 0, w, h), spp)
    assert np.array_equal(bits(outs["coop"][1]), bits(o[1])) and np.array_equal(outs["coop"][2], o[2])
    assert (cnts["coop"]["casts_normal"], cnts["coop"]["casts_shadow"]) == (o[3].casts_normal, o[3].casts_shadow)
    assert rmse(np.nan_to_num(outs["coop"][0]), np.nan_to_num(o[0])) <= RMSE_TOL


def test_many_lights_on_a_textured_scene(tmp_path):
    """project7_object with four more point lights (six shadow-casting lights: two batches): the textured MANY variant parks the surface
    in the slab between the batches.  Same bits and counters as the per-lane kernel."""
    from qaray_amd import hip
    from qaray_amd.host import load_scene_blob
    ensure_assets()
    scenes = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scenes")
    txt = open(os.path.join(scenes, "example_project7_object.xml")).read()
    lights = "".join('<light type="point" name="p%d"><intensity value="%g"/><position x="%g" y="%g" z="%g"/></light>\n' %
                     (i, 0.3 + 0.1 * i, 30 * np.cos(1.1 * i), -20 + 25 * np.sin(0.9 * i), 25 + 3 * i) for i in range(4))
    assert '<light type="direct"' in txt
    txt = txt.replace('<light type="direct"', lights + '<light type="direct"', 1)
    xml = str(tmp_path / "object_six_lights.xml")
    open(xml, "w").write(txt)
    w, h, spp = 240, 136, 2
    blob = load_scene_blob(xml, size=(w, h), asset_root=scenes)
    outs, cnts = {}, {}
    for mode in ("coop", "own"):
        c = hip.Context(0)
        c.set_option("coop", 1 if mode == "coop" else 0)
        c.upload_scene(blob)
        assert ("qa_integrate_cs" in c.kernel_name() and "TEX=1" in c.kernel_name()) == (mode == "coop"), c.kernel_name()
        c.reset_counters()
        outs[mode] = c.render_region((0, 0, w, h), spp)
        cnts[mode] = c.counters()
        c.close()
    for a, b in zip(outs["coop"], outs["own"]):
        assert np.array_equal(bits(a), bits(b))
    assert cnts["coop"] == cnts["own"]
    assert cnts["coop"]["casts_shadow"] > 0


def test_exact_repeat_is_exercised_and_invisible(ctx, tmp_path):
    """Coincident sheets: the staged integrator must send rays to wf_redo (ties / failed order checks) - and the frame
    must not show it."""
    from qaray_amd.host import load_scene_blob
    xml = _write_big_fuzz_scene(str(tmp_path), np.random.default_rng(7), "sheets")
    blob = load_scene_blob(xml, size=(256, 192), asset_root=str(tmp_path))
    ctx.upload_scene(blob)
    ctx.set_pipeline("staged")
    ctx.reset_counters()
    a = ctx.render_region((0, 0, 256, 192), 4)
    st = ctx.staged_stats()
    assert st["rays_redone"] > 0, st
    ctx.set_pipeline("mega")
    b = ctx.render_region((0, 0, 256, 192), 4, stats=True)
    for x, y in zip(a, b):
        assert np.array_equal(bits(x), bits(y))


def test_staged_strips_partition_and_determinism(ctx):
    """Round-robin 8-row strips (the multi-GPU partition) through the staged integrator == the whole frame, twice."""
    import torch
    from qaray_amd import distributed as qd, hip
    from qaray_amd.host import load_scene_blob
    ensure_assets()
    W, H, spp = 480, 270, 4
    ctx.upload_scene(load_scene_blob("trc_scene_tower.xml", size=(W, H)))
    ctx.set_pipeline("staged")
    full = ctx.render_region((0, 0, W, H), spp)
    again = ctx.render_region((0, 0, W, H), spp)
    assert all(np.array_equal(bits(a), bits(b)) for a, b in zip(full, again))
    crop = ctx.render_region((101, 33, 322, 201), spp)[0]
    assert np.array_equal(bits(crop), bits(full[0][33:201, 101:322]))
    dev = torch.device("cuda", 0)
    world = 3
    asm = np.zeros((H, W, 3), np.float32)
    for r in range(world):
        n = hip.strip_count(0, H, r, world) * 8
        rgb = torch.zeros((n, W, 3), dtype=torch.float32, device=dev)
        d = torch.zeros((n, W), dtype=torch.float32, device=dev)
        ns = torch.zeros((n, W), dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        ctx.render_strips_device((0, 0, W, H), r, world, spp, rgb, d, ns)
        ctx.synchronize()
        qd.place_strips(asm, rgb.cpu().numpy(), H, world, r)
    assert np.array_equal(bits(asm), bits(full[0]))


@pytest.mark.parametrize("scene", ["trc_scene_tower.xml", "example_project7_object.xml"])
def test_staged_tile_groups_on_streams_equal_the_megakernel(ctx, scene):
    """Frames of more than ~4000 tiles are dealt to three groups of interleaved tiles, each driving its own chain of stage
    kernels on its own stream (qa_wf.hip): same bits and counters as the megakernel, twice in a row."""
    from qaray_amd.host import load_scene_blob
    ensure_assets()
    W, H, spp = 1280, 720, 6
    ctx.upload_scene(load_scene_blob(scene, size=(W, H)))
    ctx.set_pipeline("mega")
    ctx.reset_counters()
    ref = ctx.render_region((0, 0, W, H), spp)
    cref = ctx.counters()
    ctx.set_option("staged_groups", 4)
    ctx.set_pipeline("staged")
    assert "4 tile groups" in ctx.kernel_name()
    try:
        for _ in range(2):
            ctx.reset_counters()
            out = ctx.render_region((0, 0, W, H), spp)
            c = ctx.counters()
            for a, b in zip(out, ref):
                assert np.array_equal(bits(a), bits(b))
            assert c == cref
    finally:
        ctx.set_option("staged_groups", 1)


def test_auto_is_the_megakernel_and_staged_runs_on_request_only():
    """QA_PIPE_AUTO keeps the megakernel (round 2's timed probe between the integrators is gone: the staged integrator
    lost on every scene and stays as a cross-check); the name reported after a frame is the kernel that frame ran on."""
    from qaray_amd import hip
    from qaray_amd.host import load_scene_blob
    ensure_assets()
    W, H, spp = 640, 360, 8
    blob = load_scene_blob("trc_scene_tower.xml", size=(W, H))
    c = hip.Context(0)
    c.upload_scene(blob)
    c.set_pipeline("auto")
    assert "qa_integrate_cs" in c.kernel_name()
    a = c.render_region((0, 0, W, H), spp)
    assert "qa_integrate_cs" in c.kernel_name() and "probe" not in c.kernel_name()
    c.set_option("coop", 0)
    assert c.kernel_name().startswith("qa_integrate<RES=0")
    b = c.render_region((0, 0, W, H), spp)
    assert c.kernel_name().startswith("qa_integrate<RES=0")
    c.set_pipeline("staged")
    assert c.kernel_name().startswith("staged") and "1 tile group" in c.kernel_name()
    s_ = c.render_region((0, 0, W, H), spp)
    assert c.kernel_name().startswith("staged")
    # a frame the staged integrator refuses (counting kernel) is named after what really ran
    c.render_region((0, 0, 64, 64), 1, stats=True)
    assert c.kernel_name().startswith("qa_integrate<") and "counting variant" in c.kernel_name()
    c.close()
    for x, y, z in zip(a, b, s_):
        assert np.array_equal(bits(x), bits(y)) and np.array_equal(bits(x), bits(z))


def test_many_lights_are_pooled_in_batches(tmp_path):
    """33 non-ambient lights over a global-memory mesh: the cooperative kernel pools their shadow queries four lights at a
    time, the surface waiting in a global slab between batches (round 2's kernel kept one occlusion bit per light in a 32-bit
    mask: lights 32.. aliased).  Same bits as the per-lane kernel, depth / cast counts as the oracle."""
    from oracle import binding as oracle
    from qaray_amd import hip
    from qaray_amd.host import load_scene_blob
    xml = _write_big_fuzz_scene(str(tmp_path), np.random.default_rng(5), "shell")
    txt = open(xml).read()
    lights = "".join('<light type="point" name="p%d"><intensity value="6"/><position x="%g" y="%g" z="%g"/></light>\n' %
                     (i, 9 * np.cos(0.7 * i), -9 + 3 * np.sin(1.3 * i), 6 + (i % 5)) for i in range(32))
    txt = txt.replace('<light type="direct"', lights + '<light type="direct"')
    open(xml, "w").write(txt)
    w, h, spp = 96, 72, 2
    blob = load_scene_blob(xml, size=(w, h), asset_root=str(tmp_path))
    outs, cnts = {}, {}
    for mode in ("coop", "own"):
        c = hip.Context(0)
        c.set_option("coop", 1 if mode == "coop" else 0)
        c.upload_scene(blob)
        assert ("qa_integrate_cs" in c.kernel_name()) == (mode == "coop"), c.kernel_name()
        c.reset_counters()
        outs[mode] = c.render_region((0, 0, w, h), spp)
        cnts[mode] = c.counters()
        c.close()
    for a, b in zip(outs["coop"], outs["own"]):
        assert np.array_equal(bits(a), bits(b))
    assert cnts["coop"] == cnts["own"]
    o = oracle.render(blob, (0, 0, w, h), spp)
    assert np.array_equal(bits(outs["coop"][1]), bits(o[1])) and np.array_equal(outs["coop"][2], o[2])
    assert (cnts["coop"]["casts_normal"], cnts["coop"]["casts_shadow"]) == (o[3].casts_normal, o[3].casts_shadow)
    assert rmse(np.nan_to_num(outs["coop"][0]), np.nan_to_num(o[0])) <= RMSE_TOL


def test_stop_request_ends_a_staged_frame(ctx):
    from qaray_amd.host import load_scene_blob
    ensure_assets()
    ctx.upload_scene(load_scene_blob("trc_scene_tower.xml", size=(160, 90)))
    ctx.set_pipeline("staged")
    ctx.request_stop()
    rgb, depth, ns = ctx.render_region((0, 0, 160, 90), 4)
    assert (ns == 0).all()
    ctx.clear_stop()
    rgb, depth, ns = ctx.render_region((0, 0, 160, 90), 4)
    assert (ns == 4).all()
