#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz from the REAL reference (oracle/_ref/ref_harness, built by
oracle/Makefile from /root/reference).  Only runs in the dev container; the fixtures it writes are
data (float32 radiance / depth / sample counts + cast counters) and travel with the repo.

Every fixture records: scene file (under scenes/), image size, crop, spp range, max bounce, seed.
The per-pixel RNG stream is include/qa_seed.h's qa_pixel_seed(seed, j*W+i).
"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
HARNESS = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
SCENES = os.path.join(ROOT, "scenes")

# name, scene, (W,H), crop or None, spp_min, spp_max, bounce
CASES = [
    ("c1_sphere_256x256_1spp", "example_project3_sphere.xml", (256, 256), None, 1, 1, 5),   # BASELINE config 0
    ("c2_box_64x64_4spp", "example_project12_box.xml", (64, 64), None, 4, 4, 5),
    ("c2_box_1080p_crop_8spp", "example_project12_box.xml", (1920, 1080), (900, 500, 948, 532), 8, 8, 5),
    ("c2_box_1080p_edge_crop_64spp", "example_project12_box.xml", (1920, 1080), (1896, 1064, 1920, 1080), 64, 64, 5),
    ("blinn_48x36_4spp", "example_project2_blinn.xml", (48, 36), None, 4, 4, 5),
    ("box3_48x36_4spp", "example_project3_box.xml", (48, 36), None, 4, 4, 5),
    ("project4_48x36_4spp", "example_project4.xml", (48, 36), None, 4, 4, 5),
    ("glass_48x36_8spp", "trc_mtl_glass.xml", (48, 36), None, 8, 8, 5),
    ("glossy_48x36_8spp", "trc_mtl_glossy.xml", (48, 36), None, 8, 8, 5),
    ("coffee_48x36_4spp_bounce2", "trc_mtl_coffee.xml", (48, 36), None, 4, 4, 2),
    ("sphere_adaptive_64x48_4to32spp", "example_project3_sphere.xml", (64, 48), None, 4, 32, 5),
    ("textures_80x60_2spp", "custom_textures.xml", (80, 60), None, 2, 2, 5),
    ("softshadow_dof_60x45_2spp", "custom_softshadow.xml", (60, 45), None, 2, 2, 5),
    # BASELINE configs 2-4 on the synthetic stand-in assets (scenes/gen_assets.py), crops at full size
    ("c3_object_1080p_crop_2spp", "example_project7_object.xml", (1920, 1080), (840, 560, 888, 592), 2, 2, 5),
    ("c4_caustics_4k_crop_4spp", "example_project12_caustics_glossy.xml", (3840, 2160), (1900, 1300, 1948, 1332), 4, 4, 5),
    ("c5_tower_4k_crop_2spp", "trc_scene_tower.xml", (3840, 2160), (1800, 1000, 1848, 1032), 2, 2, 5),
    # the same configs at the spp BASELINE.json quotes them on (sample indices, Halton rows, RNG stream depth and the
    # running mean / variance at large n are only exercised there)
    ("c2_box_1080p_crop_512spp", "example_project12_box.xml", (1920, 1080), (952, 532, 968, 548), 512, 512, 5),
    ("c3_object_1080p_crop_256spp", "example_project7_object.xml", (1920, 1080), (860, 570, 868, 578), 256, 256, 5),
    ("c4_caustics_4k_crop_1024spp", "example_project12_caustics_glossy.xml", (3840, 2160), (1920, 1310, 1928, 1318), 1024, 1024, 5),
    ("c5_tower_4k_crop_2048spp", "trc_scene_tower.xml", (3840, 2160), (1820, 1010, 1828, 1018), 2048, 2048, 5),
]
SEED = 0x51A7A7

# the reference's 8-bit products of whole frames (tests/golden/eightbit/): the tail of PixelRender with the reference's
# LinearToSRGB, and its FrameBuffer's z-buffer / sample-count visualisations (src/renderers/renderer.cpp:347-365,
# src/fb/framebuffer.cpp:62-107): name, scene, (W,H), spp_min, spp_max, use sRGB
EIGHTBIT_CASES = [
    ("fb_box_96x64_4spp_srgb", "example_project12_box.xml", (96, 64), 4, 4, 1),
    ("fb_sphere_adaptive_64x48_linear", "example_project3_sphere.xml", (64, 48), 4, 32, 0),
    ("fb_project4_background_48x36_srgb", "example_project4.xml", (48, 36), 2, 2, 1),
]

# -use-photon-map cases (tests/golden/photon/): name, scene, (W,H), spp, (size, bounce, radius) of the photon map
# and of the caustics map.  One RNG stream per emission, include/qa_photon.h.
PHOTON_CASES = [
    ("pm_glass_48x36_2spp", "trc_mtl_glass.xml", (48, 36), 2, (2000, 20, 2.0), (300, 20, 3.0)),
    ("pm_custom_64x48_2spp", "custom_photon.xml", (64, 48), 2, (5000, 20, 0.2), (800, 20, 1.0)),   # default radii
    ("pm_glossy_shortpaths_48x36_2spp", "trc_mtl_glossy.xml", (48, 36), 2, (1500, 4, 1.0), (200, 6, 2.0)),
]
PHOTON_DTYPE = np.dtype([("pos", np.float32, 3), ("power", np.float32), ("rgb", np.uint8, 3), ("plane_dirz", np.uint8),
                         ("dirx", np.int16), ("diry", np.int16)])   # cy::PhotonMap::Photon, 24 bytes


def main():
    subprocess.run([sys.executable, os.path.join(SCENES, "gen_assets.py")], check=True, stdout=subprocess.DEVNULL)
    if not os.path.exists(HARNESS):
        sys.exit(f"{HARNESS} missing: run `make -C oracle ref` in the dev container")
    only = set(sys.argv[1:])     # optional: names of the fixtures to (re)generate
    for name, scene, (w, h), crop, smin, smax, bounce in CASES:
        if only and name not in only:
            continue
        with tempfile.TemporaryDirectory() as td:
            out = os.path.join(td, "g")
            cmd = [HARNESS, scene, "--size", str(w), str(h), "--spp-min", str(smin), "--spp-max", str(smax),
                   "--bounce", str(bounce), "--seed", str(SEED), "--threads", "8", "--out", out]
            if crop:
                cmd += ["--crop"] + [str(c) for c in crop]
            subprocess.run(cmd, cwd=SCENES, check=True, stdout=subprocess.DEVNULL)
            meta = json.load(open(out + ".json"))
            x0, y0, x1, y1 = meta["crop"]
            ch, cw = y1 - y0, x1 - x0
            rgb = np.fromfile(out + ".rgb.f32", np.float32).reshape(ch, cw, 3)
            depth = np.fromfile(out + ".depth.f32", np.float32).reshape(ch, cw)
            ns = np.fromfile(out + ".ns.u32", np.uint32).reshape(ch, cw)
        info = dict(scene=scene, width=w, height=h, crop=[x0, y0, x1, y1], spp_min=smin, spp_max=smax,
                    bounce=bounce, seed=SEED, samples=meta["samples"], casts_normal=meta["casts_normal"],
                    casts_shadow=meta["casts_shadow"], producer="oracle/_ref/ref_harness (reference code)")
        np.savez_compressed(os.path.join(HERE, name + ".npz"), rgb=rgb, depth=depth, ns=ns,
                            meta=np.frombuffer(json.dumps(info).encode(), dtype=np.uint8))
        print(f"{name}: {cw}x{ch} samples={meta['samples']} casts={meta['casts_normal']}+{meta['casts_shadow']}")
    os.makedirs(os.path.join(HERE, "eightbit"), exist_ok=True)
    for name, scene, (w, h), smin, smax, srgb in EIGHTBIT_CASES:
        if only and name not in only:
            continue
        with tempfile.TemporaryDirectory() as td:
            out = os.path.join(td, "g")
            cmd = [HARNESS, scene, "--size", str(w), str(h), "--spp-min", str(smin), "--spp-max", str(smax), "--seed", str(SEED),
                   "--threads", "8", "--out", out, "--eight-bit", str(srgb)]
            subprocess.run(cmd, cwd=SCENES, check=True, stdout=subprocess.DEVNULL)
            arrays = dict(color=np.fromfile(out + ".color.u8", np.uint8).reshape(h, w, 3), zimg=np.fromfile(out + ".zimg.u8", np.uint8).reshape(h, w),
                          count=np.fromfile(out + ".count.u8", np.uint8).reshape(h, w), countimg=np.fromfile(out + ".countimg.u8", np.uint8).reshape(h, w),
                          rgb=np.fromfile(out + ".rgb.f32", np.float32).reshape(h, w, 3), depth=np.fromfile(out + ".depth.f32", np.float32).reshape(h, w),
                          ns=np.fromfile(out + ".ns.u32", np.uint32).reshape(h, w))
        info = dict(scene=scene, width=w, height=h, spp_min=smin, spp_max=smax, bounce=5, seed=SEED, srgb=srgb,
                    producer="oracle/_ref/ref_harness --eight-bit (the reference's LinearToSRGB and FrameBuffer code)")
        np.savez_compressed(os.path.join(HERE, "eightbit", name + ".npz"), meta=np.frombuffer(json.dumps(info).encode(), dtype=np.uint8), **arrays)
        print(f"{name}: color {arrays['color'].mean():.2f} zimg {arrays['zimg'].mean():.2f} countimg {arrays['countimg'].mean():.2f}")
    os.makedirs(os.path.join(HERE, "photon"), exist_ok=True)
    for name, scene, (w, h), spp, pm, cm in PHOTON_CASES:
        if only and name not in only:
            continue
        with tempfile.TemporaryDirectory() as td:
            out = os.path.join(td, "g")
            cmd = [HARNESS, scene, "--size", str(w), str(h), "--spp", str(spp), "--seed", str(SEED), "--threads", "8",
                   "--out", out, "--photon-map", str(pm[0]), str(cm[0]), "--photon-bounce", str(pm[1]), str(cm[1]),
                   "--photon-radius", repr(pm[2]), repr(cm[2])]
            subprocess.run(cmd, cwd=SCENES, check=True, stdout=subprocess.DEVNULL)
            meta = json.load(open(out + ".json"))
            rgb = np.fromfile(out + ".rgb.f32", np.float32).reshape(h, w, 3)
            depth = np.fromfile(out + ".depth.f32", np.float32).reshape(h, w)
            ns = np.fromfile(out + ".ns.u32", np.uint32).reshape(h, w)
            photon = np.fromfile(out + ".photonmap.bin", PHOTON_DTYPE)
            caustics = np.fromfile(out + ".caustics.bin", PHOTON_DTYPE)
        info = dict(scene=scene, width=w, height=h, crop=[0, 0, w, h], spp_min=spp, spp_max=spp, bounce=5, seed=SEED,
                    samples=meta["samples"], casts_normal=meta["casts_normal"], casts_shadow=meta["casts_shadow"],
                    photon=list(pm), caustics=list(cm), emitted=meta["photon_emitted"], emissions=meta["photon_emissions"],
                    producer="oracle/_ref/ref_harness --photon-map (reference code, one RNG stream per emission)")
        np.savez_compressed(os.path.join(HERE, "photon", name + ".npz"), rgb=rgb, depth=depth, ns=ns, photon=photon,
                            caustics=caustics, meta=np.frombuffer(json.dumps(info).encode(), dtype=np.uint8))
        print(f"{name}: photons {len(photon)}+{len(caustics)} emitted {meta['photon_emitted']} emissions {meta['photon_emissions']}")


if __name__ == "__main__":
    main()
