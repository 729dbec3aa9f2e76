"""Which inputs notice the slack constants of the own search trees?  For QA_DEBUG_SLACK_SCALE in argv (default 1 0.5 0.25 0):
default kernel vs counting kernel (reference-tree walk) on the C2 frame, the small fuzz scenes (LDS-resident own tree) and
the big fuzz meshes + C3 / C5 crops (4-wide tree, megakernel and staged)."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import ensure_assets
from test_gpu_parity import _write_fuzz_scene
from test_gpu_staged import _write_big_fuzz_scene
from qaray_amd.host import load_scene_blob
from qaray_amd import hip
ensure_assets()
scales = sys.argv[1:] or ["1", "0.5", "0.25", "0"]
ctx = hip.Context(0)

def differs(blob, w, h, spp, pipelines=("mega",)):
    out = []
    for p in pipelines:
        ctx.set_pipeline(p)
        ctx.upload_scene(blob)
        a = ctx.render_region((0, 0, w, h), spp)
        b = ctx.render_region((0, 0, w, h), spp, stats=True)
        out.append(int((a[0].view(np.uint32) != b[0].view(np.uint32)).any(axis=2).sum() + (a[1].view(np.uint32) != b[1].view(np.uint32)).sum()))
    return out

tmp = tempfile.mkdtemp()
cases = []
cases.append(("c2 1080p@128", load_scene_blob("example_project12_box.xml", size=(1920, 1080)), 1920, 1080, 128, ("mega",)))
for kind in ("soup", "sheets", "needles", "duplicates"):
    for seed in (0, 1, 2):
        d = os.path.join(tmp, f"s_{kind}{seed}"); os.makedirs(d)
        rng = np.random.default_rng(10 * seed + {"soup": 1, "sheets": 2, "needles": 3, "duplicates": 4}[kind])
        xml = _write_fuzz_scene(d, rng, kind)
        cases.append((f"small {kind} {seed}", load_scene_blob(xml, size=(192, 144), asset_root=d), 192, 144, 8, ("mega",)))
for kind in ("sheets", "shell", "soup"):
    for seed in (0, 1):
        d = os.path.join(tmp, f"b_{kind}{seed}"); os.makedirs(d)
        rng = np.random.default_rng(100 * seed + len(kind))
        xml = _write_big_fuzz_scene(d, rng, kind)
        cases.append((f"big {kind} {seed}", load_scene_blob(xml, size=(256, 192), asset_root=d), 256, 192, 8, ("mega", "staged")))
cases.append(("c3 480x270@16", load_scene_blob("example_project7_object.xml", size=(480, 270)), 480, 270, 16, ("mega", "staged")))
cases.append(("c5 960x540@16", load_scene_blob("trc_scene_tower.xml", size=(960, 540)), 960, 540, 16, ("mega", "staged")))
for sc in scales:
    os.environ["QA_DEBUG_SLACK_SCALE"] = sc
    line = []
    for name, blob, w, h, spp, pipes in cases:
        d = differs(blob, w, h, spp, pipes)
        if any(d): line.append(f"{name}: {d}")
    print(f"scale {sc}: differing pixels (rgb + depth) -> {line if line else 'none'}", flush=True)
