"""Staged integrator: step budget sweep at a spp where the steady state dominates.  python tools/gpu_budget_sweep.py c3|c5|c4 spp b1 b2 ..."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
subprocess.run([sys.executable, os.path.join(ROOT, "scenes", "gen_assets.py")], check=True, stdout=subprocess.DEVNULL)
from qaray_amd.host import load_scene_blob
from qaray_amd import hip
CASES = {"c3": ("example_project7_object.xml", (1920, 1080)), "c4": ("example_project12_caustics_glossy.xml", (3840, 2160)),
         "c5": ("trc_scene_tower.xml", (3840, 2160))}
scene, size = CASES[sys.argv[1]]
spp = int(sys.argv[2])
blob = load_scene_blob(scene, size=size)
for bud in sys.argv[3:]:
    if bud == "mega":
        os.environ["QA_PIPELINE"] = "mega"
    else:
        os.environ.pop("QA_PIPELINE", None)
        os.environ["QA_WF_BUDGET"] = bud
    ctx = hip.Context(0)
    ctx.upload_scene(blob)
    ctx.render_region((0, 0, 64, 64), 1)
    ctx.reset_kernel_time(); ctx.reset_counters()
    ctx.render_region((0, 0) + size, spp)
    ms, _ = ctx.kernel_time(); c = ctx.counters()
    line = f"{sys.argv[1]} {spp} spp budget {bud}: {ms:.1f} ms, {c['samples'] / ms * 1e-3:.1f} Msamples/s"
    if "staged" in ctx.kernel_name():
        st = ctx.staged_stats()
        line += f", passes {st['passes']}, suspended {st['jobs_suspended']}, lane util {st['lane_utilisation']:.3f}, redone {st['rays_redone']}"
    print(line, flush=True)
    ctx.close()
