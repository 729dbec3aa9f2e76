"""A/B of integrator variants on the non-resident BASELINE scenes (C3 - C5): every variant renders a small frame
(compared bitwise with the first variant) and a full-size frame (throughput).
   python tools/gpu_ab.py [--spp N] [--cases c3,c5] name:ENV=VAL,ENV=VAL ...      e.g.  ref:QA_PIPELINE=mega,QA_WIDE=0 wide:QA_PIPELINE=mega"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
subprocess.run([sys.executable, os.path.join(ROOT, "scenes", "gen_assets.py")], check=True, stdout=subprocess.DEVNULL)
from qaray_amd.host import load_scene_blob
from qaray_amd import hip

CASES = {"c3": ("example_project7_object.xml", (1920, 1080), 16), "c4": ("example_project12_caustics_glossy.xml", (3840, 2160), 16),
         "c5": ("trc_scene_tower.xml", (3840, 2160), 8)}
args = sys.argv[1:]
spp_override, cases = None, list(CASES)
while args and args[0].startswith("--"):
    if args[0] == "--spp": spp_override = int(args[1]); args = args[2:]
    elif args[0] == "--cases": cases = args[1].split(","); args = args[2:]
variants = []
for a in args:
    name, _, envs = a.partition(":")
    variants.append((name, dict(e.split("=") for e in envs.split(",") if e)))
KNOBS = ("QA_HIP_LIB", "QA_PIPELINE", "QA_WIDE", "QA_WF_BUDGET", "QA_WF_BLOCKS", "QA_SYNC", "QA_WF_STACK", "QA_WF_REFILL", "QA_WF_GROUPS", "QA_WF_GATE", "QA_WIDE_LEAF", "QA_WF_REDO_ASYNC", "QA_WF_LOGIC_BLOCKS", "QA_WF_RESERVE", "QA_COOP")
res = {}
for name, env in variants:
    for k in KNOBS: os.environ.pop(k, None)
    os.environ.update(env)
    ctx = hip.Context(0)
    # the product library reads no QA_* variable (only -DQA_DEV_KNOBS builds do): the two common ones go through the API
    if "QA_PIPELINE" in env: ctx.set_pipeline(env["QA_PIPELINE"])
    if "QA_COOP" in env: ctx.set_option("coop", int(env["QA_COOP"]))
    for tag in cases:
        scene, size, spp = CASES[tag]
        spp = spp_override or spp
        small = (size[0] // 8, size[1] // 8)
        ctx.upload_scene(load_scene_blob(scene, size=small))
        ctx.reset_counters()
        out = ctx.render_region((0, 0) + small, 4)
        res[(name, tag)] = (out, ctx.counters())
        import hashlib
        sha = hashlib.sha1(b"".join(np.ascontiguousarray(x).tobytes() for x in out)).hexdigest()[:12]
        ctx.upload_scene(load_scene_blob(scene, size=size))
        ctx.render_region((0, 0, 64, 64), 1)
        ctx.reset_kernel_time(); ctx.reset_counters()
        ctx.render_region((0, 0) + size, spp)
        ms, _ = ctx.kernel_time(); c = ctx.counters()
        casts = c["casts_normal"] + c["casts_shadow"]
        line = (f"{name:10s} {tag}: {size[0]}x{size[1]} @ {spp} spp: {ms:8.1f} ms, {c['samples'] / ms * 1e-3:7.1f} Msamples/s, "
                f"{casts / ms * 1e-6:.2f} Gcasts/s  [{ctx.kernel_name()[:24]}] small-frame sha1 {sha}")
        if "staged" in ctx.kernel_name():
            st = ctx.staged_stats()
            line += (f" passes {st['passes']} jobs {st['jobs_done']} steps/job {(st['node_steps'] + st['leaf_steps']) / max(1, st['jobs_done']):.1f} "
                     f"tri/job {st['tri_tests'] / max(1, st['jobs_done']):.1f} util {st['lane_utilisation']:.3f} susp {st['jobs_suspended']} redone {st['rays_redone']}")
        print(line, flush=True)
    ctx.close()
base = variants[0][0]
for name, _ in variants[1:]:
    for tag in cases:
        (a, ca), (b, cb) = res[(base, tag)], res[(name, tag)]
        same = [bool(np.array_equal(x.view(np.uint32), y.view(np.uint32))) for x, y in zip(a, b)]
        print(f"{name} vs {base} {tag}: rgb/depth/ns bit-equal {same}, max |drgb| {np.abs(a[0] - b[0]).max():.3e}, counters equal "
              f"{all(ca[k] == cb[k] for k in ('samples', 'casts_normal', 'casts_shadow', 'pixels'))}", flush=True)
