#!/usr/bin/env python3
"""bench.py - headline benchmark of the MI355X qaray hot path.

Metric (BASELINE.json): Msamples/s (1 sample = 1 camera path) and wall-clock to 1080p@512spp, with
the kernel's fraction of the HBM roofline, at 1/2/4/8 GPUs.

A "step" renders ONE frame of inputs/example_project12_box.xml (Cornell box, BASELINE config[1]) at
512 spp.  N=1: 1920x1080.  N>1 (weak scaling in resolution, per-GPU pixel count fixed): the same view
at 16:9 with N x 2.07 Mpixel (N=4 is exactly 3840x2160), 8-row strips dealt round-robin to the ranks,
scene blob broadcast from rank 0 over RCCL, float radiance strips gathered to rank 0 inside the step.
Inputs (scene tables) are resident in HBM before the timed region.

--config c2|c3|c4|c5 selects a BASELINE.json config by name (scene, frame, spp).  c2 is the default workload
(same as no flag).  c3 - c5 keep BASELINE's FIXED frame whatever --gpus is (strong scaling: the 8-row strips of
the same 1080p / 4K frame are dealt round-robin to the ranks, as the reference's Renderer_MPI does); their meshes
and textures are the synthetic stand-ins of scenes/gen_assets.py.  --spp overrides the config's spp (a full
C5 frame is 17 Gsamples).

One JSON line on rank 0; see DESIGN.md "Measurement" for the roofline / cpu_baseline definitions.
"""
import argparse
import json
import math
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # this process only, before torch / HIP initialise: --pipeline staged runs four tile groups on streams

import numpy as np  # noqa: E402

BYTES_PER_CAST = 144   # ray 2x32 B + hit 2x16 B + path state 2x24 B
BYTES_PER_SAMPLE = 24  # radiance accumulate read + write
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s


CONFIGS = {   # BASELINE.json configs[1..4]
    "c2": dict(scene="example_project12_box.xml", width=1920, height=1080, spp=512, strong=False,
               what="Cornell box, tinyobjloader cornell_box.obj, 36 triangles"),
    "c3": dict(scene="example_project7_object.xml", width=1920, height=1080, spp=256, strong=True,
               what="triangle meshes + textures, synthetic stand-in assets, 114 k triangles in 6 mesh instances"),
    "c4": dict(scene="example_project12_caustics_glossy.xml", width=3840, height=2160, spp=1024, strong=True,
               what="glossy caustics box, synthetic stand-in teapots, 56 k triangles in 2 mesh instances"),
    "c5": dict(scene="trc_scene_tower.xml", width=3840, height=2160, spp=2048, strong=True,
               what="glass teapot / tower / church, synthetic stand-in assets, 361 k triangles"),
}


def frame_size(n_gpus, base_w, base_h, strong=False):
    if n_gpus == 1 or strong:
        return base_w, base_h
    s = math.sqrt(n_gpus)
    return int(round(base_w * s)), int(round(base_h * s))


def cpu_baseline(scene_xml, width, height, spp, seed):
    """Reference CPU renderer (oracle/_ref/ref_harness, the reference's own code built from its
    sources) timed on this host's cores on a bounded sample of the same workload: the same frame
    at `spp` samples per pixel.  Falls back to the C restatement ("port") when the reference binary
    did not travel."""
    # the GPU box gives one GPU's job a share of the host (cgroup quota), not every hardware thread
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(cores, int(int(quota) / int(period))))
    except Exception:
        pass
    harness = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
    sample = f"{os.path.basename(scene_xml)} {width}x{height} at {spp} spp (same frame, fewer spp), seed {seed:#x}"
    if os.path.exists(harness):
        import tempfile
        with tempfile.TemporaryDirectory() as td:
            out = os.path.join(td, "cpu")
            cmd = [harness, os.path.basename(scene_xml), "--size", str(width), str(height), "--spp", str(spp),
                   "--threads", str(cores), "--seed", str(seed), "--out", out]
            env = dict(os.environ, OMP_NUM_THREADS=str(cores))
            r = subprocess.run(cmd, cwd=os.path.dirname(scene_xml), env=env, stdout=subprocess.PIPE,
                               stderr=subprocess.STDOUT, text=True)
            if r.returncode == 0:
                meta = json.load(open(out + ".json"))
                return {"value": meta["msamples_per_s"], "unit": "Msamples/s", "cores": int(meta["threads"]),
                        "kind": "reference", "sample": sample, "seconds": meta["seconds"]}
    from oracle import binding as oracle
    from qaray_amd.host import load_scene_blob
    blob = load_scene_blob(scene_xml, size=(width, height))
    t0 = time.time()
    _, _, _, cnt = oracle.render(blob, (0, 0, width, height), spp, seed=seed, threads=cores)
    dt = time.time() - t0
    return {"value": cnt.samples / dt * 1e-6, "unit": "Msamples/s", "cores": cores, "kind": "port",
            "sample": sample, "seconds": dt}


def committed_traffic(kernel_name, scene, W, H, spp):
    """Measured HBM-side bytes per launch of the same kernel on the same frame, from the rocprofv3 --pmc passes committed
    under profiles/ (tools/summarize_profiles.py writes the *_summary.json files)."""
    best = None
    try:
        prof_dir = os.path.join(ROOT, "profiles")
        for d in sorted(os.listdir(prof_dir)):
            for f in sorted(os.listdir(os.path.join(prof_dir, d))):
                if f.endswith("_summary.json"):
                    sj = json.load(open(os.path.join(prof_dir, d, f)))
                    if (sj.get("frame") == [W, H] and sj.get("spp") == spp and "hbm_traffic_bytes_per_launch" in sj
                            and sj.get("kernel_name") == kernel_name.split(" (")[0] and sj.get("scene") == scene):
                        best = (sj, os.path.join("profiles", d, f))
    except Exception:
        pass
    return best


def committed_sq(scene, W, H, spp):
    """Compute-side counters of the same frame (rocprofv3 SQ passes summarised by tools/sq_summary.py under profiles/)."""
    best = None
    try:
        prof_dir = os.path.join(ROOT, "profiles")
        for d in sorted(os.listdir(prof_dir)):
            for f in sorted(os.listdir(os.path.join(prof_dir, d))):
                if f.endswith("_sq_counters.json"):
                    sj = json.load(open(os.path.join(prof_dir, d, f)))
                    if sj.get("frame") == [W, H] and sj.get("spp") == spp and sj.get("scene") == scene and "derived" in sj:
                        best = (sj, os.path.join("profiles", d, f))
    except Exception:
        pass
    return best


def add_compute_side(roof, scene, W, H, spp):
    """valu_issue_frac = vector wave-instructions x 2 cycles / SIMD-cycles (a wave64 instruction issues over 2 cycles on a
    SIMD-32), x lane utilisation = the useful share of the vector issue bandwidth; 'bound' says what the counters say."""
    sq = committed_sq(scene, W, H, spp)
    if not sq:
        return
    dv = sq[0]["derived"]
    for k in ("valu_issue_frac", "valu_lane_utilisation", "valu_useful_frac", "wave_time_issuing", "wave_time_issue_stalled", "wave_time_waiting"):
        if k in dv:
            roof[k] = dv[k]
    roof["sq_source"] = sq[1] + " (kernel " + str(sq[0].get("kernel", "?"))[:60] + ")"
    if dv.get("valu_issue_frac", 0) + 0.0 >= 0.45:
        roof["bound"] = "valu-issue"   # half of the vector issue slots taken, a third of wave time stalled on issue: not HBM
    else:
        roof["bound"] = "latency"      # waves mostly wait (dependent loads, LDS round trips): neither HBM bandwidth nor vector issue


CPU_SAMPLE_SPP = {"c3": 24, "c4": 10, "c5": 6}   # bounded CPU samples of >= 5 s on 16 host threads (4.1 / 13.1 / 7.3 Msamples/s in round 2)


def other_config(ctx, tag, args, torch, hip, qd, load_scene_blob, SCENES_DIR, device):
    """One full-spp step of BASELINE config `tag` on this GPU (scene resident, HIP events around the launch), its roofline
    entry and a CPU-baseline sample of the same frame.  -> dict for the bench line's 'other_configs'."""
    cfg = CONFIGS[tag]
    W, H, spp = cfg["width"], cfg["height"], cfg["spp"]
    scene_xml = os.path.join(SCENES_DIR, cfg["scene"])
    blob = load_scene_blob(scene_xml, size=(W, H))
    ctx.upload_scene_device(torch.from_numpy(blob).to(device))
    bufs = qd.StripBuffers(H, W, device)
    side = torch.cuda.Stream(device=device)
    with torch.cuda.stream(side):
        ctx.render_region_device((0, 0, 64, 64), 1, bufs.rgb[:64, :64].contiguous(), bufs.depth[:64, :64].contiguous(),
                                 bufs.ns[:64, :64].contiguous(), max_bounce=args.bounce, seed=args.seed, stream=side.cuda_stream)   # warm-up (code, tables)
        torch.cuda.synchronize()
        ctx.reset_kernel_time()
        ctx.reset_counters()
        t0 = time.perf_counter()
        ctx.render_region_device((0, 0, W, H), spp, bufs.rgb, bufs.depth, bufs.ns, max_bounce=args.bounce, seed=args.seed, stream=side.cuda_stream)
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
    k_ms, launches = ctx.kernel_time()
    cnt = ctx.counters()
    kernel_name = ctx.kernel_name()
    casts = cnt["casts_normal"] + cnt["casts_shadow"]
    k_bytes = casts * BYTES_PER_CAST + cnt["samples"] * BYTES_PER_SAMPLE
    achieved = k_bytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
    out = {"workload": f"inputs/{cfg['scene']} ({cfg['what']}), {W}x{H}, {spp} spp", "steps": 1,
           "value": cnt["samples"] / elapsed * 1e-6, "unit": "Msamples/s", "ms_per_step": elapsed * 1e3,
           "casts_per_sample": casts / max(cnt["samples"], 1),
           "roofline": {"bound": "latency (waves wait 36 - 54 % of their time at ~40 % VALU issue: profiles/round03/sq_hbm_counters_16spp.txt, DESIGN.md 5 round 3)", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": achieved / HBM_PEAK_GBS, "traffic": None, "kernel": kernel_name, "kernel_ms_avg": k_ms / max(launches, 1),
                        "algorithmic_bytes_per_launch": k_bytes}}
    best = committed_traffic(kernel_name, cfg["scene"], W, H, spp)
    if best:
        out["roofline"]["traffic"] = best[0]["hbm_traffic_bytes_per_launch"]
        out["roofline"]["traffic_source"] = best[1]
    add_compute_side(out["roofline"], cfg["scene"], W, H, spp)   # SQ passes of the same frame, when committed: sets 'bound' from the counters
    add_compute_side(out["roofline"], cfg["scene"], W, H, spp)
    if args.cpu_spp > 0:
        out["cpu_baseline"] = cpu_baseline(scene_xml, W, H, CPU_SAMPLE_SPP[tag], args.seed)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", choices=sorted(CONFIGS), default=None, help="a BASELINE.json config by name (default: c2's workload, weak scaling)")
    ap.add_argument("--scene", default=None)
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--spp", type=int, default=None)
    ap.add_argument("--pipeline", choices=["auto", "mega", "staged"], default="auto",
                    help="integrator for scenes beyond LDS (same bits either way; auto = timed probe, see include/qaray_hip.h)")
    ap.add_argument("--bounce", type=int, default=5)
    ap.add_argument("--seed", type=lambda s: int(s, 0), default=0x51A7A7)
    ap.add_argument("--cpu-spp", type=int, default=128, help="spp of the bounded CPU-baseline sample (0 = skip)")
    ap.add_argument("--save-png", default=None, help="rank 0: write the last frame through FrameBuffer")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo: rehearsal of the N>1 path on a box with fewer GPUs than ranks (ranks share devices, "
                         "collectives go through host memory)")
    ap.add_argument("--photon-map", type=float, nargs=6, default=None, metavar=("N", "BOUNCE", "RADIUS", "CN", "CBOUNCE", "CRADIUS"),
                    help="probe mode (not the headline workload): build photon / caustics maps first (-use-photon-map), "
                         "e.g. --photon-map 10000 20 0.2 1000 20 1.0")
    ap.add_argument("--check", action="store_true",
                    help="rank 0: also render the whole frame alone and require the gathered image to equal it bit for bit")
    ap.add_argument("--force-collectives", action="store_true",
                    help="create the process group and run broadcast + gather with ONE rank too (the RCCL path on a one-GPU box; "
                         "launch under torch.distributed.run --nproc-per-node 1)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the one full-spp step of BASELINE's c3 / c4 / c5 that the default N=1 line carries in 'other_configs'")
    args = ap.parse_args()
    cfg = CONFIGS[args.config or "c2"]
    strong = bool(args.config) and cfg["strong"]
    args.scene = args.scene or cfg["scene"]
    args.width = args.width or cfg["width"]
    args.height = args.height or cfg["height"]
    args.spp = args.spp or cfg["spp"]
    if args.scene != "example_project12_box.xml":
        subprocess.run([sys.executable, os.path.join(ROOT, "scenes", "gen_assets.py")], check=True, stdout=subprocess.DEVNULL)

    import torch
    import torch.distributed as dist
    from qaray_amd import distributed as qd
    from qaray_amd import hip
    from qaray_amd.host import SCENES_DIR, FrameBuffer, load_scene_blob

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
    if args.backend == "gloo":
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    collective = world > 1 or args.force_collectives
    if collective:
        if "RANK" not in os.environ:   # --force-collectives without a launcher: a one-rank group of our own
            os.environ.update(RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", "29531"))
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=device)
        else:
            dist.init_process_group(backend="gloo")
    coll_device = device if args.backend == "nccl" else torch.device("cpu")

    W, H = frame_size(world, args.width, args.height, strong)
    scene_xml = args.scene if os.path.isabs(args.scene) else os.path.join(SCENES_DIR, args.scene)

    ctx = hip.Context(local_rank)
    ctx.set_pipeline(args.pipeline)
    if args.pipeline == "staged":
        ctx.set_option("staged_groups", 4)   # with GPU_MAX_HW_QUEUES=8 (set above, before HIP initialises)
    # rank 0 parses + flattens; everyone receives the blob over RCCL and adopts it from HBM
    blob = load_scene_blob(scene_xml, size=(W, H)) if rank == 0 else None
    if collective:
        dblob = qd.broadcast_blob(blob, coll_device, src=0).to(device)
    else:
        dblob = torch.from_numpy(blob).to(device)
    ctx.upload_scene_device(dblob)
    if args.photon_map:
        pm = args.photon_map   # every rank builds the same maps (deterministic in scene, parameters and seed)
        ctx.build_photon_maps((int(pm[0]), int(pm[1]), pm[2]), (int(pm[3]), int(pm[4]), pm[5]), seed=args.seed)

    nstrips = hip.strip_count(0, H, rank, world)
    maxstrips = qd.max_strips_per_rank(H, world)
    rows = maxstrips * qd.STRIP_ROWS          # equal shapes on every rank for the gather
    # colour, z-buffer and sample counts of this rank's strips in ONE flat buffer: the frame is gathered in one collective
    bufs = qd.StripBuffers(rows, W, device)
    rgb, depth, ns = bufs.rgb, bufs.depth, bufs.ns
    # One explicit stream for the kernel AND everything that consumes its output (the RCCL gather,
    # the host copy of the gloo rehearsal, the strip assembly): torch.distributed orders collectives
    # after the work already queued on torch's current stream, so the step runs with this stream
    # current.  (torch's default stream has the handle 0, which the C ABI reads as "use the context's
    # own stream" - a collective issued from the default stream would not wait for the kernel.)
    side = torch.cuda.Stream(device=device)
    stream = side.cuda_stream
    assert stream != 0
    nown = nstrips * qd.STRIP_ROWS
    full = None

    def step():
        with torch.cuda.stream(side):
            _step()

    def _step():
        nonlocal full
        if nstrips:
            ctx.render_strips_device((0, 0, W, H), rank, world, args.spp, rgb[:nown], depth[:nown], ns[:nown],
                                     max_bounce=args.bounce, seed=args.seed, stream=stream)
        if collective:
            g = qd.gather_packed(bufs.flat if args.backend == "nccl" else bufs.flat.cpu(), dst=0, force=True)
            if rank == 0:
                full = qd.assemble_frame(g, rows, W, H, world)   # (rgb, depth, sample counts)
        else:
            full = (rgb[:H], depth[:H], ns[:H])

    def fence():
        torch.cuda.synchronize()
        if collective:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    ctx.reset_kernel_time()
    ctx.reset_counters()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if collective:
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    kernel_ms, launches = ctx.kernel_time()
    cnt = ctx.counters()
    kernel_name = ctx.kernel_name()
    staged = ctx.staged_stats() if kernel_name.startswith("staged") else None
    local = torch.tensor([cnt["samples"], cnt["casts_normal"], cnt["casts_shadow"]], dtype=torch.float64, device=coll_device)
    if collective:
        dist.all_reduce(local, op=dist.ReduceOp.SUM)
    samples, casts_n, casts_s = (float(v) for v in local.tolist())

    if rank == 0:
        msamples = samples / elapsed * 1e-6
        ms_per_step = elapsed / args.steps * 1e3
        # roofline of the dominant (only) kernel, rank 0's launches: ALGORITHMIC bytes per launch /
        # average launch duration measured with HIP events on the launch stream
        k_samples = cnt["samples"] / max(launches, 1)
        k_casts = (cnt["casts_normal"] + cnt["casts_shadow"]) / max(launches, 1)
        k_bytes = k_casts * BYTES_PER_CAST + k_samples * BYTES_PER_SAMPLE
        k_ms = kernel_ms / max(launches, 1)
        achieved = k_bytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        headline = os.path.basename(scene_xml) == "example_project12_box.xml"
        what = next((c["what"] for c in CONFIGS.values() if c["scene"] == os.path.basename(scene_xml)), "probe scene, not a BASELINE config")
        if staged:
            limiter = ("critical path of the four stage kernels of a pass (launch ramp + the longest mesh walks while the chip drains), several "
                       "chains in flight; not HBM bandwidth, not arithmetic (profiles/round02/staged_timeline_16spp.txt, session3_experiments.txt)")
        elif "RES=1" in kernel_name:
            limiter = ("VALU issue (55 % of the SIMD cycles) at 57 % lane utilisation - the divergence of the per-lane tree walks (LDS-resident tree: "
                       "53 % of wave time; all correctly rounded div / sqrt removed: +5.5 % only); the wave slots are occupied 98.6 % of the kernel "
                       "since a tile's samples are handed out in chunks (88.6 % before: profiles/round03/experiments.txt 27); not HBM bandwidth")
        elif "qa_integrate_cs" in kernel_name:
            limiter = ("latency: waves wait 30 - 50 % of their time (dependent node / ray-slot reads of the cooperative walks, reloads of spilled "
                       "registers) at 38 - 48 % VALU issue; on scenes of few meshes the per-node arithmetic of the scene-graph sweeps (C4: 54 % of wave "
                       "time, profiles/round03/stamps_cs_16spp.txt; their scalar loads are not it: experiments.txt 20); what the spilled registers "
                       "cost once they leave the L2 is the first-order effect (DESIGN.md 5 round 3, sq_hbm_counters_16spp.txt); not HBM bandwidth")
        else:
            limiter = ("mesh walks in global memory at low lane occupancy and the dependent loads of the scene-graph loop; not HBM bandwidth, not "
                       "arithmetic (profiles/round02/megakernel_section_stamps.txt, session3_experiments.txt)")
        out = {
            "metric": "Msamples/s (1 sample = 1 camera path), " + ("Cornell box 1080p@512spp" if headline else f"{os.path.basename(scene_xml)} {W}x{H}@{args.spp}spp"),
            "value": msamples, "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"inputs/{os.path.basename(scene_xml)} ({what})"
                                   + (f", photon maps {args.photon_map}" if args.photon_map else "")
                                   + f", {W}x{H}, {args.spp} spp, maxBounce {args.bounce}, seed {args.seed:#x}",
                       "baseline_config": args.config or ("c2" if headline and (args.spp, args.width, args.height) == (512, 1920, 1080) else None),
                       "frame": [W, H], "spp": args.spp, "partition": f"8-row strips round-robin over {world} GPU(s)",
                       "wall_clock_s_per_frame": ms_per_step * 1e-3,
                       "casts_per_sample": (casts_n + casts_s) / max(samples, 1)},
            # achieved = ALGORITHMIC queue bytes (casts x 144 B + samples x 24 B) per launch / the launch's
            # duration (HIP events on the launch stream).  That is the contract's figure; it is NOT memory traffic: the
            # megakernel keeps path state in registers, and neither integrator is bound by HBM bandwidth or by its arithmetic
            # (profiles/round02/megakernel_section_stamps.txt, session3_experiments.txt): see 'actual_limiter'.
            # 'bound': "hbm" is what SURVEY 8(d) prices (and 'frac' is that figure); where SQ counter passes of this frame are
            # committed, add_compute_side() replaces it by what the counters say and adds the vector-issue fractions.
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "kernel": kernel_name, "kernel_ms_avg": k_ms, "launches": int(launches),
                         "algorithmic_bytes_per_launch": k_bytes,
                         "actual_limiter": limiter,
                         "note": "achieved = (casts x 144 B + samples x 24 B) / launch time, SURVEY.md 8d; measured HBM bytes are in "
                                 "'traffic' when a PMC pass of this kernel and frame is committed under profiles/"},
        }
        if staged:
            # geometry the trace stage read, from its own counters: 64-byte nodes of the 4-wide tree, 48-byte triangle records
            out["roofline"]["geometry_bytes_per_launch"] = staged["geometry_bytes"] / max(launches, 1)
            out["roofline"]["trace_lane_utilisation"] = staged["lane_utilisation"]
            out["roofline"]["staged"] = {k: staged[k] for k in ("passes", "jobs_done", "node_steps", "tri_tests", "rays_redone", "jobs_suspended")}
        # measured HBM traffic of the same kernel on the same frame, from the committed rocprofv3 --pmc passes
        best = committed_traffic(kernel_name, os.path.basename(scene_xml), W, H, args.spp)
        if best:
            out["roofline"]["traffic"] = best[0]["hbm_traffic_bytes_per_launch"]
            out["roofline"]["measured_hbm_gbs"] = best[0]["hbm_traffic_bytes_per_launch"] / (k_ms * 1e-3) / 1e9 if k_ms > 0 else None
            out["roofline"]["traffic_source"] = (best[1] + " (2 x FETCH_SIZE + WRITE_SIZE, KiB -> bytes, separate --pmc passes: an UPPER bound for HBM, "
                                                 "the counters also see Infinity-Cache hits; measured on build " + str(best[0].get("build", "?")) + ")")
        if world == 1:
            add_compute_side(out["roofline"], os.path.basename(scene_xml), W, H, args.spp)
        if world == 1 and args.cpu_spp > 0:
            out["cpu_baseline"] = cpu_baseline(scene_xml, W, H, args.cpu_spp, args.seed)
        # the default one-GPU line also carries ONE full-spp step of BASELINE's other three configs (the driver only runs this command)
        default_line = (world == 1 and not collective and args.config is None and headline and (args.spp, W, H) == (512, 1920, 1080)
                        and not args.photon_map and args.pipeline == "auto")
        if default_line and not args.no_other_configs:
            subprocess.run([sys.executable, os.path.join(ROOT, "scenes", "gen_assets.py")], check=True, stdout=subprocess.DEVNULL)
            out["other_configs"] = {}
            for tag in ("c3", "c4", "c5"):
                try:
                    out["other_configs"][tag] = other_config(ctx, tag, args, torch, hip, qd, load_scene_blob, SCENES_DIR, device)
                except Exception as e:   # never lose the headline line to a side measurement
                    out["other_configs"][tag] = {"error": repr(e)}
        if collective:
            out["collectives"] = {"backend": args.backend, "world": world, "broadcast": "flat scene blob (uint8)",
                                  "gather": "one flat float32 buffer per rank: rgb | depth | sample counts (Renderer_MPI.cpp:194-207)"}
        if args.check:
            alone = ctx.render_region((0, 0, W, H), args.spp, max_bounce=args.bounce, seed=args.seed)
            got = [t.cpu().numpy() for t in full]
            out["check"] = bool(all(np.array_equal(np.ascontiguousarray(a).view(np.uint32), np.ascontiguousarray(b).view(np.uint32))
                                    for a, b in zip(alone, got)))
        if args.save_png:
            # the reference's three images (Renderer_MPI.cpp:130-139) from the GATHERED arrays: <prefix>colorBuffer.png, ...
            fb = FrameBuffer(W, H)
            fb.deposit(0, 0, W, H, full[0].cpu().numpy(), full[1].cpu().numpy(), full[2].cpu().numpy().astype(np.uint32), args.spp, use_srgb=True)
            prefix = args.save_png[:-4] if args.save_png.endswith(".png") else args.save_png
            fb.save_image(args.save_png if args.save_png.endswith(".png") else prefix + "colorBuffer.png")
            fb.save_z_image(prefix + "depthBuffer.png")
            fb.save_sample_count_image(prefix + "sampleBuffer.png")
        print(json.dumps(out), flush=True)
    if collective:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
