"""N>1 logic on CPU: two gloo ranks partition a frame into round-robin 8-row strips, broadcast the
scene blob, render their strips and gather to rank 0.  The renderer is injected: here the CPU oracle
stands in for the HIP kernel (tests may use the oracle); bench.py plugs the HIP context into the same
functions.  Multi-rank result must equal the single-process image bit for bit (per-pixel RNG streams)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, W, H, spp, out_path):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import binding as oracle
    from qaray_amd import distributed as qd
    from qaray_amd.host import load_scene_blob
    blob = load_scene_blob("example_project3_box.xml", size=(W, H)) if rank == 0 else None
    t = qd.broadcast_blob(blob, torch.device("cpu"), src=0)
    blob = t.numpy()
    rows = qd.max_strips_per_rank(H, world) * qd.STRIP_ROWS
    # colour, z-buffer and sample counts travel in ONE flat buffer per rank (what the reference gathers: Renderer_MPI.cpp:194-207)
    bufs = qd.StripBuffers(rows, W, torch.device("cpu"))
    for k, s in enumerate(qd.own_strips(H, world, rank)):
        y0 = s * qd.STRIP_ROWS
        y1 = min(H, y0 + qd.STRIP_ROWS)
        rgb, depth, ns, _ = oracle.render(blob, (0, y0, W, y1), spp, threads=1)
        sl = slice(k * qd.STRIP_ROWS, k * qd.STRIP_ROWS + (y1 - y0))
        bufs.rgb[sl] = torch.from_numpy(rgb)
        bufs.depth[sl] = torch.from_numpy(depth)
        bufs.ns[sl] = torch.from_numpy(ns.astype(np.int32))
    g = qd.gather_packed(bufs.flat, dst=0, force=True)   # force: the collective runs with one rank too
    if rank == 0:
        rgb, depth, ns = qd.assemble_frame(g, rows, W, H, world)
        np.savez(out_path, rgb=rgb.numpy(), depth=depth.numpy(), ns=ns.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,H", [(1, 20), (2, 36), (3, 41), (4, 70), (8, 150)])   # 4 and 8: BASELINE C4 / C5 rank counts; 1: forced collectives
def test_strip_partition_gather_equals_single_process(tmp_path, world, H):
    W, spp = 40, 2
    out = str(tmp_path / "full.npz")
    mp.spawn(_worker, args=(world, _free_port(), W, H, spp, out), nprocs=world, join=True)
    from oracle import binding as oracle
    from qaray_amd.host import load_scene_blob
    ref = oracle.render(load_scene_blob("example_project3_box.xml", size=(W, H)), (0, 0, W, H), spp)
    got = np.load(out)
    for name, r in zip(("rgb", "depth", "ns"), ref[:3]):
        assert got[name].shape == r.shape, name
        assert np.array_equal(np.ascontiguousarray(got[name]).view(np.uint32), np.ascontiguousarray(r).view(np.uint32)), name


def test_strip_bookkeeping():
    from qaray_amd import distributed as qd
    for H in (1, 7, 8, 9, 1080, 2160, 3055):
        for world in (1, 2, 3, 4, 8):
            owned = [qd.own_strips(H, world, r) for r in range(world)]
            assert sorted(sum(owned, [])) == list(range(qd.num_strips(H)))
            assert max(len(o) for o in owned) == qd.max_strips_per_rank(H, world)
