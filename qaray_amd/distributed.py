"""Image-space partition across ranks (one process per GPU), mirroring the reference's MPI mode:
every rank renders tiles rank, rank+size, ... (src/renderers/renderer.cpp:383-387) and rank 0
gathers and places them (src/renderers/Renderer_MPI.cpp:142-207).  Differences by design:
the flattened scene is BROADCAST from rank 0 as one blob (the reference re-parses the XML on every
rank), the partition unit is an 8-row strip, and the gather moves float radiance, not 8-bit pixels.

The functions here are transport-only (torch.distributed: "nccl" = RCCL on GPUs, "gloo" on CPU) and
take the renderer as a callable, so the N>1 logic is testable without a GPU."""
import numpy as np

STRIP_ROWS = 8


def num_strips(height):
    return (height + STRIP_ROWS - 1) // STRIP_ROWS


def own_strips(height, world, rank):
    """Strip indices owned by `rank` (round-robin)."""
    return list(range(rank, num_strips(height), world))


def max_strips_per_rank(height, world):
    return (num_strips(height) + world - 1) // world


def place_strips(full, packed, height, world, rank):
    """Copy rank's packed strips (rows k*8..k*8+7 = strip rank+k*world) into the full image (numpy or
    torch, first axis = rows).  The analogue of PlaceImage<T> (Renderer_MPI.cpp:103-122)."""
    for k, s in enumerate(own_strips(height, world, rank)):
        r0 = s * STRIP_ROWS
        n = min(STRIP_ROWS, height - r0)
        full[r0:r0 + n] = packed[k * STRIP_ROWS:k * STRIP_ROWS + n]
    return full


def broadcast_blob(blob_np, device, src=0):
    """Rank `src` passes its flat scene (numpy uint8); every rank returns a uint8 tensor on `device`
    holding the same bytes.  One size broadcast + one payload broadcast."""
    import torch
    import torch.distributed as dist
    rank = dist.get_rank()
    n = torch.tensor([blob_np.size if rank == src else 0], dtype=torch.int64, device=device)
    dist.broadcast(n, src=src)
    if rank == src:
        t = torch.from_numpy(np.ascontiguousarray(blob_np)).to(device)
    else:
        t = torch.empty(int(n.item()), dtype=torch.uint8, device=device)
    dist.broadcast(t, src=src)
    return t


def gather_packed(packed, dst=0, force=False):
    """Gather equally-shaped packed strip tensors to rank `dst` -> list of tensors (None elsewhere).
    force: issue the collective with one rank too (exercises the RCCL path on a one-GPU box)."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(), dist.get_world_size()
    if world == 1 and not force:
        return [packed]
    out = [torch.empty_like(packed) for _ in range(world)] if rank == dst else None
    dist.gather(packed, gather_list=out, dst=dst)
    return out


class StripBuffers:
    """A rank's packed strips of ALL arrays the reference gathers (colour, z-buffer, sample count; the mask is
    `sample count != 0`: src/renderers/Renderer_MPI.cpp:194-207) in ONE flat float32 tensor, so that the frame travels in
    one collective: [rgb rows*W*3 | depth rows*W | sample counts rows*W (int32 bits)]."""

    def __init__(self, rows, width, device):
        import torch
        self.rows, self.width = rows, width
        n = rows * width
        self.flat = torch.zeros(5 * n, dtype=torch.float32, device=device)
        self.rgb = self.flat[:3 * n].view(rows, width, 3)
        self.depth = self.flat[3 * n:4 * n].view(rows, width)
        self.ns = self.flat[4 * n:].view(torch.int32).view(rows, width)

    @staticmethod
    def views(flat, rows, width):
        import torch
        n = rows * width
        return flat[:3 * n].view(rows, width, 3), flat[3 * n:4 * n].view(rows, width), flat[4 * n:].view(torch.int32).view(rows, width)


def assemble_frame(gathered_flat, rows, width, height, world):
    """Rank 0: the gathered flat buffers of every rank -> (rgb [H,W,3], depth [H,W], sample counts [H,W])."""
    parts = [StripBuffers.views(g, rows, width) for g in gathered_flat]
    return tuple(assemble([p[i] for p in parts], height, world) for i in range(3))


_ROW_INDEX_CACHE = {}


def strip_rows(height, world, rank, device=None):
    """Image rows covered by `rank`'s packed strips, in packed order (torch int64 tensor)."""
    import torch
    key = (height, world, rank, str(device))
    if key not in _ROW_INDEX_CACHE:
        rows = []
        for s in own_strips(height, world, rank):
            r0 = s * STRIP_ROWS
            rows.extend(range(r0, min(height, r0 + STRIP_ROWS)))
        _ROW_INDEX_CACHE[key] = torch.tensor(rows, dtype=torch.int64, device=device)
    return _ROW_INDEX_CACHE[key]


def assemble(gathered, height, world):
    """Rank 0: list of packed per-rank tensors -> full image tensor [height, ...].  One scatter of
    rows per rank: only the last strip of the image can be ragged, and it is the last strip of its
    owner's packed buffer, so a rank's valid rows are a prefix of that buffer."""
    import torch
    first = gathered[0]
    full = torch.empty((height,) + tuple(first.shape[1:]), dtype=first.dtype, device=first.device)
    for r in range(world):
        idx = strip_rows(height, world, r, first.device)
        if idx.numel():
            full.index_copy_(0, idx, gathered[r][:idx.numel()])
    return full
