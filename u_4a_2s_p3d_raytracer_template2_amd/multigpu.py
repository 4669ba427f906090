"""Image-space sharding across ranks: interleaved row blocks + one gather.

Pure index math and torch.distributed plumbing shared by bench.py and the tests; the pixels
themselves always come from the HIP path (p3d_render with rank/world) -- nothing here renders.
"""
import numpy as np


def block_rows(res_y, row_block, rank, world):
    """[(local_row0, y0, n_rows)] of the row blocks `rank` owns, in local order."""
    out = []
    nblocks = (res_y + row_block - 1) // row_block
    for lb, b in enumerate(range(rank, nblocks, world)):
        y0 = b * row_block
        out.append((lb * row_block, y0, min(row_block, res_y - y0)))
    return out


def padded_rows(res_y, row_block, world):
    nblocks = (res_y + row_block - 1) // row_block
    return ((nblocks + world - 1) // world) * row_block


def stitch_reference(parts, res_y, row_block):
    """CPU statement of p3d_deinterleave: parts[r] is rank r's compact [rows, W, C] array."""
    world = len(parts)
    out = np.zeros((res_y,) + tuple(parts[0].shape[1:]), parts[0].dtype)
    for r in range(world):
        for (l0, y0, n) in block_rows(res_y, row_block, r, world):
            out[y0:y0 + n] = parts[r][l0:l0 + n]
    return out


def gather_to_root(tile, dist, rank, world, gathered=None, async_op=False):
    """One collective per step: every rank's compact tile buffer -> rank 0 (direct peer->root
    transfers over xGMI, 7 links in parallel).  `tile` and `gathered` are torch
    tensors; gathered is [world, *tile.shape] on rank 0.  async_op=True returns the collective's
    work handle instead of waiting for it (wait() before touching `gathered` or reusing `tile`)."""
    if world == 1:
        return None if async_op else tile.unsqueeze(0)
    work = dist.gather(tile, gather_list=list(gathered.unbind(0)) if rank == 0 else None, dst=0, async_op=async_op)
    if async_op:
        return work
    return gathered if rank == 0 else None
