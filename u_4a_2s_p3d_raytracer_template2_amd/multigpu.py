"""Image-space sharding across ranks: interleaved row blocks + one gather.

Pure index math shared by bench.py and the tests, plus the host-memory gather bench.py falls back to when RCCL is
not available; the pixels themselves always come from the HIP path (p3d_render with rank/world) and, with RCCL up,
move by p3d_gather behind the C-ABI -- nothing here renders.
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


def gather_tiles_through_host(tiles, gathered, dist, rank, world, torch, stream=None):
    """The SAFETY NET of bench.py's N > 1 path (never the product path, and the JSON line says so): when a rank
    could not build the RCCL communicator -- or in the one-GPU rehearsal mode -- every rank's compact tile buffers
    travel to rank 0 through host memory with torch.distributed (gloo).  `tiles` is this rank's [B, rows, W, C]
    tensor (device or host), `gathered` rank 0's [world, B, rows, W, C] tensor (None elsewhere); the copy into a
    device `gathered` is enqueued on `stream`.  With RCCL up, the same bytes move by p3d_gather (include/p3d_hip.h)."""
    if tiles.is_cuda:
        torch.cuda.synchronize()
    src = tiles.cpu().contiguous()
    dst = list(torch.zeros((world,) + tuple(src.shape), dtype=src.dtype).unbind(0)) if rank == 0 else None
    dist.gather(src, gather_list=dst, dst=0)
    if rank != 0:
        return None
    stacked = torch.stack(dst)
    if gathered.is_cuda and stream is not None:
        with torch.cuda.stream(stream):
            gathered.copy_(stacked)
    else:
        gathered.copy_(stacked)
    return gathered
