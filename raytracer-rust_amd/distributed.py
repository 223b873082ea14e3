"""Multi-GPU partitioning of the render path.  One process per GPU;
torch.distributed (backend "nccl" = RCCL over xGMI, "gloo" in CPU tests) is only the transport.

  * sample split (C5): every rank renders the full frame for a disjoint range of the N*N sample
    list; the f32 partial accumulators are summed on rank 0 with ONE reduce (W*H*3*4 bytes), then
    resolved there.  This is the only exchange step of the path.
  * tile split (C4): 8x8-pixel tiles are dealt round-robin (tile t -> rank t % world); ranks render
    disjoint pixels of the same frame.  The same reduce assembles the frame on rank 0 (a sum of
    disjoint supports); no other communication.

The reference has no counterpart: it is one process (raytrace.rs:126-154)."""
import copy

import numpy as np


def sample_range(n_samples, rank, world, begin=0):
    """Contiguous, balanced split of samples [begin, begin+n_samples) -> (b, e) of `rank`."""
    base, rem = divmod(int(n_samples), int(world))
    b = begin + rank * base + min(rank, rem)
    return b, b + base + (1 if rank < rem else 0)


def tile_mask(width, height, rank, world):
    """Boolean (H, W) mask of the pixels rank renders under the tile split (row 0 = top)."""
    tx, ty = (width + 7) // 8, (height + 7) // 8
    t = np.arange(tx * ty).reshape(ty, tx)
    m = (t % world) == rank
    return np.repeat(np.repeat(m, 8, axis=0), 8, axis=1)[:height, :width]


def partition(params, rank, world, split="spp"):
    """RenderParams of one rank for a job described by `params` (whole frame, its sample range)."""
    p = copy.copy(params)
    if split == "spp":
        end = params.sample_end if params.sample_end else params.ssaa.samples_per_pixel
        p.sample_begin, p.sample_end = sample_range(end - params.sample_begin, rank, world, params.sample_begin)
        if p.sample_end == p.sample_begin:
            p.sample_end = p.sample_begin   # empty range: the C ABI treats begin == end as "nothing"
    elif split == "tiles":
        p.tile_index, p.tile_count = rank, world
    else:
        raise ValueError(split)
    return p


def bench_step_params(step, rank, world, spp, grid_n, split):
    """Partition arithmetic of one bench.py step: (sample_begin, sample_end, tile_index, tile_count) of `rank`.

    Sample split (weak scaling): every rank renders the full frame for `spp` cells of the grid_n x grid_n jitter grid
    of its own; the cell ranges of the ranks of one step are disjoint and consecutive, and successive steps move on
    through the grid (wrapping after grid_n^2 cells).  Tile split (strong scaling): every rank renders the same `spp`
    cells on its own tiles (tile t -> rank t % world)."""
    cells = grid_n * grid_n
    if spp <= 0 or cells % spp:
        raise ValueError("spp per step must divide the %d cells of the grid" % cells)
    if split == "spp":
        c0 = ((step * world + rank) * spp) % cells
        return c0, c0 + spp, 0, 1
    if split == "tiles":
        c0 = (step * spp) % cells
        return c0, c0 + spp, rank, world
    raise ValueError(split)


def reduce_to_root(accum, dst=0):
    """Sum of the ranks' partial accumulators on `dst` (in place on dst).  accum: torch tensor."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.reduce(accum, dst=dst, op=dist.ReduceOp.SUM)
    return accum


def render_distributed(render_fn, params, rank, world, split="spp"):
    """render_fn(params_for_rank) -> torch tensor (H, W, 3) of sample sums.  Returns the summed
    accumulator on rank 0 (other ranks: their partial)."""
    acc = render_fn(partition(params, rank, world, split))
    return reduce_to_root(acc, 0)
