"""N > 1 path on CPU: 2 ranks over gloo.  The per-rank renderer is the oracle (test
infrastructure standing in for the GPU), so this checks the partitioning and the reduce:
sample split and tile split must reproduce the single-process result."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from util import GOLDEN, ROOT, sub

D = sub("distributed")
api = sub("api")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _scene():
    from oracle import oracle_py as O
    d = O.load_scene_dict(os.path.join(GOLDEN, "scenes/spheres/spheres.yaml"), base_dir=GOLDEN)
    d["width"], d["height"] = 52, 37
    return O.OracleScene.from_dict(d)


def _worker(rank, world, port, split, out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    osc = _scene()
    params = api.RenderParams(api.SuperSampling("Jitter", 3), depth=4, seed=5)

    def render(p):
        r = osc.render(mode="jitter", N=3, depth=p.depth, seed=p.seed, threads=1, s_begin=p.sample_begin,
                       s_end=p.sample_end if p.sample_end else -1)["sum"]
        if p.tile_count > 1:
            r = r * D.tile_mask(osc.width, osc.height, p.tile_index, p.tile_count)[..., None]
        return torch.from_numpy(r)

    acc = D.render_distributed(render, params, rank, world, split)
    if rank == 0:
        np.save(out, acc.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("split", ["spp", "tiles"])
def test_two_ranks_reproduce_single_process(split, tmp_path):
    out = str(tmp_path / "acc.npy")
    mp.spawn(_worker, args=(2, _free_port(), split, out), nprocs=2, join=True)
    full = _scene().render(mode="jitter", N=3, depth=4, seed=5, threads=1)["sum"]
    got = np.load(out)
    assert np.allclose(got, full, rtol=1e-12, atol=1e-12)


def test_partition_helpers():
    assert [D.sample_range(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert [D.sample_range(4096, r, 8, 0) for r in range(8)][-1] == (3584, 4096)
    masks = [D.tile_mask(53, 21, r, 3) for r in range(3)]
    assert np.array_equal(sum(m.astype(int) for m in masks), np.ones((21, 53), int))
    p = api.RenderParams(api.SuperSampling("Jitter", 8), sample_begin=8, sample_end=40)
    q = D.partition(p, 1, 4, "spp")
    assert (q.sample_begin, q.sample_end) == (16, 24)
    q = D.partition(p, 2, 4, "tiles")
    assert (q.tile_index, q.tile_count, q.sample_begin, q.sample_end) == (2, 4, 8, 40)


@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_bench_step_partition_is_disjoint_and_complete(world):
    """bench.py's per-step partition (distributed.bench_step_params), without a GPU: under the sample split the cell
    ranges of the ranks of one step are disjoint, and 4096 / (64 * world) steps cover the 64x64 grid exactly once;
    under the tile split every rank traces the same cells and the tile masks are disjoint and cover the frame."""
    grid_n = 64
    for spp in (64, 512):          # 512: bench.py's default step, C5's per-GPU share of the 4096-spp frame (at world 8 ONE step is the frame)
        seen = np.zeros(grid_n * grid_n, int)
        steps = grid_n * grid_n // (spp * world)
        assert spp != 512 or world != 8 or steps == 1
        for k in range(steps):
            ranges = [D.bench_step_params(k, r, world, spp, grid_n, "spp") for r in range(world)]
            for (b, e, ti, tc) in ranges:
                assert 0 <= b < e <= grid_n * grid_n and e - b == spp and (ti, tc) == (0, 1)
                seen[b:e] += 1
        assert np.all(seen == 1)
        assert D.bench_step_params(steps, 0, world, spp, grid_n, "spp")[:2] == (0, spp)        # wraps
    spp = 64
    for k in (0, 5):
        ps = [D.bench_step_params(k, r, world, spp, grid_n, "tiles") for r in range(world)]
        assert len({p[:2] for p in ps}) == 1 and [p[2:] for p in ps] == [(r, world) for r in range(world)]
    for (w, h) in ((1920, 1080), (3840, 2160), (53, 21)):
        masks = [D.tile_mask(w, h, r, world) for r in range(world)]
        assert np.array_equal(sum(m.astype(int) for m in masks), np.ones((h, w), int))
    with pytest.raises(ValueError):
        D.bench_step_params(0, 0, world, 48, grid_n, "spp")
