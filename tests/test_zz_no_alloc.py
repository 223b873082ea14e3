"""rt_render_accumulate_noalloc (include/rtamd.h): the render entry point for callers that cannot allocate device
memory.  Called through the C ABI and compared with rt_render_accumulate on two calls in a row, the second with a
smaller input; plus its limits.  Collected last (file name) so that a failure here hides no other test."""
import numpy as np
import pytest

from util import GOLDEN, pkg, sub

pytestmark = pytest.mark.gpu

rt = pkg()
scenegen = sub("scenegen")
libmod = sub("lib")


def _both(ctx, params):
    import torch
    a = torch.zeros((ctx.height, ctx.width, 3), dtype=torch.float32, device="cuda:0")
    b = torch.zeros_like(a)
    ctx.render_accumulate(params, a)
    torch.cuda.synchronize()
    free_before = torch.cuda.mem_get_info()[0]          # hipMemGetInfo: sees the library's own hipMallocs too
    ctx.render_accumulate_noalloc(params, b)
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == free_before, "the no-alloc entry point changed the device's free memory"
    return a.cpu().numpy(), b.cpu().numpy()


def test_noalloc_matches_the_allocating_entry_point_twice():
    import os
    ctx = rt.Context(0)
    try:
        # first call: 485 spheres + 6,400 triangles at 240x136, 8 samples (261,120 paths = 48 MB of state, no lights: deterministic)
        sc = scenegen.scene_c5(width=240, height=136, nu=80, nv=40)
        ctx.upload(rt.Scene.from_objects(sc["width"], sc["height"], sc["background"], sc["camera"], sc["objects"]))
        p1 = rt.RenderParams("jitter:4", depth=5, seed=3, sample_begin=2, sample_end=10)
        a, b = _both(ctx, p1)
        assert a.any() and np.array_equal(a, b)
        # second call, smaller input, on the same scratch: scenes/cube (two lights: float atomics) at 2 samples of a
        # third of the tiles; the scratch still holds the first call's rays beyond the new queue ends
        ctx.upload(rt.Scene.load(os.path.join(GOLDEN, "scenes", "cube", "cube.yaml"), base_dir=GOLDEN))
        p2 = rt.RenderParams("jitter:2", depth=5, seed=4, sample_begin=1, sample_end=3, tile_index=1, tile_count=3)
        a, b = _both(ctx, p2)
        assert a.any() and np.allclose(a, b, rtol=1e-5, atol=1e-6)
        # and the first input again: what the second call left behind does not matter
        ctx.upload(rt.Scene.from_objects(sc["width"], sc["height"], sc["background"], sc["camera"], sc["objects"]))
        a, b = _both(ctx, p1)
        assert np.array_equal(a, b)
        st = ctx.stats()
        assert st["queue_overflow"] == 0
    finally:
        ctx.close()


def test_noalloc_limits_are_errors():
    import torch
    ctx = rt.Context(0)
    try:
        sc = scenegen.scene_c3(width=640, height=360)
        ctx.upload(rt.Scene.from_objects(sc["width"], sc["height"], sc["background"], sc["camera"], sc["objects"]))
        acc = torch.zeros((ctx.height, ctx.width, 3), dtype=torch.float32, device="cuda:0")
        # 640x360 = 230,400 paths per sample: one sample fits the 64 MiB scratch (184 B per path), two do not
        ctx.render_accumulate_noalloc(rt.RenderParams("jitter:2", depth=3, sample_begin=0, sample_end=1), acc)
        with pytest.raises(rt.RtError, match="static scratch"):
            ctx.render_accumulate_noalloc(rt.RenderParams("jitter:2", depth=3, sample_begin=0, sample_end=2), acc)
        with pytest.raises(rt.RtError, match="ONE pass"):
            ctx.render_accumulate_noalloc(rt.RenderParams("jitter:2", depth=3, sample_begin=0, sample_end=1, max_paths_per_pass=64 * 100), acc)
        with pytest.raises(rt.RtError, match="TIMING"):
            ctx.render_accumulate_noalloc(rt.RenderParams("jitter:2", depth=3, sample_begin=0, sample_end=1, timing=True), acc)
        torch.cuda.synchronize()
        # the whole frame in four calls of one sample each equals the allocating call over all four
        acc.zero_()
        for s0 in range(4):
            ctx.render_accumulate_noalloc(rt.RenderParams("jitter:2", depth=3, seed=2, sample_begin=s0, sample_end=s0 + 1), acc)
        ref = torch.zeros_like(acc)
        ctx.render_accumulate(rt.RenderParams("jitter:2", depth=3, seed=2), ref)
        torch.cuda.synchronize()
        assert torch.allclose(acc, ref, rtol=1e-5, atol=1e-6)
    finally:
        ctx.close()
