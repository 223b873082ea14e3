#!/usr/bin/env python3
"""Two half-passes on two streams against one pass on one stream: does the memory-bound k_shade of one overlap with the
VALU-bound k_closest_hit of the other?  usage: tools/overlap2_probe.py [c5|c4|c3] [spp]
Two contexts hold the same scene; each renders half of the samples into its own accumulator on its own stream."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
rt = importlib.import_module("raytracer-rust_amd")
scenegen = importlib.import_module("raytracer-rust_amd.scenegen")
w = sys.argv[1] if len(sys.argv) > 1 else "c5"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 64
grid = {"c5": 64, "c4": 8, "c3": 16}[w]
sc = {"c5": scenegen.scene_c5, "c4": scenegen.scene_c4, "c3": scenegen.scene_c3}[w]()
W, H = sc["width"], sc["height"]
scene = rt.Scene.from_objects(W, H, sc["background"], sc["camera"], sc["objects"])
dev = torch.device("cuda:0")
ctxs = [rt.Context(0).upload(scene) for _ in range(2)]
accs = [torch.zeros((H, W, 3), dtype=torch.float32, device=dev) for _ in range(2)]
streams = [torch.cuda.Stream(dev) for _ in range(2)]
def params(sb, se):
    return rt.RenderParams(rt.SuperSampling("Jitter", grid), depth=5, seed=1234, sample_begin=sb, sample_end=se)
def one(n=4):
    torch.cuda.synchronize()
    t0 = time.time()
    for k in range(n):
        ctxs[0].render_accumulate(params(0, spp), accs[0], streams[0].cuda_stream)
    torch.cuda.synchronize()
    return (time.time() - t0) / n * 1e3
def two(n=4, parts=2):
    torch.cuda.synchronize()
    t0 = time.time()
    h = spp // 2
    for k in range(n):
        ctxs[0].render_accumulate(params(0, h), accs[0], streams[0].cuda_stream)
        ctxs[1].render_accumulate(params(h, spp), accs[1], streams[1].cuda_stream)
    torch.cuda.synchronize()
    return (time.time() - t0) / n * 1e3
one(1); two(1)
for rep in range(2):
    print("%s %d spp: one stream %.2f ms, two streams of %d spp %.2f ms" % (w, spp, one(), spp // 2, two()), flush=True)
