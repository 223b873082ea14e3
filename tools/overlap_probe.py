#!/usr/bin/env python3
"""Probe: do two independent renders on two HIP streams overlap (VALU-bound closest-hit of one with the
stream-bound shade of the other)?  Compares 2 x sequential with concurrent."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
os.environ.setdefault("RTAMD_LIB", os.environ.get("RTAMD_LIB", ""))
rt = importlib.import_module("raytracer-rust_amd")
scenegen = importlib.import_module("raytracer-rust_amd.scenegen")
dev = torch.device("cuda", 0)
sc = scenegen.scene_c5(1920, 1080, 1000, 500)
scene = rt.Scene.from_objects(sc["width"], sc["height"], sc["background"], sc["camera"], sc["objects"])
spp = int(os.environ.get("SPP", "32"))
ctxs, accs, streams = [], [], []
for k in range(2):
    c = rt.Context(0); c.upload(scene); ctxs.append(c)
    accs.append(torch.zeros((1080, 1920, 3), dtype=torch.float32, device=dev))
    streams.append(torch.cuda.Stream(dev, priority=(-1 if k else 0)))
def params(k, step):
    cell0 = ((step * 2 + k) * spp) % 4096
    return rt.RenderParams(rt.SuperSampling("Jitter", 64), depth=5, seed=1234, sample_begin=cell0, sample_end=cell0 + spp)
def run(concurrent, steps=4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(steps):
        for k in range(2):
            st = streams[k] if concurrent else streams[0]
            ctxs[k].render_accumulate(params(k, s), accs[k], st.cuda_stream)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3
run(False, 1); run(True, 1)
for rep in range(2):
    print("sequential %.2f ms per pair   concurrent %.2f ms per pair" % (run(False), run(True)), flush=True)
