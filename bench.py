#!/usr/bin/env python3
"""Headline benchmark: Mrays/s of the wavefront render path at 1080p on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by torch.distributed.run, one rank per GPU, RCCL = backend "nccl")

Workload (BASELINE.json / SURVEY.md 8d): the combined scene of config C5 -- ~485 RTIOW spheres + the 1,000,000-triangle torus --
at 1920x1080, depth 5, jittered 64x64 sample grid (`jitter:64`, 4096 spp).  One STEP = one pass of the hot path over one batch:
512 samples per pixel of the full frame on every rank -- C5's per-GPU share of the 4096-spp frame, a disjoint cell range of the
64x64 grid per rank and step -- followed by the RCCL sum-reduce of the f32 accumulators to rank 0 and the resolve there.  At
N = 8 a step IS the C5 frame.  Per-GPU work is fixed as N grows => weak scaling.  `--split tiles` deals the 8x8 tiles of ONE
frame to the ranks instead (fixed total work => strong scaling; the C4 configuration).  At N = 1 the same JSON line also carries,
under "other_workloads", the other configurations with their OWN sampling: C3 (spheres only, 1080p, `jitter:16`: a step is the
256-spp frame), C4 (torus + plane, 3840x2160, `jitter:8`: a step is the 64-spp frame), C3 at depth 50 and C1 (scenes/spheres).

value = rays actually traced by all ranks (primary + scattered + shadow) / wall time of the K timed steps.  Rays are counted on
the device by the queue counters (rt_get_stats).  The timed steps carry no instrumentation; the K steps are then run ONCE MORE
with HIP events around every closest-hit launch (RT_FLAG_TIMING, on the launch stream) -- `roofline` is built on those.

  python bench.py --render OUT.png     the north-star artefact: C5 at 1920x1080, 1024 spp (jitter:32), through
                                       rt_render_image; writes the PNG and prints a JSON line with time-to-image
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0        # HBM3E 8.0 TB/s spec
L2_PEAK_GBS = 34500.0        # aggregate L2 bandwidth
VALU_PEAK_TFLOPS = 157.3     # peak FP32 vector
REHEARSAL = os.environ.get("RTAMD_BENCH_REHEARSAL", "") == "gloo"    # see main()
N_SIMD = 1024                # 256 CUs x 4 SIMD-32
CLOCK_GHZ = 2.4
VALU_PIPE_PEAK_GCYC = N_SIMD * CLOCK_GHZ     # VALU pipe cycles available per second, in 1e9

# name, metric, jitter grid of the configuration, samples per pixel of one step
WORKLOADS = {
    "c3": ("C3 RTIOW ~485 spheres", "Mrays/s (whole node) at 1080p", 16, 256),
    "c4": ("C4 1M-triangle torus + plane", "Mrays/s (whole node) at 4K", 8, 64),
    "c5": ("C5 ~485 spheres + 1M-triangle torus", "Mrays/s (whole node) at 1080p", 64, 512),
}
PMC_PROFILE = "r03_pmc.json"
ISA_PROFILE = "r03_isa_closest_hit.json"
KERNEL_SOURCES = ("rt_kernels.hip", "rt_device.h", "Makefile")


def kernel_hash():
    """sha256 over the kernel sources the PMC / ISA profiles were collected on (stored in profiles/*pmc.json)."""
    import hashlib
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        try:
            h.update(open(os.path.join(ROOT, "raytracer-rust_amd", "csrc", name), "rb").read())
        except OSError:
            return None
    return h.hexdigest()[:16]


def make_scene(scenegen, workload, small):
    if workload == "c3":
        return scenegen.scene_c3(*((320, 180) if small else (1920, 1080)))
    if workload == "c4":
        return scenegen.scene_c4(*((384, 216, 100, 50) if small else (3840, 2160, 1000, 500)))
    return scenegen.scene_c5(*((320, 180, 100, 50) if small else (1920, 1080, 1000, 500)))


def load_json(name):
    try:
        return json.load(open(os.path.join(ROOT, "profiles", name)))
    except Exception:
        return None


def isa_model(cst, launches):
    """VALU pipe cycles per closest-hit launch from the committed ISA histogram (tools/isa_histogram.py) times what the
    counters build saw the waves do: `always` counts only the straight-line part of each region (a lower bound), `total`
    every instruction of the region (an upper bound).  The PMC-measured figure lies between the two."""
    h = load_json(ISA_PROFILE)
    if not h or not cst.get("wave_node_steps"):
        return None
    r = h["regions"]
    steps = {"node": cst["wave_node_steps"], "leaf": cst["wave_leaf_steps"], "refill": cst["wave_refills"]}

    def cyc(which):
        g = lambda name: r.get(name, {}).get("valu_cycles_" + which, 0)
        leaf = g("leaf") + g("leaf_end") + (g("leaf_tri") + g("leaf_sphere") if which == "total" else min(g("leaf_tri"), g("leaf_sphere")))
        refill = g("refill") + g("refill_take")
        ctl = g("control")
        return (steps["node"] * (g("node") + ctl) + steps["leaf"] * (leaf + ctl) + steps["refill"] * refill) / max(1, launches)

    return {"valu_pipe_cycles_per_launch_lower": cyc("always"), "valu_pipe_cycles_per_launch_upper": cyc("total"),
            "wave_node_steps_per_launch": steps["node"] / max(1, launches), "wave_leaf_steps_per_launch": steps["leaf"] / max(1, launches),
            "wave_refills_per_launch": steps["refill"] / max(1, launches), "source": "profiles/%s x rt_get_stats counters" % ISA_PROFILE}


def run_workload(rt, scenegen, args, workload, rank, world, dev, steps, warmup, split):
    """Warm-up, one counters step, `steps` timed steps of `workload`, then the same steps once more with HIP events around
    the closest-hit launches; returns the result dict (rank 0) or None."""
    import torch
    import torch.distributed as dist
    D = importlib.import_module("raytracer-rust_amd.distributed")

    sc = make_scene(scenegen, workload, args.small)
    W, H = sc["width"], sc["height"]
    name, metric, grid_n, spp_default = WORKLOADS[workload]
    spp = args.spp_per_step or spp_default
    t0 = time.time()
    scene = rt.Scene.from_objects(W, H, sc["background"], sc["camera"], sc["objects"])
    ctx = rt.Context(dev.index)
    ctx.set_bvh_builder(args.bvh)
    ctx.upload(scene)
    upload_s = time.time() - t0

    accum = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
    total = torch.zeros_like(accum)
    stream = torch.cuda.current_stream(dev)
    reduce_ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)] if world > 1 and not REHEARSAL else None
    reduce_ms = []

    def step(k, timing=False, counters=False, time_reduce=False):
        """One batch: spp samples/pixel on this rank (or this rank's tiles), reduce, resolve."""
        sb, se, ti, tc = D.bench_step_params(k, rank, world, spp, grid_n, split)
        p = rt.RenderParams(rt.SuperSampling("Jitter", grid_n), depth=args.depth, seed=1234, sample_begin=sb, sample_end=se,
                            tile_index=ti, tile_count=tc, timing=timing, counters=counters, max_paths_per_pass=args.paths_per_pass)
        accum.zero_()
        ctx.render_accumulate(p, accum, stream.cuda_stream)
        if world > 1:
            # the exchange step of the sample split: sum the partial accumulators on rank 0 over xGMI
            # (tile split: the ranks' tiles are disjoint, the same reduce assembles the frame)
            total.copy_(accum)
            if REHEARSAL:
                host = total.cpu()                      # gloo has no reduce for device tensors
                dist.reduce(host, dst=0, op=dist.ReduceOp.SUM)
                total.copy_(host)
            else:
                if time_reduce:
                    reduce_ev[0].record(stream)
                dist.reduce(total, dst=0, op=dist.ReduceOp.SUM)
                if time_reduce:
                    reduce_ev[1].record(stream)
                    reduce_ev[1].synchronize()
                    reduce_ms.append(reduce_ev[0].elapsed_time(reduce_ev[1]))
            src = total
        else:
            src = accum
        if rank == 0:
            return ctx.resolve(src, spp * (world if split == "spp" else 1), stream=stream.cuda_stream)
        return None

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # ---------------------------------------------------------------- warmup, counters
    for k in range(warmup):
        step(k)
    barrier()
    ctx.stats_reset()
    step(0, counters=True, timing=True)                               # untimed: what the rays and the waves do
    cst = ctx.stats()
    nodes_per_ray = cst["nodes_visited"] / max(1, cst["rays_total"])
    prims_per_ray = cst["prims_tested"] / max(1, cst["rays_total"])
    counter_launches = max(1, cst["hit_launches"])
    barrier()
    ctx.stats_reset()

    # ---------------------------------------------------------------- timed region (no instrumentation inside)
    barrier()
    t0 = time.perf_counter()
    for k in range(steps):
        step(warmup + k)
    torch.cuda.synchronize(dev)
    own_elapsed = time.perf_counter() - t0        # this rank alone, before it waits for the others
    barrier()
    elapsed = time.perf_counter() - t0
    st_timed = ctx.stats()
    # ---------------------------------------------------------------- the same steps once more, instrumented
    ctx.stats_reset()
    for k in range(steps):
        step(warmup + k, timing=True, time_reduce=True)
    barrier()
    st = ctx.stats()                              # hit_ms / hit_launches: HIP events on the launch stream
    cdev = "cpu" if REHEARSAL else dev
    t_el = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
    t_own = torch.tensor([own_elapsed, -own_elapsed], dtype=torch.float64, device=cdev)
    rays = torch.tensor([float(st_timed["rays_total"]), float(st_timed["paths"])], dtype=torch.float64, device=cdev)
    if world > 1:
        dist.all_reduce(t_el, op=dist.ReduceOp.MAX)
        dist.all_reduce(t_own, op=dist.ReduceOp.MAX)
        dist.all_reduce(rays, op=dist.ReduceOp.SUM)
    elapsed = float(t_el.item())
    total_rays, total_paths = float(rays[0].item()), float(rays[1].item())

    out = None
    if rank == 0:
        import numpy as np
        n_planes = len([o for o in sc["objects"] if o["type"] == "Plane"])
        mesh_objects = [i for i, o in enumerate(sc["objects"]) if o["type"] == "Mesh"]
        mesh_share = None
        if not args.no_mesh_share:      # (one more closest-hit launch, outside the timed region; the profiling scripts switch it off)
            hits = ctx.primary_hits(0.5, 0.5)
            mesh_share = float(np.isin(hits["object_id"], mesh_objects).mean()) if mesh_objects else 0.0
        # algorithmic bytes of one closest-hit query ("k_closest_hit"): 32 B ray words a,b in (16 B for a primary
        # ray), 8 B hit record out, 64 B per BVH node visited, 48 B per primitive tested, 32 B per plane
        r0 = float(st["rays_per_iteration"][0]) / max(1.0, float(st["rays_total"]))
        bytes_per_ray = (16.0 * r0 + 32.0 * (1.0 - r0)) + 8.0 + 64.0 * nodes_per_ray + 48.0 * prims_per_ray + 32.0 * n_planes
        flops_per_ray = 40.0 * nodes_per_ray + 45.0 * prims_per_ray + 10.0 * n_planes
        launches = max(1, st["hit_launches"])
        rays_per_launch = st["rays_total"] / launches
        ms_per_launch = st["hit_ms"] / launches
        sec = ms_per_launch * 1e-3
        request_rate = rays_per_launch * bytes_per_ray / sec / 1e9 if sec > 0 else 0.0
        valu_tflops = rays_per_launch * flops_per_ray / sec / 1e12 if sec > 0 else 0.0
        # PMC figures of the same command (separate rocprofv3 --pmc passes, tools/pmc_profile.sh -> profiles/r03_pmc.json),
        # stored per ray so that they scale with the live ray count of this run, together with the hash of the kernel sources
        # they were collected on: a profile of other kernels is flagged, not silently used
        pmc_all = load_json(PMC_PROFILE) or {}
        pmc = pmc_all.get(workload) if not args.small else None
        khash = kernel_hash()
        pmc_stale = bool(pmc) and (khash is None or pmc.get("kernel_hash") != khash)
        model = isa_model(cst, counter_launches)
        roof = {"bound": "valu_issue", "kernel": "k_closest_hit", "unit": "G VALU-pipe-cycles/s", "peak": VALU_PIPE_PEAK_GCYC,
                "achieved": None, "frac": None, "traffic": None,
                # SURVEY.md 8d's own two fractions for the same kernel and launches, beside the binding one:
                "fp32_flops": {"achieved": valu_tflops, "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": valu_tflops / VALU_PEAK_TFLOPS,
                               "flops_per_ray": flops_per_ray,
                               "note": "SURVEY 8d flop model (40 per node, 45 per primitive, 10 per plane) x rays / launch time: counts the "
                                       "arithmetic of the tests only, about a tenth of the VALU instructions issued"},
                "hbm_counter": None,
                "pmc_stale": pmc_stale, "kernel_hash": khash,
                "launches": st["hit_launches"], "ms_per_launch": ms_per_launch, "rays_per_launch": rays_per_launch,
                "hit_ms_per_step": st["hit_ms"] / steps,
                "nodes_per_ray": nodes_per_ray, "prims_per_ray": prims_per_ray,
                "self_hit_rays_share": cst["self_hit_rays"] / max(1, cst["rays_total"]),
                "lanes_with_work_per_step": cst["step_lanes"] / max(1, 64 * (cst["wave_node_steps"] + cst["wave_leaf_steps"])),
                "hit_kernel_share_of_step": st["hit_ms"] * 1e-3 / elapsed,
                "rays_per_iteration_per_step": [x / steps for x in st["rays_per_iteration"][:8]],
                "isa_model": model}
        if pmc and sec > 0:
            cyc = pmc["valu_pipe_cycles_per_ray"] * rays_per_launch
            roof["achieved"] = cyc / sec / 1e9
            roof["frac"] = roof["achieved"] / VALU_PIPE_PEAK_GCYC
            roof["traffic"] = pmc["fetch_write_bytes_per_ray"] * rays_per_launch
            roof["valu_insts_per_ray_pmc"] = pmc["valu_insts_per_ray"]
            roof["lane_utilisation_pmc"] = pmc.get("valu_lane_utilisation")
            roof["wave_time_split_pmc"] = pmc.get("wave_time_split")
            roof["hbm_counter"] = {"bound": "hbm", "achieved": roof["traffic"] / sec / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": roof["traffic"] / sec / 1e9 / HBM_PEAK_GBS,
                                   "note": "FETCH_SIZE + WRITE_SIZE per launch (separate --pmc passes): L2-miss traffic, Infinity Cache hits "
                                           "included (MI355X_MICROARCH.md), so an upper bound of the HBM bytes"}
        elif model and sec > 0:
            # no PMC file for this workload: the ISA-histogram lower bound (still cannot exceed 1)
            roof["achieved"] = model["valu_pipe_cycles_per_launch_lower"] * (rays_per_launch / max(1.0, cst["rays_total"] / max(1, counter_launches))) / sec / 1e9
            roof["frac"] = roof["achieved"] / VALU_PIPE_PEAK_GCYC
        roof["l2_request_rate"] = {"achieved": request_rate, "peak": L2_PEAK_GBS, "unit": "GB/s", "frac": request_rate / L2_PEAK_GBS,
                                   "bytes_per_ray": bytes_per_ray, "frac_of_hbm_peak": request_rate / HBM_PEAK_GBS,
                                   "note": "algorithmic bytes (SURVEY 8d / DESIGN.md section 4) / launch time: a request rate served by L1/L2/"
                                           "Infinity Cache, priced against the aggregate L2 bandwidth; against the HBM peak it exceeds 1 "
                                           "because the scene is cache-resident"}
        roof["note"] = ("k_closest_hit is bound by VALU issue and latency: frac = VALU pipe cycles issued (2 per wave64 instruction, "
                        "4 for f64 and transcendental; SQ_INSTS_VALU* of the same command, profiles/%s) / (1024 SIMDs x "
                        "launch time x 2.4 GHz); launch time = HIP events on the launch stream over the K steps re-run right after "
                        "the timed region; the scene is served by the caches, HBM-side traffic is `hbm_counter`" % PMC_PROFILE)
        out = {
            "metric": metric, "value": total_rays / elapsed / 1e6, "unit": "Mrays/s",
            "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": elapsed / steps * 1e3,
            "higher_is_better": True, "scaling": "weak" if split == "spp" else "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s, %dx%d, depth %d, %d spp/step%s of the jitter:%d grid, %s split%s" %
                                   (name, W, H, args.depth, spp, "/GPU" if split == "spp" else "", grid_n, split,
                                    " [SMALL: not a measurement]" if args.small else "") + (" [REHEARSAL over gloo on shared GPUs: not a measurement]" if REHEARSAL else ""),
                       "paths_per_step": total_paths / steps, "rays_per_path": total_rays / max(1.0, total_paths),
                       "samples_per_s": total_paths / elapsed, "mesh_share_of_primary_hits": mesh_share,
                       "self_hit_rays_share": roof["self_hit_rays_share"],
                       "mesh_winding": "outward (lit torus)", "bvh_nodes": st["bvh_nodes"], "bvh_depth": st["bvh_depth"],
                       "bvh_build_ms": st["bvh_build_ms"], "bvh_builder": args.bvh, "scene_upload_s": upload_s},
            "roofline": roof,
        }
        if world > 1:
            # what a scaling loss would be made of: the slowest and the fastest rank's own time per step (before the barrier),
            # and the exchange step alone (HIP events around the RCCL reduce on rank 0, instrumented steps)
            out["ranks"] = {"ms_per_step_max": float(t_own[0].item()) / steps * 1e3, "ms_per_step_min": -float(t_own[1].item()) / steps * 1e3,
                            "reduce_ms_rank0": (sum(reduce_ms) / len(reduce_ms)) if reduce_ms else None,
                            "reduce_bytes": int(W) * int(H) * 3 * 4}
    ctx.close()
    del accum, total
    torch.cuda.empty_cache()
    return out, sc


def run_c1(rt, dev, frames=20):
    """C1: scenes/spheres as it is, 500x500, `--ssaa uniform:2` (= jitter 2x2), depth 5; the reference's own
    CPU-runnable case.  1 M paths per frame: a launch-bound job on this part, reported for completeness."""
    import torch
    golden = os.path.join(ROOT, "tests", "golden")
    ctx = rt.Context(dev.index)
    ctx.upload(rt.Scene.load(os.path.join(golden, "scenes", "spheres", "spheres.yaml"), base_dir=golden))
    accum = torch.zeros((500, 500, 3), dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream(dev)
    p = lambda k: rt.RenderParams("jitter:2", depth=5, seed=100 + k)
    ctx.render_accumulate(p(0), accum, stream.cuda_stream)
    torch.cuda.synchronize(dev)
    ctx.stats_reset()
    t0 = time.perf_counter()
    for k in range(frames):
        accum.zero_()
        ctx.render_accumulate(p(k), accum, stream.cuda_stream)
        ctx.resolve(accum, 4, stream=stream.cuda_stream)
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    st = ctx.stats()
    ctx.close()
    return {"workload": "C1 scenes/spheres, 500x500, jitter:2 (4 spp), depth 5, one frame per step", "value": st["rays_total"] / dt / 1e6, "unit": "Mrays/s",
            "ms_per_step": dt / frames * 1e3, "steps": frames, "samples_per_s": st["paths"] / dt, "rays_per_path": st["rays_total"] / max(1, st["paths"])}


def render_north_star(rt, scenegen, out_png=None, n=32, depth=5, seed=2026, device=0, small=False):
    """BASELINE.json's target artefact: C5 at 1920x1080, n*n = 1024 spp, through rt_render_image (host outputs).
    Returns (accum (H,W,3) float32 sample sums, rgb8, info dict).  No oracle here: the gate on this very frame is
    tests/test_gpu_northstar.py, which calls this function."""
    sc = make_scene(scenegen, "c5", small)
    t0 = time.perf_counter()
    scene = rt.Scene.from_objects(sc["width"], sc["height"], sc["background"], sc["camera"], sc["objects"])
    ctx = rt.Context(device)
    ctx.upload(scene)
    t_upload = time.perf_counter() - t0
    ctx.render_image(rt.RenderParams("jitter:1", depth=depth, seed=seed))         # first-launch costs (code objects) are not the image's
    ctx.stats_reset()
    t1 = time.perf_counter()
    acc, rgb = ctx.render_image(rt.RenderParams("jitter:%d" % n, depth=depth, seed=seed))
    t_render = time.perf_counter() - t1
    st = ctx.stats()
    t2 = time.perf_counter()
    if out_png:
        rt.write_png(out_png, rgb)
    t_png = time.perf_counter() - t2
    info = {"artefact": "C5 (%d spheres + 1,000,000-triangle torus), %dx%d, %d spp (jitter:%d), depth %d" %
                        (len([o for o in sc["objects"] if o["type"] == "Sphere"]), sc["width"], sc["height"], n * n, n, depth),
            "time_to_image_s": t_render, "scene_upload_s": t_upload, "png_write_s": t_png, "rays": st["rays_total"], "paths": st["paths"],
            "mrays_per_s": st["rays_total"] / t_render / 1e6, "msamples_per_s": st["paths"] / t_render / 1e6,
            "note": "time_to_image = rt_render_image wall time: all passes, resolve and the copies of the f32 sums (24.9 MB) and the "
                    "RGB8 frame (6.2 MB) to the host", "png": out_png}
    ctx.close()
    return acc, rgb, sc, info


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c5", choices=["c3", "c4", "c5"])
    ap.add_argument("--spp-per-step", type=int, default=0, help="samples per pixel of one step (0 = the configuration's own: C5 512, C3 256, C4 64)")
    ap.add_argument("--depth", type=int, default=5)
    ap.add_argument("--split", default=None, choices=["spp", "tiles"])
    ap.add_argument("--paths-per-pass", type=int, default=0, help="rt_render_params.max_paths_per_pass (0 = library default)")
    ap.add_argument("--bvh", default="host", choices=["host", "gpu"], help="BVH builder: host binned SAH (default) or GPU linear BVH")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-workloads", action="store_true", help="skip the C3 / C4 legs of the N = 1 line")
    ap.add_argument("--no-mesh-share", action="store_true", help="skip the rt_primary_hits probe that reports the mesh's share of the primary hits")
    ap.add_argument("--cpu-sample-step", type=int, default=5, help="cpu_baseline (C5) traces every n-th pixel and row")
    ap.add_argument("--small", action="store_true", help="tiny configuration for plumbing checks (not a valid measurement)")
    ap.add_argument("--render", default=None, metavar="OUT.png", help="render the north-star frame instead of benchmarking")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rt = importlib.import_module("raytracer-rust_amd")
    scenegen = importlib.import_module("raytracer-rust_amd.scenegen")

    if args.render:
        _, _, _, info = render_north_star(rt, scenegen, args.render, small=args.small)
        print(json.dumps(info), flush=True)
        return

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d" % (args.gpus, world, args.gpus))
    if REHEARSAL:
        # rehearsal of the N > 1 control flow on a box with fewer GPUs than ranks: every rank on a GPU it shares, the
        # collectives over gloo on host copies.  NOT a measurement (the JSON line says so); the driver's runs use RCCL.
        local_rank %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if REHEARSAL:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=dev)

    split = args.split or ("tiles" if args.workload == "c4" else "spp")
    if world > 1:
        # the host BVH builder threads over min(16, cores): with several ranks on one host each takes its share
        os.environ.setdefault("RTAMD_BVH_THREADS", str(max(1, min(16, (os.cpu_count() or 16) // world))))
    out, sc = run_workload(rt, scenegen, args, args.workload, rank, world, dev, args.steps, args.warmup, split)
    if rank == 0:
        if world == 1 and not args.no_other_workloads:
            # the other BASELINE configurations with their own sampling, driver-timed in the same run (fewer steps: they are
            # context, not the headline)
            others = {}
            spp0 = args.spp_per_step
            args.spp_per_step = 0
            for w in ("c3", "c4", "c3_depth50"):
                if w == args.workload:
                    continue
                wl, depth0 = w, args.depth
                if w == "c3_depth50":                     # C3 at the reference's default depth 5 and at RTIOW's 50
                    wl, args.depth = "c3", 50
                o, _ = run_workload(rt, scenegen, args, wl, 0, 1, dev, max(2, args.steps // 2), 1, "tiles" if wl == "c4" else "spp")
                args.depth = depth0
                r = o["roofline"]
                others[w] = {"workload": o["config"]["workload"], "value": o["value"], "unit": o["unit"], "ms_per_step": o["ms_per_step"],
                             "steps": o["steps"], "samples_per_s": o["config"]["samples_per_s"], "rays_per_path": o["config"]["rays_per_path"],
                             "mesh_share_of_primary_hits": o["config"]["mesh_share_of_primary_hits"],
                             "closest_hit_ms_per_launch": r["ms_per_launch"], "rays_per_launch": r["rays_per_launch"],
                             "nodes_per_ray": r["nodes_per_ray"], "prims_per_ray": r["prims_per_ray"],
                             "roofline_frac": r["frac"], "pmc_stale": r["pmc_stale"], "fp32_flops_frac": r["fp32_flops"]["frac"],
                             "hit_kernel_share_of_step": r["hit_kernel_share_of_step"]}
            args.spp_per_step = spp0
            if not args.small:
                others["c1"] = run_c1(rt, dev)
            out["other_workloads"] = others
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(sc, scenegen, args)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(sc, scenegen, args):
    """The oracle (the reference's own algorithm: linear object scan, brute-force triangles behind one AABB, f64; its optional
    cull stays OFF) timed on this box's host cores on bounded samples: the headline workload (every n-th pixel and row, 1 spp),
    C1 in full (500x500 at 4 spp) and a labelled crop of C3."""
    from oracle import oracle_py as O
    threads = min(os.cpu_count() or 1, 16)        # the GPU box grants 16 cores per GPU

    def timed(osc, w, h, step, depth, N=1):
        t0 = time.perf_counter()
        r = osc.render_crop((0, w, 0, h), mode="jitter", N=N, depth=depth, seed=7, threads=threads, step=step)
        dt = time.perf_counter() - t0
        npx = len(range(0, w, step)) * len(range(0, h, step))
        return r["rays"] / dt / 1e6, npx, r["rays"], dt

    step = max(1, args.cpu_sample_step if not args.small else 4)
    v, npx, rays, dt = timed(O.OracleScene.from_dict(scenegen.to_oracle_dict(sc)), sc["width"], sc["height"], step, args.depth)
    out = {"value": v, "unit": "Mrays/s", "cores": threads, "kind": "port",
           "sample": "every %dth pixel and row of the same frame (%d pixels), 1 spp, depth %d: %d rays in %.1f s" % (step, npx, args.depth, rays, dt)}
    if not args.small:
        golden = os.path.join(ROOT, "tests", "golden")
        v, npx, rays, dt = timed(O.OracleScene.from_yaml(os.path.join(golden, "scenes", "spheres", "spheres.yaml"), base_dir=golden), 500, 500, 1, 5, N=16)
        others = {"c1": {"value": v, "unit": "Mrays/s", "cores": threads, "kind": "port",
                         "sample": "C1's whole frame (scenes/spheres 500x500, depth 5) at jitter:16 = 256 spp, 64 x the 4 spp of C1 (which take "
                                   "about 10 ms here): %d rays in %.2f s" % (rays, dt)}}
        c3 = make_scene(scenegen, "c3", False)
        v, npx, rays, dt = timed(O.OracleScene.from_dict(scenegen.to_oracle_dict(c3)), c3["width"], c3["height"], 1, args.depth, N=4)
        others["c3"] = {"value": v, "unit": "Mrays/s", "cores": threads, "kind": "port",
                        "sample": "C3's whole 1920x1080 frame at jitter:4 = 16 of its 256 spp, depth %d: %d rays in %.2f s" % (args.depth, rays, dt)}
        out["others"] = others
    return out


if __name__ == "__main__":
    main()
