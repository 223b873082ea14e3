#!/usr/bin/env python3
"""Counters of k_closest_hit per iteration: renders C5 (or --workload c3/c4) at 1080p, 16 spp, with depth 1, 2, ... 5 and the counters
build, and prints the differences -- nodes visited, primitives tested, wave steps and lanes with work of each iteration's rays."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rt = importlib.import_module("raytracer-rust_amd")
scenegen = importlib.import_module("raytracer-rust_amd.scenegen")
w = sys.argv[sys.argv.index("--workload") + 1] if "--workload" in sys.argv else "c5"
sc = {"c3": lambda: scenegen.scene_c3(1920, 1080), "c4": lambda: scenegen.scene_c4(1920, 1080, 1000, 500), "c5": lambda: scenegen.scene_c5(1920, 1080, 1000, 500)}[w]()
ctx = rt.Context(0)
ctx.upload(rt.Scene.from_objects(sc["width"], sc["height"], sc["background"], sc["camera"], sc["objects"]))
prev = None
keys = ("rays_total", "nodes_visited", "prims_tested", "wave_node_steps", "wave_leaf_steps", "wave_refills", "step_lanes", "self_hit_rays")
print("%2s %10s %7s %7s %9s %9s %8s %6s %6s" % ("it", "rays", "nodes/r", "prims/r", "nodesteps", "leafsteps", "refills", "lanes", "self"))
for depth in range(1, 6):
    ctx.stats_reset()
    ctx.render_image(rt.RenderParams("jitter:4", depth=depth, seed=5, counters=True))
    st = ctx.stats()
    cur = {k: st[k] for k in keys}
    d = {k: cur[k] - (prev[k] if prev else 0) for k in keys}
    prev = cur
    steps = d["wave_node_steps"] + d["wave_leaf_steps"]
    print("%2d %10d %7.2f %7.2f %9d %9d %8d %6.3f %6.3f" % (depth - 1, d["rays_total"], d["nodes_visited"] / d["rays_total"], d["prims_tested"] / d["rays_total"],
                                                       d["wave_node_steps"], d["wave_leaf_steps"], d["wave_refills"], d["step_lanes"] / max(1, 64 * steps), d["self_hit_rays"] / d["rays_total"]))
ctx.close()
