#!/usr/bin/env python3
"""Per-launch view of the last bench step of a tools/kernel_stats.sh run: tools/launch_trace.py gpurun_out/kstats_<tag>

For every iteration: rays in the queue (bench line's rays_per_iteration_per_step), the closest-hit and shade launch times,
ns per ray, and for k_shade the algorithmic bytes per second (queue record in + hit record + queue record out for the
rays that go on, 16-byte radiance store for those that end; the shading records it gathers are not counted)."""
import csv, glob, json, os, sys

d = sys.argv[1]
trace = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(trace)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
line = [json.loads(l) for l in open(os.path.join(d, "bench.log")) if l.startswith('{"metric"')][0]
per_it = line["roofline"]["rays_per_iteration_per_step"]
# the last TIMED step: bench.py's counters step (k_closest_hit<true>) comes after it and is left out
gens = [i for i, r in enumerate(rows) if "k_raygen" in r["Kernel_Name"]] + [len(rows)]
steps = [rows[a:b] for a, b in zip(gens[:-1], gens[1:])]
step = [s for s in steps if any("k_closest_hit<false" in r["Kernel_Name"] for r in s)][-1]
step = step[:max(i for i, r in enumerate(step) if "k_accumulate" in r["Kernel_Name"] or "k_shade" in r["Kernel_Name"]) + 1]
hit = [r for r in step if "k_closest_hit" in r["Kernel_Name"]]
shade = [r for r in step if "k_shade" in r["Kernel_Name"]]
passes = max(1, round(line["roofline"]["launches"] / (line["steps"] * max(1, len(hit)))))   # a step of C4 is several passes: the trace shows the last one
per_it = [x / passes for x in per_it]
dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
print("last pass of the last timed step (%d pass%s per step); raygen %.0f us" % (passes, "" if passes == 1 else "es", dur(step[0])))
print(" it        rays   hit us  ns/ray  shade us  ns/ray  shade GB/s (algorithmic)")
for it, (h, s) in enumerate(zip(hit, shade)):
    n = per_it[it] if it < len(per_it) else 0
    nxt = per_it[it + 1] if it + 1 < len(per_it) else 0
    rec_in = 16 if it == 0 else 48
    byts = n * (rec_in + 8) + nxt * 48 + (n - nxt) * 16
    print("%3d %11.0f %8.0f %7.3f %9.0f %7.3f %8.0f" % (it, n, dur(h), dur(h) * 1e3 / max(n, 1), dur(s), dur(s) * 1e3 / max(n, 1), byts / dur(s) / 1e3))
t0, t1 = int(step[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in step)
print("step %.2f ms, kernels %.2f ms" % ((t1 - t0) / 1e6, sum(dur(r) for r in step) / 1e3))
