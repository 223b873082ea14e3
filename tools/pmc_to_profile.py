#!/usr/bin/env python3
"""Turns a tools/pmc_profile.sh run (gpurun_out/pmc_<tag>/) into the per-ray PMC figures bench.py reads:

  python tools/pmc_to_profile.py <workload> gpurun_out/pmc_<tag> [--out profiles/r03_pmc.json]

Every pass of pmc_profile.sh is `bench.py --steps 1 --warmup 0`: the k_closest_hit<false> dispatches of a pass are the
launches of the one timed step, whose ray count is in the bench line of the pass's log.  Per-ray figures therefore
scale to any run of the same workload.  VALU pipe cycles: a wave64 VALU instruction occupies its SIMD-32 pipe for
2 cycles, f64 arithmetic and f32 transcendentals for 4, f64 transcendentals for 8 (MI355X_MICROARCH.md, cycle constants).
FETCH_SIZE is in KiB; for this path's access pattern (gathers of whole 64-byte records) it was calibrated against a
known byte count (profiles/r01_fetch_size_calibration.txt: ratio 1.001), so the guide's x2 for wide streaming reads
does not apply."""
import json
import os
import sys

workload, root = sys.argv[1], sys.argv[2]
out_path = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r03_pmc.json")
summ = json.load(open(os.path.join(root, "summary.json")))
line = None
for l in open(os.path.join(root, "sq1.log")):
    if l.startswith('{"metric"'):
        line = json.loads(l)
r = line["roofline"]
res = {}
for kern in ("k_closest_hit", "k_shade"):
    v = summ[kern]
    n = v["_dispatches"]
    # every step of a pass (timed, instrumented; for k_shade also bench.py's counters step, whose k_closest_hit is a variant of
    # its own that the summary keeps apart) traces the same rays: dispatches x mean rays per launch
    rays = r["rays_per_launch"] * n
    g = lambda k: v.get(k, 0.0)
    f64 = g("SQ_INSTS_VALU_ADD_F64") + g("SQ_INSTS_VALU_MUL_F64") + g("SQ_INSTS_VALU_FMA_F64")
    cycles = 2.0 * g("SQ_INSTS_VALU") + 2.0 * (f64 + g("SQ_INSTS_VALU_TRANS_F32")) + 6.0 * g("SQ_INSTS_VALU_TRANS_F64")
    gui = g("GRBM_GUI_ACTIVE")
    e = {"dispatches": n, "rays": rays,
         "valu_insts_per_ray": g("SQ_INSTS_VALU") / rays, "valu_f64_insts_per_ray": f64 / rays,
         "valu_trans_f32_insts_per_ray": g("SQ_INSTS_VALU_TRANS_F32") / rays, "salu_insts_per_ray": g("SQ_INSTS_SALU") / rays,
         "vmem_rd_insts_per_ray": g("SQ_INSTS_VMEM_RD") / rays, "lds_insts_per_ray": g("SQ_INSTS_LDS") / rays,
         "valu_pipe_cycles_per_ray": cycles / rays,
         "fetch_write_bytes_per_ray": (g("FETCH_SIZE") + g("WRITE_SIZE")) * 1024.0 / rays,
         "fetch_bytes_per_ray": g("FETCH_SIZE") * 1024.0 / rays, "write_bytes_per_ray": g("WRITE_SIZE") * 1024.0 / rays,
         "valu_lane_utilisation": g("SQ_THREAD_CYCLES_VALU") / (64.0 * g("SQ_ACTIVE_INST_VALU")) if g("SQ_ACTIVE_INST_VALU") else None,
         "waves_per_simd": g("SQ_WAVE_CYCLES") * 4.0 / (1024.0 * g("GRBM_GUI_ACTIVE") / 8.0) if gui else None,
         "valu_pipe_busy_from_cycles": cycles / (1024.0 * gui / 8.0) if gui else None,
         "wave_time_split": {"waiting_s_waitcnt": g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"), "issue_stall": (g("SQ_WAIT_INST_ANY")) / g("SQ_WAVE_CYCLES"),
                             "issuing": g("SQ_ACTIVE_INST_ANY") / g("SQ_WAVE_CYCLES"), "issuing_valu": g("SQ_ACTIVE_INST_VALU") / g("SQ_WAVE_CYCLES")} if g("SQ_WAVE_CYCLES") else None,
         "l2_hit_rate": g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum")) if g("TCC_HIT_sum") + g("TCC_MISS_sum") else None}
    res[kern] = e
all_ = {}
if os.path.exists(out_path):
    all_ = json.load(open(out_path))
top = dict(res["k_closest_hit"])
top["k_shade"] = res["k_shade"]
top["source"] = "tools/pmc_profile.sh (one rocprofv3 --pmc pass per run of `bench.py --steps 1 --warmup 0 --workload %s`), summary: %s" % (workload, os.path.join(root, "summary.json"))
top["bench_line_of_the_sq1_pass"] = {"value": line["value"], "ms_per_launch": r["ms_per_launch"], "rays_per_launch": r["rays_per_launch"]}
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
top["kernel_hash"] = bench.kernel_hash()      # bench.py flags a profile collected on other kernel sources ("pmc_stale")
all_[workload] = top
json.dump(all_, open(out_path, "w"), indent=1)
print(json.dumps({k: top[k] for k in ("valu_insts_per_ray", "valu_pipe_cycles_per_ray", "fetch_write_bytes_per_ray", "valu_lane_utilisation", "waves_per_simd", "valu_pipe_busy_from_cycles", "wave_time_split")}, indent=1))
