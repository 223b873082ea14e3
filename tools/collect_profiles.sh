#!/bin/bash
# After tools/profile_all.sh (and a plain `python bench.py > gpurun_out/bench_default.log`) on the GPU box: copies what is to be
# judged from gpurun_out/ into profiles/r03_* and rebuilds profiles/r03_pmc.json (the per-ray PMC figures bench.py reads).
set -e
cd "$(dirname "$0")/.."
for w in c5 c4 c3; do
  python3 tools/pmc_to_profile.py $w gpurun_out/pmc_r3_$w > /dev/null
  cp gpurun_out/kstats_r3_$w/kernel_stats.csv profiles/r03_kernel_stats_$w.csv
  cp gpurun_out/kstats_r3_$w/launch_trace.txt profiles/r03_launch_trace_$w.txt
  cp gpurun_out/pmc_r3_$w/summary.json profiles/r03_pmc_summary_$w.json
  grep '^{"metric"' gpurun_out/kstats_r3_$w/bench.log > profiles/r03_bench_under_rocprof_$w.json
done
[ -f gpurun_out/bench_default.log ] && grep '^{"metric"' gpurun_out/bench_default.log > profiles/r03_bench_default_line.json
[ -f gpurun_out/pytest_gpu_verbose.log ] && cp gpurun_out/pytest_gpu_verbose.log profiles/r03_pytest_gpu_verbose.txt
[ -f gpurun_out/parity_record.json ] && cp gpurun_out/parity_record.json profiles/r03_parity.json
python3 tools/isa_histogram.py > /dev/null 2>&1 || true
echo collected
