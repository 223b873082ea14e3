set -e
for w in c5 c4 c3; do
  tools/kernel_stats.sh r3_$w --no-other-workloads --workload $w
  python3 tools/launch_trace.py gpurun_out/kstats_r3_$w > gpurun_out/kstats_r3_$w/launch_trace.txt
  tools/pmc_profile.sh r3_$w --no-other-workloads --workload $w > gpurun_out/r3_pmc_$w.log 2>&1
  tail -3 gpurun_out/r3_pmc_$w.log | head -1
done
