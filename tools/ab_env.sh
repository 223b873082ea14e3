#!/bin/bash
# Same-box A/B of environment switches on the built library: tools/ab_env.sh name1:VAR=val name2:VAR=val ... ("name:" = no variable);
# AB_ARGS="--workload c4" adds bench.py arguments
set -u
mkdir -p gpurun_out
for rep in 1 2; do
  for kv in "$@"; do
    name=${kv%%:*}; setting=${kv#*:}
    env $setting timeout -k 10 150 python3 bench.py --steps 4 --warmup 1 --spp-per-step 64 --no-cpu-baseline --no-other-workloads ${AB_ARGS:-} > gpurun_out/env_${name}_$rep.log 2>&1 || { echo "$name failed"; tail -5 gpurun_out/env_${name}_$rep.log; exit 1; }
    python3 - "$name" gpurun_out/env_${name}_$rep.log <<'PY'
import json,sys
for line in open(sys.argv[2]):
    if line.startswith('{"metric"'):
        j=json.loads(line); r=j["roofline"]
        print(f'{sys.argv[1]:>10}: {j["value"]:8.1f} Mrays/s  hit {r["ms_per_launch"]:.4f} ms/launch  nodes/ray {r["nodes_per_ray"]:.2f} prims/ray {r["prims_per_ray"]:.2f} lanes {r["lanes_with_work_per_step"]:.3f} step {j["ms_per_step"]:.2f} ms')
PY
  done
done
