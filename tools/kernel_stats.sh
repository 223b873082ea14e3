#!/bin/bash
# Per-kernel times of one bench.py run under rocprofv3 --kernel-trace --stats: tools/kernel_stats.sh <tag> [bench.py args...]
# -> gpurun_out/kstats_<tag>/ (copy the *_kernel_stats.csv you want judged into profiles/)
set -u
TAG=$1; shift
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/kstats_$TAG
mkdir -p $OUT
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --steps 4 --warmup 1 --spp-per-step 64 --no-cpu-baseline --no-mesh-share --no-other-workloads "$@" > $OUT/bench.log 2>&1 || { tail -5 $OUT/bench.log; exit 1; }
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
cp $f $OUT/kernel_stats.csv
python3 - $OUT/kernel_stats.csv <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:8]:
    print("%-60s calls %5s avg %10.1f us  total %8.2f ms  %5.1f %%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6, float(r["Percentage"])))
PY
grep '^{"metric"' $OUT/bench.log | python3 -c 'import json,sys; j=json.loads(sys.stdin.read()); print(round(j["value"]),"Mrays/s", round(j["ms_per_step"],2),"ms/step")'
