#!/bin/bash
# L2 <-> memory counters (request sizes, stalls, queue levels) of the bench workload, one rocprofv3 pass per run: tools/pmc_tcc.sh <tag> [bench args]
set -u
TAG=$1; shift
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmct_$TAG
mkdir -p $OUT
run_pass() {
  name=$1; shift
  timeout -s KILL 150 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-mesh-share --spp-per-step 64 --no-other-workloads $BENCH_ARGS > $OUT/$name.log 2>&1
  rc=$?; echo "pass $name rc=$rc"; return $rc
}
BENCH_ARGS="$*"
run_pass t1 TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum || exit 1
run_pass t2 TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum TCC_BUSY_sum || exit 1
run_pass t3 TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_CYCLE_sum || exit 1
run_pass t4 TCC_SRC_FIFO_FULL_sum TCC_LATENCY_FIFO_FULL_sum TCC_IB_STALL_sum TCC_BUBBLE_sum || exit 1
run_pass t5 TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_STREAMING_REQ_sum || exit 1
run_pass t6 TCC_READ_sum TCC_WRITE_sum TCC_ATOMIC_sum TCC_WRITEBACK_sum || exit 1
python3 tools/pmc_summary.py $OUT > $OUT/summary.json
