#!/bin/bash
# Collects rocprofv3 PMC counters for the bench workload, one pass per run (no tracing flags).
# usage: tools/pmc_profile.sh <tag> [bench args...]   -> gpurun_out/pmc_<tag>/<pass>/..., summary via tools/pmc_summary.py
set -u
TAG=$1; shift
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
run_pass() {
  name=$1; shift
  timeout -s KILL 150 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-mesh-share --spp-per-step 64 $BENCH_ARGS > $OUT/$name.log 2>&1
  rc=$?
  echo "pass $name rc=$rc"
  return $rc
}
BENCH_ARGS="$*"
run_pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU || exit 1
run_pass sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 || exit 1
run_pass sq3 SQ_INSTS_VALU_TRANS_F64 SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM GRBM_GUI_ACTIVE || exit 1
run_pass tcc1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum || exit 1
run_pass fetch FETCH_SIZE || exit 1
run_pass write WRITE_SIZE || exit 1
run_pass tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum || exit 1
run_pass tcc2 TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum || exit 1
python3 tools/pmc_summary.py $OUT > $OUT/summary.json && cat $OUT/summary.json
