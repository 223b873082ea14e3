#!/bin/bash
# One SQ pass (lane utilisation, VALU busy, waits) for the current library: tools/pmc_quick.sh <tag>
set -u
TAG=$1
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmcq_$TAG
mkdir -p $OUT
timeout -s KILL 150 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT/sq1 -- python3 bench.py --steps 1 --warmup 0 --spp-per-step 64 --no-cpu-baseline --no-other-workloads ${PMC_ARGS:-} > $OUT/sq1.log 2>&1 || exit 1
timeout -s KILL 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SMEM GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq2 -- python3 bench.py --steps 1 --warmup 0 --spp-per-step 64 --no-cpu-baseline --no-other-workloads ${PMC_ARGS:-} > $OUT/sq2.log 2>&1 || exit 1
python3 tools/pmc_summary.py $OUT > $OUT/summary.json
python3 - $OUT/summary.json <<'PY'
import json,sys
j=json.load(open(sys.argv[1]))
for k in ('k_closest_hit','k_shade'):
    v=j[k]; n=v['_dispatches']
    print(k, 'lane util %.3f'%(v['SQ_THREAD_CYCLES_VALU']/(64*v['SQ_ACTIVE_INST_VALU'])), 'VALU busy %.3f'%(v['SQ_ACTIVE_INST_VALU']*4/(1024*v['GRBM_GUI_ACTIVE']/8)),
          'waves/SIMD %.2f'%(v['SQ_WAVE_CYCLES']*4/(1024*v['GRBM_GUI_ACTIVE']/8)), 'VALU/launch %.3e'%(v['SQ_INSTS_VALU']/n), 'SALU/launch %.3e'%(v['SQ_INSTS_SALU']/n),
          'VMEM/launch %.3e'%(v['SQ_INSTS_VMEM_RD']/n), 'LDS/launch %.3e'%(v['SQ_INSTS_LDS']/n), 'wait_any %.3f'%(v['SQ_WAIT_ANY']/v['SQ_WAVE_CYCLES']))
PY
