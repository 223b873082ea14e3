#!/bin/bash
# Soak of fresh seeds on the kernels as they stand: tools/soak_final.sh <first seed offset>; logs under gpurun_out/soak_last_*.log
B=${1:-0}
run() { name=$1; shift; timeout -k 10 $1 "${@:2}" > gpurun_out/soak_last_$name.log 2>&1; rc=$?; tail -1 gpurun_out/soak_last_$name.log; echo "$name rc=$rc"; [ $rc -le 1 ] || exit $rc; }
run objects 300 python -u tools/soak_objects.py $((20001 + B)) $((20700 + B))
SOAK_TILT=1 run tilt 120 python -u tools/soak_objects.py $((21001 + B)) $((21150 + B))
run images 150 python -u tools/soak_images.py $((22001 + B)) $((22200 + B))
SOAK_GLASS_MESH=1 run glass 120 python -u tools/soak_images.py $((22301 + B)) $((22400 + B))
SOAK_ROOM=1 run rooms 120 python -u tools/soak_images.py $((22501 + B)) $((22540 + B))
run soups 120 python -u tools/soak_soups.py $((23001 + B)) $((24500 + B))
