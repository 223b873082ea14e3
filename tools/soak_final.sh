run() { name=$1; shift; timeout -k 10 $1 "${@:2}" > gpurun_out/soak_r3_last_$name.log 2>&1; rc=$?; tail -1 gpurun_out/soak_r3_last_$name.log; echo "$name rc=$rc"; [ $rc -le 1 ] || exit $rc; }
run objects 420 python -u tools/soak_objects.py 9401 10600
run images 200 python -u tools/soak_images.py 12001 12250
SOAK_GLASS_MESH=1 run glass 150 python -u tools/soak_images.py 12301 12400
run soups 150 python -u tools/soak_soups.py 13001 15000
