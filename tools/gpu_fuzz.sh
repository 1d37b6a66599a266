#!/bin/bash
# GPU box: adversarial fuzz of mcrt_intersect in every mode (tools/fuzz_walk.py --gpu), summary lines into gpurun_out/<tag>_gpu_fuzz.txt
# usage: tools/gpu_fuzz.sh <tag>
TAG=$1; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${TAG}_gpu_fuzz.txt
: > $OUT
run() { echo "# fuzz_walk.py $*" >> $OUT; timeout -k 10 500 python -u tools/fuzz_walk.py --gpu "$@" >> $OUT 2>&1 || { echo "FAILED: $*" >> $OUT; exit 1; }; }
run --rays 1000000 --seed 11 --levels 3 --scene config4_2k
run --rays 1000000 --seed 12 --levels 5 --scene config4_2k
run --rays 1000000 --seed 13 --levels 4 --scene config4_2k --accel 1,12,0
run --rays 1000000 --seed 14 --levels 3 --scene config4_2k --accel 2,6,1e-6
run --rays 2000000 --seed 15 --levels 3 --scene cornell --scene config2 --scene config3 --scene hexagon
run --rays 2000000 --seed 16 --levels 5 --scene cornell --scene config2 --scene three_lights --scene cornell_sphere_light
# octrees that live in HBM: mcrt_intersect runs the dealt walk of the render kernel
run --rays 1000000 --seed 21 --levels 4 --soup 12000
run --rays 1000000 --seed 22 --levels 5 --soup 60000
run --rays 1000000 --seed 23 --levels 3 --soup 30000 --accel 4,10,0
tail -3 $OUT
