#!/bin/bash
# Bit-identity soak (scripts/gpu_soak.py) of every configuration whose kernels hand data from workgroup to workgroup inside a
# launch; -> gpurun_out/soak_$TAG.txt.  usage: scripts/gpu_soak.sh TAG [REPEATS]
TAG=${1:-r05}
R=${2:-40}
mkdir -p gpurun_out
OUT=gpurun_out/soak_$TAG.txt
: > $OUT
run() { echo "== $*" >> $OUT; timeout -k 10 ${TMO:-280} python scripts/gpu_soak.py "$@" >> $OUT 2>&1; echo "   exit $?" >> $OUT; tail -2 $OUT; }
run lattice8:mixed --repeats $R --iters 5 --rebuild 10 --poison
run lattice8:f32 --repeats $R --iters 5 --rebuild 10 --poison
run lattice:mixed,lattice:f32 --repeats $R --iters 5 --rebuild 10 --poison
run sphere2500,sphere8,intel,input_M3500_g2o,dlr,torus3D,parking-garage --repeats $R --iters 6 --rebuild 10 --poison
run grid100x100:8,grid60x40:4,grid60x40:2 --repeats $R --iters 6 --rebuild 10 --poison
