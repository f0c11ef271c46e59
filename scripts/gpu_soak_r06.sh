#!/bin/bash
# r06: the bit-identity soak on the build whose rr_pgo_optimize loops on the device (scripts/gpu_soak.py drives single handles through
# optimize()): the small graphs 200 repeats with handles rebuilt on junk-filled memory, the lattices and the sharded forms 40.
TAG=${1:-r06z}
mkdir -p gpurun_out
OUT=gpurun_out/soak_$TAG.txt
: > $OUT
run() { echo "== $*" >> $OUT; timeout -k 10 ${TMO:-400} python scripts/gpu_soak.py "$@" >> $OUT 2>&1; echo "   exit $?" >> $OUT; tail -2 $OUT; }
run sphere2500,intel,input_M3500_g2o,dlr,torus3D,parking-garage,intel:mixed --repeats 200 --iters 10 --rebuild 10 --poison
run sphere8,grid100x100:8,grid60x40:4,grid60x40:2,grid100x100 --repeats 40 --iters 6 --rebuild 10 --poison
run lattice:mixed,lattice:f32 --repeats 40 --iters 5 --rebuild 10 --poison
run lattice8:mixed --repeats 20 --iters 5 --rebuild 10 --poison
