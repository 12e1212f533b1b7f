#!/bin/bash
# dev (GPU box): tile walk orders / the 1024-thread tracer shape against the product order on the 2-D bench lattices
R=$(pwd); out=$R/gpurun_out/walk2d; mkdir -p $out
B=tools/dev/_build
timeout 1500 python tools/dev/ab2d.py ${ROUNDS:-2} $B/lib_base.so $B/lib_b2.so $B/lib_b4.so $B/lib_b2s.so $B/lib_b4s.so $B/lib_b8s.so 2>&1 | tee $out/walk.log
LBMPM_RK2D_SHAPE=2 timeout 600 python tools/dev/ab2d.py 2 $B/lib_tall.so 2>&1 | tee $out/tall.log
