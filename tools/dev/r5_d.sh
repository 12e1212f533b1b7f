#!/bin/bash
R=$(pwd); out=$R/gpurun_out/r5d; mkdir -p $out
timeout 900 python -m pytest tests/test_rk3d_gpu.py -x -q -k "bench_line or two_process" > $out/pytest.log 2>&1; tail -n 3 $out/pytest.log
SLAB_CALIBRATE=2 timeout 900 python tools/slab_rank_cost.py 512 8 > $out/rank_cost_cal2.log 2>&1
grep "^rank\|single\|sum over\|re-cut" $out/rank_cost_cal2.log
