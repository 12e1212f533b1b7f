#!/bin/bash
R=$(pwd); out=$R/gpurun_out/r5d; mkdir -p $out
timeout 900 python -m pytest tests/test_rk3d_gpu.py tests/test_long_parity_gpu.py -x -q -k "slab or two_process or pipelined or wide or dies or bench_line or chunk_border" > $out/pytest.log 2>&1; tail -n 3 $out/pytest.log
SLAB_CALIBRATE=2 timeout 900 python tools/slab_rank_cost.py 512 8 > $out/rank_cost_cal2.log 2>&1
grep "^rank\|single\|sum over\|re-cut" $out/rank_cost_cal2.log
timeout 600 python tools/slabbench_pipelined.py 512 8 > $out/pipelined.log 2>&1; tail -n 2 $out/pipelined.log
