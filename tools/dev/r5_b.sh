#!/bin/bash
R=$(pwd); out=$R/gpurun_out/r5b; mkdir -p $out
timeout 900 python -m pytest tests/test_rk3d_gpu.py -x -q > $out/pytest_rk3d.log 2>&1; echo "rk3d tests rc=$?" | tee -a $out/summary.txt
tail -n 6 $out/pytest_rk3d.log
LBMPM_RK3D_SLAB_LAUNCHES=3 timeout 600 python tools/slab_rank_cost.py 512 8 > $out/rank_cost_3launch.log 2>&1
timeout 600 python tools/slab_rank_cost.py 512 8 > $out/rank_cost_fused.log 2>&1
grep "^rank\|single\|sum over" $out/rank_cost_3launch.log $out/rank_cost_fused.log
timeout 600 python tools/slabbench_pipelined.py 512 8 > $out/pipelined_fused.log 2>&1; tail -n 4 $out/pipelined_fused.log
