#!/bin/bash
R=$(pwd); out=$R/gpurun_out/r5c; mkdir -p $out
for v in koepi hip; do
LBMPM_RK3D_AUX_PRIO=1 LBMPM_LIBRARY=$R/tools/dev/_build/lib_$v.so timeout 600 python tools/slab_rank_cost.py 512 8 > $out/rank_cost_prio_$v.log 2>&1
grep "^rank [0347]\|single" $out/rank_cost_prio_$v.log | sed "s/^/prio $v /"
done
LBMPM_RK3D_AUX_PRIO=1 LBMPM_RK3D_SLAB_LAUNCHES=3 timeout 600 python tools/slab_rank_cost.py 512 8 > $out/rank_cost_prio_3l.log 2>&1
grep "^rank [0347]\|single" $out/rank_cost_prio_3l.log | sed "s/^/prio 3launch /"
