#!/bin/bash
R=$(pwd); out=$R/gpurun_out/r5e; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_slab
SLAB_RANKS= timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_slab -o x -- python $R/tools/slab_rank_cost.py 512 8 20 > $out/run.log 2>&1
python $R/tools/rocprof_summary.py $(find $R/gpurun_out/prof_slab -name x_results.db | head -1) > $out/slab_kernel_trace.txt
rm -rf $R/gpurun_out/prof_slab
head -20 $out/slab_kernel_trace.txt | cut -c1-150
