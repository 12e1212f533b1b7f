# kernel trace of the 3-D CSF model in z-slabs on the GPU box (all slabs on this GPU): bash tools/csf3d_slab_trace.sh [edge=512] [slabs=8]
# -> gpurun_out/csf3d_slab_trace_<edge>_<slabs>.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
EDGE=${1:-512}; K=${2:-8}
rm -rf $R/gpurun_out/csf_trace
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/csf_trace -o x -- python $R/tools/csf3d_slab_cost.py $EDGE 10 MRT $K > $R/gpurun_out/csf3d_slab_trace_${EDGE}_${K}.log 2>&1
db=$(find $R/gpurun_out/csf_trace -name "x_results.db" | head -1)
python $R/tools/rocprof_summary.py $db > $R/gpurun_out/csf3d_slab_trace_${EDGE}_${K}.txt
rm -rf $R/gpurun_out/csf_trace
grep -h "csf3d" $R/gpurun_out/csf3d_slab_trace_${EDGE}_${K}.txt | cut -c1-70,108-160
tail -1 $R/gpurun_out/csf3d_slab_trace_${EDGE}_${K}.log
