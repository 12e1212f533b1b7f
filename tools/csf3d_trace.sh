# kernel trace of the 3-D CSF step on the GPU box: bash tools/csf3d_trace.sh [edge=512] [relax=MRT] -> gpurun_out/csf3d_trace_<edge>_<relax>.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
EDGE=${1:-512}; RELAX=${2:-MRT}
rm -rf $R/gpurun_out/csf_trace
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/csf_trace -o x -- python $R/tools/csf3d_bench.py $EDGE 10 $RELAX > $R/gpurun_out/csf3d_trace_${EDGE}_${RELAX}.log 2>&1
db=$(find $R/gpurun_out/csf_trace -name "x_results.db" | head -1)
python $R/tools/rocprof_summary.py $db > $R/gpurun_out/csf3d_trace_${EDGE}_${RELAX}.txt
rm -rf $R/gpurun_out/csf_trace
grep -h "csf3d" $R/gpurun_out/csf3d_trace_${EDGE}_${RELAX}.txt | cut -c1-70,108-160
tail -1 $R/gpurun_out/csf3d_trace_${EDGE}_${RELAX}.log
