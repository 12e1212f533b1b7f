cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CMD="python $R/tools/csf3d_bench.py 512 4 MRT mixed"
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf $R/gpurun_out/pmcm_$set
  timeout 600 rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/pmcm_$set -o x -- $CMD > /dev/null 2>&1
  db=$(find $R/gpurun_out/pmcm_$set -name "x_results.db" | head -1)
  python $R/tools/rocprof_summary.py $db --pmc | grep -E "csf3d_(collide<false|phase<false|gradient|solid)" | cut -c1-60,88-150
  rm -rf $R/gpurun_out/pmcm_$set
done
