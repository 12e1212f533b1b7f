# PMC passes over the 3-D CSF step (one counter group per pass, kernel-trace only); on the GPU box:
#   bash tools/pmc_csf3d.sh [edge=512] ; tables under gpurun_out/pmc_csf_*.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
EDGE=${1:-512}
CMD="python $R/tools/csf3d_bench.py $EDGE 4 MRT"
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rm -rf $R/gpurun_out/pmc_csf_$tag
  timeout 600 rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/pmc_csf_$tag -o x -- $CMD > $R/gpurun_out/pmc_csf_$tag.log 2>&1
  db=$(find $R/gpurun_out/pmc_csf_$tag -name "x_results.db" | head -1)
  [ -n "$db" ] && python $R/tools/rocprof_summary.py $db --pmc > $R/gpurun_out/pmc_csf_$tag.txt
  rm -rf $R/gpurun_out/pmc_csf_$tag
done
grep -h "csf3d_\(collide\|phase\|gradient\)" $R/gpurun_out/pmc_csf_*.txt | grep -v "Lb1E" | cut -c1-60,90-140
