# PMC passes over the c5 bench (one counter group per pass, kernel-trace only); run on the GPU box:
#   bash tools/pmc_c5.sh ; results under gpurun_out/pmc_*/ ; summarise with tools/rocprof_summary.py --pmc
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CMD="python $R/bench.py --steps 6 --warmup 2 --no-secondary --no-cpu-baseline"
KERN=${KERN:-fused}
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_WAIT_ANY"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rm -rf $R/gpurun_out/pmc_$tag
  timeout 600 rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/pmc_$tag -o x -- $CMD > $R/gpurun_out/pmc_$tag.log 2>&1
  db=$(find $R/gpurun_out/pmc_$tag -name "x_results.db" | head -1)
  [ -n "$db" ] && python $R/tools/rocprof_summary.py $db --pmc | grep -E "$KERN.*false" | grep -v "^#" | awk '{print $(NF-2), $(NF-1), $NF}'
done
