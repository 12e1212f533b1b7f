cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CMD="python $R/bench.py --steps 6 --warmup 2 --no-secondary --no-cpu-baseline"
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "TCC_EA_RDREQ_sum TCC_EA_WRREQ_sum TCC_EA_RDREQ_32B_sum TCC_EA_WRREQ_64B_sum"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 600 rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/pmc_$tag -o x -- $CMD > $R/gpurun_out/pmc_$tag.log 2>&1
  python $R/tools/rocprof_summary.py $(ls $R/gpurun_out/pmc_$tag/*/x_results.db $R/gpurun_out/pmc_$tag/x_results.db 2>/dev/null | head -1) --pmc | grep -E "counter|rk3d_fused" | grep -v "^#"
done
