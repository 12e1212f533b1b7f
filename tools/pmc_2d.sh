# Utilisation counters of the 2-D fused kernels (run on the GPU box): bash tools/pmc_2d.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for wl in c3 c4 c2; do
  CMD="python $R/bench.py --workload $wl --steps 30 --warmup 3"
  for set in "VALUBusy" "MemUnitBusy" "MemUnitStalled" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "GRBM_GUI_ACTIVE SQ_WAVES"; do
    tag=${wl}_$(echo $set | tr ' ' '_' | cut -c1-30)
    rm -rf $R/gpurun_out/pmc2_$tag
    timeout 300 rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/pmc2_$tag -o x -- $CMD > $R/gpurun_out/pmc2_$tag.log 2>&1
    db=$(find $R/gpurun_out/pmc2_$tag -name "x_results.db" | head -1)
    echo "== $wl : $set"
    [ -n "$db" ] && python $R/tools/rocprof_summary.py $db --pmc | grep -E "fused" | grep -v "^#" | head -4
  done
done
