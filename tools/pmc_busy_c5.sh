cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CMD="python $R/bench.py --steps 6 --warmup 2 --no-secondary --no-cpu-baseline"
for set in "VALUBusy" "MemUnitBusy" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE"; do
  tag=c5_$(echo $set | tr ' ' '_' | cut -c1-30)
  rm -rf $R/gpurun_out/pmc3_$tag
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/pmc3_$tag -o x -- $CMD > $R/gpurun_out/pmc3_$tag.log 2>&1
  db=$(find $R/gpurun_out/pmc3_$tag -name "x_results.db" | head -1)
  echo "== c5 : $set"
  [ -n "$db" ] && python $R/tools/rocprof_summary.py $db --pmc | grep -E "fused" | grep -v "^#" | cut -c1-40,100-170 | head -5
done
