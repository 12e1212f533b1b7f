# utilisation counters of the c5 kernel (run on the GPU box from the repo root):  bash tools/pmc_busy_c5.sh [initial|mixed|graded] [outfile]
state=${1:-initial}
R=$(pwd)
outf=${2:-$R/gpurun_out/pmc_busy_c5_$state.txt}; case $outf in /*) ;; *) outf=$R/$outf;; esac
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 6 --warmup 2 --no-secondary --no-cpu-baseline --c5-state $state"
: > $outf
for set in "VALUBusy" "MemUnitBusy" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD" \
           "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY" "SQ_WAIT_ANY SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE"; do
  tag=c5_${state}_$(echo $set | tr ' ' '_' | cut -c1-30)
  rm -rf $R/gpurun_out/pmc3_$tag
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/pmc3_$tag -o x -- $CMD > $R/gpurun_out/pmc3_$tag.log 2>&1
  db=$(find $R/gpurun_out/pmc3_$tag -name "x_results.db" | head -1)
  echo "== c5 ($state) : $set" >> $outf
  [ -n "$db" ] && python $R/tools/rocprof_summary.py $db --pmc | grep -E "fused" | grep -v "^#" | cut -c1-60,90-170 >> $outf
  rm -rf $R/gpurun_out/pmc3_$tag
done
cat $outf
