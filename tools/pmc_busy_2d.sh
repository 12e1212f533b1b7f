# address-path / L1 counters of the 2-D bench kernels (run on the GPU box from the repo root):  bash tools/pmc_busy_2d.sh [outfile]
R=$(pwd)
outf=${1:-$R/gpurun_out/pmc_busy_2d.txt}; case $outf in /*) ;; *) outf=$R/$outf;; esac
cd /tmp && export TMPDIR=/tmp
export LBMPM_BENCH_SECONDARY_STEPS=20 LBMPM_NO_GRAPH=1
: > $outf
rocprofv3 --list-avail 2>/dev/null | grep -oE "\b(TA_[A-Z_0-9a-z]+|TCP_[A-Z_0-9a-z]+|MemUnit[A-Za-z]+|VALUBusy|MemUnitStalled|WriteUnitStalled|L2CacheHit)\b" | sort -u | tr '\n' ' ' > $R/gpurun_out/pmc_avail_ta_tcp.txt
for set in "VALUBusy" "MemUnitBusy" "MemUnitStalled" "WriteUnitStalled" "TA_BUSY_avr" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "GRBM_GUI_ACTIVE"; do
  tag=2d_$(echo $set | tr ' ' '_' | cut -c1-30)
  rm -rf $R/gpurun_out/pmc4_$tag
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/pmc4_$tag -o x -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-c5-legs > $R/gpurun_out/pmc4_$tag.log 2>&1
  db=$(find $R/gpurun_out/pmc4_$tag -name "x_results.db" | head -1)
  echo "== $set" >> $outf
  [ -n "$db" ] && python $R/tools/rocprof_summary.py $db --pmc | grep -E "rk2d_fused|rk2dp_fused|sc2d_fused<true|rk3dq_fused<false" | grep -v "^#" | cut -c1-64,100-180 >> $outf
  rm -rf $R/gpurun_out/pmc4_$tag
done
cat $outf
