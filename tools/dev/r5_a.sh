#!/bin/bash
# round 5, GPU call A: the new tests + a bench line + the counters' calibration rows
R=$(pwd); out=$R/gpurun_out/r5a; mkdir -p $out
timeout 900 python -m pytest tests/test_rk3d_gpu.py -x -q -k "dies_mid_run or named_transport or two_process or transport_selftest or matching_neighbours" > $out/pytest_transport.log 2>&1; echo "transport tests rc=$?" | tee -a $out/summary.txt
timeout 1200 python -m pytest tests/test_long_parity_gpu.py -x -q > $out/pytest_long.log 2>&1; echo "long parity rc=$?" | tee -a $out/summary.txt
timeout 600 python -m pytest tests/test_sc2d_gpu.py tests/test_rk2d_gpu.py tests/test_rk2d_pert_gpu.py -x -q > $out/pytest_2d.log 2>&1; echo "2d rc=$?" | tee -a $out/summary.txt
timeout 600 python bench.py --no-cpu-baseline > $out/bench.json 2> $out/bench.log; echo "bench rc=$?" | tee -a $out/summary.txt
cd /tmp && export TMPDIR=/tmp
export LBMPM_BENCH_SECONDARY_STEPS=20 LBMPM_NO_GRAPH=1
CMD="python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-c5-legs"
for set in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/prof_$set
  timeout 600 rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/prof_$set -o x -- $CMD > /dev/null 2> $out/pmc_$set.log
  python $R/tools/rocprof_summary.py $(find $R/gpurun_out/prof_$set -name x_results.db | head -1) --pmc > $out/r05a_pmc_$set.txt
done
cd $R
python tools/pmc_to_json.py $out/r05a_pmc_FETCH_SIZE.txt $out/r05a_pmc_WRITE_SIZE.txt > $out/pmc_traffic.json 2> $out/pmc_json.log
rm -rf $R/gpurun_out/prof_FETCH_SIZE $R/gpurun_out/prof_WRITE_SIZE
tail -n 5 $out/pytest_transport.log $out/pytest_long.log $out/pytest_2d.log; head -c 900 $out/bench.json; grep -A8 calibration $out/pmc_traffic.json | head -60
