#!/bin/bash
# Profiles of the default bench for profiles/ (run ON THE GPU BOX from the repo root):
#   bash tools/profile_round.sh r01
# pass 1: rocprofv3 --kernel-trace --stats; passes 2-4: PMC counters, one group per pass (never
# together with the trace domains gpurun refuses).  Raw rocpd databases stay in gpurun_out/;
# the text/JSON summaries land in gpurun_out/profiles_<tag>/ and are copied into profiles/ by hand.
tag=${1:-rNN}
R=$(pwd)
out=$R/gpurun_out/profiles_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline"
if [ -z "$SKIP_TRACE" ]; then
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_trace -o x -- $CMD > $out/${tag}_bench_under_rocprof.json 2> $out/trace.log
python $R/tools/rocprof_summary.py $(find $R/gpurun_out/prof_trace -name x_results.db | head -1) > $out/${tag}_bench_kernel_trace.txt
fi
# counter passes: a few steps of every workload are enough, and every dispatch costs ~0.1 s there; no graph replay
export LBMPM_BENCH_SECONDARY_STEPS=20 LBMPM_NO_GRAPH=1
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  t=$(echo $set | tr ' ' '_')
  rm -rf $R/gpurun_out/prof_$t
  rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/prof_$t -o x -- $CMD > /dev/null 2> $out/pmc_$t.log
  python $R/tools/rocprof_summary.py $(find $R/gpurun_out/prof_$t -name x_results.db | head -1) --pmc > $out/${tag}_pmc_$t.txt
done
unset LBMPM_BENCH_SECONDARY_STEPS LBMPM_NO_GRAPH
cd $R
python tools/pmc_to_json.py $out/${tag}_pmc_FETCH_SIZE.txt $out/${tag}_pmc_WRITE_SIZE.txt > $out/pmc_traffic.json
python bench.py > $out/${tag}_bench_c5_n1.json 2> $out/bench.log
tail -c 600 $out/${tag}_bench_c5_n1.json
