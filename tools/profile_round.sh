#!/bin/bash
# Profiles of the default bench for profiles/ (run ON THE GPU BOX from the repo root):
#   bash tools/profile_round.sh r04
# pass 1: rocprofv3 --kernel-trace --stats; passes 2-4: PMC counters, one group per pass (never
# together with the trace domains gpurun refuses).  Then the same for the c5 state with both colours
# in every cell (`--c5-state mixed`), a `--steps 500` run of the default line, and the logs behind the
# figures DESIGN.md quotes (mixed / graded states, slab cost per rank).  Raw rocpd databases stay in
# gpurun_out/; the text/JSON summaries land in gpurun_out/profiles_<tag>/ and are copied into profiles/ by hand.
tag=${1:-rNN}
R=$(pwd)
out=$R/gpurun_out/profiles_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-c5-legs"
CMDX="python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary --c5-state mixed"
if [ -z "$SKIP_TRACE" ]; then
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_trace -o x -- $CMD > $out/${tag}_bench_under_rocprof.json 2> $out/trace.log
python $R/tools/rocprof_summary.py $(find $R/gpurun_out/prof_trace -name x_results.db | head -1) > $out/${tag}_bench_kernel_trace.txt
rm -rf $R/gpurun_out/prof_trace_mixed
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_trace_mixed -o x -- $CMDX > $out/${tag}_bench_mixed_under_rocprof.json 2> $out/trace_mixed.log
python $R/tools/rocprof_summary.py $(find $R/gpurun_out/prof_trace_mixed -name x_results.db | head -1) > $out/${tag}_bench_mixed_kernel_trace.txt
fi
# counter passes: a few steps of every workload are enough, and every dispatch costs ~0.1 s there; no graph replay
export LBMPM_BENCH_SECONDARY_STEPS=20 LBMPM_NO_GRAPH=1
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  t=$(echo $set | tr ' ' '_')
  rm -rf $R/gpurun_out/prof_$t
  rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/prof_$t -o x -- $CMD > /dev/null 2> $out/pmc_$t.log
  python $R/tools/rocprof_summary.py $(find $R/gpurun_out/prof_$t -name x_results.db | head -1) --pmc > $out/${tag}_pmc_$t.txt
done
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf $R/gpurun_out/prof_mixed_$set
  rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/prof_mixed_$set -o x -- $CMDX > /dev/null 2> $out/pmc_mixed_$set.log
  python $R/tools/rocprof_summary.py $(find $R/gpurun_out/prof_mixed_$set -name x_results.db | head -1) --pmc > $out/${tag}_pmc_mixed_$set.txt
done
# the graded state and the SRT relaxation: their legs of the bench line get counted bytes too (no `counted_frac: null`)
CMDG="python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary --c5-state graded"
CMDS="python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary --relax SRT"
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf $R/gpurun_out/prof_graded_$set $R/gpurun_out/prof_srt_$set
  rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/prof_graded_$set -o x -- $CMDG > /dev/null 2> $out/pmc_graded_$set.log
  python $R/tools/rocprof_summary.py $(find $R/gpurun_out/prof_graded_$set -name x_results.db | head -1) --pmc > $out/${tag}_pmc_graded_$set.txt
  rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/prof_srt_$set -o x -- $CMDS > /dev/null 2> $out/pmc_srt_$set.log
  python $R/tools/rocprof_summary.py $(find $R/gpurun_out/prof_srt_$set -name x_results.db | head -1) --pmc > $out/${tag}_pmc_srt_$set.txt
done
unset LBMPM_BENCH_SECONDARY_STEPS LBMPM_NO_GRAPH
cd $R
python tools/pmc_to_json.py $out/${tag}_pmc_FETCH_SIZE.txt $out/${tag}_pmc_WRITE_SIZE.txt \
       mixed $out/${tag}_pmc_mixed_FETCH_SIZE.txt $out/${tag}_pmc_mixed_WRITE_SIZE.txt \
       graded $out/${tag}_pmc_graded_FETCH_SIZE.txt $out/${tag}_pmc_graded_WRITE_SIZE.txt \
       - $out/${tag}_pmc_srt_FETCH_SIZE.txt $out/${tag}_pmc_srt_WRITE_SIZE.txt > $out/pmc_traffic.json
cp $out/pmc_traffic.json profiles/pmc_traffic.json      # (on the box: the bench runs below read it; copy it back by hand as well)
python bench.py > $out/${tag}_bench_c5_n1.json 2> $out/bench.log
if [ -z "$SKIP_LONG" ]; then
python bench.py --steps 500 --warmup 50 --no-secondary --no-cpu-baseline > $out/${tag}_bench_c5_steps500.json 2>> $out/bench.log
python tools/dev/k3mixed.py 512 > $out/${tag}_k3mixed.log 2>&1
python tools/slab_rank_cost.py 512 8 > $out/${tag}_slab_rank_cost_512_8.log 2>&1
SLAB_CALIBRATE=2 python tools/slab_rank_cost.py 512 8 > $out/${tag}_slab_rank_cost_512_8_calibrated.log 2>&1
python tools/slabbench_pipelined.py 512 8 > $out/${tag}_slabbench_pipelined_512_8.log 2>&1
python tools/slabbench_pipelined.py 512 2 > $out/${tag}_slabbench_pipelined_512_2.log 2>&1
fi
tail -c 600 $out/${tag}_bench_c5_n1.json
