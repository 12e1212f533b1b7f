#!/bin/bash
# dev (GPU box): counted HBM traffic of c3 / c4 per walk order:  LIBS="base b4s" bash tools/dev/walk2d_pmc.sh
R=$(pwd); out=$R/gpurun_out/walk2d; mkdir -p $out
cd /tmp && export TMPDIR=/tmp LBMPM_NO_GRAPH=1
for lib in ${LIBS:-base b4s}; do
  for set in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
    t=$(echo $set | tr ' ' '_')
    rm -rf $R/gpurun_out/w2_${lib}_$t
    LBMPM_LIBRARY=$R/tools/dev/_build/lib_$lib.so timeout 300 rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/w2_${lib}_$t -o x -- python $R/tools/dev/run2d.py 20 > /dev/null 2> $out/pmc_${lib}_$t.log
    echo "== $lib $set"
    python $R/tools/rocprof_summary.py $(find $R/gpurun_out/w2_${lib}_$t -name x_results.db | head -1) --pmc | grep -E "fused" | grep -E "$(echo $set | tr ' ' '|')" | cut -c1-60,90-140
  done
done 2>&1 | tee $out/pmc.log
