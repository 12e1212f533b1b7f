#!/bin/bash
R=$(pwd); out=$R/gpurun_out/r5f; mkdir -p $out
timeout 2400 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1; echo "gpu suite rc=$?"; tail -n 4 $out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 2
timeout 900 python bench.py > $out/bench.json 2> $out/bench.log; echo "bench rc=$?"; head -c 600 $out/bench.json
