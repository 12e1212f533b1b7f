#!/bin/bash
# dev (GPU box): the GPU's clocks and power while a workload runs:  bash tools/dev/clocks.sh c5|c3|c4 [LBMPM_K3_RELAX=...]
wl=${1:-c5}
if [ $wl = c5 ]; then python bench.py --steps ${STEPS:-6000} --warmup 10 --no-secondary --no-cpu-baseline --no-live-traffic $EXTRA > /tmp/wl.log 2>&1 &
else python bench.py --workload $wl --steps 150000 --warmup 10 > /tmp/wl.log 2>&1 & fi
pid=$!
for i in $(seq 1 14); do
  sleep 5
  echo "t=$((i*5))s $(rocm-smi --showclocks --showpower 2>/dev/null | grep -E 'sclk|mclk|fclk|Power' | sed -e 's/.*GPU\[0\]//' | tr -s ' \t' ' ' | tr '\n' ';')"
done
kill $pid 2>/dev/null; wait $pid 2>/dev/null
tail -c 300 /tmp/wl.log
