#!/bin/bash
# dev (GPU box): as diet3d.sh, plus the per-rank slab cost of both builds
bash tools/dev/diet3d.sh
for lib in tools/dev/_build/lib_prev.so openlbmpm_amd/liblbmpm_hip.so; do
  echo "== $lib"; LBMPM_LIBRARY=$PWD/$lib python tools/slab_rank_cost.py 512 8 2>&1 | grep -v amdgpu | tail -10
done
