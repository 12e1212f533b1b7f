#!/bin/bash
# dev (GPU box): a new build of the 2-D kernels against the previous one: bit equality on the bench lattices, then times
B=tools/dev/_build; out=gpurun_out/diet2d; mkdir -p $out
LBMPM_LIBRARY=$PWD/$B/lib_prev.so python tools/dev/eager_equal.py $out/a.npz > /dev/null
python tools/dev/eager_equal.py $out/b.npz > /dev/null
python tools/dev/eager_equal.py $out/a.npz $out/b.npz
LBMPM_LIBRARY=$PWD/$B/lib_prev.so python tools/dev/equal2d_sc.py $out/c.npz > /dev/null; python tools/dev/equal2d_sc.py $out/d.npz | tail -1; python tools/dev/equal2d_sc.py $out/c.npz $out/d.npz
python tools/dev/ab2d.py ${ROUNDS:-3} $B/lib_prev.so openlbmpm_amd/liblbmpm_hip.so
rm -f $out/*.npz
