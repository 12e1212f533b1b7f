#!/bin/bash
# dev (GPU box): a new build of the 3-D kernel against the previous one: bit equality on a c5-model lattice, then times
B=tools/dev/_build; out=gpurun_out/diet3d; mkdir -p $out
LBMPM_LIBRARY=$PWD/$B/lib_prev.so python tools/dev/equal3d.py $out/a.npz > /dev/null
python tools/dev/equal3d.py $out/b.npz > /dev/null
python tools/dev/equal3d.py $out/a.npz $out/b.npz
AB_RELAX=${AB_RELAX:-MRT} python tools/dev/ab.py 512 ${ROUNDS:-3} $B/lib_prev.so openlbmpm_amd/liblbmpm_hip.so
rm -f $out/a.npz $out/b.npz
