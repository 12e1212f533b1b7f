cd $GRAFT_REPO_ROOT
python -m pytest tests/test_rk3d_gpu.py -x -q -m gpu 2>&1 | tail -4
for b in 2 3 4 8; do echo "BOUNDARY=$b"; LBMPM_RK3D_BOUNDARY=$b python tools/slabbench.py 512 1 8 2>&1 | grep "k="; done
