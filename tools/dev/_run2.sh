export LBMPM_K3_RELAX=MRT
echo "== rank cost"; timeout 900 python tools/slab_rank_cost.py 512 8 30 2>&1 | grep -v amdgpu | tail -11
echo "== XCC=0"; LBMPM_RK3D_XCC=0 timeout 900 python tools/slabbench_pipelined.py 512 8 20 2>&1 | grep -v amdgpu | tail -2 | head -1
echo "== XCC=1"; timeout 900 python tools/slabbench_pipelined.py 512 8 20 2>&1 | grep -v amdgpu | tail -2 | head -1
echo "== COMM_CUS=16"; LBMPM_RK3D_COMM_CUS=16 timeout 900 python tools/slabbench_pipelined.py 512 8 20 2>&1 | grep -v amdgpu | tail -2 | head -1
echo "== BOUNDARY=2"; LBMPM_RK3D_BOUNDARY=2 timeout 900 python tools/slabbench_pipelined.py 512 8 20 2>&1 | grep -v amdgpu | tail -2
echo "== BOUNDARY=4"; LBMPM_RK3D_BOUNDARY=4 timeout 900 python tools/slabbench_pipelined.py 512 8 20 2>&1 | grep -v amdgpu | tail -2
echo "== cluster"; timeout 900 python tools/slabbench.py 512 8 2>&1 | grep -v amdgpu | tail -4
