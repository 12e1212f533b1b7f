"""dev: rk2d_fused (c2 lattice) over lattice sizes -- does a state that fits the 256 MB Infinity Cache run faster per node?  (no)"""
import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
for n in (512, 640, 768, 832, 896, 960, 1024, 1152, 1280, 1536, 2048, 3072):
    s,_,_ = bench.build_c2(n, n, 0)
    k = max(200, int(600 * (1024/n)**2))
    w, mt, md = bench.time_solver_2d(s, k, k//10)
    nf = s.num_fluid_nodes
    mb = n*n*18*8*2/1e6
    print("c2 %4d^2  pops(2 copies) %6.0f MB  %.4f ms/step  %6.0f MLUPS  %.2f ns/node" % (n, mb, md/k, nf*k/w/1e6, md/k*1e6/nf), flush=True)
    s.close()
