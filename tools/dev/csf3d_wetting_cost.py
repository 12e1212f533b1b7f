import sys, os, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from openlbmpm_amd.rk3dcsf import RK3DCSFSolver
size=(512,512,512)
dom=bench.c5_domain(size); dom[0]=dom[1]; dom[-1]=dom[-2]
for state in ("mixed","initial"):
    rR,rB=bench.c5_state(dom,0,512,state)
    for w in (2,0,2,0):
        s=RK3DCSFSolver(dom, dict(relax="MRT", tauB=0.8, wetting=w)); s.set_macro(rR,rB); s.step(5); s.sync()
        t,_=s.step_timed(10); print(state, "wetting", w, round(t/10,3)); s.close()
