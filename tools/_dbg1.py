import sys
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
from openlbmpm_amd.rk2d import RK2DSolver
from openlbmpm_amd.geometry import simple_geometry, initial_densities_rk
for n in (256, 512, 1024):
    dom = simple_geometry(n, n)
    rR, rB = initial_densities_rk(dom, False, 10, mode="intrusion")
    for v in (0,1):
        s = RK2DSolver(dom, None, variant=v)
        s.set_macro(rR, rB)
        a=s.get("rhoR"); print(n,v,'init rhoR sum',a.sum(), rR.sum())
        s.step(5)
        a=s.get("rhoR"); b=s.get("rhoB"); f=s.get("fR")
        print(n,v,'rhoR sum',a.sum(),'rhoB',b.sum(),'fR',f.sum(), 'steps',s.steps_done)
        s.close()
