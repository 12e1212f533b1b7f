"""dev: ms per step of rk3dq_fused on small lattices (the reference ini's 32 x 32 x 96 duct, 100^3, 128 x 128 x 256): chunk_planes of csrc/rk3d.hip"""
import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from openlbmpm_amd.rk3d import RK3DSlab
from openlbmpm_amd.RKColorGradientD3Q19 import duct
from openlbmpm_amd.geometry import porous_spheres, initial_densities_rk3d
for name, dom in (("duct 32x32x96 (the shipped ini)", duct(32,32,96)), ("porous 100^3", porous_spheres(100,100,100,seed=20260928)), ("porous 128x128x256", porous_spheres(128,128,256,seed=1))):
    rR,rB=initial_densities_rk3d(dom,10)
    s=RK3DSlab(dom,0,dom.shape[0],dict(relax="SRT")); s.set_density(rR,rB); s.step_single(5); s.sync()
    ms,_=s.step_timed(200); print("%-34s %.4f ms per step  %.0f MLUPS" % (name, ms/200, s.num_fluid_nodes*200/ms/1e3)); s.close()
