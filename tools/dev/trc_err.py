import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
from test_tr_coupled import scenario
from helpers import rel_err
from openlbmpm_amd.rk2d import RK2DSolver
for name in ("capillary","porous"):
    d, flow, tr, to_dense = scenario(name)
    s = RK2DSolver(d["isDomain"], flow, diagnostics=True)
    s.set_macro(to_dense(d["init_rhoR"]), to_dense(d["init_rhoB"]))
    s.configure_tracers(**tr); s.set_tracer(0, to_dense(d["init_conc"][0]))
    done=0; worst={}
    for k in d["snaps"]:
        s.step(int(k)-1-done); done=int(k)-1
        for key in ("rhoR","rhoB"):
            worst[key]=max(worst.get(key,0), rel_err(s.get_compact("rec_"+key), d["s%d_%s"%(k,key)]))
        s.step(1); done=int(k)
        worst["conc"]=max(worst.get("conc",0), rel_err(s.get_tracer(0, compact=True), d["s%d_conc"%k][0]))
        for key in ("vx","vy","phi","Gx","Gy","Fx","Fy"):
            worst[key]=max(worst.get(key,0), rel_err(s.get_compact(key), d["s%d_%s"%(k,key)]))
    print(name, {k:"%.1e"%v for k,v in worst.items()})
