"""Non-interactive command line (replaces the reference's interactive, un-importable main.py):

    python -m openlbmpm_amd rk  <ini-dir> [--out DIR] [--steps N] [--device D]
    python -m openlbmpm_amd sc  <ini-dir> [--out DIR] [--steps N] [--device D]
    python -m openlbmpm_amd tr  <ini-dir> ...      colour gradient + tracers (RKtwophasesetup2D.ini + transportsetup.ini)
    python -m openlbmpm_amd rk3d <ini-dir> ...     D3Q19 colour gradient (RKtwophasesetup3D.ini); under torchrun: z-slabs, one per GPU
"""
import argparse
import sys
import time


def main(argv=None):
    ap = argparse.ArgumentParser(prog="python -m openlbmpm_amd")
    ap.add_argument("model", choices=["rk", "sc", "tr", "rk3d"], help="rk = colour gradient (RKtwophasesetup2D.ini); "
                                                       "sc = Shan-Chen / EFS (twophasesetup.ini + efs2D.ini|shanchen2D.ini)")
    ap.add_argument("ini_dir")
    ap.add_argument("--out", default=None, help="result directory (default ~/LBMResults)")
    ap.add_argument("--steps", type=int, default=None, help="override the ini's number of time steps")
    ap.add_argument("--device", type=int, default=0)
    a = ap.parse_args(argv)
    t0 = time.time()
    if a.model == "rk":
        from .RKD2Q9 import RKColorGradientLBM
        sim = RKColorGradientLBM(a.ini_dir, output_dir=a.out, device=a.device)
        if a.steps is not None:
            sim.timeSteps = a.steps
        path = sim.runRKColorGradient2D()
        steps, nodes = sim.timeSteps, sim.voidSpace
    elif a.model == "rk3d":
        import os
        from .RKColorGradientD3Q19 import RKColorGradient3D
        device = a.device
        if int(os.environ.get("WORLD_SIZE", "1")) > 1:          # launched by torchrun: one rank per GPU
            import torch
            import torch.distributed as dist
            device = int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count()
            torch.cuda.set_device(device)
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if os.environ.get("LBMPM_DIST_BACKEND", "nccl") == "nccl":
                dist.init_process_group(backend="nccl", device_id=torch.device("cuda", device))
            else:                                                # "gloo": several ranks rehearsing on one GPU
                dist.init_process_group(backend=os.environ["LBMPM_DIST_BACKEND"])
        sim = RKColorGradient3D(a.ini_dir, output_dir=a.out, device=device)
        if a.steps is not None:
            sim.timeSteps = a.steps
        path = sim.runRKColorGradient3D()
        steps, nodes = sim.timeSteps, sim.voidSpace
    elif a.model == "tr":
        from .Transport2DRK import Transport2DRK
        sim = Transport2DRK(a.ini_dir, output_dir=a.out, device=a.device)
        if a.steps is not None:
            sim.timeSteps = a.steps
        path = " and ".join(sim.runTransport2DMPMCRKNew())
        steps, nodes = sim.timeSteps, sim.voidSpace
    else:
        from .ShanChenD2Q9 import ShanChenD2Q9
        sim = ShanChenD2Q9(a.ini_dir, output_dir=a.out, device=a.device)
        if a.steps is not None:
            sim.numTimeStep = a.steps
        path = sim.runTypeSCmodel()
        steps, nodes = sim.numTimeStep + 1, int(sim.isDomain.sum())
    dt = time.time() - t0
    print("%d steps on %d fluid nodes in %.2f s (%.1f MLUPS incl. output); results in %s"
          % (steps, nodes, dt, steps * nodes / dt / 1e6, path))
    return 0


if __name__ == "__main__":
    sys.exit(main())
