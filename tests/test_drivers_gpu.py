"""GPU tests of the driver counterparts (ini in -> result file out with the reference's dataset
names) against what the real reference drivers recorded (golden 'h5|' captures)."""
import numpy as np
import pytest

from helpers import golden_files, rel_err
from ini_fixtures import write_rk, write_sc

pytestmark = pytest.mark.gpu


def test_rk_driver_records_match_reference(tmp_path):
    from openlbmpm_amd.RKD2Q9 import RKColorGradientLBM
    from openlbmpm_amd.results import load_results
    d = np.load([f for f in golden_files("rk_") if f.endswith("rk_csf_mrt_capillary.npz")][0])
    write_rk(str(tmp_path), nx=20, ny=48, steps=60, interval=25)
    sim = RKColorGradientLBM(str(tmp_path), output_dir=str(tmp_path / "out"))
    path = sim.runRKColorGradient2D()
    res = load_results(path)
    assert sim.records == 3          # steps 0, 25, 50
    for name in ("FluidMacro/FluidDensityRin0", "FluidMacro/FluidDensityBin0", "FluidPDF/FluidPDFRat0",
                 "FluidPDF/FluidPDFBat0", "FluidVelocity/FluidVelocityXAt0", "FluidVelocity/FluidVelocityYAt0"):
        g = d["h5|SimulationResultsRK.h5:/" + name]
        assert rel_err(res["/" + name], g) < 1e-9, name
    assert "/FluidMacro/FluidDensityRin2" in res


def test_sc_driver_records_match_reference(tmp_path):
    from openlbmpm_amd.ShanChenD2Q9 import ShanChenD2Q9
    from openlbmpm_amd.results import load_results
    d = np.load([f for f in golden_files("sc_") if f.endswith("sc_sc_srt_convective.npz")][0])
    write_sc(str(tmp_path), inter="ShanChen", nx=20, ny=48, steps=80, outlet="Convective")
    sim = ShanChenD2Q9(str(tmp_path), output_dir=str(tmp_path / "out"))
    path = sim.runTypeSCmodel()
    res = load_results(path)
    assert sim.records == 2          # passes 1 and 81 of 81 (ShanChenD2Q9.py:1561)
    for rec in (0, 1):
        for name in ("FluidMacro/FluidDensityType0in%d", "FluidMacro/FluidDensityType1in%d",
                     "FluidVelocity/FluidVelocityXAt%d", "FluidVelocity/FluidVelocityYAt%d"):
            g = d["h5|SimulationResults.h5:/" + name % rec]
            assert rel_err(res["/" + name % rec], g) < 1e-9, name % rec


def test_cli_runs(tmp_path):
    from openlbmpm_amd.__main__ import main
    write_sc(str(tmp_path), inter="EFS", steps=40, relax="MRT")
    assert main(["sc", str(tmp_path), "--out", str(tmp_path / "o")]) == 0
