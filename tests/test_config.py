"""CPU tests of the ini reader (reference grammar: quoted strings, case-insensitive keys, the key
skews of the shipped files) and of the drivers' argument checking."""
import os

import pytest

from ini_fixtures import write_rk, write_sc
from openlbmpm_amd import config


def test_rk_ini(tmp_path):
    write_rk(str(tmp_path), nx=20, ny=200, steps=500001, interval=2500)
    p = config.read_rk2d(str(tmp_path))
    assert (p["nx"], p["ny"], p["steps"], p["interval"]) == (20, 200, 500001, 2500)
    assert p["relax"] == "MRT" and p["inlet"] == "Neumann" and p["outlet"] == "Dirichlet"
    assert p["sigma"] == 0.1          # shipped spelling "SurfaceTension" accepted for "SurfaceTensionValue"
    assert p["vyR"] == -1.0e-4 and p["rhoRL"] == 5e-8 and p["wetting"] == 2 and not p["image"]


def test_rk_ini_errors(tmp_path):
    with pytest.raises(config.ConfigError):
        config.read_rk2d(str(tmp_path))                      # file missing
    write_rk(str(tmp_path))
    path = os.path.join(str(tmp_path), "RKtwophasesetup2D.ini")
    text = open(path).read()
    open(path, "w").write(text.replace("'CSF'", "'Perturbation'", 1))       # read: runRKColorGradient2DPerturbation runs that loop
    q = config.read_rk2d(str(tmp_path))
    assert q["tension_type"] == "Perturbation" and q["AkR"] == 0.14 and q["solidPhi"] == 0.5
    # the perturbation loop has no solver-side validation behind it: a misspelt relaxation / inlet / outlet is refused by the reader
    pert = text.replace("'CSF'", "'Perturbation'", 1)
    for good, typo in (("'MRT'", "'TRT'"), ("'Neumann'", "'Neuman'"), ("'Dirichlet'", "'Dirichlett'")):
        assert good in pert
        open(path, "w").write(pert.replace(good, typo, 1))
        with pytest.raises(config.ConfigError):
            config.read_rk2d(str(tmp_path))
    open(path, "w").write(text.replace("'CSF'", "'Level-set'", 1))
    with pytest.raises(config.ConfigError):
        config.read_rk2d(str(tmp_path))
    open(path, "w").write(text.replace("TauR = 1.0", "TauR = fast"))
    with pytest.raises(config.ConfigError):
        config.read_rk2d(str(tmp_path))


@pytest.mark.parametrize("inter", ["EFS", "ShanChen"])
def test_sc_ini(tmp_path, inter):
    write_sc(str(tmp_path), inter=inter, steps=300)
    p = config.read_sc2d(str(tmp_path))
    assert p["inter"] == inter and p["steps"] == 300 and (p["tau0"], p["tau1"]) == (1.0, 1.0)
    assert p["G"] == (0.20 if inter == "EFS" else 3.8)
    assert p["method"] == "ZouHe" and p["outlet"] in ("Dirichlet", "Convective")
    # the loop alternates: 'Freeflow' outlet and 'Chang' inlet run for the explicit forcing loop (SRT for 'Freeflow')
    for kw, ok in ((dict(outlet="Freeflow"), inter == "EFS"), (dict(method="Chang"), True),
                   (dict(outlet="Freeflow", relax="MRT"), False), (dict(method="Guo"), False), (dict(outlet="Open"), False)):
        write_sc(str(tmp_path), inter=inter, steps=300, **kw)
        if kw.get("outlet") == "Freeflow" and inter != "EFS":         # no such branch in the original loop: no outlet rule, with a warning
            with pytest.warns(UserWarning, match="no branch"):
                assert config.read_sc2d(str(tmp_path))["outlet"] == "Dirichlet"
        elif ok:
            q = config.read_sc2d(str(tmp_path))
            assert (q["outlet"], q["method"]) == (kw.get("outlet", "Dirichlet" if inter == "EFS" else q["outlet"]), kw.get("method", "ZouHe"))
        else:
            with pytest.raises(config.ConfigError):
                config.read_sc2d(str(tmp_path))


def test_transport_ini(tmp_path):
    from ini_fixtures import write_transport, TRANSPORT_INI
    from openlbmpm_amd import config
    write_transport(str(tmp_path))
    t = config.read_transport(str(tmp_path))
    assert t["num_tracers"] == 2 and t["diffX"] == [1. / 6., 0.12] and t["dXY"] == 0.01 and t["beta"] == 0.8
    for old, new in (("'MPMC'", "'Single'"), ("NumberSchemes = 5", "NumberSchemes = 9"), ("'Freeflow'", "'Outflow'"), ("InletType = 'Dirichlet'", "InletType = 'Neumann'"),
                     ("Reaction = 'no'", "Reaction = 'yes'\n[Reaction]\nNumberReaction = 1\nReactionRate = 0.1"), ("0.12", "0.12, 0.3")):
        (tmp_path / "transportsetup.ini").write_text(TRANSPORT_INI.replace(old, new))
        with pytest.raises(config.ConfigError):
            config.read_transport(str(tmp_path))
    # the spelling the reference's reader knows ('FreeFlow', Transport2DRK.py:191) is not the one its loop acts on (:1363)
    (tmp_path / "transportsetup.ini").write_text(TRANSPORT_INI.replace("'Freeflow'", "'FreeFlow'"))
    assert config.read_transport(str(tmp_path))["outlet_type"] == "FreeFlow"


def test_force_scheme_key(tmp_path):
    write_sc(str(tmp_path), inter="EFS", scheme=8)
    assert config.read_sc2d(str(tmp_path))["scheme"] == 8
    write_sc(str(tmp_path), inter="EFS", scheme=6)
    with pytest.raises(config.ConfigError):
        config.read_sc2d(str(tmp_path))
    write_sc(str(tmp_path), inter="ShanChen", scheme=10)       # original Shan-Chen ignores the key
    assert config.read_sc2d(str(tmp_path))["scheme"] == 4


def test_rk3d_ini(tmp_path):
    from ini_fixtures import write_rk3d
    write_rk3d(str(tmp_path), relax="MRT")
    p = config.read_rk3d(str(tmp_path))
    assert (p["nx"], p["ny"], p["nz"], p["steps"]) == (32, 32, 96, 1000)
    assert p["relax"] == "MRT" and p["AkR"] == 7.0e-3 and p["tauB"] == 0.9 and p["velocityZB"] == -1.0e-4
    assert p["SolidRhoR"] == 0.7 and p["densityRL"] == 1.0e-8 and not p["image"]
    assert p["cycle"] is False and p["last_step"] == 350          # [CyclesSetup] as shipped (RKtwophasesetup3D.ini:57-59)
    assert p["inlet"] == "Neumann"
    import re
    ini = tmp_path / "RKtwophasesetup3D.ini"
    text = ini.read_text()
    ini.write_text(re.sub(r"(?m)^(\s*BoundaryTypeInlet\s*=).*$", r"\1 'Dirichlet'\ndensityRH = 1e-8\ndensityBH = 1.003", text))
    q = config.read_rk3d(str(tmp_path))
    assert q["inlet"] == "Dirichlet" and q["densityBH"] == 1.003 and q["densityRH"] == 1e-8
    ini.write_text(re.sub(r"(?m)^(\s*BoundaryTypeOutlet\s*=).*$", r"\1 'Convective'", text))
    assert config.read_rk3d(str(tmp_path))["outlet"] == "Convective"
    ini.write_text(re.sub(r"(?m)^(\s*BoundaryTypeOutlet\s*=).*$", r"\1 'Freeflow'", text))
    with pytest.raises(config.ConfigError, match="Convective"):
        config.read_rk3d(str(tmp_path))
    write_rk3d(str(tmp_path), alpha="0.2")          # read, warned about, without effect (AcceleratedRKGPU2D.py:1140: loaded, never used)
    with pytest.warns(UserWarning, match="no effect"):
        assert config.read_rk3d(str(tmp_path))["AlphaR"] == 0.2
    write_rk3d(str(tmp_path), relax="TRT")
    with pytest.raises(config.ConfigError):
        config.read_rk3d(str(tmp_path))


def test_rk3d_ini_with_the_2d_files_surface_tension_section(tmp_path):
    """[SurfaceTension] SurfaceTensionType = 'CSF' in the 3-D file selects the CSF loop carried to D3Q19 (lbmpm_rk3dcsf_*)"""
    from ini_fixtures import write_rk3d, write_rk3d_csf
    write_rk3d(str(tmp_path))
    assert config.read_rk3d(str(tmp_path))["tension_type"] == "Perturbation"          # the shipped file has no such section
    write_rk3d_csf(str(tmp_path), sigma=0.03, theta=35.0, relax="MRT")
    p = config.read_rk3d(str(tmp_path))
    assert p["tension_type"] == "CSF" and p["sigma"] == 0.03 and p["theta"] == 35.0 and p["wetting"] == 2
    assert p["delta"] == 0.98 and p["tautype"] == 2 and p["relax"] == "MRT"
    write_rk3d_csf(str(tmp_path), wetting=1)
    with pytest.raises(config.ConfigError, match="WettingType"):
        config.read_rk3d(str(tmp_path))
