"""Reader for the reference's ini files (IniFiles/*.ini), same grammar and key names.

String values keep their single quotes in the files (`Type = 'MRT'`; the reference compares
against "'MRT'", RKD2Q9.py:36); option names are case-insensitive (configparser).  Known skews of
the shipped files are accepted both ways: `SurfaceTension` vs
`SurfaceTensionValue`; missing `[BodyForce] isBodyForce`.  Errors raise ConfigError (the
reference prints and sys.exit()s).
"""
import configparser
import os


class ConfigError(ValueError):
    pass


def _unquote(v):
    v = v.strip()
    if len(v) >= 2 and v[0] == "'" and v[-1] == "'":
        return v[1:-1]
    return v


class Ini:
    def __init__(self, path):
        if not os.path.isfile(path):
            raise ConfigError("ini file not found: %s" % path)
        self.path = path
        self.cp = configparser.ConfigParser()
        self.cp.read(path)

    def raw(self, section, *keys, default=None):
        if section not in self.cp:
            if default is not None:
                return default
            raise ConfigError("%s: missing section [%s]" % (self.path, section))
        for k in keys:
            if k in self.cp[section]:
                return self.cp[section][k]
        if default is not None:
            return default
        raise ConfigError("%s: missing key %s in [%s]" % (self.path, "/".join(keys), section))

    def str(self, section, *keys, default=None):
        return _unquote(self.raw(section, *keys, default=default))

    def _num(self, conv, section, keys, default):
        v = self.raw(section, *keys, default=None if default is None else repr(default))
        try:
            return conv(_unquote(v).split(";")[0])
        except ValueError:
            raise ConfigError("%s: [%s] %s = %r is not a %s" % (self.path, section, keys[0], v, conv.__name__))

    def float(self, section, *keys, default=None):
        return self._num(float, section, keys, default)

    def int(self, section, *keys, default=None):
        return self._num(int, section, keys, default)

    def floats(self, section, key, count=None):
        v = [p for p in self.raw(section, key).split(",")]
        try:
            out = [float(p) for p in v]
        except ValueError:
            raise ConfigError("%s: [%s] %s must be a comma list of numbers" % (self.path, section, key))
        if count is not None and len(out) != count:
            raise ConfigError("%s: [%s] %s needs %d values" % (self.path, section, key, count))
        return out


def read_rk2d(ini_dir):
    """RKtwophasesetup2D.ini -> dict (keys consumed by RKD2Q9.py:26-297)."""
    c = Ini(os.path.join(ini_dir, "RKtwophasesetup2D.ini"))
    p = {}
    p["image"] = c.str("ImageSetup", "Existance") == "yes"
    if not p["image"]:
        p["nx"] = c.int("DomainSize", "xDomain"); p["ny"] = c.int("DomainSize", "yDomain")
    p["nbuf"] = c.int("DomainSize", "numBufferingLayers")
    p["ratio"] = c.float("DomainSize", "ratioTopToBottom")
    p["tension_type"] = c.str("SurfaceTension", "SurfaceTensionType")
    if p["tension_type"] not in ("CSF", "Perturbation"):
        raise ConfigError("SurfaceTensionType must be 'CSF' or 'Perturbation'")
    if p["tension_type"] == "Perturbation":
        # the reference's runRKColorGradient2DPerturbation stops at its first inlet launch (RKD2Q9.py:1099 passes 10
        # arguments to a 12-argument kernel); RKColorGradientLBM.runRKColorGradient2DPerturbation here runs that loop
        # with the four call-site repairs its golden captures list (tests/golden/gen/make_golden_rk_pert.py)
        p["AkR"] = c.float("RKParameters", "AkR"); p["AkB"] = c.float("RKParameters", "AkB")      # RKD2Q9.py:116-124
        p["solidPhi"] = c.float("SolidBoundarySetup", "SolidColorDiff")                            # RKD2Q9.py:182
    p["sigma"] = c.float("SurfaceTension", "SurfaceTensionValue", "SurfaceTension", default=0.0 if p["tension_type"] == "Perturbation" else None)
    p["theta"] = c.float("SurfaceTension", "ContactAngle", default=90.0 if p["tension_type"] == "Perturbation" else None)
    p["wetting"] = c.int("SurfaceTension", "WettingType", default=2 if p["tension_type"] == "Perturbation" else None)
    p["beta"] = c.float("RKParameters", "BetaThickness")
    p["delta"] = c.float("RKParameters", "DeltaValue")
    p["tauR"] = c.float("FluidParameters", "TauR"); p["tauB"] = c.float("FluidParameters", "TauB")
    p["rho0R"] = c.float("FluidParameters", "InitialRhoR"); p["rho0B"] = c.float("FluidParameters", "InitialRhoB")
    p["tautype"] = c.int("FluidParameters", "TauType")
    if c.str("BodyForce", "isBodyForce", default="'no'") == "yes":
        raise ConfigError("body force is read but never used by the reference's CSF loop; not supported")
    p["steps"] = c.int("TimeSetup", "TimeSteps"); p["interval"] = c.int("TimeSetup", "TimeInterval")
    p["relax"] = c.str("RelaxationType", "Type")
    p["inlet"] = c.str("BoundaryCondition", "BoundaryTypeInlet")
    p["outlet"] = c.str("BoundaryCondition", "BoundaryTypeOutlet")
    p["vyR"] = c.float("BoundaryCondition", "VelocityYR", default=0.0)
    p["vyB"] = c.float("BoundaryCondition", "VelocityYB", default=0.0)
    p["rhoBH"] = c.float("BoundaryCondition", "densityBH", default=1.0)
    p["rhoRH"] = c.float("BoundaryCondition", "densityRH", default=1.0)
    p["rhoBL"] = c.float("BoundaryCondition", "densityBL", default=1.0)
    p["rhoRL"] = c.float("BoundaryCondition", "densityRL", default=1.0)
    p["cycle"] = c.str("CyclesSetup", "IsCycle", default="'no'") == "yes"
    p["last_step"] = c.int("CyclesSetup", "LastStep", default=0)
    if p["tension_type"] == "Perturbation":
        # the perturbation driver does not go through RK2DSolver's checks; a misspelt value must not pick a physics path silently
        # (the reference's own outlet switch is if / elif without else: any other value would run with NO outlet rule, RKD2Q9.py:1064-1088)
        for key, allowed in (("relax", ("SRT", "MRT")), ("inlet", ("Neumann", "Dirichlet")), ("outlet", ("Convective", "Dirichlet"))):
            if p[key] not in allowed:
                raise ConfigError("SurfaceTensionType 'Perturbation': %s must be one of %s, got %r"
                                  % ({"relax": "[RelaxationType] Type", "inlet": "BoundaryTypeInlet", "outlet": "BoundaryTypeOutlet"}[key], allowed, p[key]))
    return p


def read_rk3d(ini_dir):
    """RKtwophasesetup3D.ini -> dict.  The module that consumed this file (RKColorGradientD3Q19,
    main.py:22) is not in the reference tree; the keys are read by name as the ini spells them."""
    c = Ini(os.path.join(ini_dir, "RKtwophasesetup3D.ini"))
    p = {}
    p["image"] = c.str("ImageSetup", "Existance", default="'no'") == "yes"
    if not p["image"]:
        p["nx"] = c.int("DomainSize", "xDomain"); p["ny"] = c.int("DomainSize", "yDomain")
        p["nz"] = c.int("DomainSize", "zDomain")
    for k in ("AlphaR", "AlphaB"):
        # the kernels of the perturbation loop this model extends load the rest weights C_i(alpha) and never use them
        # (AcceleratedRKGPU2D.py:1140 / :1190; equilibria are calEquilibriumRK2D, :170): any value gives the same numbers
        p[k] = c.float("RKParameters", k, default=0.0)
        if p[k] != 0.0:
            import warnings
            warnings.warn("[RKParameters] %s = %g has no effect, as in the reference's perturbation kernels "
                          "(AcceleratedRKGPU2D.py:1125-1267 load the rest weights and do not use them)" % (k, p[k]))
    # [SurfaceTension] is not in the shipped 3-D file (its parameters are the perturbation loop's AkR / AkB); with the 2-D file's section
    # (RKtwophasesetup2D.ini, RKD2Q9.py:72-101) the 3-D driver runs the CSF loop carried to D3Q19 (lbmpm_rk3dcsf_*)
    p["tension_type"] = c.str("SurfaceTension", "SurfaceTensionType", default="'Perturbation'")
    if p["tension_type"] not in ("CSF", "Perturbation"):
        raise ConfigError("SurfaceTensionType must be 'CSF' or 'Perturbation'")
    if p["tension_type"] == "CSF":
        p["sigma"] = c.float("SurfaceTension", "SurfaceTensionValue", "SurfaceTension")
        p["theta"] = c.float("SurfaceTension", "ContactAngle")
        p["wetting"] = c.int("SurfaceTension", "WettingType", default=2)
        if p["wetting"] != 2:
            raise ConfigError("3-D CSF: WettingType 2 (Akai et al. 2018); WettingType 1 (AcceleratedRKGPU2D.py:1639-1679) rotates the normal in the plane "
                              "and has no 3-D form")
        p["delta"] = c.float("RKParameters", "DeltaValue")
        p["tautype"] = c.int("FluidParameters", "TauType", default=2)
        if p["tautype"] not in (1, 2):
            raise ConfigError("[FluidParameters] TauType must be 1 or 2")
        p["AkR"] = c.float("RKParameters", "AkR", default=0.0); p["AkB"] = c.float("RKParameters", "AkB", default=0.0)      # read, unused by the CSF loop
    else:
        p["AkR"] = c.float("RKParameters", "AkR"); p["AkB"] = c.float("RKParameters", "AkB")
    p["beta"] = c.float("RKParameters", "BetaThickness")
    p["tauR"] = c.float("FluidParameters", "TauR"); p["tauB"] = c.float("FluidParameters", "TauB")
    p["rho0R"] = c.float("FluidParameters", "InitialRhoR", default=1.0)
    p["rho0B"] = c.float("FluidParameters", "InitialRhoB", default=1.0)
    csf = p["tension_type"] == "CSF"      # (the CSF loop takes phi on the walls from the fluid next to them, AcceleratedRKGPU2D.py:1560-1581)
    p["SolidRhoR"] = c.float("BoundariesSetup", "SolidRhoR", default=0.5 if csf else None); p["SolidRhoB"] = c.float("BoundariesSetup", "SolidRhoB", default=0.5 if csf else None)
    if p["SolidRhoR"] + p["SolidRhoB"] == 0.0:
        raise ConfigError("[BoundariesSetup] SolidRhoR + SolidRhoB must not be zero")
    p["inlet"] = c.str("BoundaryCondition", "BoundaryTypeInlet")
    if p["inlet"] not in ("Neumann", "Dirichlet") or (p["inlet"] == "Neumann" and c.str("BoundaryCondition", "NeumannType", default="'ZouHe'") != "ZouHe"):
        raise ConfigError("3-D inlet: BoundaryTypeInlet = 'Neumann' with NeumannType = 'ZouHe' (velocityZR / velocityZB) or 'Dirichlet' "
                          "(densityRH / densityBH: Zou-He pressure per colour, the keys of RKtwophasesetup2D.ini)")
    p["densityRH"] = c.float("BoundaryCondition", "densityRH", default=1.0)
    p["densityBH"] = c.float("BoundaryCondition", "densityBH", default=1.0)
    if p["inlet"] == "Dirichlet" and not (p["densityRH"] > 0.0 and p["densityBH"] > 0.0):
        raise ConfigError("3-D pressure inlet: densityRH and densityBH must be positive (the closure divides by them; the absent colour gets e.g. 1e-8)")
    p["outlet"] = c.str("BoundaryCondition", "BoundaryTypeOutlet")
    if p["outlet"] not in ("Dirichlet", "Convective"):
        raise ConfigError("3-D outlet: BoundaryTypeOutlet = 'Dirichlet' (densityRL / densityBL) or 'Convective' (the planes 0 .. 2 copy plane 3)")
    p["velocityZR"] = c.float("BoundaryCondition", "velocityZR", default=0.0)
    p["velocityZB"] = c.float("BoundaryCondition", "velocityZB", default=0.0)
    p["densityBL"] = c.float("BoundaryCondition", "densityBL", default=1.0)
    p["densityRL"] = c.float("BoundaryCondition", "densityRL", default=1.0)
    if c.str("GradientType", "Type", default="'Isotropic'") != "Isotropic":
        raise ConfigError("[GradientType] Type: only 'Isotropic'")
    p["steps"] = c.int("TimeSteps", "TimeSteps")
    p["interval"] = c.int("TimeSteps", "TimeInterval", default=0)       # not in the shipped file
    p["relax"] = c.str("RelaxationType", "Type", default="'SRT'")
    if p["relax"] not in ("SRT", "MRT"):
        raise ConfigError("[RelaxationType] Type must be 'SRT' or 'MRT'")
    # [CyclesSetup] (RKtwophasesetup3D.ini:57-59): the 2-D rules of RKD2Q9.py:491-559 carried to z (RKColorGradientD3Q19.py)
    p["cycle"] = c.str("CyclesSetup", "IsCycle", default="'no'") == "yes"
    p["last_step"] = c.int("CyclesSetup", "LastStep", default=0)
    return p


def read_sc2d(ini_dir):
    """twophasesetup.ini + efs2D.ini | shanchen2D.ini -> dict (ShanChenD2Q9.py:42-157, :172-499)."""
    c = Ini(os.path.join(ini_dir, "twophasesetup.ini"))
    p = {}
    p["image"] = c.str("PictureSetup", "Exist") == "yes"
    if not p["image"]:
        p["nx"] = c.int("SeparationBorder", "xGrid"); p["ny"] = c.int("SeparationBorder", "yGrid")
    if c.int("FluidsTypes", "NumberOfFluids") != 2:
        raise ConfigError("NumberOfFluids must be 2 (the reference's outlet kernel is hard-wired to two fluids)")
    p["inter"] = c.str("InterType", "InteractionType")
    if p["inter"] not in ("ShanChen", "EFS"):
        raise ConfigError("InteractionType must be 'ShanChen' or 'EFS'")
    p["relax"] = c.str("RelaxationType", "Type")
    p["duplicate"] = None
    if c.str("DuplicateDomain", "Option", default="'no'") == "yes":
        # the reference asks for the two numbers with input() (ShanChenD2Q9.py:573-574); here they are two more
        # keys of the same section, or the `duplicate=` argument of the driver
        p["duplicate"] = (c.int("DuplicateDomain", "xDirectionNumber", default=1), c.int("DuplicateDomain", "yDirectionNumber", default=1))
    p["cycle"] = c.str("DICycles", "Option", default="'no'") == "yes"          # ShanChenD2Q9.py:148-157
    p["last_step"] = c.int("DICycles", "LastStep", default=0)
    if p["cycle"] and not p["image"]:
        raise ConfigError("[DICycles] Option = 'yes' initialises pore-image domains only (ShanChenD2Q9.py:788)")
    m = Ini(os.path.join(ini_dir, "efs2D.ini" if p["inter"] == "EFS" else "shanchen2D.ini"))
    sec = "EFSParameters" if p["inter"] == "EFS" else "ShanChenParameters"
    p["rho0"], p["rho1"] = m.floats("FluidProperties", "InitialDensities", 2)
    p["bg0"], p["bg1"] = m.floats("FluidProperties", "BackgroundDensities", 2)
    p["tau0"], p["tau1"] = m.floats("FluidProperties", "FluidsTau", 2)
    p["G"] = m.floats(sec, "InteractionFluid", 1)[0]
    p["Gs0"], p["Gs1"] = m.floats(sec, "InteractionSolid", 2)
    if m.str("BoundaryDefinition", "BoundaryTypeInlet") != "Neumann":
        raise ConfigError("only BoundaryTypeInlet 'Neumann' runs in the reference "
                          "(the Dirichlet inlet references undefined attributes, ShanChenD2Q9.py:1497)")
    p["method"] = m.str("BoundaryDefinition", "BoundaryMethod")
    if p["method"] not in ("ZouHe", "Chang"):
        raise ConfigError("[BoundaryDefinition] BoundaryMethod must be 'ZouHe' or 'Chang'")
    p["outlet"] = m.str("BoundaryDefinition", "BoundaryTypeOutlet")
    if p["outlet"] not in ("Dirichlet", "Convective", "Freeflow"):
        raise ConfigError("[BoundaryDefinition] BoundaryTypeOutlet must be 'Dirichlet', 'Convective' or 'Freeflow'")
    if p["inter"] != "EFS" and p["outlet"] == "Freeflow":
        # the original Shan-Chen loop tests for 'Convective' only (ShanChenD2Q9.py:1599): any other value, this one included,
        # runs without an outlet rule -- which is what 'Dirichlet' selects in the solver for that loop
        import warnings
        warnings.warn("BoundaryTypeOutlet 'Freeflow' has no branch in the original Shan-Chen loop (ShanChenD2Q9.py:1599): "
                      "running without an outlet rule, as the reference would")
        p["outlet"] = "Dirichlet"
    if p["outlet"] == "Freeflow" and p["relax"] != "SRT":
        raise ConfigError("BoundaryTypeOutlet 'Freeflow' with MRT: the reference's loop copies the outlet rows after its moment "
                          "transforms (ShanChenD2Q9.py:1855-1884) and the run turns NaN; use SRT")
    p["vy0"], p["vy1"] = m.floats("VelocityBoundary", "velocityY", 2)
    p["scheme"] = m.int("ForceScheme", "ExplicitScheme") if p["inter"] == "EFS" else 4
    if p["scheme"] not in (4, 8, 10):
        raise ConfigError("[ForceScheme] ExplicitScheme must be 4, 8 or 10")
    p["steps"] = m.int("Time", "numberTimeStep")
    return p


def read_transport(ini_dir):
    """transportsetup.ini -> dict (keys consumed by Transport2DRK.py:31-311; the file itself is not
    shipped with the reference).  Only the combinations its working loop runTransport2DMPMCRKNew
    can execute are accepted: multiphase flow system, D2Q5, MRT, Dirichlet (Inamuro) inlet or none,
    free-flow outlet or none."""
    c = Ini(os.path.join(ini_dir, "transportsetup.ini"))
    p = {}
    if c.str("SystemType", "Option") != "MPMC":
        raise ConfigError("[SystemType] Option: only 'MPMC' (tracers in the two-phase flow) is part of the GPU path")
    p["reaction_rate"] = 0.0
    if c.str("SystemType", "Reaction") == "yes":
        # A + B -> C between tracers 0, 1, 2; the kernel uses the first rate only (AccelerateTransport2DRK.py:105-107)
        nr = c.int("Reaction", "NumberReaction")
        rates = c.floats("Reaction", "ReactionRate")
        if nr < 1 or len(rates) < 1:
            raise ConfigError("[Reaction] needs NumberReaction >= 1 and ReactionRate")
        p["reaction_rate"] = rates[0]
    p["precipitation"] = c.str("SystemType", "Precipitation")      # read (Transport2DRK.py:91) and never used by the reference either
    if c.int("SystemType", "NumberSchemes") != 5:
        # NumberSchemes = 9: the loop hands the D2Q9 kernels its five-entry D2Q5 direction tables (Transport2DRK.py:1395 ->
        # AccelerateTransport2DRK.py:1084 indexes them with 0..8: IndexError under emulation, out-of-bounds reads on a GPU)
        raise ConfigError("[SystemType] NumberSchemes: only the D2Q5 scheme runs in the reference (Transport2DRK.py:1343-1384); "
                          "its D2Q9 branch indexes five-entry direction tables with nine directions (:1395)")
    n = c.int("TransportParameters", "NumberTracers")
    if not 1 <= n <= 4:
        raise ConfigError("[TransportParameters] NumberTracers must be 1..4")
    p["num_tracers"] = n
    if p["reaction_rate"] and n != 3:
        raise ConfigError("[SystemType] Reaction = 'yes' couples exactly three tracers (NumberTracers = 3)")
    p["diffJ"] = c.floats("TransportParameters", "DiffusionJ", n)
    p["tau"] = c.floats("TransportParameters", "Tau", n)
    p["beta"] = c.float("TransportParameters", "BetaInterface")
    # The loop acts on exact spellings (Transport2DRK.py:1363, 1378) that differ from the ones its reader knows
    # (:156-191): 'Dirichlet' -> Inamuro inlet row, 'Freeflow' -> calFreeConcBoundary3 on the outlet rows; any other
    # value the reader accepts runs WITHOUT that boundary rule (the tracer streaming wraps around in y then).
    p["inlet_type"] = c.str("BoundaryCondition", "InletType")
    if p["inlet_type"] == "Neumann":
        raise ConfigError("[BoundaryCondition] InletType = 'Neumann': the reference's reader fails on it "
                          "(Transport2DRK.py:160 uses the undefined self.concGradientUpper)")
    p["inlet_conc"] = c.floats("BoundaryCondition", "ConcentrationInlet", n) if p["inlet_type"] == "Dirichlet" else [1.0] * n
    p["outlet_type"] = c.str("BoundaryCondition", "OutletType")
    if p["outlet_type"] not in ("Freeflow", "FreeFlow", "Dirichlet", "VonNeumann"):
        raise ConfigError("[BoundaryCondition] OutletType must be 'Freeflow' (the spelling the loop acts on), 'FreeFlow', "
                          "'Dirichlet' or 'VonNeumann' (read, then ignored by the loop)")
    p["init_type"] = c.str("InitialCondition", "Type")
    if p["init_type"] == "Homogeneous":
        p["init_conc"] = c.floats("InitialCondition", "TracerConc", n)
    p["fluid"] = c.int("FluidForTransport", "FluidType")            # ditto (Transport2DRK.py:215)
    if c.str("RelaxationType", "Relaxation") != "MRT":
        raise ConfigError("[RelaxationType] Relaxation: the D2Q5 path is MRT (calCollisionTransportLinearEqlMRTGPU)")
    p["diffX"] = c.floats("TransportMRT", "DiffusionX", n)
    p["diffY"] = c.floats("TransportMRT", "DiffusionY", n)
    # the reference indexes these two by tracer when reading and uses the whole list as a scalar
    # when filling S (Transport2DRK.py:334-335): one value for all tracers is what it can run with
    p["dXY"] = c.floats("TransportMRT", "DiffusionXY", n)[0]
    p["dYX"] = c.floats("TransportMRT", "DiffusionYX", n)[0]
    return p
