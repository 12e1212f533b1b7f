"""Own ini fixtures in the reference's grammar (quoted strings, comments, key spellings of the
shipped files incl. their skews) for the config / driver tests."""

RK_INI = """[ImageSetup]
Existance = 'no'

[DomainSize]
xDomain = {nx}
yDomain = {ny}
numBufferingLayers = 10
ratioTopToBottom = 0.5

[SurfaceTension]
SurfaceTensionType = 'CSF'
;;'CSF'-> continuum surface force
SurfaceTension = 0.1
ContactAngle = 60
WettingType = 2
;;1 - Y. Xu et al 2017

[RKParameters]
AlphaR = 0.44444444
AlphaB = 0.44444444
BetaThickness = 0.7
AkR = 1.4e-1
AkB = 1.4e-1
DeltaValue = 0.98

[FluidParameters]
TauR = 1.0
TauB = 1.0
InitialRhoR = 1.0
InitialRhoB = 1.0
TauType = 2

[BodyForce]
bodyForceX = 0.0
bodyForceY = 0.0

[SolidBoundarySetup]
SolidColorDiff = 0.5

[BoundaryCondition]
BoundaryTypeInlet = 'Neumann'
NeumannType = 'ZouHe'
velocityYR = -1.0e-4
densityBH = 5e-8
densityRH = 1.00536
velocityYB = 0.0
BoundaryTypeOutlet = 'Dirichlet'
densityBL = 1.0
densityRL = 5e-8

[GradientType]
Type = 'Isotropic'

[TimeSetup]
TimeSteps = {steps}
TimeInterval = {interval}

[Parallelism]
Parallel = 'yes'
xDimension = 128
ThreadsNum = 32

[RelaxationType]
Type = '{relax}'
;;MRT

[CyclesSetup]
IsCycle = '{cycle}'
LastStep = {last}
"""

TWOPHASE_INI = """[PictureSetup]
Exist = '{image}'
[SeparationBorder]
xGrid = {nx}
yGrid = {ny}
[FluidsTypes]
NumberOfFluids = 2
[InterType]
InteractionType = '{inter}'
;;ShanChen: original Shan-Chen Model
[Parallelism]
Parallel = 'yes'
xDimension = 256
ThreadsNum = 32
[RelaxationType]
Type = '{relax}'
[DuplicateDomain]
Option = 'no'
[DICycles]
Option = '{cycle}'
LastStep =  {last}
"""

MODEL_INI = """[FluidProperties]
InitialDensities = 1.0,1.0
BackgroundDensities = {bg},{bg}
FluidsTau = 1.,1.

[{section}]
interactionFluid = {G}
interactionSolid = {Gs0},{Gs1}
potentialType = 'Simple'

[BoundaryDefinition]
BoundaryTypeInlet = 'Neumann'
BoundaryMethod = '{method}'
BoundaryTypeOutlet = '{outlet}'

[VelocityBoundary]
velocityX = 0.0,0.0
velocityY = 0.0,{vy1}

[PressureBoundary]
PressureInlet = 0.0, 0.0
PressureOutlet = 1.0, 0.0

[ForceScheme]
ExplicitScheme = {scheme}

[BodyForce]
Option = 'no'
forceXG = 0.0
forceYG = 0.0

[Time]
numberTimeStep = {steps}
"""


def write_rk(d, nx=20, ny=48, steps=60, interval=25, relax="MRT", cycle="no", last=100):
    import os
    with open(os.path.join(d, "RKtwophasesetup2D.ini"), "w") as fh:
        fh.write(RK_INI.format(nx=nx, ny=ny, steps=steps, interval=interval, relax=relax, cycle=cycle, last=last))


def write_sc(d, inter="EFS", nx=20, ny=48, steps=80, relax="SRT", outlet="Dirichlet", scheme=4, image="no", cycle="no",
             last=0, method="ZouHe"):
    import os
    with open(os.path.join(d, "twophasesetup.ini"), "w") as fh:
        fh.write(TWOPHASE_INI.format(nx=nx, ny=ny, inter=inter, relax=relax, image=image, cycle=cycle, last=last))
    efs = inter == "EFS"
    with open(os.path.join(d, "efs2D.ini" if efs else "shanchen2D.ini"), "w") as fh:
        fh.write(MODEL_INI.format(section="EFSParameters" if efs else "ShanChenParameters", bg=0.02 if efs else 0.06,
                                  G=0.20 if efs else 3.8, Gs0=-0.14 if efs else -0.40, Gs1=0.14 if efs else 0.40,
                                  outlet=outlet, vy1=-5.03e-4 if efs else -1.01e-3, steps=steps, scheme=scheme, method=method))


TRANSPORT_INI = """[SystemType]
Option = 'MPMC'
Reaction = 'no'
Precipitation = 'no'
NumberSchemes = 5

[TransportParameters]
NumberTracers = 2
DiffusionJ = 0.3333333333333333, 0.3333333333333333
Tau = 1.0, 1.0
BetaInterface = 0.8

[BoundaryCondition]
InletType = 'Dirichlet'
ConcentrationInlet = 1.0, 0.25
OutletType = 'Freeflow'

[InitialCondition]
Type = 'Homogeneous'
TracerConc = 1.0, 0.5

[FluidForTransport]
FluidType = 0

[RelaxationType]
Relaxation = 'MRT'

[TransportMRT]
DiffusionX = 0.16666666666666666, 0.12
DiffusionY = 0.16666666666666666, 0.2
DiffusionXY = 0.01, 0.01
DiffusionYX = 0.02, 0.02
"""


def write_transport(d):
    import os
    with open(os.path.join(d, "transportsetup.ini"), "w") as fh:
        fh.write(TRANSPORT_INI)


# key spellings of IniFiles/RKtwophasesetup3D.ini (own values)
RK3D_INI = """[ImageSetup]
Existance = 'no'

[DomainSize]
xDomain = {nx}
yDomain = {ny}
zDomain = {nz}

[RKParameters]
AlphaR = {alpha}
AlphaB = 0.
BetaThickness = 1.
AkR = 7.0e-3
AkB = 7.0e-3
DeltaValue = 0.98

[FluidParameters]
TauR = 1.0
TauB = 0.9
InitialRhoR = 1.0
InitialRhoB = 1.0

[BoundariesSetup]
SolidRhoR = 0.7
SolidRhoB = 0.0

[BoundaryCondition]
BoundaryTypeInlet = 'Neumann'
NeumannType = 'ZouHe'
velocityZR = 0.0
;;-2.5e-4
velocityZB = -1.0e-4
BoundaryTypeOutlet = 'Dirichlet'
densityBL = 1.0
densityRL = 1.0e-8

[GradientType]
Type = 'Isotropic'
;;'Isotropic'

[TimeSteps]
TimeSteps = {steps}

[Parallelism]
Parallel = 'yes'
xDimension = 128
ThreadsNum = 32

[RelaxationType]
Type = '{relax}'
;;MRT

[CyclesSetup]
IsCycle = 'no'
LastStep = 350
"""


# the [SurfaceTension] section of RKtwophasesetup2D.ini (:14-18) and its TauType: with them the 3-D driver runs the CSF loop
RK3D_CSF_EXTRA = """
[SurfaceTension]
SurfaceTensionType = 'CSF'
SurfaceTension = {sigma}
ContactAngle = {theta}
WettingType = {wetting}
"""


def write_rk3d_csf(d, sigma=0.05, theta=60.0, wetting=2, **kw):
    import os
    write_rk3d(d, **kw)
    with open(os.path.join(d, "RKtwophasesetup3D.ini"), "a") as fh:
        fh.write(RK3D_CSF_EXTRA.format(sigma=sigma, theta=theta, wetting=wetting))


def write_rk3d(d, nx=32, ny=32, nz=96, steps=1000, relax="SRT", alpha="0."):
    import os
    with open(os.path.join(d, "RKtwophasesetup3D.ini"), "w") as fh:
        fh.write(RK3D_INI.format(nx=nx, ny=ny, nz=nz, steps=steps, relax=relax, alpha=alpha))
