#!/usr/bin/env python3
"""One-off generator of the kernel-level (drop-in) boundary from ONE spec table:
  include/lbmpm_kernels.h                         C declarations, one per reference kernel
  openlbmpm_amd/csrc/sparse_entry_gen.h           extern "C" definitions forwarding to the launchers
  openlbmpm_amd/_kernel_specs.py                  ctypes signatures for the Python shim
Argument lists are the reference kernels' own (module, name, citation), minus the launch
configuration.  kinds: i = int64, d = float64, I = int64*, D = float64*, B = boolean* (one byte per entry, numpy bool),
L = int64* iterated over its whole length by the reference kernel (`for m in list`): the C entry point takes the length right after it
(<name>_len), the Python shim passes the array's size."""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
A = "RKCG2D/AcceleratedRKGPU2D.py"; O = "ShanChen2D/OptimizedD2Q9GPU.py"; E = "ShanChen2D/ExplicitD2Q9GPU.py"
T = "RKCG2D/AccelerateTransport2DRK.py"; B = "RKCG2D/RKGPU2DBoundary.py"; DE = "ShanChen2D/AccelerateGPU2D.py"

# (module tag, reference kernel, file:line, "name:kind ...", launcher call)
SPEC = [
 ("rk", "fillNeighboringNodes", A + ":15", "totalNodes:i nx:i ny:i xDim:i fluidNodes:I domainNewIndex:I neighboringNodes:I",
  "launch_rk_fill_neighbors(st, totalNodes, nx, ny, fluidNodes, domainNewIndex, neighboringNodes)"),
 ("rk", "fillNeighboringWettingNodes", A + ":58", "totalWettingNodes:i nx:i ny:i xDim:i wettingNodes:I domainNewIndex:I neighboringWettingNodes:I",
  "launch_rk_fill_neighbors(st, totalWettingNodes, nx, ny, wettingNodes, domainNewIndex, neighboringWettingNodes)"),
 ("rk", "calMacroDensityRKGPU2D", A + ":103", "totalNodes:i xDim:i fluidPDFR:D fluidPDFB:D fluidRhoR:D fluidRhoB:D",
  "launch_rk_macro_density(st, totalNodes, fluidPDFR, fluidPDFB, fluidRhoR, fluidRhoB)"),
 ("rk", "calStreaming1GPU", A + ":340", "totalNum:i xDim:i fluidNodes:I neighboringNodes:I fluidPDF:D fluidPDFNew:D",
  "launch_rk_stream1(st, totalNum, neighboringNodes, fluidPDF, fluidPDFNew)"),
 ("rk", "calStreaming2GPU", A + ":409", "totalNum:i xDim:i fluidPDFNew:D fluidPDF:D",
  "launch_rk_stream2(st, totalNum, fluidPDFNew, fluidPDF)"),
 ("rk", "ghostPointsConstantVelocityRK", A + ":607", "totalNodes:i nx:i ny:i xDim:i fluidNodes:I neighboringNodes:I fluidRhoR:D fluidRhoB:D fluidPDFR:D fluidPDFB:D forceX:D forceY:D",
  "launch_rk_ghost_inlet_velocity(st, totalNodes, nx, ny, fluidNodes, neighboringNodes, fluidRhoR, fluidRhoB, fluidPDFR, fluidPDFB)"),
 ("rk", "convectiveOutletGPU", A + ":700", "totalNodes:i nx:i xDim:i fluidNodes:I neighboringNodes:I fluidPDFR:D fluidPDFB:D fluidRhoR:D fluidRhoB:D",
  "launch_rk_outlet_convective_row(st, totalNodes, nx, 2, fluidNodes, neighboringNodes, fluidPDFR, fluidPDFB, fluidRhoR, fluidRhoB)"),
 ("rk", "convectiveOutletGhost2GPU", A + ":731", "totalNodes:i nx:i xDim:i fluidNodes:I neighboringNodes:I fluidPDFR:D fluidPDFB:D fluidRhoR:D fluidRhoB:D",
  "launch_rk_outlet_convective_row(st, totalNodes, nx, 1, fluidNodes, neighboringNodes, fluidPDFR, fluidPDFB, fluidRhoR, fluidRhoB)"),
 ("rk", "convectiveOutletGhost3GPU", A + ":762", "totalNodes:i nx:i xDim:i fluidNodes:I neighboringNodes:I fluidPDFR:D fluidPDFB:D fluidRhoR:D fluidRhoB:D",
  "launch_rk_outlet_convective_row(st, totalNodes, nx, 0, fluidNodes, neighboringNodes, fluidPDFR, fluidPDFB, fluidRhoR, fluidRhoB)"),
 ("rk", "calConstPressureInletGPU", A + ":925", "totalNodes:i nx:i ny:i xDim:i constPHB:d constPHR:d fluidNodes:I fluidRhoB:D fluidRhoR:D fluidPDFB:D fluidPDFR:D",
  "launch_rk_inlet_pressure(st, totalNodes, nx, ny, constPHB, constPHR, fluidNodes, fluidRhoB, fluidRhoR, fluidPDFB, fluidPDFR)"),
 ("rk", "ghostPointsConstPressureInletRK", A + ":968", "totalNodes:i nx:i ny:i xDim:i fluidNodes:I neighboringNodes:I fluidRhoR:D fluidRhoB:D fluidPDFR:D fluidPDFB:D",
  "launch_rk_ghost_inlet_pressure(st, totalNodes, nx, ny, fluidNodes, neighboringNodes, fluidRhoR, fluidRhoB, fluidPDFR, fluidPDFB)"),
 ("rk", "ghostPointsConstPressureLowerRK", A + ":1045", "totalNodes:i nx:i xDim:i fluidNodes:I neighboringNodes:I fluidRhoR:D fluidRhoB:D fluidPDFR:D fluidPDFB:D",
  "launch_rk_ghost_outlet_pressure(st, totalNodes, nx, neighboringNodes, fluidRhoR, fluidRhoB, fluidPDFR, fluidPDFB)"),
 ("rk", "calPhaseFieldPhi", A + ":1348", "totalNodes:i xDim:i fluidRhoR:D fluidRhoB:D phiValue:D",
  "launch_rk_phase_field(st, totalNodes, fluidRhoR, fluidRhoB, phiValue)"),
 ("rk", "calTotalFluidPDF", A + ":1414", "totalNodes:i xDim:i fluidPDFR:D fluidPDFB:D fluidPDFTotal:D",
  "launch_rk_total_pdf(st, totalNodes, fluidPDFR, fluidPDFB, fluidPDFTotal)"),
 ("rk", "calColorValueOnSolid", A + ":1560", "totalSolidWetting:i xDim:i neighboringWettingSolid:I weightsCoeff:D colorValueFluid:D colorValueSolid:D",
  "launch_rk_color_on_solid(st, totalSolidWetting, neighboringWettingSolid, colorValueFluid, colorValueSolid)"),
 ("rk", "calRKInitialGradient", A + ":1584", "totalNodes:i xDim:i numColorSolid:i fluidNodes:I neighboringNodes:I weightsCoeff:D unitEX:D unitEY:D colorValueFluid:D colorValueSolid:D gradientX:D gradientY:D",
  "launch_rk_gradient(st, totalNodes, neighboringNodes, colorValueFluid, colorValueSolid, gradientX, gradientY)"),
 ("rk", "updateColorGradientOnWetting", A + ":1639", "totalFluidWettingNodes:i xDim:i cosTheta:d sinTheta:d fluidNodesWetting:I unitVectorNsx:D unitVectorNsy:D gradientX:D gradientY:D",
  "launch_rk_wetting1(st, totalFluidWettingNodes, cosTheta, sinTheta, fluidNodesWetting, unitVectorNsx, unitVectorNsy, gradientX, gradientY)"),
 ("rk", "updateColorGradientOnWettingNew", A + ":2430", "totalFluidWettingNodes:i xDim:i cosTheta:d sinTheta:d fluidNodesWetting:I unitVectorNsx:D unitVectorNsy:D gradientX:D gradientY:D",
  "launch_rk_wetting2(st, totalFluidWettingNodes, cosTheta, sinTheta, fluidNodesWetting, unitVectorNsx, unitVectorNsy, gradientX, gradientY)"),
 ("rk", "calForceTermInColorGradient2D", A + ":1686", "totalNodes:i xDim:i surfaceTension:d neighboringNodes:I weightsCoeff:D unitEX:D unitEY:D gradientX:D gradientY:D forceX:D forceY:D KValue:D",
  "launch_rk_force(st, totalNodes, 1, surfaceTension, neighboringNodes, gradientX, gradientY, forceX, forceY, KValue)"),
 ("rk", "calForceTermInColorGradientNew2D", A + ":2499", "totalNodes:i xDim:i surfaceTension:d neighboringNodes:I weightsCoeff:D unitEX:D unitEY:D gradientX:D gradientY:D forceX:D forceY:D KValue:D",
  "launch_rk_force(st, totalNodes, 2, surfaceTension, neighboringNodes, gradientX, gradientY, forceX, forceY, KValue)"),
 ("rk", "calPerturbationFromForce2D", A + ":1743", "totalNodes:i xDim:i optionF:i tauR:d tauB:d deltaValue:d weightsCoeff:D unitEX:D unitEY:D physicalVX:D physicalVY:D forceX:D forceY:D colorValue:D fluidTotalPDF:D fluidRhoR:D fluidRhoB:D",
  "launch_rk_force_srt(st, totalNodes, (int)optionF, tauR, tauB, deltaValue, physicalVX, physicalVY, forceX, forceY, colorValue, fluidTotalPDF, fluidRhoR, fluidRhoB)"),
 ("rk", "calRKCollision1TotalGPU2DSRTM", A + ":1804", "totalNodes:i xDim:i optionF:i tauR:d tauB:d deltaValue:d unitEX:D unitEY:D weightsCoeff:D physicalVX:D physicalVY:D fluidRhoR:D fluidRhoB:D ColorValue:D fluidPDFTotal:D",
  "launch_rk_collide_srt(st, totalNodes, (int)optionF, tauR, tauB, deltaValue, physicalVX, physicalVY, fluidRhoR, fluidRhoB, ColorValue, fluidPDFTotal)"),
 ("rk", "calRecoloringProcessM", A + ":1857", "totalNodes:i xDim:i betaValue:d weightsCoeff:D fluidRhoR:D fluidRhoB:D unitEX:D unitEY:D gradientX:D gradientY:D fluidPDFR:D fluidPDFB:D fluidPDFTotal:D",
  "launch_rk_recolor(st, totalNodes, betaValue, fluidRhoR, fluidRhoB, gradientX, gradientY, fluidPDFR, fluidPDFB, fluidPDFTotal)"),
 ("rk", "calRKCollision1TotalGPU2DMRTM", A + ":1938", "totalNodes:i xDim:i optionF:i tauR:d tauB:d deltaValue:d unitEX:D unitEY:D weightsCoeff:D physicalVX:D physicalVY:D fluidRhoR:D fluidRhoB:D ColorValue:D fluidPDFTotal:D transformationM:D inverseTM:D collisionS:D",
  "launch_rk_collide_mrt(st, totalNodes, (int)optionF, tauR, tauB, deltaValue, physicalVX, physicalVY, fluidRhoR, fluidRhoB, ColorValue, fluidPDFTotal, transformationM, inverseTM, collisionS)"),
 ("rk", "calPerturbationFromForce2DMRT", A + ":2027", "totalNodes:i xDim:i optionF:i tauR:d tauB:d deltaValue:d weightsCoeff:D unitEX:D unitEY:D physicalVX:D physicalVY:D forceX:D forceY:D colorValue:D fluidTotalPDF:D transformationM:D inverseTM:D collisionS:D fluidRhoR:D fluidRhoB:D",
  "launch_rk_force_mrt(st, totalNodes, (int)optionF, tauR, tauB, deltaValue, physicalVX, physicalVY, forceX, forceY, colorValue, fluidTotalPDF, transformationM, inverseTM, collisionS, fluidRhoR, fluidRhoB)"),
 ("rk", "constantTotalVelocityInlet", A + ":2348", "totalNodes:i nx:i ny:i xDim:i specificVY:d fluidNodes:I neighboringNodes:I fluidRhoR:D fluidRhoB:D fluidPDFR:D fluidPDFB:D fluidPDFTotal:D physicalVY:D",
  "launch_rk_inlet_velocity_total(st, totalNodes, nx, ny, specificVY, fluidNodes, fluidRhoR, fluidRhoB, fluidPDFR, fluidPDFB, fluidPDFTotal, physicalVY)"),
 ("rk", "calConstPressureLowerGPUTotal", A + ":2560", "totalNodes:i nx:i xDim:i constPL:d fluidNodes:I fluidPDFTotal:D physicalVY:D fluidRhoR:D fluidRhoB:D fluidPDFR:D fluidPDFB:D",
  "launch_rk_outlet_pressure_total(st, totalNodes, nx, constPL, fluidNodes, fluidPDFTotal, physicalVY, fluidRhoR, fluidRhoB, fluidPDFR, fluidPDFB)"),
 ("rk", "calPhysicalVelocityRKGPU2DNew1", A + ":2634", "totalNodes:i xDim:i fluidPDFTotal:D fluidRhoR:D fluidRhoB:D physicalVX:D physicalVY:D forceX:D forceY:D",
  "launch_rk_velocity(st, totalNodes, fluidPDFTotal, fluidRhoR, fluidRhoB, physicalVX, physicalVY, forceX, forceY)"),
 # ---------------- the perturbation loop's kernels (RKD2Q9.py:1046-1223; the 2-D loop the D3Q19 model extends)
 ("rk", "calPhysicalVelocityRKGPU2D", A + ":125", "totalNodes:i xDim:i fluidPDFR:D fluidPDFB:D fluidRhoR:D fluidRhoB:D physicalVX:D physicalVY:D",
  "launch_rk_pert_velocity(st, totalNodes, fluidPDFR, fluidPDFB, fluidRhoR, fluidRhoB, physicalVX, physicalVY)"),
 ("rk", "constantVelocityZHBoundaryHigherRK", A + ":657", "totalNodes:i nx:i ny:i xDim:i specificVYR:d specificVYB:d fluidNodes:I fluidRhoR:D fluidRhoB:D fluidPDFR:D fluidPDFB:D",
  "launch_rk_pert_inlet_velocity(st, totalNodes, nx, ny, specificVYR, specificVYB, fluidNodes, fluidRhoR, fluidRhoB, fluidPDFR, fluidPDFB)"),
 ("rk", "calConstPressureLowerGPU", A + ":1008", "totalNodes:i nx:i xDim:i constPLB:d constPLR:d fluidNodes:I fluidRhoB:D fluidRhoR:D fluidPDFB:D fluidPDFR:D",
  "launch_rk_pert_outlet_pressure(st, totalNodes, nx, constPLB, constPLR, fluidNodes, fluidRhoB, fluidRhoR, fluidPDFB, fluidPDFR, 0)"),
 ("rk", "calRKCollision1GPU2DSRTNew", A + ":1125", "totalNodes:i xDim:i delta:d tauR:d tauB:d unitEX:D unitEY:D constantCR:D constantCB:D weightsCoeff:D physicalVX:D physicalVY:D fluidRhoR:D fluidRhoB:D phiValue:D fluidPDFR:D fluidPDFB:D collisionR1:D collisionB1:D",
  "launch_rk_pert_collide1_srt(st, totalNodes, tauR, tauB, physicalVX, physicalVY, fluidRhoR, fluidRhoB, phiValue, fluidPDFR, fluidPDFB)"),
 ("rk", "calRKCollision23GPUNew", A + ":1169", "totalNodes:i xDim:i betaCoeff:d AkR:d AkB:d solidPhi:d fluidNodes:I neighboringNodes:I constantB:D weightsCoeff:D unitEX:D unitEY:D schemeGradient:D fluidRhoR:D fluidRhoB:D phiValue:D constantCR:D constantCB:D fluidPDFR:D fluidPDFB:D CGX:D CGY:D fluidPDFTotal:D",
  "launch_rk_pert_collide23(st, totalNodes, betaCoeff, AkR, AkB, solidPhi, neighboringNodes, constantB, weightsCoeff, fluidRhoR, fluidRhoB, fluidPDFR, fluidPDFB, fluidPDFTotal)"),
 ("rk", "calRKCollision1GPU2DMRTNew", A + ":1272", "totalNodes:i xDim:i delta:d tauR:d tauB:d bodyFX:d bodyFY:d unitEX:D unitEY:D constantCR:D constantCB:D weightsCoeff:D physicalVX:D physicalVY:D fluidRhoR:D fluidRhoB:D phiValue:D fluidPDFTotal:D transformationM:D inverseTM:D collisionS:D",
  "launch_rk_pert_collide1_mrt(st, totalNodes, tauR, tauB, bodyFX, bodyFY, physicalVX, physicalVY, fluidRhoR, fluidRhoB, phiValue, fluidPDFTotal, transformationM, inverseTM, collisionS)"),
 ("rk", "convectiveAverageBoundaryGPU", A + ":791", "totalNodes:i nx:i xDim:i fluidNodes:I neighboringNodes:I normalVelocity:D fluidPDFR:D fluidPDFB:D fluidPDFROld:D fluidPDFBOld:D",
  "launch_rk_outlet_average_row(st, totalNodes, nx, 2, fluidNodes, neighboringNodes, normalVelocity, fluidPDFR, fluidPDFB, fluidPDFROld, fluidPDFBOld)"),
 ("rk", "convectiveAverageBoundaryGPU2", A + ":820", "totalNodes:i nx:i xDim:i fluidNodes:I neighboringNodes:I normalVelocity:D fluidPDFR:D fluidPDFB:D fluidPDFROld:D fluidPDFBOld:D",
  "launch_rk_outlet_average_row(st, totalNodes, nx, 1, fluidNodes, neighboringNodes, normalVelocity, fluidPDFR, fluidPDFB, fluidPDFROld, fluidPDFBOld)"),
 ("rk", "convectiveAverageBoundaryGPU3", A + ":852", "totalNodes:i nx:i xDim:i fluidNodes:I neighboringNodes:I normalVelocity:D fluidPDFR:D fluidPDFB:D fluidPDFROld:D fluidPDFBOld:D",
  "launch_rk_outlet_average_row(st, totalNodes, nx, 0, fluidNodes, neighboringNodes, normalVelocity, fluidPDFR, fluidPDFB, fluidPDFROld, fluidPDFBOld)"),
 ("rk", "calConstPressureHighGPU", A + ":1087", "totalNodes:i nx:i ny:i xDim:i constPHB:d constPHR:d fluidNodes:I fluidPDFB:D fluidPDFR:D",
  "launch_rk_pressure_high(st, totalNodes, nx, ny, constPHB, constPHR, fluidNodes, fluidPDFB, fluidPDFR)"),
 ("rk", "constantVelocityZHBoundaryHigherNewRK", A + ":2307", "totalNodes:i nx:i ny:i xDim:i specificVYR:d specificVYB:d fluidNodes:I neighboringNodes:I fluidRhoR:D fluidRhoB:D fluidPDFR:D fluidPDFB:D",
  "launch_rk_inlet_velocity_red(st, totalNodes, nx, ny, specificVYR, fluidNodes, neighboringNodes, fluidRhoR, fluidPDFR, fluidPDFB, 0)"),
 # ---------------- AcceleratedRKGPU2D.py, kernels no loop launches any more (csrc/sparse_rest_rk.h)
 ("rk", "calRKCollision1GPU2DSRT", A + ":194", "totalNodes:i xDim:i delta:d tauR:d tauB:d unitEX:D unitEY:D constantCR:D constantCB:D weightsCoeff:D physicalVX:D physicalVY:D fluidRhoR:D fluidRhoB:D fluidPDFR:D fluidPDFB:D",
  "launch_rk_old_collide1_srt(st, totalNodes, delta, tauR, tauB, constantCR, constantCB, weightsCoeff, physicalVX, physicalVY, fluidRhoR, fluidRhoB, fluidPDFR, fluidPDFB)"),
 ("rk", "calRKCollision1GPU2DMRT", A + ":429", "totalNodes:i xDim:i delta:d tauR:d tauB:d unitEX:D unitEY:D constantCR:D constantCB:D weightsCoeff:D physicalVX:D physicalVY:D fluidRhoR:D fluidRhoB:D fluidPDFR:D fluidPDFB:D transformationM:D inverseTM:D collisionS:D",
  "launch_rk_old_collide1_mrt(st, totalNodes, delta, tauR, tauB, constantCR, constantCB, weightsCoeff, physicalVX, physicalVY, fluidRhoR, fluidRhoB, fluidPDFR, fluidPDFB, transformationM, inverseTM, collisionS)"),
 ("rk", "calRKCollision23GPU", A + ":511", "totalNodes:i xDim:i betaCoeff:d AkR:d AkB:d solidDiff:d fluidNodes:I neighboringNodes:I constantB:D weightsCoeff:D unitEX:D unitEY:D schemeGradient:D fluidRhoR:D fluidRhoB:D constantCR:D constantCB:D fluidPDFR:D fluidPDFB:D CGX:D CGY:D",
  "launch_rk_old_collide23(st, totalNodes, betaCoeff, AkR, AkB, solidDiff, neighboringNodes, constantB, weightsCoeff, schemeGradient, fluidRhoR, fluidRhoB, constantCR, constantCB, fluidPDFR, fluidPDFB, CGX, CGY)"),
 ("rk", "copyFluidPDFLastStep", A + ":887", "totalNodes:i nx:i xDim:i fluidNodes:I fluidPDFR:D fluidPDFB:D fluidPDFROld:D fluidPDFBOld:D",
  "launch_rk_copy_outlet_rows(st, totalNodes, nx, fluidNodes, fluidPDFR, fluidPDFB, fluidPDFROld, fluidPDFBOld)"),
 ("rk", "copyFluidPDFRecoverOutlet", A + ":906", "totalNodes:i nx:i xDim:i fluidNodes:I fluidPDFR:D fluidPDFB:D fluidPDFROld:D fluidPDFBOld:D",
  "launch_rk_copy_outlet_rows(st, totalNodes, nx, fluidNodes, fluidPDFROld, fluidPDFBOld, fluidPDFR, fluidPDFB)"),
 ("rk", "calNeumannPhiOutlet", A + ":1363", "totalNodes:i xDim:i nx:i fluidNodes:I neighboringNodes:I phiValue:D",
  "launch_rk_neumann_phi_outlet(st, totalNodes, nx, fluidNodes, neighboringNodes, phiValue)"),
 ("rk", "calModifiedPeriodicBoundary", A + ":1382", "totalNodes:i nx:i ny:i xDim:i fluidNodes:I neighboringNodes:I fluidPDFR:D fluidPDFB:D",
  "launch_rk_modified_periodic(st, totalNodes, nx, ny, fluidNodes, fluidPDFR, fluidPDFB)"),
 ("rk", "calRKCollision1TotalGPU2DSRT", A + ":1430", "totalNodes:i xDim:i tauR:d tauB:d unitEX:D unitEY:D constantCR:D constantCB:D weightsCoeff:D physicalVX:D physicalVY:D fluidRhoR:D fluidRhoB:D phiValue:D fluidPDFTotal:D collisionTotal1:D",
  "launch_rk_total_collide1_srt(st, totalNodes, tauR, tauB, weightsCoeff, physicalVX, physicalVY, fluidRhoR, fluidRhoB, phiValue, fluidPDFTotal, collisionTotal1)"),
 ("rk", "calRKCollision2TotalGPUNew", A + ":1468", "totalNodes:i xDim:i surfaceTA:d solidPhi:d fluidNodes:I neighboringNodes:I constantB:D weightsCoeff:D unitEX:D unitEY:D phiValue:D collisionTotal2:D gradientX:D gradientY:D",
  "launch_rk_total_collide2(st, totalNodes, surfaceTA, solidPhi, neighboringNodes, constantB, weightsCoeff, phiValue, collisionTotal2, gradientX, gradientY)"),
 ("rk", "calRecoloringProcess", A + ":1519", "totalNodes:i xDim:i betaValue:d weightsCoeff:D fluidRhoR:D fluidRhoB:D unitEX:D unitEY:D gradientX:D gradientY:D collisionTotal1:D collisionTotal2:D fluidPDFR:D fluidPDFB:D fluidPDFTotal:D",
  "launch_rk_recolor_add(st, totalNodes, betaValue, weightsCoeff, fluidRhoR, fluidRhoB, gradientX, gradientY, collisionTotal1, collisionTotal2, fluidPDFR, fluidPDFB)"),
 ("rk", "calPhysicalVelocityRKGPU2DVNew", A + ":1907", "totalNodes:i nx:i ny:i xDim:i fluidNodes:I fluidPDFTotal:D fluidRhoR:D fluidRhoB:D physicalVX:D physicalVY:D forceX:D forceY:D",
  "launch_rk_velocity_below_inlet(st, totalNodes, nx, ny, fluidNodes, fluidPDFTotal, fluidRhoR, fluidRhoB, physicalVX, physicalVY, forceX, forceY)"),
 ("rk", "calMacroDensityRKGPU2DNew", A + ":2610", "totalNodes:i nx:i ny:i xDim:i fluidNodes:I fluidPDFR:D fluidPDFB:D fluidRhoR:D fluidRhoB:D",
  "launch_rk_density_below_inlet(st, totalNodes, nx, ny, fluidNodes, fluidPDFR, fluidPDFB, fluidRhoR, fluidRhoB)"),
 # ---------------- RKGPU2DBoundary.py: the same-named kernels of AcceleratedRKGPU2D.py, four of them with other semantics
 ("rkb", "constantVelocityZHBoundaryHigherRK", B + ":11", "totalNodes:i nx:i ny:i xDim:i specificVYR:d specificVYB:d fluidNodes:I fluidRhoR:D fluidRhoB:D fluidPDFR:D fluidPDFB:D",
  "launch_rk_pert_inlet_velocity(st, totalNodes, nx, ny, specificVYR, specificVYB, fluidNodes, fluidRhoR, fluidRhoB, fluidPDFR, fluidPDFB)"),
 ("rkb", "ghostPointsConstantVelocityRK", B + ":58", "totalNodes:i nx:i ny:i xDim:i fluidNodes:I neighboringNodes:I fluidRhoR:D fluidRhoB:D fluidPDFR:D fluidPDFB:D",
  "launch_rk_ghost_inlet_velocity(st, totalNodes, nx, ny, fluidNodes, neighboringNodes, fluidRhoR, fluidRhoB, fluidPDFR, fluidPDFB)"),
 ("rkb", "convectiveOutletGPU", B + ":112", "totalNodes:i nx:i xDim:i fluidNodes:I neighboringNodes:I fluidPDFR:D fluidPDFB:D fluidRhoR:D fluidRhoB:D",
  "launch_rk_outlet_convective_row(st, totalNodes, nx, 2, fluidNodes, neighboringNodes, fluidPDFR, fluidPDFB, fluidRhoR, fluidRhoB)"),
 ("rkb", "convectiveOutletGhost2GPU", B + ":149", "totalNodes:i nx:i xDim:i fluidNodes:I neighboringNodes:I fluidPDFR:D fluidPDFB:D fluidRhoR:D fluidRhoB:D",
  "launch_rk_outlet_convective_row(st, totalNodes, nx, 1, fluidNodes, neighboringNodes, fluidPDFR, fluidPDFB, fluidRhoR, fluidRhoB)"),
 ("rkb", "convectiveOutletGhost3GPU", B + ":187", "totalNodes:i nx:i xDim:i fluidNodes:I neighboringNodes:I fluidPDFR:D fluidPDFB:D fluidRhoR:D fluidRhoB:D",
  "launch_rk_outlet_convective_row(st, totalNodes, nx, 0, fluidNodes, neighboringNodes, fluidPDFR, fluidPDFB, fluidRhoR, fluidRhoB)"),
 ("rkb", "convectiveAverageBoundaryGPU", B + ":222", "totalNodes:i nx:i xDim:i fluidNodes:I neighboringNodes:I normalVelocity:D fluidPDFR:D fluidPDFB:D fluidPDFROld:D fluidPDFBOld:D",
  "launch_rk_outlet_average_row(st, totalNodes, nx, 2, fluidNodes, neighboringNodes, normalVelocity, fluidPDFR, fluidPDFB, fluidPDFROld, fluidPDFBOld)"),
 ("rkb", "convectiveAverageBoundaryGPU2", B + ":254", "totalNodes:i nx:i xDim:i fluidNodes:I neighboringNodes:I normalVelocity:D fluidPDFR:D fluidPDFB:D fluidPDFROld:D fluidPDFBOld:D",
  "launch_rk_outlet_average_row(st, totalNodes, nx, 1, fluidNodes, neighboringNodes, normalVelocity, fluidPDFR, fluidPDFB, fluidPDFROld, fluidPDFBOld)"),
 ("rkb", "convectiveAverageBoundaryGPU3", B + ":289", "totalNodes:i nx:i xDim:i fluidNodes:I neighboringNodes:I normalVelocity:D fluidPDFR:D fluidPDFB:D fluidPDFROld:D fluidPDFBOld:D",
  "launch_rk_outlet_average_row(st, totalNodes, nx, 0, fluidNodes, neighboringNodes, normalVelocity, fluidPDFR, fluidPDFB, fluidPDFROld, fluidPDFBOld)"),
 ("rkb", "calConstPressureInletGPU", B + ":325", "totalNodes:i nx:i ny:i xDim:i constPHB:d constPHR:d fluidNodes:I fluidRhoB:D fluidRhoR:D fluidPDFB:D fluidPDFR:D",
  "launch_rk_inlet_pressure(st, totalNodes, nx, ny, constPHB, constPHR, fluidNodes, fluidRhoB, fluidRhoR, fluidPDFB, fluidPDFR)"),
 ("rkb", "ghostPointsConstPressureInletRK", B + ":371", "totalNodes:i nx:i ny:i xDim:i fluidNodes:I neighboringNodes:I fluidRhoR:D fluidRhoB:D fluidPDFR:D fluidPDFB:D",
  "launch_rk_ghost_inlet_pressure(st, totalNodes, nx, ny, fluidNodes, neighboringNodes, fluidRhoR, fluidRhoB, fluidPDFR, fluidPDFB)"),
 ("rkb", "calConstPressureLowerGPU", B + ":414", "totalNodes:i nx:i xDim:i constPLB:d constPLR:d fluidNodes:I fluidRhoB:D fluidRhoR:D fluidPDFB:D fluidPDFR:D",
  "launch_rk_pert_outlet_pressure(st, totalNodes, nx, constPLB, constPLR, fluidNodes, fluidRhoB, fluidRhoR, fluidPDFB, fluidPDFR, 1)"),
 ("rkb", "ghostPointsConstPressureLowerRK", B + ":452", "totalNodes:i nx:i xDim:i fluidNodes:I neighboringNodes:I fluidRhoR:D fluidRhoB:D fluidPDFR:D fluidPDFB:D",
  "launch_rk_ghost_outlet_pressure_grid(st, totalNodes, nx, fluidNodes, neighboringNodes, fluidRhoR, fluidRhoB, fluidPDFR, fluidPDFB)"),
 ("rkb", "calConstPressureHighGPU", B + ":496", "totalNodes:i nx:i ny:i xDim:i constPHB:d constPHR:d fluidNodes:I fluidPDFB:D fluidPDFR:D",
  "launch_rk_pressure_high(st, totalNodes, nx, ny, constPHB, constPHR, fluidNodes, fluidPDFB, fluidPDFR)"),
 ("rkb", "constantVelocityZHBoundaryHigherNewRK", B + ":535", "totalNodes:i nx:i ny:i xDim:i specificVYR:d specificVYB:d fluidNodes:I neighboringNodes:I fluidRhoR:D fluidRhoB:D fluidPDFR:D fluidPDFB:D",
  "launch_rk_inlet_velocity_red(st, totalNodes, nx, ny, specificVYR, fluidNodes, neighboringNodes, fluidRhoR, fluidPDFR, fluidPDFB, 1)"),
 ("rkb", "calConstPressureLowerGPUTotal", B + ":581", "totalNodes:i nx:i xDim:i constPL:d fluidNodes:I fluidPDFTotal:D physicalVY:D fluidRhoR:D fluidRhoB:D fluidPDFR:D fluidPDFB:D",
  "launch_rk_outlet_pressure_total(st, totalNodes, nx, constPL, fluidNodes, fluidPDFTotal, physicalVY, fluidRhoR, fluidRhoB, fluidPDFR, fluidPDFB)"),
 ("rkb", "constantTotalVelocityInlet", B + ":623", "totalNodes:i nx:i ny:i xDim:i specificVY:d fluidNodes:I neighboringNodes:I fluidRhoR:D fluidRhoB:D fluidPDFR:D fluidPDFB:D fluidPDFTotal:D physicalVY:D",
  "launch_rk_inlet_velocity_total(st, totalNodes, nx, ny, specificVY, fluidNodes, fluidRhoR, fluidRhoB, fluidPDFR, fluidPDFB, fluidPDFTotal, physicalVY)"),
 # ---------------- Shan-Chen / EFS (two fluids)
 ("sc", "fillNeighboringNodes", O + ":22", "totalNodes:i nx:i ny:i xDim:i fluidNodes:I domainNewIndex:I neighboringNodes:I",
  "launch_rk_fill_neighbors(st, totalNodes, nx, ny, fluidNodes, domainNewIndex, neighboringNodes)"),
 ("sc", "savePDFLastStep", O + ":70", "totalNodes:i numFluids:i xDim:i fluidPDF:D fluidPDFOld:D",
  "sc_check_nf(numFluids); LBMPM_HIP_TRY(hipMemcpyAsync(fluidPDFOld, fluidPDF, sizeof(double) * 2 * 9 * (size_t)totalNodes, hipMemcpyDeviceToDevice, st))"),
 ("sc", "calMacroWholeVelocity", O + ":336", "totalNodes:i numFluids:i xDim:i tau:D fluidRho:D fluidPDF:D primeVX:D primeVY:D",
  "sc_check_nf(numFluids); launch_sc_macro_whole_velocity(st, totalNodes, tau, fluidRho, fluidPDF, primeVX, primeVY)"),
 ("sc", "calFluidRhoGPU", O + ":84", "totalNodes:i numFluids:i xDim:i fluidRho:D fluidPDF:D",
  "sc_check_nf(numFluids); launch_sc_rho(st, totalNodes, fluidRho, fluidPDF)"),
 ("sc", "calFluidPotentialGPUEql", O + ":99", "totalNodes:i numFluids:i xDim:i fluidRho:D fluidPotential:D",
  "sc_check_nf(numFluids); LBMPM_HIP_TRY(hipMemcpyAsync(fluidPotential, fluidRho, sizeof(double) * 2 * (size_t)totalNodes, hipMemcpyDeviceToDevice, st))"),
 ("sc", "calPhysicalVelocity", O + ":156", "totalNodes:i numFluids:i xDim:i fluidPDF:D fluidRho:D forceX:D forceY:D velocityPX:D velocityPY:D",
  "sc_check_nf(numFluids); launch_sc_physical_velocity(st, totalNodes, fluidPDF, fluidRho, forceX, forceY, velocityPX, velocityPY)"),
 ("sc", "calStreaming1GPU", O + ":452", "totalNum:i numFluids:i xDim:i fluidNodes:I neighboringNodes:I fluidPDF:D fluidPDFNew:D",
  "sc_check_nf(numFluids); launch_sc_stream1(st, totalNum, neighboringNodes, fluidPDF, fluidPDFNew)"),
 ("sc", "calStreaming2GPU", O + ":539", "totalNum:i numFluids:i xDim:i fluidPDFNew:D fluidPDF:D",
  "sc_check_nf(numFluids); launch_sc_stream2(st, totalNum, fluidPDFNew, fluidPDF)"),
 ("sc", "constantPressureZouHeBoundaryLower", O + ":555", "totalNodes:i numFluids:i nx:i xDim:i densityL:d fluidNodes:I fluidRho:D fluidPDF:D",
  "sc_check_nf(numFluids); launch_sc_outlet_pressure_row(st, totalNodes, nx, 1, fluidNodes, fluidRho, fluidPDF)"),
 # scheme 8: the same rules one row further inside, two ghost rows
 ("sc", "constantPressureZouHeBoundaryLower8", O + ":590", "totalNodes:i numFluids:i nx:i xDim:i densityL:d fluidNodes:I fluidRho:D fluidPDF:D",
  "sc_check_nf(numFluids); launch_sc_outlet_pressure_row(st, totalNodes, nx, 2, fluidNodes, fluidRho, fluidPDF)"),
 ("sc", "ghostPointsConstantPressureOutlet8", O + ":775", "totalNodes:i numFluids:i nx:i xDim:i fluidNodes:I neighboringNodes:I fluidRho:D fluidPDF:D",
  "sc_check_nf(numFluids); launch_sc_ghost_outlet_row(st, totalNodes, nx, 1, fluidNodes, neighboringNodes, fluidRho, fluidPDF)"),
 ("sc", "ghostPointsConstantPressureOutlet82", O + ":807", "totalNodes:i numFluids:i nx:i xDim:i fluidNodes:I neighboringNodes:I fluidRho:D fluidPDF:D",
  "sc_check_nf(numFluids); launch_sc_ghost_outlet_row(st, totalNodes, nx, 0, fluidNodes, neighboringNodes, fluidRho, fluidPDF)"),
 ("sc", "constantVelocityZouHeBoundaryHigher8", O + ":868", "totalNodes:i numFluids:i nx:i ny:i xDim:i specificVY:D fluidNodes:I fluidRho:D fluidPDF:D",
  "sc_check_nf(numFluids); launch_sc_inlet_velocity_row(st, totalNodes, nx, ny - 3, specificVY, fluidNodes, fluidRho, fluidPDF)"),
 ("sc", "ghostPointsConstantVelocity8", O + ":897", "totalNodes:i numFluids:i nx:i ny:i xDim:i fluidNodes:I neighboringNodes:I fluidRho:D fluidPDF:D",
  "sc_check_nf(numFluids); launch_sc_ghost_inlet_row(st, totalNodes, nx, ny - 2, fluidNodes, neighboringNodes, fluidRho, fluidPDF)"),
 ("sc", "ghostPointsConstantVelocity82", O + ":927", "totalNodes:i numFluids:i nx:i ny:i xDim:i fluidNodes:I neighboringNodes:I fluidRho:D fluidPDF:D",
  "sc_check_nf(numFluids); launch_sc_ghost_inlet_row(st, totalNodes, nx, ny - 1, fluidNodes, neighboringNodes, fluidRho, fluidPDF)"),
 ("sc", "ghostPointsConstantVelocityInlet", O + ":710", "totalNodes:i numFluids:i nx:i ny:i xDim:i fluidNodes:I neighboringNodes:I fluidRho:D fluidPDF:D",
  "sc_check_nf(numFluids); launch_sc_ghost_inlet_row(st, totalNodes, nx, ny - 1, fluidNodes, neighboringNodes, fluidRho, fluidPDF)"),
 ("sc", "ghostPointsConstantPressureOutlet", O + ":743", "totalNodes:i numFluids:i nx:i xDim:i fluidNodes:I neighboringNodes:I fluidRho:D fluidPDF:D",
  "sc_check_nf(numFluids); launch_sc_ghost_outlet_row(st, totalNodes, nx, 0, fluidNodes, neighboringNodes, fluidRho, fluidPDF)"),
 ("sc", "constantVelocityZouHeBoundaryHigher", O + ":839", "totalNodes:i numFluids:i nx:i ny:i xDim:i specificVY:D fluidNodes:I fluidRho:D fluidPDF:D",
  "sc_check_nf(numFluids); launch_sc_inlet_velocity_row(st, totalNodes, nx, ny - 2, specificVY, fluidNodes, fluidRho, fluidPDF)"),
 ("sc", "convectiveOutletGPU", O + ":960", "totalNodes:i numFluids:i nx:i xDim:i fluidNodes:I neighboringNodes:I fluidPDFNew:D fluidRho:D",
  "sc_check_nf(numFluids); launch_sc_outlet_copy_row(st, totalNodes, nx, 2, fluidNodes, neighboringNodes, fluidPDFNew, fluidRho)"),
 ("sc", "convectiveOutletGhost2GPU", O + ":988", "totalNodes:i numFluids:i nx:i xDim:i fluidNodes:I neighboringNodes:I fluidPDFNew:D fluidRho:D",
  "sc_check_nf(numFluids); launch_sc_outlet_copy_row(st, totalNodes, nx, 1, fluidNodes, neighboringNodes, fluidPDFNew, fluidRho)"),
 ("sc", "convectiveOutletGhost3GPU", O + ":1016", "totalNodes:i numFluids:i nx:i xDim:i fluidNodes:I neighboringNodes:I fluidPDFNew:D fluidRho:D",
  "sc_check_nf(numFluids); launch_sc_outlet_copy_row(st, totalNodes, nx, 0, fluidNodes, neighboringNodes, fluidPDFNew, fluidRho)"),
 ("sc", "convectiveOutletEachGPU", O + ":1044", "totalNodes:i numFluids:i nx:i xDim:i fluidNodes:I neighboringNodes:I fluidPDFNew:D fluidPDFOld:D fluidRho:D physicalVY:D",
  "sc_check_nf(numFluids); launch_sc_outlet_convective_row(st, totalNodes, nx, 2, fluidNodes, neighboringNodes, fluidPDFNew, fluidPDFOld, fluidRho, physicalVY)"),
 ("sc", "convectiveOutletEach2GPU", O + ":1070", "totalNodes:i numFluids:i nx:i xDim:i fluidNodes:I neighboringNodes:I fluidPDFNew:D fluidPDFOld:D fluidRho:D physicalVY:D",
  "sc_check_nf(numFluids); launch_sc_outlet_convective_row(st, totalNodes, nx, 1, fluidNodes, neighboringNodes, fluidPDFNew, fluidPDFOld, fluidRho, physicalVY)"),
 ("sc", "convectiveOutletEach3GPU", O + ":1098", "totalNodes:i numFluids:i nx:i xDim:i fluidNodes:I neighboringNodes:I fluidPDFNew:D fluidPDFOld:D fluidRho:D physicalVY:D",
  "sc_check_nf(numFluids); launch_sc_outlet_convective_row(st, totalNodes, nx, 0, fluidNodes, neighboringNodes, fluidPDFNew, fluidPDFOld, fluidRho, physicalVY)"),
 ("sc", "interactionCollisionProcess", O + ":1274", "totalNodes:i numFluids:i xDim:i weightInter:D tau:D interCoeff:D interSolid:D weightsCoeff:D fluidRho:D fluidPotential:D fluidPDF:D fluidPDFNew:D fluidNodes:I neighboringNodes:I forceX:D forceY:D",
  "sc_check_nf(numFluids); launch_sc_interaction_collision(st, totalNodes, tau, interCoeff, interSolid, fluidRho, fluidPotential, fluidPDF, neighboringNodes, forceX, forceY)"),
 ("sc", "calExplicit4thOrderScheme", E + ":51", "totalNodes:i numFluids:i xDim:i fluidNodes:I neighboringNodes:I weightInter:D interactionCoeff:D interactionSolid:D fluidPotential:D forceX:D forceY:D",
  "sc_check_nf(numFluids); launch_sc_efs_force4(st, totalNodes, neighboringNodes, interactionCoeff, interactionSolid, fluidPotential, forceX, forceY)"),
 ("sc", "fillNeighboringNodesISO8", E + ":392", "totalNodes:i nx:i ny:i xDim:i fluidNodes:I domainNewIndex:I neighboringNodes:I",
  "launch_sc_fill_neighbors_iso(st, totalNodes, nx, ny, 24, fluidNodes, domainNewIndex, neighboringNodes)"),
 ("sc", "fillNeighboringNodesISO10", E + ":488", "totalNodes:i nx:i ny:i xDim:i fluidNodes:I domainNewIndex:I neighboringNodes:I",
  "launch_sc_fill_neighbors_iso(st, totalNodes, nx, ny, 36, fluidNodes, domainNewIndex, neighboringNodes)"),
 ("sc", "calExplicit8thOrderScheme", E + ":627", "totalNodes:i numFluids:i xDim:i fluidNodes:I neighboringNodes:I weightInter:D interactionCoeff:D interactionSolid:D fluidPotential:D forceX:D forceY:D",
  "sc_check_nf(numFluids); launch_sc_efs_force_iso(st, totalNodes, 24, neighboringNodes, weightInter, interactionCoeff, interactionSolid, fluidPotential, forceX, forceY)"),
 ("sc", "calExplicit10thOrderScheme", E + ":957", "totalNodes:i numFluids:i xDim:i fluidNodes:I neighboringNodes:I weightInter:D interactionCoeff:D interactionSolid:D fluidPotential:D forceX:D forceY:D",
  "sc_check_nf(numFluids); launch_sc_efs_force_iso(st, totalNodes, 36, neighboringNodes, weightInter, interactionCoeff, interactionSolid, fluidPotential, forceX, forceY)"),
 ("sc", "calEquilibriumFuncEFGPU", E + ":227", "totalNum:i numFluids:i xDim:i weightCoeff:D EX:D EY:D fluidRho:D equilibriumVX:D equilibriumVY:D fEq:D",
  "sc_check_nf(numFluids); launch_sc_efs_feq(st, totalNum, fluidRho, equilibriumVX, equilibriumVY, fEq)"),
 ("sc", "calForceDistrGPU", E + ":255", "totalNodes:i numFluids:i xDim:i EX:D EY:D equilibriumVX:D equilibriumVY:D fluidRho:D forceX:D forceY:D fEq:D fForce:D",
  "sc_check_nf(numFluids); launch_sc_efs_fforce(st, totalNodes, equilibriumVX, equilibriumVY, fluidRho, forceX, forceY, fEq, fForce)"),
 ("sc", "transformPDFGPU", E + ":278", "totalNodes:i numFluids:i xDim:i fluidPDF:D fForce:D",
  "sc_check_nf(numFluids); launch_sc_efs_transform(st, totalNodes, fluidPDF, fForce)"),
 ("sc", "calCollisionEXGPU", E + ":294", "totalNodes:i numFluids:i xDim:i tau:D fluidPDF:D fEq:D fForce:D",
  "sc_check_nf(numFluids); launch_sc_efs_collide_srt(st, totalNodes, tau, fluidPDF, fEq, fForce)"),
 ("sc", "calEquilibriumVEFGPU", E + ":340", "totalNodes:i numFluids:i xDim:i tau:D EX:D EY:D fluidRho:D forceX:D forceY:D fluidPDF:D eqVX:D eqVY:D",
  "sc_check_nf(numFluids); launch_sc_efs_ueq(st, totalNodes, tau, 1, fluidRho, forceX, forceY, fluidPDF, eqVX, eqVY)"),
 ("sc", "transformEquilibriumVelocity", E + ":1426", "totalNodes:i numFluids:i xDim:i EX:D EY:D fluidRho:D forceX:D forceY:D fluidPDF:D conserveS:D eqVX:D eqVY:D",
  "sc_check_nf(numFluids); launch_sc_efs_ueq(st, totalNodes, conserveS, 0, fluidRho, forceX, forceY, fluidPDF, eqVX, eqVY)"),
 ("sc", "transformPDFandEquil", E + ":1379", "totalNodes:i numFluids:i xDim:i fluidPDF:D fEq:D collisionMatrix:D fluidPDFM:D",
  "sc_check_nf(numFluids); launch_sc_mrt_transform_pdf_eq(st, totalNodes, fluidPDF, fEq, collisionMatrix, fluidPDFM)"),
 ("sc", "transfromForceTerm", E + ":1404", "totalNodes:i numFluids:i xDim:i fForce:D collisionMatrix:D fForceM:D",
  "sc_check_nf(numFluids); launch_sc_mrt_transform_force(st, totalNodes, fForce, collisionMatrix, fForceM)"),
 ("sc", "calAfterCollisionMRT", E + ":1457", "totalNodes:i numFluids:i xDim:i fluidPDF:D fForce:D fEq:D fluidPDFM:D fForceM:D",
  "sc_check_nf(numFluids); launch_sc_mrt_after_collision(st, totalNodes, fluidPDF, fForce, fEq, fluidPDFM, fForceM)"),
 # ---------------- tracer transport
 # ---------------- OptimizedD2Q9GPU.py / ExplicitD2Q9GPU.py, kernels no loop launches (csrc/sparse_rest_sc.h)
 ("sc", "calFluidPotentialGPUPR", O + ":112", "totalNodes:i numFluids:i xDim:i constR:d temperatureT:d coeffA:d coeffB:d coeffAlpha:d constC0:d constG:d fluidRho:D fluidPotential:D",
  "sc_check_nf(numFluids); launch_sc_potential_pr(st, totalNodes, constR, temperatureT, coeffA, coeffB, coeffAlpha, constC0, constG, fluidRho, fluidPotential)"),
 ("sc", "calMacroPressure", O + ":135", "totalNodes:i numFluids:i xDim:i interactionCoeff:D fluidRho:D fluidPressure:D",
  "sc_check_nf(numFluids); launch_sc_pressure(st, totalNodes, 0, interactionCoeff, fluidRho, fluidRho, fluidPressure)"),
 ("sc", "calInteractionForce", O + ":186", "totaNodes:i numFluids:i nx:i ny:i xDim:i fluidNodes:I neighboringNodes:I weightInter:D interactionCoeff:D interactionSolid:D fluidPotential:D forceX:D forceY:D",
  "sc_check_nf(numFluids); launch_sc_product_force(st, totaNodes, neighboringNodes, weightInter, interactionCoeff, interactionSolid, fluidPotential, forceX, forceY)"),
 ("sc", "addBodyForceGPU", O + ":320", "totalNum:i numFluids:i xDim:i bodyFX:d bodyFY:d forceX:D forceY:D fluidRho:D",
  "sc_check_nf(numFluids); launch_sc_add_body_force(st, totalNum, bodyFX, bodyFY, forceX, forceY, fluidRho)"),
 ("sc", "calEquilibriumVGPU", O + ":361", "totalNum:i numFluids:i xDim:i tau:D fluidRho:D forceX:D forceY:D mixtureVX:D mixtureVY:D equilibriumVX:D equilibriumVY:D",
  "sc_check_nf(numFluids); launch_sc_equilibrium_velocity(st, totalNum, tau, fluidRho, forceX, forceY, mixtureVX, mixtureVY, equilibriumVX, equilibriumVY)"),
 ("sc", "calEquilibriumFuncGPU", O + ":379", "totalNum:i numFluids:i xDim:i weightCoeff:D fluidRho:D equilibriumVX:D equilibriumVY:D fEq:D",
  "sc_check_nf(numFluids); launch_sc_equilibrium(st, totalNum, weightCoeff, fluidRho, equilibriumVX, equilibriumVY, fEq)"),
 ("sc", "calCollisionSRTGPU", O + ":435", "totalNum:i numFluids:i xDim:i tau:D fluidPDF:D fEq:D",
  "sc_check_nf(numFluids); launch_sc_collide_srt(st, totalNum, tau, fluidPDF, fEq)"),
 ("sc", "constantPressureZouHeBoundaryHigher", O + ":625", "totalNodes:i numFluids:i nx:i ny:i xDim:i densityH:d fluidNodes:I fluidRho:D fluidPDF:D",
  "sc_check_nf(numFluids); launch_sc_inlet_pressure_row(st, totalNodes, nx, ny, densityH, fluidNodes, fluidRho, fluidPDF)"),
 ("sc", "ghostPointsConstantPressureInlet", O + ":659", "totalNodes:i numFluids:i nx:i ny:i xDim:i fluidNodes:I neighboringNodes:I fluidRho:D fluidPDF:D",
  "sc_check_nf(numFluids); launch_sc_ghost_pressure_inlet(st, totalNodes, nx, ny, fluidNodes, neighboringNodes, fluidRho, fluidPDF)"),
 ("sc", "calVelocityBoundaryHigherChangGPU", O + ":1127", "totalNodes:i numFluids:i nx:i ny:i xDim:i specificVY:D fluidNodes:I fluidRho:D forceX:D forceY:D fluidPDFOld:D fluidPDFNew:D",
  "sc_check_nf(numFluids); launch_sc_chang_velocity_high(st, totalNodes, nx, ny, specificVY, fluidNodes, fluidRho, fluidPDFOld, fluidPDFNew)"),
 ("sc", "calPressureBoundaryHigherChangGPU", O + ":1172", "totalNodes:i numFluids:i nx:i ny:i xDim:i specificRhoH:d fluidNodes:I fluidRho:D forceX:D forceY:D fluidPDFOld:D fluidPDFNew:D",
  "sc_check_nf(numFluids); launch_sc_chang_pressure(st, totalNodes, nx, ny - 2, 1, specificRhoH, fluidNodes, fluidRho, forceX, forceY, fluidPDFOld, fluidPDFNew)"),
 ("sc", "calPressureBoundaryLowerChangGPU", O + ":1222", "totalNodes:i numFluids:i nx:i ny:i xDim:i specificRhoL:d fluidNodes:I fluidRho:D forceX:D forceY:D fluidPDFOld:D fluidPDFNew:D",
  "sc_check_nf(numFluids); launch_sc_chang_pressure(st, totalNodes, nx, 1, 0, specificRhoL, fluidNodes, fluidRho, forceX, forceY, fluidPDFOld, fluidPDFNew)"),
 ("sc", "interactionCollisionEOFProcess", O + ":1454", "totalNodes:i numFluids:i xDim:i weightInter:D tauReverse:D interCoeff:D interSolid:D weigthCoeff:D fluidRho:D fluidPotential:D fluidPDF:D fluidPDFNew:D fluidNodes:I neighboringNodes:I forceX:D forceY:D",
  "sc_check_nf(numFluids); launch_sc_eof_collision(st, totalNodes, weightInter, tauReverse, interCoeff, interSolid, fluidRho, fluidPotential, fluidPDF, neighboringNodes, forceX, forceY)"),
 ("sc", "calStreaming1withLinkGPU", O + ":1674", "totalNum:i numFluids:i xDim:i fluidNodes:I neighboringNodes:I fluidRho:D fluidPDF:D fluidPDFNew:D physicalVX:D physicalVY:D weightCoeff:D",
  "sc_check_nf(numFluids); launch_sc_stream1_link(st, totalNum, neighboringNodes, fluidRho, fluidPDF, fluidPDFNew, physicalVX, physicalVY, weightCoeff)"),
 ("sc", "interactionForceGuo", O + ":1804", "totalNodes:i numFluids:i xDim:i weightInter:D interCoeff:D interSolid:D weightsCoeff:D fluidPotential:D fluidNodes:I neighboringNodes:I forceX:D forceY:D",
  "sc_check_nf(numFluids); launch_sc_product_force(st, totalNodes, neighboringNodes, weightInter, interCoeff, interSolid, fluidPotential, forceX, forceY)"),
 ("sc", "calCollisionGuo", O + ":1917", "totalNodes:i numFluids:i xDim:i tau:D weightsCoeff:D unitEX:D unitEY:D fluidRho:D forceX:D forceY:D physicalVX:D physicalVY:D fluidPDF:D",
  "sc_check_nf(numFluids); launch_sc_collide_guo(st, totalNodes, tau, weightsCoeff, fluidRho, forceX, forceY, physicalVX, physicalVY, fluidPDF)"),
 ("sc", "calMacroPressureEX", E + ":19", "totalNodes:i numFluids:i xDim:i interactionCoeff:D fluidRho:D fluidPotential:D fluidPressure:D",
  "sc_check_nf(numFluids); launch_sc_pressure(st, totalNodes, 2, interactionCoeff, fluidRho, fluidPotential, fluidPressure)"),
 ("sc", "calEffectiveMassPR", E + ":38", "totalNodes:i numFluids:i xDim:i temperature:d interactionCoeff:D fluidRho:D fluidPsi:D",
  "sc_check_nf(numFluids); launch_sc_effective_mass_pr(st, totalNodes, temperature, interactionCoeff, fluidRho, fluidPsi)"),
 ("sc", "calTotalVelocityGPU", E + ":311", "totalNodes:i numFluids:i xDim:i EX:D EY:D forceX:D forceY:D fluidPDF:D totalVX:D totalVY:D",
  "sc_check_nf(numFluids); launch_sc_total_velocity(st, totalNodes, forceX, forceY, fluidPDF, totalVX, totalVY)"),
 ("sc", "calPressureExpGPU", E + ":371", "totalNodes:i numFluids:i xDim:i interCoeff:D fluidRho:D fluidPotential:D fluidPressure:D",
  "sc_check_nf(numFluids); launch_sc_pressure(st, totalNodes, 1, interCoeff, fluidRho, fluidPotential, fluidPressure)"),
 ("sc", "convectiveOutletGPUEFS", E + ":1476", "totalNodes:i numFluids:i nx:i xDim:i fluidNodes:I neighboringNodes:I fluidPDFNew:D fluidRho:D fForce:D fEq:D",
  "sc_check_nf(numFluids); launch_sc_freeflow_row(st, totalNodes, nx, 2, fluidNodes, neighboringNodes, fluidPDFNew, fluidRho, fForce, fEq)"),
 ("sc", "convectiveOutletGhost2GPUEFS", E + ":1506", "totalNodes:i numFluids:i nx:i xDim:i fluidNodes:I neighboringNodes:I fluidPDFNew:D fluidRho:D fForce:D fEq:D",
  "sc_check_nf(numFluids); launch_sc_freeflow_row(st, totalNodes, nx, 1, fluidNodes, neighboringNodes, fluidPDFNew, fluidRho, fForce, fEq)"),
 ("sc", "convectiveOutletGhost3GPUEFS", E + ":1537", "totalNodes:i numFluids:i nx:i xDim:i fluidNodes:I neighboringNodes:I fluidPDFNew:D fluidRho:D fForce:D fEq:D",
  "sc_check_nf(numFluids); launch_sc_freeflow_row(st, totalNodes, nx, 0, fluidNodes, neighboringNodes, fluidPDFNew, fluidRho, fForce, fEq)"),
 ("tr", "fillNeighboringNodesTransport", T + ":51", "totalNodes:i nx:i ny:i xDim:i fluidNodes:I domainNewIndex:I neighboringNodes:I",
  "launch_tr_fill_neighbors(st, totalNodes, nx, ny, fluidNodes, domainNewIndex, neighboringNodes)"),
 ("tr", "calReactionTracersGPU", T + ":95", "totalNodes:i numTracers:i xDim:i reactionRate:D diffJcoeffs:D tracerConc:D tracerPDF:D",
  "launch_tr_reaction(st, totalNodes, (int)numTracers, reactionRate, diffJcoeffs, tracerConc, tracerPDF)"),
 ("tr", "calConcentrationGPU", T + ":78", "totalNodes:i numTracers:i xDim:i numSchemes:i tracerConc:D tracerPDF:D",
  "tr_check_q5(numSchemes); launch_tr_concentration(st, totalNodes, (int)numTracers, tracerConc, tracerPDF)"),
 ("tr", "calCollisionTransportLinearEqlMRTGPU", T + ":535", "totalNodes:i xDim:i numTracers:i unitVX:D unitVY:D velocityVX:D velocityVY:D tracerConc:D tracerPDF:D transportM:D inverseRelaxationMS:D weightsCoeff:D",
  "launch_tr_collide_mrt(st, totalNodes, (int)numTracers, velocityVX, velocityVY, tracerConc, tracerPDF, transportM, inverseRelaxationMS)"),
 ("tr", "calStreamingTransportGPU", T + ":139", "totalNodes:i xDim:i numTracers:i neighboringNodes:I tracerPDF:D tracerPDFNew:D",
  "launch_tr_stream1(st, totalNodes, (int)numTracers, neighboringNodes, tracerPDF, tracerPDFNew)"),
 ("tr", "calStreamingTransport2GPU", T + ":184", "totalNum:i numTracers:i xDim:i tracerPDFNew:D tracerPDF:D",
  "launch_tr_stream2(st, totalNum, (int)numTracers, tracerPDFNew, tracerPDF)"),
 ("tr", "calFreeConcBoundary3", T + ":461", "totalNodes:i numTracers:i nx:i xDim:i fluidNodes:I neighboringNodes:I tracerConc:D tracerPDF:D",
  "launch_tr_free_outlet(st, totalNodes, (int)numTracers, nx, fluidNodes, neighboringNodes, tracerPDF)"),
 ("tr", "calInamuroConstConcBoundary", T + ":682", "totalNodes:i xDim:i numTracers:i ny:i nx:i fluidNodes:I neighboringTRNodes:I concBoundary:D weightsCoeff:D tracerPDF:D",
  "launch_tr_inlet_inamuro(st, totalNodes, (int)numTracers, ny, nx, fluidNodes, concBoundary, tracerPDF)"),
 ("tr", "calValueTransportDomain", T + ":957", "totalNodes:i xDim:i critiriaValue:d valueTransportDomain:D fluidRhoR:D",
  "launch_tr_indicator(st, totalNodes, critiriaValue, valueTransportDomain, fluidRhoR)"),
 ("tr", "calTransportWithInterfaceD2Q5", T + ":976", "totalNodes:i xDim:i numTracers:i betaTracer:D valueTransportDomain:D unitEX:D unitEY:D gradientX:D gradientY:D weightsCoeff:D tracerConc:D tracerPDF:D",
  "launch_tr_interface(st, totalNodes, (int)numTracers, betaTracer, valueTransportDomain, gradientX, gradientY, tracerConc, tracerPDF)"),
 # ---------------- AccelerateTransport2DRK.py, kernels the working loop does not launch (csrc/sparse_rest_tr.h)
 ("tr", "calCollisionTransportGPU", T + ":118", "totalNodes:i xDim:i numTracers:i numScheme:i unitVX:D unitVY:D velocityVX:D velocityVY:D tauTransport:D valueJDE:D tracerConc:D tracerPDF:D tracerPDFNew:D",
  "tr_check_q5(numScheme); launch_tr_collide_bgk(st, totalNodes, (int)numTracers, velocityVX, velocityVY, tauTransport, valueJDE, tracerConc, tracerPDF)"),
 ("tr", "calUpdateDistributionGPU", T + ":197", "totalNodes:i xDim:i criteriaFluid:d fluidRhoR:D distriField:B",
  "launch_tr_update_distribution(st, totalNodes, criteriaFluid, fluidRhoR, distriField)"),
 ("tr", "calUpdateConcOnNewNodesGPU", T + ":216", "totalNodes:i xDim:i numTracers:i newFluidList:L surroundingNodes:I tracerConc:D distrField:B",
  "launch_tr_conc_on_new_nodes(st, totalNodes, (int)numTracers, newFluidList, newFluidList_len, surroundingNodes, tracerConc, distrField)"),
 ("tr", "calUpdateConcOnOldNodesGPU", T + ":245", "totalNodes:i xDim:i numTracers:i oldFluidList:L tracerConc:D tracerPDF:D",
  "launch_tr_clear_listed(st, totalNodes, (int)numTracers, oldFluidList, oldFluidList_len, tracerConc, tracerPDF)"),
 ("tr", "calUpdateConcOnAllNewNodesGPU", T + ":267", "totalNodes:i xDim:i numTracers:i transportDomain:B tracerConc:D tracerPDF:D",
  "launch_tr_clear_outside(st, totalNodes, (int)numTracers, transportDomain, tracerConc, tracerPDF)"),
 ("tr", "calUpdateConcWholeDomainGPU", T + ":285", "totalNodes:i nx:i xDim:i numTracers:i randomPert:d fluidNodes:I sumOldConc:D sumOldList:D sumNewList:D tracerConcNew:D tracerConc:D distrField:B",
  "launch_tr_rescale_whole_domain(st, totalNodes, (int)numTracers, randomPert, sumOldConc, sumOldList, sumNewList, tracerConcNew, tracerConc)"),
 ("tr", "calTransportInterfaceGPU", T + ":310", "totalNodes:i xDim:i numTracers:i numScheme:i neighboringNodes:I velocityVX:D velocityVY:D tracerConc:D tracerPDF:D distriField:B",
  "tr_check_q5(numScheme); launch_tr_interface_exchange<5>(st, totalNodes, (int)numTracers, neighboringNodes, tracerPDF, distriField)"),
 ("tr", "calUpdatedPDFWithNewRho", T + ":389", "totalNodes:i xDim:i numTracers:i newList:L unitX:D unitY:D velocityX:D velocityY:D tracerConc:D tracerConcNew:D valueJDE:D tracerPDF:D distrField:B",
  "launch_tr_pdf_with_new_rho(st, totalNodes, (int)numTracers, newList, newList_len, velocityX, velocityY, tracerConc, tracerConcNew, valueJDE, tracerPDF, distrField)"),
 ("tr", "calFreeConcBoundary1", T + ":419", "totalNodes:i numTracers:i nx:i xDim:i fluidNodes:I neighboringNodes:I tracerConc:D tracerPDF:D",
  "launch_tr_free_row(st, totalNodes, (int)numTracers, nx, 2, fluidNodes, neighboringNodes, tracerPDF)"),
 ("tr", "calFreeConcBoundary2", T + ":440", "totalNodes:i numTracers:i nx:i xDim:i fluidNodes:I neighboringNodes:I tracerConc:D tracerPDF:D",
  "launch_tr_free_row(st, totalNodes, (int)numTracers, nx, 1, fluidNodes, neighboringNodes, tracerPDF)"),
 ("tr", "calZeroConcenBoundary", T + ":480", "totalNodes:i numTracers:i nx:i ny:i xDim:i fluidNodes:I tracerConc:D tracerPDF:D neighboringNodes:I",
  "launch_tr_zero_gradient_inlet(st, totalNodes, (int)numTracers, nx, ny, fluidNodes, tracerConc, tracerPDF, neighboringNodes)"),
 ("tr", "calUpdateConcInTransportDomainByV", T + ":500", "totalNodes:i numTracers:i xDim:i totalTracer:D totalOld:D transportDomain:B physicalVX:D physicalVY:D unitVX:D unitVY:D weightsCoeff:D tracerConc:D tracerPDF:D",
  "launch_tr_conc_by_velocity(st, totalNodes, (int)numTracers, totalTracer, totalOld, transportDomain, physicalVX, physicalVY, weightsCoeff, tracerConc, tracerPDF)"),
 ("tr", "calCollisionTransportQuadraticEqlMRTGPU", T + ":596", "totalNodes:i xDim:i numTracers:i unitVX:D unitVY:D velocityVX:D velocityVY:D tracerConc:D tracerPDF:D transportM:D inverseRelaxationMS:D weightsCoeff:D",
  "if (totalNodes > 5) { set_error(\"calCollisionTransportQuadraticEqlMRTGPU indexes the 5-entry unitVY by node (AccelerateTransport2DRK.py:624): defined for totalNodes <= 5 only\"); return LBMPM_ERR_UNSUPPORTED; } "
  "launch_tr_collide_mrt_quadratic(st, totalNodes, (int)numTracers, velocityVY, tracerConc, tracerPDF, transportM, inverseRelaxationMS, weightsCoeff)"),
 ("tr", "calAntiCollisionConcBoundary", T + ":661", "totalNodes:i xDim:i numTracers:i ny:i nx:i fluidNodes:I neighboringTRNodes:I concBoundary:D weightsCoeff:D tracerPDF:D",
  "launch_tr_anti_bounce_inlet(st, totalNodes, (int)numTracers, ny, nx, fluidNodes, neighboringTRNodes, concBoundary, weightsCoeff, tracerPDF)"),
 ("tr", "calCollisionQ9", T + ":704", "totalNodes:i xDim:i numTracers:i unitVX:D unitVY:D velocityVX:D velocityVY:D tauDiff:D tracerConc:D tracerPDF:D weightsCoeff:D",
  "launch_tr9_collide_bgk(st, totalNodes, (int)numTracers, velocityVX, velocityVY, tauDiff, tracerConc, tracerPDF, weightsCoeff)"),
 ("tr", "calStreaming1GPU", T + ":736", "totalNum:i numFluids:i xDim:i fluidNodes:I neighboringNodes:I fluidPDF:D fluidPDFNew:D",
  "launch_tr9_stream1(st, totalNum, (int)numFluids, neighboringNodes, fluidPDF, fluidPDFNew)"),
 ("tr", "calStreaming2GPU", T + ":823", "totalNum:i numFluids:i xDim:i fluidPDFNew:D fluidPDF:D",
  "launch_tr9_stream2(st, totalNum, (int)numFluids, fluidPDFNew, fluidPDF)"),
 ("tr", "calTransportInterfaceQ9GPU", T + ":839", "totalNodes:i xDim:i numTracers:i numScheme:i neighboringNodes:I velocityVX:D velocityVY:D tracerConc:D tracerPDF:D distriField:B",
  "if (numScheme != 9) { set_error(\"numScheme must be 9 (D2Q9); got %lld\", (long long)numScheme); return LBMPM_ERR_UNSUPPORTED; } "
  "launch_tr_interface_exchange<9>(st, totalNodes, (int)numTracers, neighboringNodes, tracerPDF, distriField)"),
 ("tr", "calUpdateConcInTransportDomainByVQ9", T + ":927", "totalNodes:i numTracers:i xDim:i totalTracer:D totalOld:D transportDomain:B physicalVX:D physicalVY:D unitVX:D unitVY:D weightsCoeff:D tracerConc:D tracerPDF:D",
  "set_error(\"calUpdateConcInTransportDomainByVQ9 cannot run in the reference: it stores 9 weights in a 5-entry shared array (AccelerateTransport2DRK.py:936-939) and indexes unitVY by node (:953)\"); return LBMPM_ERR_UNSUPPORTED"),
 ("tr", "calTransportWithInterfaceD2Q9", T + ":1019", "totalNodes:i xDim:i numTracers:i betaTracer:D valueTransportDomain:D unitEX:D unitEY:D gradientX:D gradientY:D weightsCoeff:D tracerConc:D tracerPDF:D",
  "launch_tr9_interface(st, totalNodes, (int)numTracers, betaTracer, valueTransportDomain, gradientX, gradientY, weightsCoeff, tracerConc, tracerPDF)"),
 ("tr", "calCollisionTransportLinearEqlMRTGPUD2Q9", T + ":1053", "totalNodes:i xDim:i numTracers:i unitVX:D unitVY:D velocityVX:D velocityVY:D tracerConc:D tracerPDF:D transportM:D inverseRelaxationMS:D weightsCoeff:D",
  "launch_tr9_collide_mrt(st, totalNodes, (int)numTracers, velocityVX, velocityVY, tracerConc, tracerPDF, transportM, inverseRelaxationMS, weightsCoeff)"),
 # ---------------- ShanChen2D/AccelerateGPU2D.py: the legacy DENSE explicit-forcing pipeline (:1336-2487, :2698) + the two macro kernels its
 # driver chains in front (csrc/dense_ef.h); dense direction-major f[9][ny * nx], boolean masks, periodic wrap inside the kernels
 ("de", "calMacroDensityGPU1D", DE + ":54", "nx:i ny:i fluidDensity:D fluidDistrC:D fluidDistrN:D isDomain:B",
  "launch_de_macro_density(st, nx, ny, fluidDensity, fluidDistrC, fluidDistrN, isDomain)"),
 ("de", "calMacroVelocityGPU1D", DE + ":80", "nx:i ny:i fluidVelocityX:D fluidVelocityY:D fluidDensity:D fluidDistr:D isDomain:B",
  "launch_de_macro_velocity(st, nx, ny, fluidVelocityX, fluidVelocityY, fluidDensity, fluidDistr)"),
 ("de", "calStreamingStep1", DE + ":1336", "nx:i ny:i fluidDistrOld:D fluidDistrMiddle:D",
  "launch_de_stream1(st, nx, ny, fluidDistrOld, fluidDistrMiddle)"),
 ("de", "calStreamingStep2", DE + ":1372", "nx:i ny:i fluidDistrNew:D fluidDistrMiddle:D",
  "launch_de_stream2(st, nx, ny, fluidDistrNew, fluidDistrMiddle)"),
 ("de", "calInteractionForceEFGPU", DE + ":1392", "nx:i ny:i constC:d interactionFluids:d potentialFluid0:D potentialFluid1:D externalForce0X:D externalForce0Y:D externalForce1X:D externalForce1Y:D isDomain1D:B isSolid:B",
  "launch_de_force(st, nx, ny, constC, interactionFluids, potentialFluid0, potentialFluid1, externalForce0X, externalForce0Y, externalForce1X, externalForce1Y, isDomain1D)"),
 ("de", "calExternalForceSolid", DE + ":2209", "nx:i ny:i interactionS0:d interactionS1:d potentialFluid0:D potentialFluid1:D externalForce0X:D externalForce0Y:D externalForce1X:D externalForce1Y:D isSolid:B",
  "launch_de_force_solid<false>(st, nx, ny, interactionS0, interactionS1, potentialFluid0, potentialFluid1, externalForce0X, externalForce0Y, externalForce1X, externalForce1Y, nullptr, isSolid)"),
 ("de", "calExternalForceSolidEF", DE + ":2257", "nx:i ny:i interactionS0:d interactionS1:d potentialFluid0:D potentialFluid1:D externalForce0X:D externalForce0Y:D externalForce1X:D externalForce1Y:D isDomain:B isSolid:B",
  "launch_de_force_solid<true>(st, nx, ny, interactionS0, interactionS1, potentialFluid0, potentialFluid1, externalForce0X, externalForce0Y, externalForce1X, externalForce1Y, isDomain, isSolid)"),
 ("de", "calEffectiveVGPU", DE + ":2309", "nx:i ny:i tau0:d tau1:d fluidDensity0:D fluidDensity1:D velocityX0:D velocityY0:D velocityX1:D velocityY1:D effectiveVX:D effectiveVY:D isDomain:B",
  "launch_de_effective_v<false>(st, nx, ny, tau0, tau1, fluidDensity0, fluidDensity1, velocityX0, velocityY0, velocityX1, velocityY1, effectiveVX, effectiveVY, isDomain)"),
 ("de", "calEffectiveVGPUMRT", DE + ":2332", "nx:i ny:i conserveS0:d conserveS1:d fluidDensity0:D fluidDensity1:D velocityX0:D velocityY0:D velocityX1:D velocityY1:D effectiveVX:D effectiveVY:D isDomain:B",
  "launch_de_effective_v<true>(st, nx, ny, conserveS0, conserveS1, fluidDensity0, fluidDensity1, velocityX0, velocityY0, velocityX1, velocityY1, effectiveVX, effectiveVY, isDomain)"),
 ("de", "calEquilibriumFuncEFGPU", DE + ":2354", "nx:i ny:i fluidDensity:D effectiveVX:D effectiveVY:D equilibriumFunc:D isDomain:B",
  "launch_de_equilibrium(st, nx, ny, fluidDensity, effectiveVX, effectiveVY, equilibriumFunc, isDomain)"),
 ("de", "calForcingTermEFGPU", DE + ":2403", "nx:i ny:i fluidDensity:D externalForceX:D externalForceY:D effectiveVX:D effectiveVY:D equilibriumFunc:D forcingTerm:D isDomain:B",
  "launch_de_forcing_term(st, nx, ny, fluidDensity, externalForceX, externalForceY, effectiveVX, effectiveVY, equilibriumFunc, forcingTerm, isDomain)"),
 ("de", "calTransformedDistrFuncGPU", DE + ":2444", "nx:i ny:i fluidDistr:D forcingTerm:D isDomain:B",
  "launch_de_transform(st, nx, ny, fluidDistr, forcingTerm, isDomain)"),
 ("de", "calMacroVelocityEFGPU", DE + ":2460", "nx:i ny:i fluidDensity:D externalFX:D externalFY:D distrFunc:D velocityX:D velocityY:D isDomain:B",
  "launch_de_velocity_ef(st, nx, ny, fluidDensity, externalFX, externalFY, distrFunc, velocityX, velocityY, isDomain)"),
 ("de", "calCollisionEFGPU", DE + ":2487", "nx:i ny:i tau:d fluidDistrOld:D equilibriumFunc:D forcingTerm:D isDomain:B",
  "launch_de_collision(st, nx, ny, tau, fluidDistrOld, equilibriumFunc, forcingTerm, isDomain)"),
 ("de", "calHalfWallBounceBack", DE + ":2698", "nx:i ny:i fluidDistr:D isDomain:B isSolid:B",
  "launch_de_bounce_back(st, nx, ny, fluidDistr, isDomain, isSolid)"),
]

CT = {"i": "int64_t", "d": "double", "I": "int64_t *", "D": "double *", "B": "uint8_t *"}
PY = {"i": "C.c_int64", "d": "C.c_double", "I": "C.c_void_p", "D": "C.c_void_p", "B": "C.c_void_p"}


def main():
    hdr = ['''/*
 * lbmpm_kernels.h -- kernel-level (drop-in) C ABI of liblbmpm_hip.so: one entry point per @cuda.jit KERNEL of the
 * reference's five kernel modules (RKCG2D/AcceleratedRKGPU2D.py -> lbmpm_rk_*, RKCG2D/RKGPU2DBoundary.py -> lbmpm_rkb_*,
 * ShanChen2D/OptimizedD2Q9GPU.py and ExplicitD2Q9GPU.py -> lbmpm_sc_*, RKCG2D/AccelerateTransport2DRK.py -> lbmpm_tr_*), plus the
 * explicit-forcing pipeline of the legacy dense file ShanChen2D/AccelerateGPU2D.py -> lbmpm_de_* (15 kernels on its own dense
 * direction-major arrays f[9][ny * nx]; the rest of that file is unreachable dead code, see DESIGN.md section 7),
 * the ones its working loops launch and the ones nothing launches alike, on the reference's own sparse
 * arrays (AoS f[N][9] / f[nF][N][9] float64, int64 neighbour tables, boolean masks one byte per entry), same argument
 * order as the Numba signature minus the launch configuration `[grid, block]`; a list argument the reference kernel
 * iterates over as a whole (`for m in newFluidList`) is followed by its length (<name>_len).
 * GENERATED by tools/gen_shim.py.
 *
 * All pointers are DEVICE pointers (see lbmpm_device_malloc & co below); `stream` is a hipStream_t
 * (NULL = the legacy default stream, which is what Numba's default-stream launches use).
 * Arguments the reference passes but the arithmetic does not need (xDim, arrays the reference kernel itself never reads) are
 * accepted and ignored.  The lattice-constant arrays (direction vectors, weights) are built into the kernels: each such argument is
 * CHECKED against the built-in table the first time its device pointer is seen (one small device-to-host copy, verdict cached per
 * pointer) and refused with LBMPM_ERR_UNSUPPORTED if it holds anything else -- never silently ignored.  Every call enqueues asynchronously and returns 0 or a
 * negative lbmpm_status.  Shan-Chen entry points require numFluids == 2 (like the reference's
 * outlet kernel); tracer entry points that take numScheme(s) the value that belongs to them (5, or 9 for the D2Q9 kernel).
 */
#ifndef LBMPM_KERNELS_H
#define LBMPM_KERNELS_H
#include "lbmpm.h"
#ifdef __cplusplus
extern "C" {
#endif

/* device-memory facade used by the numba.cuda-shaped Python shim (to_device / copy_to_host) */
int lbmpm_device_malloc(int64_t bytes, void **out);
int lbmpm_device_free(void *ptr);
int lbmpm_memcpy_h2d(void *dst, const void *src, int64_t bytes);
int lbmpm_memcpy_d2h(void *dst, const void *src, int64_t bytes);   /* synchronises, like copy_to_host */
int lbmpm_device_synchronize(void);
''']
    ent = ["// GENERATED by tools/gen_shim.py -- extern \"C\" kernel-level entry points\n#pragma once\n"]
    py = ['"""GENERATED by tools/gen_shim.py: ctypes signatures of the kernel-level entry points:\n(module tag, reference kernel) -> (C symbol, ctypes of the arguments, the reference kernel\'s own argument names,\nkind letters: i int64, d float64, I int64[], D float64[], B boolean[])."""\nimport ctypes as C\n\nKERNELS = {']
    import re
    Q9 = {"unitEX": "LT_D2Q9_EX", "EX": "LT_D2Q9_EX", "unitEY": "LT_D2Q9_EY", "EY": "LT_D2Q9_EY",
          "weightsCoeff": "LT_D2Q9_W", "weightCoeff": "LT_D2Q9_W", "weigthCoeff": "LT_D2Q9_W"}
    # (the tracer kernels' unitEX / unitEY are the flow lattice's or the tracer's depending on the kernel: left alone)
    Q5 = {"unitVX": "LT_D2Q5_VX", "unitX": "LT_D2Q5_VX", "unitVY": "LT_D2Q5_VY", "unitY": "LT_D2Q5_VY"}
    nchecks = 0
    for mod, name, cite, args, call in SPEC:
        al = [a.split(":") for a in args.split()]
        # lattice-constant arrays the launcher does not take: checked against the built-in table (once per device pointer)
        used = set(re.findall(r"\b[A-Za-z_]\w*\b", call))
        table = Q9 if mod in ("rk", "rkb", "sc") or "Q9" in name else ({k: v for k, v in Q5.items()} if mod == "tr" else {})
        if mod == "tr" and "Q9" in name:
            table = {k: v for k, v in Q9.items() if k not in ("weightsCoeff", "unitEX", "unitEY")}
            table.update({"unitVX": "LT_D2Q9_EX", "unitVY": "LT_D2Q9_EY", "unitX": "LT_D2Q9_EX", "unitY": "LT_D2Q9_EY"})
        checks = "".join("    if (int rc_ = lbmpm::check_lattice_constant(st, \"%s\", \"%s\", %s, lbmpm::%s)) return rc_;\n" % (name, n, n, table[n])
                         for n, k in al if k == "D" and n not in used and n in table)
        if "cannot run in the reference" in call:      # an entry point that refuses outright says why; nothing to check first
            checks = ""
        nchecks += checks.count("check_lattice_constant")
        call = "\n" + checks + "    " + call if checks else call
        cargs = ", ".join("int64_t *%s, int64_t %s_len" % (n, n) if k == "L" else "%s%s" % (CT[k] + ("" if CT[k].endswith("*") else " "), n) for n, k in al)
        sym = "lbmpm_%s_%s" % (mod, name)
        hdr.append("/* %s %s */\nint %s(void *stream, %s);" % (cite, name, sym, cargs))
        ent.append("extern \"C\" int %s(void *stream, %s)\n{\n    hipStream_t st = static_cast<hipStream_t>(stream);\n"
                   "    (void)st;%s\n    %s;\n    LBMPM_HIP_TRY(hipGetLastError());\n    return LBMPM_OK;\n}\n"
                   % (sym, cargs, "".join(" (void)%s;" % n + (" (void)%s_len;" % n if k == "L" else "") for n, k in al), call))
        py.append("    (%r, %r): (%r, [%s], %r, %r)," % (mod, name, sym, ", ".join("C.c_void_p, C.c_int64" if k == "L" else PY[k] for n, k in al), tuple(n for n, k in al),
                                                   "".join(k for n, k in al)))
    hdr.append("\n#ifdef __cplusplus\n}\n#endif\n#endif /* LBMPM_KERNELS_H */\n")
    py.append("}\n")
    open(os.path.join(ROOT, "include", "lbmpm_kernels.h"), "w").write("\n".join(hdr))
    open(os.path.join(ROOT, "openlbmpm_amd", "csrc", "sparse_entry_gen.h"), "w").write("\n".join(ent))
    open(os.path.join(ROOT, "openlbmpm_amd", "_kernel_specs.py"), "w").write("\n".join(py))
    print(len(SPEC), "entry points,", nchecks, "lattice-constant checks")


if __name__ == "__main__":
    main()
