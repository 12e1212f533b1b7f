/*
 * lbmpm.h -- C ABI of liblbmpm_hip.so, the MI355X (gfx950) replacement for the GPU
 * collision-streaming path of PorousMediaSimulation/openLBMPM.
 *
 * The reference has no FFI: its "operator interface" is a set of Numba @cuda.jit kernel
 * objects launched from Python drivers, one launch per algebraic step
 * (e.g. RKCG2D/RKD2Q9.py:1295-1490, 20 launches per time step).  This header is what a
 * maintainer binds instead (ctypes stub in INTEGRATION.md).  Two levels:
 *
 *   (1) FUSED solvers (performance path).  One context per simulation; the library owns
 *       the device memory (dense SoA lattice, q-major) and advances whole time steps in
 *       fused HIP kernels.  Host arrays cross the boundary in the reference's own dense
 *       result layout (the arrays its drivers write to HDF5: [ny][nx] and [ny][nx][9]
 *       float64, zeros at solid nodes; RKD2Q9.py:902-957).
 *
 *   (2) KERNEL-LEVEL entry points (drop-in path): one C function per reference kernel on
 *       the reference's sparse arrays (declared in lbmpm_kernels.h).
 *
 * Conventions: every function returns 0 on success or a negative lbmpm_status; the text
 * of the last failure on the calling thread is available from lbmpm_last_error().
 * All floating point is IEEE binary64.  No function takes or returns a C++/torch type.
 * A context is not re-entrant; distinct contexts are independent (one per GPU/process).
 */
#ifndef LBMPM_H
#define LBMPM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum lbmpm_status {
    LBMPM_OK = 0,
    LBMPM_ERR_INVALID = -1,     /* bad argument / configuration               */
    LBMPM_ERR_HIP = -2,         /* a HIP runtime call failed                   */
    LBMPM_ERR_NOMEM = -3,       /* device or host allocation failed            */
    LBMPM_ERR_STATE = -4,       /* call not valid in the context's state       */
    LBMPM_ERR_UNSUPPORTED = -5, /* option exists in the reference but not here */
    LBMPM_ERR_TIMEOUT = -6      /* a neighbour rank's face message did not arrive in time (lbmpm_rk3d_sync_deadline) */
} lbmpm_status;

const char *lbmpm_last_error(void);
/* "liblbmpm_hip <version> gfx950" */
const char *lbmpm_version(void);
/* number of visible HIP devices, or a negative status */
int lbmpm_device_count(void);
/* measured HBM ceiling of the device for the roofline: copy (read + write, two buffers), read-only and
 * in-place update (read + write of the same lines; may be NULL) GB/s over buffers of
 * bytes_per_buffer each (choose them far larger than the 256 MB Infinity Cache) */
int lbmpm_hbm_stream_test(int device, int64_t bytes_per_buffer, int reps, double *copy_gbs, double *read_gbs, double *inplace_gbs);

/* ------------------------------------------------------------------------------------
 * Colour-gradient D2Q9 two-phase solver with continuum-surface-force (CSF) tension.
 * Replaces the kernel sequence of RKColorGradientLBM.runRKColorGradient2DCSF
 * (RKCG2D/RKD2Q9.py:1225-1490) = the @cuda.jit kernels of RKCG2D/AcceleratedRKGPU2D.py:
 *   constantTotalVelocityInlet :2348, ghostPointsConstantVelocityRK :607,
 *   calConstPressureInletGPU :925, ghostPointsConstPressureInletRK :968,
 *   calConstPressureLowerGPUTotal :2560, ghostPointsConstPressureLowerRK :1045,
 *   convectiveOutletGPU/Ghost2GPU/Ghost3GPU :700/:731/:762,
 *   calTotalFluidPDF :1414, calPhysicalVelocityRKGPU2DNew1 :2634, calPhaseFieldPhi :1348,
 *   calColorValueOnSolid :1560, calRKInitialGradient :1584,
 *   updateColorGradientOnWetting :1639 / ...New :2430,
 *   calForceTermInColorGradient2D :1686 / ...New2D :2499,
 *   calRKCollision1TotalGPU2DSRTM :1804 + calPerturbationFromForce2D :1743,
 *   calRKCollision1TotalGPU2DMRTM :1938 + calPerturbationFromForce2DMRT :2027,
 *   calRecoloringProcessM :1857, calStreaming1GPU :340, calStreaming2GPU :409,
 *   calMacroDensityRKGPU2D :103
 * and the host set-up of RKD2Q9.py:657-892 (compaction, neighbour tables, wetting lists,
 * solid normals), which becomes implicit in the dense mask.
 * ---------------------------------------------------------------------------------- */

enum { LBMPM_RELAX_SRT = 0, LBMPM_RELAX_MRT = 1 };
enum { LBMPM_INLET_VELOCITY = 0,      /* BoundaryTypeInlet 'Neumann'   */
       LBMPM_INLET_PRESSURE = 1 };    /* BoundaryTypeInlet 'Dirichlet' */
enum { LBMPM_OUTLET_PRESSURE = 0,     /* BoundaryTypeOutlet 'Dirichlet'  */
       LBMPM_OUTLET_CONVECTIVE = 1,   /* BoundaryTypeOutlet 'Convective' */
       LBMPM_OUTLET_NONE = 2,         /* sc2d only: no boundary kernels at all, inlet included: fully periodic box
                                         (the static-droplet Laplace case of the reference's CPU path SimpleD2Q9) */
       LBMPM_OUTLET_FREEFLOW = 3 };   /* sc2d, explicit forcing, SRT (any ExplicitScheme): BoundaryTypeOutlet 'Freeflow'
                                         (ShanChenD2Q9.py:1865-1884, ExplicitD2Q9GPU.py:1476-1563): before every collision
                                         rows 2, 1, 0 take f-bar, F_i and f_eq of the row above, i.e. they leave the
                                         collision with the populations of row 3 */
enum { LBMPM_INLET_ZOUHE = 0,         /* BoundaryMethod 'ZouHe'                                                     */
       LBMPM_INLET_CHANG = 1 };       /* sc2d, explicit forcing, ExplicitScheme 4 only: BoundaryMethod 'Chang' -- the velocity
                                         inlet of Chang et al. 2009 (OptimizedD2Q9GPU.py:1127-1161, ShanChenD2Q9.py:1803,
                                         :1999): unknown populations from this step's streamed and the last step's final ones */

typedef struct lbmpm_rk2d_config {
    int64_t nx, ny;            /* xDomain, yDomain (incl. ghost rows 0 and ny-1)            */
    double surface_tension;    /* [SurfaceTension] SurfaceTensionValue                      */
    double contact_angle_deg;  /* [SurfaceTension] ContactAngle                             */
    int32_t wetting_type;      /* [SurfaceTension] WettingType 1 (Xu 2017) | 2 (Akai 2018)  */
    double beta;               /* [RKParameters] BetaThickness                              */
    double delta;              /* [RKParameters] DeltaValue                                 */
    double tau_r, tau_b;       /* [FluidParameters] TauR, TauB                              */
    int32_t tau_type;          /* [FluidParameters] TauType 1 | 2                           */
    int32_t relaxation;        /* [RelaxationType] Type: LBMPM_RELAX_*                      */
    int32_t inlet_type;        /* LBMPM_INLET_*                                             */
    int32_t outlet_type;       /* LBMPM_OUTLET_*                                            */
    double inlet_velocity_y;   /* VelocityYR + VelocityYB (RKD2Q9.py:1300)                  */
    double inlet_rho_r;        /* densityRH                                                 */
    double inlet_rho_b;        /* densityBH                                                 */
    double outlet_rho_total;   /* densityBL + densityRL (RKD2Q9.py:1344)                    */
    int32_t device;            /* HIP device ordinal                                        */
    int32_t variant;           /* 0 = default kernel schedule; see DESIGN.md                */
} lbmpm_rk2d_config;

typedef struct lbmpm_rk2d lbmpm_rk2d;

/* Field ids for lbmpm_rk2d_get_field.  "current" fields describe the lattice exactly as
 * the reference's device arrays hold it after the last completed time step. */
typedef enum lbmpm_rk2d_field {
    LBMPM_RK_PDF_R = 0,   /* [ny][nx][9] post-streaming f_R  (deviceFluidPDFR)      */
    LBMPM_RK_PDF_B = 1,   /* [ny][nx][9] post-streaming f_B  (deviceFluidPDFB)      */
    LBMPM_RK_RHO_R = 2,   /* [ny][nx]   sum_i f_R            (deviceFluidRhoR)      */
    LBMPM_RK_RHO_B = 3,   /* [ny][nx]                         (deviceFluidRhoB)      */
    LBMPM_RK_VX = 4,      /* [ny][nx]   u of the last step   (devicePhysicalVX)     */
    LBMPM_RK_VY = 5,      /*                                  (devicePhysicalVY)     */
    LBMPM_RK_PHI = 6,     /* [ny][nx]   phase field           (deviceColorValue)     */
    LBMPM_RK_GX = 7,      /* [ny][nx]   colour gradient       (deviceGradientX)      */
    LBMPM_RK_GY = 8,
    LBMPM_RK_FX = 9,      /* [ny][nx]   CSF force             (deviceForceX)         */
    LBMPM_RK_FY = 10,
    LBMPM_RK_K = 11,      /* [ny][nx]   curvature             (deviceKValue)         */
    /* what resultInHDF5 (RKD2Q9.py:938-957) would record at the START of the next step
     * (after that step's boundary kernels, velocity and phase field): */
    LBMPM_RK_REC_PDF_R = 20, LBMPM_RK_REC_PDF_B = 21, LBMPM_RK_REC_RHO_R = 22,
    LBMPM_RK_REC_RHO_B = 23, LBMPM_RK_REC_VX = 24, LBMPM_RK_REC_VY = 25
} lbmpm_rk2d_field;

/* is_domain: host [ny][nx] uint8, 1 = void/fluid, 0 = solid (isDomain, RKD2Q9.py:417-443).
 * Size limit: roundup(nx, 32) * ny < 2^28 nodes (the kernels address a node inside a lattice plane with 32 bits); beyond it LBMPM_ERR_INVALID. */
int lbmpm_rk2d_create(const lbmpm_rk2d_config *cfg, const uint8_t *is_domain,
                      lbmpm_rk2d **out);
void lbmpm_rk2d_destroy(lbmpm_rk2d *ctx);

/* Initial (or restart) state: dense host arrays [ny][nx][9] as fluidPDFR/fluidPDFB
 * (RKD2Q9.py:449-450); values at solid nodes are ignored.  Resets the step counter. */
int lbmpm_rk2d_set_pdf(lbmpm_rk2d *ctx, const double *pdf_r, const double *pdf_b);
/* Same from densities and velocity through f = rho w (1 + 3eu + 4.5(eu)^2 - 1.5u^2)
 * (RKD2Q9.py:577-601).  vx/vy may be NULL (= 0). */
int lbmpm_rk2d_set_macro(lbmpm_rk2d *ctx, const double *rho_r, const double *rho_b,
                         const double *vx, const double *vy);

/* Advance nsteps time steps (asynchronous on the context's stream). */
int lbmpm_rk2d_step(lbmpm_rk2d *ctx, int64_t nsteps);
/* Same, bracketed by HIP events on the context's stream; *ms_total = elapsed device time,
 * *ms_dominant = summed duration of the dominant (collision-streaming) kernel only. */
int lbmpm_rk2d_step_timed(lbmpm_rk2d *ctx, int64_t nsteps, double *ms_total,
                          double *ms_dominant);
int lbmpm_rk2d_sync(lbmpm_rk2d *ctx);
/* Diagnostics (u and curvature of the last step of each lbmpm_rk2d_step call are kept
 * for LBMPM_RK_VX/VY/K).  Off by default: costs three extra stores per node on that step. */
int lbmpm_rk2d_enable_diagnostics(lbmpm_rk2d *ctx, int on);
/* Use a caller-owned hipStream_t (e.g. torch's current stream); NULL = library stream. */
int lbmpm_rk2d_set_stream(lbmpm_rk2d *ctx, void *hip_stream);

/* Copy a field to a dense host array (synchronises).  out must hold ny*nx (or ny*nx*9)
 * doubles.  Solid nodes read 0. */
int lbmpm_rk2d_get_field(lbmpm_rk2d *ctx, int field, double *out);
/* --- D2Q5 tracer transport coupled to the colour-gradient flow (BASELINE config 4).
 * Replaces the D2Q5-MRT tracer sub-step of Transport2DRK.runTransport2DMPMCRKNew
 * (RKCG2D/Transport2DRK.py:1341-1418), i.e. RKCG2D/AccelerateTransport2DRK.py:
 *   calValueTransportDomain :957, calCollisionTransportLinearEqlMRTGPU :535,
 *   calTransportWithInterfaceD2Q5 :976, calFreeConcBoundary3 :461, calStreamingTransportGPU :139,
 *   calStreamingTransport2GPU :184, calInamuroConstConcBoundary :682, calConcentrationGPU :78,
 *   fillNeighboringNodesTransport :51 (implicit in the mask)
 * and the host matrices of Transport2DRK.py:313-347.  Fused into the flow kernel (+80 B per node
 * and tracer).  Keys follow the (non-shipped) transportsetup.ini read at Transport2DRK.py:35-311. */
typedef struct lbmpm_tracer_config {
    int32_t num_tracers;            /* [TransportParameters] NumberTracers, 1..4          */
    double diffusion_x[4];          /* [TransportMRT] DiffusionX                          */
    double diffusion_y[4];          /* [TransportMRT] DiffusionY                          */
    double diffusion_xy, diffusion_yx; /* [TransportMRT] DiffusionXY, DiffusionYX         */
    double beta_interface[4];       /* [TransportParameters] BetaInterface                */
    double criteria_rho;            /* criteriaFluidRho: tracer lives where rhoR <= this  */
    double inlet_concentration[4];  /* [BoundaryCondition] ConcentrationInlet             */
    int32_t dirichlet_inlet;        /* InletType 'Dirichlet' (Inamuro) on ghost row ny-1  */
    int32_t free_outlet;            /* OutletType 'Freeflow' on row 0                     */
    double reaction_rate;           /* [Reaction] ReactionRate (first value): A + B -> C between tracers 0, 1, 2
                                       (calReactionTracersGPU, AccelerateTransport2DRK.py:95); 0 = no reaction */
    double diffusion_j[4];          /* [TransportParameters] DiffusionJ: rest weight J_0 of every tracer; the
                                       reaction source is spread with J_0, (1-J_0)/4 x 4 (Transport2DRK.py:404) */
} lbmpm_tracer_config;

/* [SurfaceTension] SurfaceTensionType = 'Perturbation': the loop of RKColorGradientLBM.runRKColorGradient2DPerturbation
 * (RKCG2D/RKD2Q9.py:978-1223; kernels AcceleratedRKGPU2D.py:1125-1343 + the per-colour Zou-He rows :657-695, :1008-1039) as one
 * fused launch per time step instead of the CSF step the context was created for.  Uses tau_r, tau_b, beta, relaxation of the
 * create-time config (its inlet_type / outlet_type pick the loop's boundary kernels: velocity :657-695 or pressure :925-962 inlet per
 * colour from inlet_rho_r / inlet_rho_b, pressure outlet per colour :1008-1039 or convective outlet :700-784); the step streams FIRST
 * (set_pdf gives the state before the first streaming).  The rows 0, 1 (convective outlet: 0 .. 3), ny-2 and ny-1 must hold no solid
 * node, else LBMPM_ERR_UNSUPPORTED (the kernel-level entry points run such a lattice).
 * Fields: LBMPM_RK_PDF_R/B = the stored (recoloured) populations; RHO_R/B, VX, VY, PHI, GX, GY = those of the last step (needs
 * diagnostics); REC_* = what the next step records after streaming + boundary kernels. */
typedef struct lbmpm_rk2d_perturbation {
    double ak_r, ak_b;             /* [RKParameters] AkR, AkB                                  */
    double solid_phi;              /* phase field carried by solid neighbours (solidPhi)       */
    double inlet_velocity_y_r;     /* [BoundaryDefinition] VelocityYR                          */
    double inlet_velocity_y_b;     /*                      VelocityYB                          */
    double outlet_rho_r;           /*                      densityRL                           */
    double outlet_rho_b;           /*                      densityBL                           */
} lbmpm_rk2d_perturbation;
int lbmpm_rk2d_set_perturbation(lbmpm_rk2d *ctx, const lbmpm_rk2d_perturbation *par);
int lbmpm_rk2d_tracer_configure(lbmpm_rk2d *ctx, const lbmpm_tracer_config *cfg);
/* dense [ny][nx] concentration; g = C w (Transport2DRK.py:399-470); before the first step */
int lbmpm_rk2d_tracer_set_concentration(lbmpm_rk2d *ctx, int tracer, const double *conc);
/* deviceTracerConc[tracer] after the last completed step, dense [ny][nx] */
int lbmpm_rk2d_tracer_get_concentration(lbmpm_rk2d *ctx, int tracer, double *out);

int64_t lbmpm_rk2d_num_fluid_nodes(const lbmpm_rk2d *ctx);
int64_t lbmpm_rk2d_steps_done(const lbmpm_rk2d *ctx);
/* Name of the dominant kernel as it appears in rocprofv3 --kernel-trace output. */
const char *lbmpm_rk2d_dominant_kernel(const lbmpm_rk2d *ctx);
/* Device bytes held by the context. */
int64_t lbmpm_rk2d_device_bytes(const lbmpm_rk2d *ctx);

/* ------------------------------------------------------------------------------------
 * Two-component Shan-Chen D2Q9 solver: original Shan-Chen (velocity-shift forcing) and the
 * explicit forcing scheme "EFS" (Porter et al. 2012), SRT or MRT.
 * Replaces the kernel sequences of ShanChenD2Q9.runOptimizedLBM (ShanChen2D/ShanChenD2Q9.py
 * :1433-1629) and ShanChenD2Q9.runOptimizedEFLBM (:1631-2087), i.e. the @cuda.jit kernels
 *   ShanChen2D/OptimizedD2Q9GPU.py: savePDFLastStep :70, calFluidRhoGPU :84,
 *     calFluidPotentialGPUEql :99, calPhysicalVelocity :156, calMacroWholeVelocity :336,
 *     calStreaming1GPU :452, calStreaming2GPU :539, constantPressureZouHeBoundaryLower :555,
 *     ghostPointsConstantVelocityInlet :710, ghostPointsConstantPressureOutlet :743,
 *     constantVelocityZouHeBoundaryHigher :839, convectiveOutletGPU/Ghost2GPU/Ghost3GPU
 *     :960/:988/:1016, convectiveOutletEachGPU/Each2GPU/Each3GPU :1044/:1070/:1098,
 *     interactionCollisionProcess :1274
 *   ShanChen2D/ExplicitD2Q9GPU.py: calExplicit4thOrderScheme :51, calEquilibriumFuncEFGPU :227,
 *     calForceDistrGPU :255, transformPDFGPU :278, calCollisionEXGPU :294,
 *     calEquilibriumVEFGPU :340, transformPDFandEquil :1379, transfromForceTerm :1404,
 *     transformEquilibriumVelocity :1426, calAfterCollisionMRT :1457
 * Two fluids (NumberOfFluids = 2, like every shipped ini and the reference's outlet kernel),
 * potential psi = rho ('Simple'), ExplicitScheme = 4, inlet 'Neumann'/'ZouHe'.
 * ---------------------------------------------------------------------------------- */
enum { LBMPM_SC_MODEL_SHANCHEN = 0, LBMPM_SC_MODEL_EFS = 1 };

typedef struct lbmpm_sc2d_config {
    int64_t nx, ny;
    int32_t model;             /* [InterType] InteractionType: LBMPM_SC_MODEL_*           */
    int32_t relaxation;        /* [RelaxationType] Type: LBMPM_RELAX_* (MRT: EFS only)     */
    double tau[2];             /* [FluidProperties] FluidsTau                              */
    double g_fluid;            /* InteractionFluid (G_01 = G_10; G_kk = 0)                 */
    double g_solid[2];         /* InteractionSolid                                         */
    int32_t outlet_type;       /* LBMPM_OUTLET_PRESSURE ('Dirichlet': densities 1.0/0.02 as
                                  hard-coded in OptimizedD2Q9GPU.py:560) | _CONVECTIVE    */
    double inlet_velocity_y[2];/* [VelocityBoundary] velocityY                             */
    int32_t device;
    int32_t variant;           /* 0 = default                                              */
    int32_t force_scheme;      /* [ForceScheme] ExplicitScheme: 4 (also 0), 8 or 10 (EFS only).
                                  8: boundary rows one row further inside + two ghost rows;
                                  10: no boundary kernels (as the reference's loop); both run as
                                  two sweeps per step instead of the fused kernel              */
    int32_t inlet_method;      /* [BoundaryDefinition] BoundaryMethod: LBMPM_INLET_* (fills the struct's tail padding:
                                  size and the offsets of the older fields are unchanged)      */
} lbmpm_sc2d_config;

typedef struct lbmpm_sc2d lbmpm_sc2d;

/* Fields describe the lattice as the reference's device arrays hold it at the END of the last
 * completed loop iteration (EFS: after lbmpm_sc2d_step(n) the iterations 0..n-1 are complete). */
typedef enum lbmpm_sc2d_field {
    LBMPM_SC_PDF0 = 0, LBMPM_SC_PDF1 = 1,    /* [ny][nx][9]  deviceFluidPDF[k]             */
    LBMPM_SC_RHO0 = 2, LBMPM_SC_RHO1 = 3,    /* [ny][nx]     deviceFluidRho[k]             */
    LBMPM_SC_VX = 4, LBMPM_SC_VY = 5,        /* devicePhysicalVX/VY                        */
    LBMPM_SC_FX0 = 6, LBMPM_SC_FX1 = 7, LBMPM_SC_FY0 = 8, LBMPM_SC_FY1 = 9, /* deviceForceX/Y[k] */
    LBMPM_SC_UEQX = 10, LBMPM_SC_UEQY = 11,  /* EFS: deviceEquilibriumVX/VY                */
    /* original Shan-Chen only: what the driver records mid-iteration, after the inlet kernels of
     * the NEXT iteration (ShanChenD2Q9.py:1523-1572) */
    LBMPM_SC_REC_PDF0 = 20, LBMPM_SC_REC_PDF1 = 21, LBMPM_SC_REC_RHO0 = 22, LBMPM_SC_REC_RHO1 = 23
} lbmpm_sc2d_field;

/* (size limit as lbmpm_rk2d_create: roundup(nx, 32) * ny < 2^28 nodes) */
int lbmpm_sc2d_create(const lbmpm_sc2d_config *cfg, const uint8_t *is_domain, lbmpm_sc2d **out);
void lbmpm_sc2d_destroy(lbmpm_sc2d *ctx);
/* dense host arrays [ny][nx][9] per component (fluidPDF[k], ShanChenD2Q9.py:738) */
int lbmpm_sc2d_set_pdf(lbmpm_sc2d *ctx, const double *pdf0, const double *pdf1);
/* f_k = w rho_k (ShanChenD2Q9.py:757-766) */
int lbmpm_sc2d_set_density(lbmpm_sc2d *ctx, const double *rho0, const double *rho1);
int lbmpm_sc2d_step(lbmpm_sc2d *ctx, int64_t nsteps);
int lbmpm_sc2d_step_timed(lbmpm_sc2d *ctx, int64_t nsteps, double *ms_total, double *ms_dominant);
int lbmpm_sc2d_sync(lbmpm_sc2d *ctx);
/* keep forces / velocities of every step for lbmpm_sc2d_get_field (off by default) */
int lbmpm_sc2d_enable_diagnostics(lbmpm_sc2d *ctx, int on);
int lbmpm_sc2d_get_field(lbmpm_sc2d *ctx, int field, double *out);
int64_t lbmpm_sc2d_num_fluid_nodes(const lbmpm_sc2d *ctx);
int64_t lbmpm_sc2d_steps_done(const lbmpm_sc2d *ctx);
const char *lbmpm_sc2d_dominant_kernel(const lbmpm_sc2d *ctx);
int64_t lbmpm_sc2d_device_bytes(const lbmpm_sc2d *ctx);

/* ------------------------------------------------------------------------------------
 * D3Q19 colour-gradient solver (perturbation operator), z-slab capable.
 * The reference ships only IniFiles/RKtwophasesetup3D.ini for this model (main.py:22 imports a
 * module that is not in the tree), so there is no reference interface to mirror: the model
 * is the D3Q19 extension of the 2-D kernels AcceleratedRKGPU2D.py:1125 (BGK), :1169
 * (gradient + perturbation + recolouring), :657/:1008 (Zou-He per colour), :607/:1045 (ghost
 * planes), :340/:409 (streaming).  Pinned by reduction to that 2-D loop
 * (tests/test_rk3d_reduction.py) -- as repaired: the reference's runRKColorGradient2DPerturbation stops at
 * its first inlet launch, the captures come from it with four call-site repairs (tests/golden/gen/make_golden_rk_pert.py),
 * one of which (R3) moves calTotalFluidPDF behind collision 1 and so changes the loop's numbers.
 * A context owns the planes [z_offset, z_offset + nz_local) of a global lattice of
 * nz_global planes plus one halo plane on each side.  One time step of a slab is
 *     pack_halo -> [caller moves F_SEND_* to the neighbours' F_RECV_*] -> unpack_halo
 *     -> phase_field -> [caller moves PHI_SEND_* to the neighbours' PHI_RECV_*] -> collide
 * (the first step after set_density needs no population exchange).  With a single slab
 * lbmpm_rk3d_step does all of it.  variant 0 (default): collide is one fused z-marching kernel
 * that computes the phase field itself, phase_field(ctx, 0) then only produces the planes the
 * neighbours need; variant 1: phase_field and collide are two full sweeps.
 *
 * Scope of the 3-D model against SURVEY.md 8 a17 ("extend a3-a11 to D3Q19").  Built: streaming + half-way bounce-back, densities /
 * phase field / velocity, solid phase field (SolidRhoR, SolidRhoB), isotropic gradient, BGK and MRT, perturbation operator,
 * recolouring, velocity inlet and pressure outlet as z planes with their ghost planes -- everything IniFiles/RKtwophasesetup3D.ini
 * parametrises -- plus, since round 6, the loop's other two boundary kernels as z-plane rules: the pressure INLET per colour
 * (AcceleratedRKGPU2D.py:925-962, inlet_type) and the convective OUTLET (:700-784, outlet_type), each pinned by reduction to the 2-D fused
 * loop, whose kernels are pinned to the reference's one by one; and state in / out and restart (below).
 * [SurfaceTension] SurfaceTensionType = 'CSF' in 3-D (curvature force + wetting rule) is a model of its own: lbmpm_rk3dcsf_* at the end of this
 * header.  Not in either: body force (read and never used by the reference's colour-gradient loops).
 * ---------------------------------------------------------------------------------- */
typedef struct lbmpm_rk3d_config {
    int64_t nx, ny, nz_local, nz_global, z_offset;
    double ak_r, ak_b;          /* [RKParameters] AkR, AkB                         */
    double beta;                /* BetaThickness                                   */
    double tau_r, tau_b;        /* [FluidParameters] TauR, TauB                    */
    double solid_phi;           /* (SolidRhoR-SolidRhoB)/(SolidRhoR+SolidRhoB)     */
    double inlet_vz_r, inlet_vz_b;     /* [BoundaryCondition] velocityZR, velocityZB */
    double outlet_rho_r, outlet_rho_b; /* densityRL, densityBL                     */
    int32_t device;
    int32_t variant;
    int32_t relaxation;         /* [RelaxationType] Type: 0 'SRT', 1 'MRT' (RKtwophasesetup3D.ini:53-55;
                                 * D3Q19 moment basis of d'Humieres et al. 2002: s_e 1.19, s_eps = s_pi 1.4,
                                 * s_q = s_m 1.2, stress moments at 1/tau) */
    int32_t inlet_type;         /* [BoundaryCondition] BoundaryTypeInlet: LBMPM_INLET_VELOCITY 'Neumann' (the shipped ini; inlet_vz_*) |
                                 * LBMPM_INLET_PRESSURE 'Dirichlet' (inlet_rho_*: Zou-He pressure per colour on the plane nz-2, the z-plane
                                 * form of AcceleratedRKGPU2D.py:925-962; ghost plane as :968-1002) */
    double recolor_axis, recolor_diag;  /* weights w_i / |e_i| of the recolouring term beta rhoR rhoB / rho^2 w_i cos(theta_i)
                                 * (AcceleratedRKGPU2D.py:1241-1267) for the directions with |e_i| = 1 and sqrt 2.
                                 * 0 = the model's own, 1/18 and 1/(36 sqrt 2).  Other values exist for the parity pin:
                                 * recolor_axis = 1/9 - 2/(36 sqrt 2) makes a y-uniform lattice project exactly onto the
                                 * reference's D2Q9 perturbation loop (tests/test_rk3d_reduction.py) */
    double inlet_rho_r, inlet_rho_b;    /* densityRH, densityBH (pressure inlet only) */
    int32_t outlet_type;        /* [BoundaryCondition] BoundaryTypeOutlet: LBMPM_OUTLET_PRESSURE 'Dirichlet' (the shipped ini; outlet_rho_*) |
                                 * LBMPM_OUTLET_CONVECTIVE 'Convective': the planes z = 2, 1, 0 take the streamed populations of plane 3 and
                                 * re-sum their densities (AcceleratedRKGPU2D.py:700-784 as z planes).  Needs the masks of the planes
                                 * z = 0 .. 3 to coincide, nz_global >= 8, and the lattice's bottom slab to own >= 6 planes */
    int32_t reserved;
} lbmpm_rk3d_config;

typedef struct lbmpm_rk3d lbmpm_rk3d;

enum { LBMPM_RK3D_PHI = 0, LBMPM_RK3D_RHO_R = 1, LBMPM_RK3D_RHO_B = 2, LBMPM_RK3D_VX = 3,
       LBMPM_RK3D_VY = 4, LBMPM_RK3D_VZ = 5 };
enum { LBMPM_RK3D_BUF_F_SEND_UP = 0, LBMPM_RK3D_BUF_F_SEND_DOWN = 1,
       LBMPM_RK3D_BUF_F_RECV_FROM_BELOW = 2, LBMPM_RK3D_BUF_F_RECV_FROM_ABOVE = 3,
       LBMPM_RK3D_BUF_PHI_SEND_UP = 4, LBMPM_RK3D_BUF_PHI_SEND_DOWN = 5,
       LBMPM_RK3D_BUF_PHI_RECV_FROM_BELOW = 6, LBMPM_RK3D_BUF_PHI_RECV_FROM_ABOVE = 7 };

/* is_domain_with_halo: host [nz_local + 2][ny][nx] uint8 (1 = fluid): the owned planes plus
 * the plane below and above them; planes outside the global lattice must be 0.
 * Compact storage (the default for every nx: 23 doubles per fluid cell, rows cut into ceil(nx / 64) segments of <= 64 cells):
 * (nz_local + 2) * ny * ceil(nx / 64) row segments per slab must stay below 2^31; nx, ny <= 32767. */
int lbmpm_rk3d_create(const lbmpm_rk3d_config *cfg, const uint8_t *is_domain_with_halo, lbmpm_rk3d **out);
void lbmpm_rk3d_destroy(lbmpm_rk3d *ctx);
int lbmpm_rk3d_set_stream(lbmpm_rk3d *ctx, void *hip_stream);
/* owned planes [nz_local][ny][nx]; f = w rho at rest */
int lbmpm_rk3d_set_density(lbmpm_rk3d *ctx, const double *rho_r, const double *rho_b);
/* ---- State in and out (the reference's 2-D drivers: initial f from densities + velocity RKD2Q9.py:577-601; restart from recorded
 * populations, [CyclesSetup] IsCycle = 'yes', RKD2Q9.py:491-559; populations recorded RKD2Q9.py:938-957).  Host arrays cover the owned
 * planes, dense, zeros / ignored off the fluid.  Every set_* resets the step counter (set_state: to the value given).
 *
 * "post_collision": what the populations are.  0 = the lattice as a time step finds it after streaming (what set_density / set_macro
 * give: the first step streams nothing); 1 = the lattice as a time step leaves it -- collided and recoloured, to be streamed by the
 * next step: what the reference's arrays hold when it records them, and what get_pdf / get_state return after at least one step.
 *
 * set_macro: f_c,i = rho_c w_i (1 + 3 e_i.u + 4.5 (e_i.u)^2 - 1.5 u^2); vx, vy, vz [nz_local][ny][nx] or NULL (= 0; all NULL is
 *   set_density bit for bit).
 * get_pdf / set_pdf: f_R, f_B as [nz_local][ny][nx][19] each -- the 3-D analogue of fluidPDFR / fluidPDFB [ny][nx][9]; direction order
 *   rest, +x -x +y -y +z -z, (x,y) ++ -- +- -+, (x,z) ++ -- +- -+, (y,z) ++ -- +- -+.  The 23-value storage converts inside
 *   (f_R,i = k_R g_i + c_i e_i.A; back: g = f_R + f_B, k_R = rho_R / rho, A from the first moment of f_R): the round trip is good to
 *   rounding (1e-15), not to the bit, and populations that are not a state of the model are projected onto one.
 * get_state / set_state: the doubles a cell stores, as stored, [nz_local][ny][nx][S] with S = state_info out[0] (23: g_0..18, k_R,
 *   A_x, A_y, A_z; 38: f_R,0..18, f_B,0..18): a run continued from (get_state, steps, post_collision) in a NEW context -- same lattice,
 *   same parameters, same storage, any slab decomposition -- equals the uninterrupted run bit for bit.
 * state_info: out[0] doubles per cell of get_state, out[1] steps done, out[2] post_collision of the current state. */
int lbmpm_rk3d_set_macro(lbmpm_rk3d *ctx, const double *rho_r, const double *rho_b, const double *vx, const double *vy, const double *vz);
int lbmpm_rk3d_set_pdf(lbmpm_rk3d *ctx, const double *pdf_r, const double *pdf_b, int post_collision);
int lbmpm_rk3d_get_pdf(lbmpm_rk3d *ctx, double *pdf_r, double *pdf_b);
int lbmpm_rk3d_state_info(const lbmpm_rk3d *ctx, int64_t *out);
int lbmpm_rk3d_get_state(lbmpm_rk3d *ctx, double *state);
int lbmpm_rk3d_set_state(lbmpm_rk3d *ctx, const double *state, int64_t doubles_per_cell, int64_t steps_done, int post_collision);
int lbmpm_rk3d_pack_halo(lbmpm_rk3d *ctx);
int lbmpm_rk3d_unpack_halo(lbmpm_rk3d *ctx, int have_below, int have_above);
/* with_diagnostics: also keep rhoR, rhoB, u of the streamed, boundary-corrected lattice */
int lbmpm_rk3d_phase_field(lbmpm_rk3d *ctx, int with_diagnostics);
int lbmpm_rk3d_collide(lbmpm_rk3d *ctx);
/* Overlap of the halo exchange with the bulk of the step: call collide_interior FIRST in a step
 * (it collides the planes that do not depend on the neighbours on a second stream), then
 * pack_halo .. unpack_halo .. phase_field .. [phi exchange] as above on the context's stream, then
 * collide_boundary (the planes next to the slab faces; joins the two streams and ends the
 * step).  collide_boundary without a preceding collide_interior equals lbmpm_rk3d_collide. */
int lbmpm_rk3d_collide_interior(lbmpm_rk3d *ctx);
int lbmpm_rk3d_collide_boundary(lbmpm_rk3d *ctx);
/* One call = n whole time steps of a slab with the schedule above, the transfers left to the caller's transport:
 * exchange(user, what) -- what 0: populations (F_SEND_* -> the neighbours' F_RECV_*), 1: phase field (PHI_*) --
 * must ENQUEUE the transfers on the context's stream (lbmpm_rk3d_set_stream) and return 0; it is called twice per
 * step from the calling thread (not at all on a slab without neighbours; the first step moves no populations).
 * timed != 0: HIP events around the step, the interior launch, the exchange chain and the boundary launches of
 * (up to 256) steps; read the averages with lbmpm_rk3d_slab_timing (out[5]: step, interior, pack..phi exchange,
 * boundary [ms], steps averaged). */
/* (exchange == NULL on a slab with neighbours: the transport connected to the context -- see below -- moves the messages) */
typedef int (*lbmpm_rk3d_exchange_fn)(void *user, int what);
int lbmpm_rk3d_step_slab(lbmpm_rk3d *ctx, int64_t nsteps, int has_below, int has_above, lbmpm_rk3d_exchange_fn exchange,
                         void *user, int timed);
int lbmpm_rk3d_slab_timing(lbmpm_rk3d *ctx, double *out);

/* ---- Transport of the slab exchange INSIDE the library ("ncclGroupStart; ncclSend / ncclRecv ...; ncclGroupEnd on a
 * communication stream").  With a transport connected, lbmpm_rk3d_step_slab(ctx, n, has_below, has_above, NULL, NULL, timed) and
 * lbmpm_rk3d_halo_exchange drive the face messages themselves: no callback, no host work per step beyond the enqueues.  The compact
 * 23-value storage only (one face message per cut and step).  Two transports:
 *
 * LBMPM_TRANSPORT_IPC (one node): every rank owns a landing area (two slots per face, by the parity of the message's sequence number,
 *   plus one 64-bit flag per slot in fine-grained memory) that its neighbours map with hipIpcOpenMemHandle.  A message is ONE
 *   hipMemcpyAsync from the sender's packed face into the receiver's slot -- a copy-engine (SDMA) transfer over xGMI that needs no
 *   CU while the interior launch holds them all -- followed by hipStreamWriteValue64(flag, seq); the receiver's stream waits with
 *   hipStreamWaitValue64(flag >= seq) before it unpacks.  (Where the device cannot do stream value operations, one-lane kernels write
 *   and poll the flag instead.)  Two slots suffice without an acknowledgement: a rank sends message s + 2 only after it has received
 *   its neighbour's message s + 1, which that neighbour sent after unpacking message s.
 *     1. lbmpm_rk3d_ipc_init(ctx, blob): allocates the landing area, writes LBMPM_IPC_BLOB_BYTES describing it,
 *     2. the caller moves every rank's blob to its two neighbours by any byte transport (torch.distributed, MPI, a file),
 *     3. lbmpm_rk3d_ipc_connect(ctx, blob_of_the_rank_below | NULL, blob_of_the_rank_above | NULL).
 *   Blobs of the calling process itself (several slabs in one process) connect by plain pointers.
 *
 * LBMPM_TRANSPORT_RCCL: ncclSend / ncclRecv to the ranks rank - 1 and rank + 1 in one group per message, on the context's stream;
 *   librccl is opened at run time (dlopen: `librccl_path` or "librccl.so.1" / "librccl.so"), so the library does not link against it.
 *     1. rank 0: lbmpm_rccl_unique_id(id, path), 2. the caller broadcasts the LBMPM_RCCL_ID_BYTES, 3. every rank:
 *     lbmpm_rk3d_rccl_connect(ctx, id, rank, nranks, path)  (collective: ncclCommInitRank).
 *
 * Every rank of a run must make the same sequence of exchanging calls (step_slab with the same step counts, halo_exchange). */
enum { LBMPM_TRANSPORT_NONE = 0, LBMPM_TRANSPORT_IPC = 1, LBMPM_TRANSPORT_RCCL = 2 };
#define LBMPM_IPC_BLOB_BYTES 256
#define LBMPM_RCCL_ID_BYTES 128
int lbmpm_rk3d_ipc_init(lbmpm_rk3d *ctx, void *blob_out);
int lbmpm_rk3d_ipc_connect(lbmpm_rk3d *ctx, const void *blob_below, const void *blob_above);
int lbmpm_rccl_unique_id(void *id_out, const char *librccl_path);
int lbmpm_rk3d_rccl_connect(lbmpm_rk3d *ctx, const void *id, int rank, int nranks, const char *librccl_path);
/* drop the connected transport (closes the mapped handles / destroys the communicator); the callback path applies again */
int lbmpm_rk3d_transport_disconnect(lbmpm_rk3d *ctx);
/* LBMPM_TRANSPORT_* of the context; *value_ops (may be NULL): 1 when the IPC flags go through hipStreamWriteValue64 / WaitValue64 */
int lbmpm_rk3d_transport_kind(lbmpm_rk3d *ctx, int *value_ops);
/* pack -> exchange over the connected transport -> unpack of the CURRENT state's face planes, enqueued on the context's stream (what
 * step_slab does before its first step; also used before diagnostics of a distributed run) */
int lbmpm_rk3d_halo_exchange(lbmpm_rk3d *ctx);
/* IPC only, for a set-up self-test with a deadline: makes every wait of this context's stream on an incoming message return
 * (the host writes the largest sequence number into the context's own flags).  The transport is unusable afterwards: disconnect. */
int lbmpm_rk3d_ipc_release_waits(lbmpm_rk3d *ctx);
/* Probe of the connected transport between the real neighbours (set-up, before set_density): `rounds` patterned messages each way
 * through the send buffers and landing slots -- every slot several times, so that a stale cached line would show -- enqueued on the
 * context's stream with a comparing kernel behind the transport's waits.  The caller waits for the stream under a deadline (IPC:
 * lbmpm_rk3d_ipc_release_waits frees a stuck wait), then reads the number of doubles that arrived wrong. */
int lbmpm_rk3d_transport_probe(lbmpm_rk3d *ctx, int rounds);
int lbmpm_rk3d_transport_probe_result(lbmpm_rk3d *ctx, int64_t *mismatches);
/* Transport-level self-test on ONE GPU: three messages of `bytes` bytes sent "up" and "down" to the caller itself through the given
 * transport -- IPC: the landing area connected to itself (by pointer: same process), both slot parities, copies + flag operations as
 * in a run; RCCL: a one-rank communicator, ncclSend / ncclRecv to rank 0 in one group -- and compared with what was sent. */
int lbmpm_transport_selftest(int kind, int device, int64_t bytes, const char *librccl_path);
int lbmpm_rk3d_step(lbmpm_rk3d *ctx, int64_t nsteps);
int lbmpm_rk3d_step_timed(lbmpm_rk3d *ctx, int64_t nsteps, double *ms_total, double *ms_dominant);
int lbmpm_rk3d_sync(lbmpm_rk3d *ctx);
/* The steady-state watchdog of a slab run.  Like lbmpm_rk3d_sync, but the host polls the context's streams and gives up after
 * `seconds` WITHOUT PROGRESS -- the exchange chain of every slab step ends by writing its step number into a pinned host word, and the
 * deadline counts from the last time that word moved (from the call, if it never does): however many steps are queued and however
 * slow a healthy neighbour is, only a stream that sits in a wait for a face message that will not come (the neighbour died, or hangs
 * itself) runs into it.
 * IPC transport: the host releases the waits (lbmpm_rk3d_ipc_release_waits), drains the streams and returns LBMPM_ERR_TIMEOUT;
 * RCCL transport: the communicator is aborted (ncclCommAbort) first.  The context's lattice state is garbage afterwards and the
 * transport is marked dead (every later exchange returns LBMPM_ERR_TIMEOUT): the caller reports and ends the run, or sets it up
 * again from a record.  Every rank of a run whose neighbour chain is broken reaches its own deadline: call it on all of them.
 * (The reference has no multi-GPU path; this belongs to the slab decomposition of SURVEY section 8e.) */
int lbmpm_rk3d_sync_deadline(lbmpm_rk3d *ctx, double seconds);
/* device pointer + size of a halo buffer (LBMPM_RK3D_BUF_*) for the caller's transport */
int lbmpm_rk3d_buffer(lbmpm_rk3d *ctx, int which, void **device_ptr, int64_t *bytes);
/* owned planes [nz_local][ny][nx] */
int lbmpm_rk3d_get_field(lbmpm_rk3d *ctx, int field, double *out);
/* out[4]: doubles stored per fluid cell (38, or 23 with the compressed compact storage: 19 colour-blind populations + k_R + the
 * recolouring vector, from which the pull rebuilds both colours -- AcceleratedRKGPU2D.py:1241-1267 makes the two lattices an affine
 * image of those), fluid cells owned, those of them in row segments flagged single-colour (no records kept), bytes one step moves
 * for the owned cells by the storage's own count */
int lbmpm_rk3d_storage_info(lbmpm_rk3d *ctx, int64_t *out);
/* development aid: with LBMPM_RK3D_TRACE set at create time every workgroup of the last rk3dq_fused launch leaves four words
 * (start, prologue done, end on the 100 MHz clock; first << 32 | last plane): copied to out[4 * nblocks] */
int lbmpm_rk3d_debug_trace(lbmpm_rk3d *ctx, unsigned long long *out, int64_t nblocks);
/* development aid: one stored component of plane zl (0 .. nz_local+1, halo planes included) of the current state as a dense nx x ny
 * plane: comp 0..18 populations, 19..22 the record {k_R, A}, 23 the phase-field array, 24 the row flags (out[y * nseg + s]) */
int lbmpm_rk3d_debug_plane(lbmpm_rk3d *ctx, int comp, int zl, double *out);
int64_t lbmpm_rk3d_num_fluid_nodes(const lbmpm_rk3d *ctx);
int64_t lbmpm_rk3d_steps_done(const lbmpm_rk3d *ctx);
const char *lbmpm_rk3d_dominant_kernel(const lbmpm_rk3d *ctx);
int64_t lbmpm_rk3d_device_bytes(const lbmpm_rk3d *ctx);

/* ------------------------------------------------------------------------------------
 * Colour-gradient D3Q19 two-phase solver with continuum-surface-force (CSF) tension: [SurfaceTension] SurfaceTensionType = 'CSF' in
 * three dimensions ("extend a3-a11 to D3Q19 ... CSF kappa = -div n in 3-D").  The reference has no 3-D source; this
 * is its 2-D CSF loop RKColorGradientLBM.runRKColorGradient2DCSF (RKCG2D/RKD2Q9.py:1295-1490 -- the kernel list is the one above
 * lbmpm_rk2d_config) carried to D3Q19 kernel by kernel with z as the flow axis: the inlet acts on the plane nz-2 (ghost plane nz-1),
 * the outlet on the plane 1 (ghost plane 0), every edge wraps periodically like fillNeighboringNodes (walls are what the mask says).
 * Pinned by reduction: a lattice uniform along y reproduces the capture of the real 2-D driver (SRT) and the pinned 2-D oracle
 * (tests/test_oracle_rk3d_csf.py, tests/test_rk3d_csf_gpu.py).  What differs from the 2-D loop:
 *   - wetting rule: WettingType 2 (Akai et al. 2018, AcceleratedRKGPU2D.py:2430-2492; a vector rule, valid as it stands in 3-D) with the
 *     solid normals from the 3-D E8 stencil of Sbragaglia et al. 2007 (92 points), whose sums along one axis are the reference's
 *     24-point weights (RKD2Q9.py:811-885).  WettingType 1 (Xu 2017, :1639-1679) is a rotation in the plane: LBMPM_ERR_UNSUPPORTED.
 *     "Wetting solids" / "fluid next to solid" are taken over the 18 lattice neighbours (2-D: the 8 of the 3 x 3 block).
 *   - curvature K = -(I - n n) : grad n (the 2-D formula :2512-2551 is its restriction), Zou-He closures with the transverse terms of
 *     Hecht & Harting 2010 (their 2-D image is the 1/2 (f_1 - f_3) of :943-944),
 *   - MRT: moment basis of d'Humieres et al. 2002, rates s_e 1.19, s_eps = s_pi 1.4, s_q = s_m 1.2 (as lbmpm_rk3d_config), stress
 *     moments at 1/tau, conserved moments 0; Guo source in moment space M^-1 (I - S/2) M (:2027-2113).  mrt_rates overrides (tests).
 * Slabs along z (one context per GPU / rank): ghost_lo = ghost_hi = 2 make the two planes at either end of the context's lattice images of
 * the neighbouring slabs' edge planes.  The curvature reads n one cell around and n reads phi one cell around that, so a step has three
 * face messages instead of the perturbation model's one (all for fluid cells only; the receiver recomputes phi on the walls of its first
 * ghost plane): phi of two planes (after the phase field), n of one plane (after the gradient),
 * the five populations per colour that cross the face (after the collision; behind them a byte per fluid cell of two planes: whether the
 * cell's block handed on one colour alone, so that the bulk path -- see `variant` -- runs up to the faces).  lbmpm_rk3dcsf_stage runs a third of a step;
 * lbmpm_rk3dcsf_face_copy hands a message to a context of the same process, lbmpm_rk3dcsf_face_pack / _unpack go through a device buffer
 * the caller sends (RCCL / MPI / torch.distributed).  Bit-equal to the undivided lattice (tests/test_rk3d_csf_gpu.py).
 * ---------------------------------------------------------------------------------- */
typedef struct lbmpm_rk3dcsf_config {
    int64_t nx, ny, nz;        /* xDomain, yDomain, zDomain (incl. the ghost planes 0 and nz-1); nz >= 8 */
    double surface_tension;    /* [SurfaceTension] SurfaceTensionValue                      */
    double contact_angle_deg;  /* [SurfaceTension] ContactAngle (measured through the blue fluid with rule 2; at exactly 0 or 180 the rule's two
                                * candidates coincide and its choice -- AcceleratedRKGPU2D.py:2488-2492 -- is decided by rounding, as in the reference) */
    double beta, delta;        /* [RKParameters] BetaThickness, DeltaValue                  */
    double tau_r, tau_b;       /* [FluidParameters] TauR, TauB                              */
    double inlet_velocity_z;   /* velocityZR + velocityZB (2-D: RKD2Q9.py:1300)             */
    double inlet_rho_r, inlet_rho_b;   /* densityRH, densityBH (pressure inlet)             */
    double outlet_rho_total;   /* densityBL + densityRL (RKD2Q9.py:1344)                    */
    int32_t wetting_type;      /* 2 (Akai 2018) | 0: no correction of the gradient at walls */
    int32_t tau_type;          /* [FluidParameters] TauType 1 | 2                           */
    int32_t relaxation;        /* LBMPM_RELAX_*                                             */
    int32_t inlet_type;        /* LBMPM_INLET_VELOCITY | LBMPM_INLET_PRESSURE               */
    int32_t outlet_type;       /* LBMPM_OUTLET_PRESSURE | LBMPM_OUTLET_CONVECTIVE           */
    int32_t device;
    int32_t variant;           /* 0: blocks of 256 fluid cells deep inside one colour skip the phase-field pull, the gradient and the curvature
                                * (exact: phi = +-1, G = n = K = F = 0 there, bit for bit); 1: every cell takes the full path (cross-check) */
    double mrt_rates[6];       /* all 0: the model's own; else s_e, s_eps, s_q, s_pi, s_m, rate of the conserved moments */
    double bulk_epsilon;       /* 0: a colour whose density is below 2^-51 of the total is absent (the rounding of the total: densities, phi, u, G,
                                * populations equal the loop's to 1e-15; K, ill-conditioned where |G| sits at the loop's 1e-8 threshold, to 1e-8).
                                * OPT-IN, larger values (<= 1e-3): below ~1e-8 of the total the loop's own threshold switches the recolouring off
                                * and a colour's tail spreads by plain diffusion, ~ 8 sqrt(t / 6) cells, taking blocks off the bulk path; cut at
                                * bulk_epsilon the bulk path keeps its share in long runs (1e-10: + 35 % after 2 000 steps of the 512^3 drainage).
                                * The cut is a sink at the tail's end: after 3 000 steps densities, phi, u differ from the exact loop by 8e-8 at
                                * 1e-10 and 5e-5 at 1e-7 (tests/test_rk3d_csf_gpu.py, profiles/r06_soak_csf3d.txt) */
    int32_t ghost_lo, ghost_hi; /* 0, 0: the undivided lattice.  2, 2: a slab of it -- the planes 0, 1 and nz-2, nz-1 of this context are images of
                                * the neighbouring slabs' edge planes; is_domain and the arrays of set_macro / set_pdf / get_field carry them like any
                                * other plane (cut from the undivided lattice).  The slabs form a RING: the loop wraps z like x and y (the walls of
                                * the ghost plane 0 average phi over the plane nz-1), so the first slab's low face is the last slab's high face */
    int64_t slab_z0, global_nz; /* a slab: its first own plane (this context's plane 2) and the number of planes of the undivided lattice; the open
                                * planes are the undivided lattice's 0, 1 (0 .. 3 convective) and nz-2, nz-1; >= 4 own planes per slab */
} lbmpm_rk3dcsf_config;

typedef struct lbmpm_rk3dcsf lbmpm_rk3dcsf;

/* "current" fields: the lattice as the reference's device arrays would hold it after the last completed step (populations streamed,
 * densities re-summed; phi, G, F, K, u of that step).  REC_*: what the next step's first half makes of it -- boundary planes,
 * velocity with half the force, phase field -- i.e. what the reference records (RKD2Q9.py:1382-1393).  All [nz][ny][nx]
 * (populations [nz][ny][nx][19], direction order as lbmpm_rk3d_get_pdf), zeros off the fluid. */
typedef enum lbmpm_rk3dcsf_field {
    LBMPM_RK3DCSF_PDF_R = 0, LBMPM_RK3DCSF_PDF_B = 1, LBMPM_RK3DCSF_RHO_R = 2, LBMPM_RK3DCSF_RHO_B = 3,
    LBMPM_RK3DCSF_VX = 4, LBMPM_RK3DCSF_VY = 5, LBMPM_RK3DCSF_VZ = 6,        /* need enable_diagnostics */
    LBMPM_RK3DCSF_PHI = 7,                                                    /* carries phi_s on the wetting solids */
    LBMPM_RK3DCSF_GX = 8, LBMPM_RK3DCSF_GY = 9, LBMPM_RK3DCSF_GZ = 10,
    LBMPM_RK3DCSF_FX = 11, LBMPM_RK3DCSF_FY = 12, LBMPM_RK3DCSF_FZ = 13,
    LBMPM_RK3DCSF_K = 14,                                                     /* needs enable_diagnostics */
    LBMPM_RK3DCSF_NSX = 15, LBMPM_RK3DCSF_NSY = 16, LBMPM_RK3DCSF_NSZ = 17,   /* solid normal at the fluid cells next to solid */
    LBMPM_RK3DCSF_KIND = 18,   /* 0 solid, 1 fluid, 2 wetting solid, 3 fluid next to solid */
    LBMPM_RK3DCSF_REC_PDF_R = 30, LBMPM_RK3DCSF_REC_PDF_B = 31, LBMPM_RK3DCSF_REC_RHO_R = 32, LBMPM_RK3DCSF_REC_RHO_B = 33,
    LBMPM_RK3DCSF_REC_VX = 34, LBMPM_RK3DCSF_REC_VY = 35, LBMPM_RK3DCSF_REC_VZ = 36, LBMPM_RK3DCSF_REC_PHI = 37
} lbmpm_rk3dcsf_field;

/* is_domain: host [nz][ny][nx] uint8, 1 = fluid.  The ghost planes must carry the mask of the plane they copy (nz-1 <- nz-2; 0 <- 1,
 * or 0, 1, 2 <- 3 with the convective outlet). */
int lbmpm_rk3dcsf_create(const lbmpm_rk3dcsf_config *cfg, const uint8_t *is_domain, lbmpm_rk3dcsf **out);
void lbmpm_rk3dcsf_destroy(lbmpm_rk3dcsf *ctx);
/* f_c,i = rho_c w_i (1 + 3 e_i.u + 4.5 (e_i.u)^2 - 1.5 u^2) (RKD2Q9.py:577-601); vx, vy, vz may be NULL (= 0); the force of "the step
 * before" starts at zero.  Resets the step counter. */
int lbmpm_rk3dcsf_set_macro(lbmpm_rk3dcsf *ctx, const double *rho_r, const double *rho_b, const double *vx, const double *vy, const double *vz);
/* restart ([CyclesSetup] IsCycle, RKD2Q9.py:491-559): streamed populations [nz][ny][nx][19] per colour as LBMPM_RK3DCSF_PDF_* returns
 * them, and the force of the last step (LBMPM_RK3DCSF_F*; NULL = 0): a run continued from them equals the uninterrupted run bit for bit */
int lbmpm_rk3dcsf_set_pdf(lbmpm_rk3dcsf *ctx, const double *pdf_r, const double *pdf_b, const double *fx, const double *fy, const double *fz);
int lbmpm_rk3dcsf_step(lbmpm_rk3dcsf *ctx, int64_t nsteps);          /* undivided lattices only (ghost_lo = ghost_hi = 0) */
/* One third of a time step, in order 0, 1, 2: (0) the blocks' lists, the bulk's collision on its own stream, the phase field;
 * (1) phi on the wetting solids, gradient, n; (2) the collision of the full path, buffers swapped.  Between them the face messages:
 * LBMPM_CSF_MSG_PHI after stage 0, LBMPM_CSF_MSG_NORMAL after stage 1, LBMPM_CSF_MSG_PDF after stage 2.  face: 0 = low z, 1 = high z. */
enum { LBMPM_CSF_MSG_PDF = 0, LBMPM_CSF_MSG_PHI = 1, LBMPM_CSF_MSG_NORMAL = 2 };
int lbmpm_rk3dcsf_stage(lbmpm_rk3dcsf *ctx, int stage);
/* doubles of a message through that face (0 for the undivided lattice); the two sides of a face must agree (same mask on both) */
int64_t lbmpm_rk3dcsf_face_doubles(const lbmpm_rk3dcsf *ctx, int msg, int face);
/* ... of the message that comes in through that face (the populations travel for the fluid cells of the sender's edge plane only: the
 * plane this context holds an image of, not the one it sends) */
int64_t lbmpm_rk3dcsf_face_doubles_in(const lbmpm_rk3dcsf *ctx, int msg, int face);
/* the message this context sends through `face` into / the message it receives through `face` out of a DEVICE buffer of face_doubles
 * doubles, on the context's stream (lbmpm_rk3dcsf_sync before the buffer leaves; unpack before the next stage) */
int lbmpm_rk3dcsf_face_pack(lbmpm_rk3dcsf *ctx, int msg, int face, double *device_buffer);
int lbmpm_rk3dcsf_face_unpack(lbmpm_rk3dcsf *ctx, int msg, int face, const double *device_buffer);
/* same process: src's message through src_face straight into dst's ghost planes at the opposite face (ordered by events against both
 * contexts' streams; nothing to wait for on the host) */
int lbmpm_rk3dcsf_face_copy(lbmpm_rk3dcsf *src, int src_face, lbmpm_rk3dcsf *dst, int msg);
/* ms_total: HIP events around the steps; ms_dominant: the steps without their bookkeeping launches -- csf3d_collide_deep for the bulk on a
 * second stream beside phase field, solid phi, gradient and csf3d_collide for the blocks on the full path */
int lbmpm_rk3dcsf_step_timed(lbmpm_rk3dcsf *ctx, int64_t nsteps, double *ms_total, double *ms_dominant);
int lbmpm_rk3dcsf_sync(lbmpm_rk3dcsf *ctx);
/* keep u and K of every step (four more stores per cell) */
int lbmpm_rk3dcsf_enable_diagnostics(lbmpm_rk3dcsf *ctx, int on);
int lbmpm_rk3dcsf_get_field(lbmpm_rk3dcsf *ctx, int field, double *out);
int64_t lbmpm_rk3dcsf_num_fluid_nodes(const lbmpm_rk3dcsf *ctx);
int64_t lbmpm_rk3dcsf_num_wetting_solids(const lbmpm_rk3dcsf *ctx);
/* fluid cells whose block took the bulk path in the last step (variant 0; see lbmpm_rk3dcsf_config.variant) */
int64_t lbmpm_rk3dcsf_bulk_cells(lbmpm_rk3dcsf *ctx);
int64_t lbmpm_rk3dcsf_steps_done(const lbmpm_rk3dcsf *ctx);
int64_t lbmpm_rk3dcsf_device_bytes(const lbmpm_rk3dcsf *ctx);
const char *lbmpm_rk3dcsf_dominant_kernel(const lbmpm_rk3dcsf *ctx);

#ifdef __cplusplus
}
#endif
#endif /* LBMPM_H */
