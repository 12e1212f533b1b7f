"""ctypes binding of liblbmpm_hip.so (the C ABI declared in include/lbmpm.h).

The library is the product: there is NO CPU fallback.  If it is missing or cannot be
loaded this module raises immediately (build it with `python -m openlbmpm_amd.build`).
"""
import ctypes as C
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
# LBMPM_LIBRARY: another build of the same sources (tools/dev/devlib.py: the -DLBMPM_DEV build with time stamps and knock-outs)
LIB_PATH = os.environ.get("LBMPM_LIBRARY") or os.path.join(_PKG, "liblbmpm_hip.so")

F64P = C.POINTER(C.c_double)
U8P = C.POINTER(C.c_uint8)
I64P = C.POINTER(C.c_int64)


class LbmpmError(RuntimeError):
    """status: the C function's return value (LBMPM_ERR_* of include/lbmpm.h; None when no call was involved)"""

    def __init__(self, msg, status=None):
        super().__init__(msg)
        self.status = status


# status codes of include/lbmpm.h
OK, ERR_INVALID, ERR_HIP, ERR_NOMEM, ERR_STATE, ERR_UNSUPPORTED, ERR_TIMEOUT = 0, -1, -2, -3, -4, -5, -6


class RK2DConfig(C.Structure):
    # mirrors struct lbmpm_rk2d_config (include/lbmpm.h)
    _fields_ = [("nx", C.c_int64), ("ny", C.c_int64),
                ("surface_tension", C.c_double), ("contact_angle_deg", C.c_double),
                ("wetting_type", C.c_int32),
                ("beta", C.c_double), ("delta", C.c_double),
                ("tau_r", C.c_double), ("tau_b", C.c_double),
                ("tau_type", C.c_int32), ("relaxation", C.c_int32),
                ("inlet_type", C.c_int32), ("outlet_type", C.c_int32),
                ("inlet_velocity_y", C.c_double), ("inlet_rho_r", C.c_double),
                ("inlet_rho_b", C.c_double), ("outlet_rho_total", C.c_double),
                ("device", C.c_int32), ("variant", C.c_int32)]


class RK2DPerturbation(C.Structure):
    # mirrors struct lbmpm_rk2d_perturbation (include/lbmpm.h)
    _fields_ = [("ak_r", C.c_double), ("ak_b", C.c_double), ("solid_phi", C.c_double), ("inlet_velocity_y_r", C.c_double),
                ("inlet_velocity_y_b", C.c_double), ("outlet_rho_r", C.c_double), ("outlet_rho_b", C.c_double)]


class TracerConfig(C.Structure):
    # mirrors struct lbmpm_tracer_config (include/lbmpm.h)
    _fields_ = [("num_tracers", C.c_int32), ("diffusion_x", C.c_double * 4), ("diffusion_y", C.c_double * 4),
                ("diffusion_xy", C.c_double), ("diffusion_yx", C.c_double), ("beta_interface", C.c_double * 4),
                ("criteria_rho", C.c_double), ("inlet_concentration", C.c_double * 4),
                ("dirichlet_inlet", C.c_int32), ("free_outlet", C.c_int32),
                ("reaction_rate", C.c_double), ("diffusion_j", C.c_double * 4)]


class SC2DConfig(C.Structure):
    # mirrors struct lbmpm_sc2d_config (include/lbmpm.h)
    _fields_ = [("nx", C.c_int64), ("ny", C.c_int64), ("model", C.c_int32), ("relaxation", C.c_int32),
                ("tau", C.c_double * 2), ("g_fluid", C.c_double), ("g_solid", C.c_double * 2),
                ("outlet_type", C.c_int32), ("inlet_velocity_y", C.c_double * 2),
                ("device", C.c_int32), ("variant", C.c_int32), ("force_scheme", C.c_int32),
                ("inlet_method", C.c_int32)]


class RK3DConfig(C.Structure):
    # mirrors struct lbmpm_rk3d_config (include/lbmpm.h)
    _fields_ = [(n, C.c_int64) for n in ("nx", "ny", "nz_local", "nz_global", "z_offset")] + \
               [(n, C.c_double) for n in ("ak_r", "ak_b", "beta", "tau_r", "tau_b", "solid_phi", "inlet_vz_r",
                                          "inlet_vz_b", "outlet_rho_r", "outlet_rho_b")] + \
               [("device", C.c_int32), ("variant", C.c_int32), ("relaxation", C.c_int32), ("inlet_type", C.c_int32),
                ("recolor_axis", C.c_double), ("recolor_diag", C.c_double), ("inlet_rho_r", C.c_double), ("inlet_rho_b", C.c_double),
                ("outlet_type", C.c_int32), ("reserved", C.c_int32)]


class RK3DCSFConfig(C.Structure):
    # mirrors struct lbmpm_rk3dcsf_config (include/lbmpm.h)
    _fields_ = [(n, C.c_int64) for n in ("nx", "ny", "nz")] + \
               [(n, C.c_double) for n in ("surface_tension", "contact_angle_deg", "beta", "delta", "tau_r", "tau_b", "inlet_velocity_z",
                                          "inlet_rho_r", "inlet_rho_b", "outlet_rho_total")] + \
               [(n, C.c_int32) for n in ("wetting_type", "tau_type", "relaxation", "inlet_type", "outlet_type", "device", "variant")] + \
               [("mrt_rates", C.c_double * 6), ("bulk_epsilon", C.c_double), ("ghost_lo", C.c_int32), ("ghost_hi", C.c_int32),
                ("slab_z0", C.c_int64), ("global_nz", C.c_int64)]


EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int)      # lbmpm_rk3d_exchange_fn
IPC_BLOB_BYTES, RCCL_ID_BYTES = 256, 128                     # LBMPM_IPC_BLOB_BYTES, LBMPM_RCCL_ID_BYTES
TRANSPORT_NONE, TRANSPORT_IPC, TRANSPORT_RCCL = 0, 1, 2

_lib = None

# every symbol include/lbmpm.h declares (checked by tests/test_abi.py)
_SIGNATURES = {
    "lbmpm_last_error": (C.c_char_p, []),
    "lbmpm_version": (C.c_char_p, []),
    "lbmpm_device_count": (C.c_int, []),
    "lbmpm_rk2d_create": (C.c_int, [C.POINTER(RK2DConfig), U8P, C.POINTER(C.c_void_p)]),
    "lbmpm_rk2d_destroy": (None, [C.c_void_p]),
    "lbmpm_rk2d_set_pdf": (C.c_int, [C.c_void_p, F64P, F64P]),
    "lbmpm_rk2d_set_macro": (C.c_int, [C.c_void_p, F64P, F64P, F64P, F64P]),
    "lbmpm_rk2d_step": (C.c_int, [C.c_void_p, C.c_int64]),
    "lbmpm_rk2d_step_timed": (C.c_int, [C.c_void_p, C.c_int64, F64P, F64P]),
    "lbmpm_rk2d_sync": (C.c_int, [C.c_void_p]),
    "lbmpm_rk2d_enable_diagnostics": (C.c_int, [C.c_void_p, C.c_int]),
    "lbmpm_rk2d_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "lbmpm_rk2d_get_field": (C.c_int, [C.c_void_p, C.c_int, F64P]),
    "lbmpm_rk2d_num_fluid_nodes": (C.c_int64, [C.c_void_p]),
    "lbmpm_rk2d_steps_done": (C.c_int64, [C.c_void_p]),
    "lbmpm_rk2d_dominant_kernel": (C.c_char_p, [C.c_void_p]),
    "lbmpm_rk2d_device_bytes": (C.c_int64, [C.c_void_p]),
    "lbmpm_rk2d_tracer_configure": (C.c_int, [C.c_void_p, C.POINTER(TracerConfig)]),
    "lbmpm_rk2d_set_perturbation": (C.c_int, [C.c_void_p, C.POINTER(RK2DPerturbation)]),
    "lbmpm_rk2d_tracer_set_concentration": (C.c_int, [C.c_void_p, C.c_int, F64P]),
    "lbmpm_rk2d_tracer_get_concentration": (C.c_int, [C.c_void_p, C.c_int, F64P]),
    "lbmpm_sc2d_create": (C.c_int, [C.POINTER(SC2DConfig), U8P, C.POINTER(C.c_void_p)]),
    "lbmpm_sc2d_destroy": (None, [C.c_void_p]),
    "lbmpm_sc2d_set_pdf": (C.c_int, [C.c_void_p, F64P, F64P]),
    "lbmpm_sc2d_set_density": (C.c_int, [C.c_void_p, F64P, F64P]),
    "lbmpm_sc2d_step": (C.c_int, [C.c_void_p, C.c_int64]),
    "lbmpm_sc2d_step_timed": (C.c_int, [C.c_void_p, C.c_int64, F64P, F64P]),
    "lbmpm_sc2d_sync": (C.c_int, [C.c_void_p]),
    "lbmpm_sc2d_enable_diagnostics": (C.c_int, [C.c_void_p, C.c_int]),
    "lbmpm_sc2d_get_field": (C.c_int, [C.c_void_p, C.c_int, F64P]),
    "lbmpm_sc2d_num_fluid_nodes": (C.c_int64, [C.c_void_p]),
    "lbmpm_sc2d_steps_done": (C.c_int64, [C.c_void_p]),
    "lbmpm_sc2d_dominant_kernel": (C.c_char_p, [C.c_void_p]),
    "lbmpm_sc2d_device_bytes": (C.c_int64, [C.c_void_p]),
    "lbmpm_rk3d_create": (C.c_int, [C.POINTER(RK3DConfig), U8P, C.POINTER(C.c_void_p)]),
    "lbmpm_rk3d_destroy": (None, [C.c_void_p]),
    "lbmpm_rk3d_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "lbmpm_rk3d_set_density": (C.c_int, [C.c_void_p, F64P, F64P]),
    "lbmpm_rk3d_set_macro": (C.c_int, [C.c_void_p, F64P, F64P, F64P, F64P, F64P]),
    "lbmpm_rk3d_set_pdf": (C.c_int, [C.c_void_p, F64P, F64P, C.c_int]),
    "lbmpm_rk3d_get_pdf": (C.c_int, [C.c_void_p, F64P, F64P]),
    "lbmpm_rk3d_state_info": (C.c_int, [C.c_void_p, I64P]),
    "lbmpm_rk3d_get_state": (C.c_int, [C.c_void_p, F64P]),
    "lbmpm_rk3d_set_state": (C.c_int, [C.c_void_p, F64P, C.c_int64, C.c_int64, C.c_int]),
    "lbmpm_rk3d_pack_halo": (C.c_int, [C.c_void_p]),
    "lbmpm_rk3d_unpack_halo": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "lbmpm_rk3d_phase_field": (C.c_int, [C.c_void_p, C.c_int]),
    "lbmpm_rk3d_collide": (C.c_int, [C.c_void_p]),
    "lbmpm_rk3d_collide_interior": (C.c_int, [C.c_void_p]),
    "lbmpm_hbm_stream_test": (C.c_int, [C.c_int, C.c_int64, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "lbmpm_rk3d_collide_boundary": (C.c_int, [C.c_void_p]),
    "lbmpm_rk3d_step_slab": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int]),
    "lbmpm_rk3d_slab_timing": (C.c_int, [C.c_void_p, F64P]),
    "lbmpm_rk3d_ipc_init": (C.c_int, [C.c_void_p, C.c_void_p]),
    "lbmpm_rk3d_ipc_connect": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "lbmpm_rccl_unique_id": (C.c_int, [C.c_void_p, C.c_char_p]),
    "lbmpm_rk3d_rccl_connect": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_char_p]),
    "lbmpm_rk3d_transport_disconnect": (C.c_int, [C.c_void_p]),
    "lbmpm_rk3d_transport_kind": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "lbmpm_rk3d_halo_exchange": (C.c_int, [C.c_void_p]),
    "lbmpm_rk3d_ipc_release_waits": (C.c_int, [C.c_void_p]),
    "lbmpm_rk3d_transport_probe": (C.c_int, [C.c_void_p, C.c_int]),
    "lbmpm_rk3d_transport_probe_result": (C.c_int, [C.c_void_p, I64P]),
    "lbmpm_transport_selftest": (C.c_int, [C.c_int, C.c_int, C.c_int64, C.c_char_p]),
    "lbmpm_rk3d_step": (C.c_int, [C.c_void_p, C.c_int64]),
    "lbmpm_rk3d_step_timed": (C.c_int, [C.c_void_p, C.c_int64, F64P, F64P]),
    "lbmpm_rk3d_sync": (C.c_int, [C.c_void_p]),
    "lbmpm_rk3d_sync_deadline": (C.c_int, [C.c_void_p, C.c_double]),
    "lbmpm_rk3d_buffer": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), I64P]),
    "lbmpm_rk3d_get_field": (C.c_int, [C.c_void_p, C.c_int, F64P]),
    "lbmpm_rk3d_num_fluid_nodes": (C.c_int64, [C.c_void_p]),
    "lbmpm_rk3d_steps_done": (C.c_int64, [C.c_void_p]),
    "lbmpm_rk3d_dominant_kernel": (C.c_char_p, [C.c_void_p]),
    "lbmpm_rk3d_device_bytes": (C.c_int64, [C.c_void_p]),
    "lbmpm_rk3d_storage_info": (C.c_int, [C.c_void_p, I64P]),
    "lbmpm_rk3d_debug_trace": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    "lbmpm_rk3d_debug_plane": (C.c_int, [C.c_void_p, C.c_int, C.c_int, F64P]),
    "lbmpm_rk3dcsf_create": (C.c_int, [C.POINTER(RK3DCSFConfig), U8P, C.POINTER(C.c_void_p)]),
    "lbmpm_rk3dcsf_destroy": (None, [C.c_void_p]),
    "lbmpm_rk3dcsf_set_macro": (C.c_int, [C.c_void_p, F64P, F64P, F64P, F64P, F64P]),
    "lbmpm_rk3dcsf_set_pdf": (C.c_int, [C.c_void_p, F64P, F64P, F64P, F64P, F64P]),
    "lbmpm_rk3dcsf_step": (C.c_int, [C.c_void_p, C.c_int64]),
    "lbmpm_rk3dcsf_step_timed": (C.c_int, [C.c_void_p, C.c_int64, F64P, F64P]),
    "lbmpm_rk3dcsf_stage": (C.c_int, [C.c_void_p, C.c_int]),
    "lbmpm_rk3dcsf_face_doubles": (C.c_int64, [C.c_void_p, C.c_int, C.c_int]),
    "lbmpm_rk3dcsf_face_doubles_in": (C.c_int64, [C.c_void_p, C.c_int, C.c_int]),
    "lbmpm_rk3dcsf_face_pack": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "lbmpm_rk3dcsf_face_unpack": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "lbmpm_rk3dcsf_face_copy": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int]),
    "lbmpm_rk3dcsf_sync": (C.c_int, [C.c_void_p]),
    "lbmpm_rk3dcsf_enable_diagnostics": (C.c_int, [C.c_void_p, C.c_int]),
    "lbmpm_rk3dcsf_get_field": (C.c_int, [C.c_void_p, C.c_int, F64P]),
    "lbmpm_rk3dcsf_num_fluid_nodes": (C.c_int64, [C.c_void_p]),
    "lbmpm_rk3dcsf_num_wetting_solids": (C.c_int64, [C.c_void_p]),
    "lbmpm_rk3dcsf_bulk_cells": (C.c_int64, [C.c_void_p]),
    "lbmpm_rk3dcsf_steps_done": (C.c_int64, [C.c_void_p]),
    "lbmpm_rk3dcsf_device_bytes": (C.c_int64, [C.c_void_p]),
    "lbmpm_rk3dcsf_dominant_kernel": (C.c_char_p, [C.c_void_p]),
}


def _share_hip_runtime_with_torch():
    """PyTorch-ROCm wheels bundle their own libamdhip64/libhsa-runtime64.  Two HIP/HSA
    runtimes in one process cannot both open the GPU, so when torch is installed we make its
    copies the process-wide ones BEFORE liblbmpm_hip.so resolves its DT_NEEDED entries (same
    SONAMEs), whether or not torch itself has been imported yet.  Without torch the system
    ROCm runtime under /opt/rocm is used."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


_loaded = {}


def _load(path):
    if path in _loaded:
        return _loaded[path]
    if not os.path.exists(path):
        raise LbmpmError("%s is missing: the HIP library is the only compute path "
                         "(no CPU fallback). Build it: python -m openlbmpm_amd.build" % path)
    _share_hip_runtime_with_torch()
    try:
        L = C.CDLL(path)
    except OSError as e:
        raise LbmpmError("cannot load %s: %s" % (path, e))
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _loaded[path] = L
    return L


def lib():
    """Load (once) and return the shared library; raise LbmpmError if unavailable."""
    global _lib
    if _lib is None:
        _lib = _load(LIB_PATH)
    return _lib


def use_library(path=None):
    """Tests and dev tools: contexts created from now on come from the library at `path` (None: the product library again).  Objects
    keep the library they were created with; two builds of the sources can serve one process side by side."""
    global _lib
    _lib = _load(path or LIB_PATH)
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = lib().lbmpm_last_error().decode("utf-8", "replace")
        raise LbmpmError("%s failed (status %d): %s" % (what or "lbmpm call", rc, msg), status=rc)
