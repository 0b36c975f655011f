"""ctypes binding of the C-ABI shared library (include/fastmpc.h).

The library is built in-tree (`mpc-sensorlessao_amd/lib/libfastmpc.so`, see csrc/Makefile).  If it
is missing this module raises: there is no Python or CPU fallback for the solve path.
"""
from __future__ import annotations

import ctypes as C
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FMPC_LIB") or os.path.join(_HERE, "lib", "libfastmpc.so")

FMPC_OK = 0
FMPC_W_LINESEARCH = 1
FMPC_E_NULL = -1
FMPC_E_DIM = -2
FMPC_E_UNSUPPORTED = -3
FMPC_E_NOT_PD_PHI = -4
FMPC_E_NOT_PD_SCHUR = -5
FMPC_E_HIP = -6
FMPC_E_ALLOC = -7
FMPC_E_NO_DEVICE = -8
# fmpc_last_dispatch paths
FMPC_PATH_GENERIC = 0
FMPC_PATH_WAVE = 1
FMPC_PATH_SHARED = 2
FMPC_PATH_PANEL = 3
FMPC_PATH_RAMP = 4
FMPC_PATH_TILED = 5
FMPC_PATH_TILED_F32 = 6
FMPC_PREC_F64 = 0
FMPC_PREC_F32_MIXED = 1

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)
_vp = C.c_void_p

# name -> (restype, argtypes); every symbol include/fastmpc.h declares
SIGNATURES = {
    "fmpc_version": (C.c_int, []),
    "fmpc_strerror": (C.c_char_p, [C.c_int]),
    "fmpc_create": (C.c_int, [C.POINTER(_vp), C.c_int, C.c_int, C.c_int, C.c_int] + [_vp] * 14 + [C.c_int]),
    "fmpc_destroy": (C.c_int, [_vp]),
    "fmpc_dims": (C.c_int, [_vp, _ip, _ip, _ip, _ip, _ip]),
    "fmpc_solve": (C.c_int, [_vp, C.c_int] + [_vp] * 5 + [C.c_int, C.c_double] + [_vp] * 5),
    "fmpc_step_ld": (C.c_int, [C.c_int]),
    "fmpc_solve_device": (C.c_int, [_vp, C.c_int] + [_vp] * 5 + [C.c_int, C.c_double] + [_vp] * 5 + [_vp]),
    "fmpc_unpack": (C.c_int, [_vp, C.c_int] + [_vp] * 4),
    "fmpc_unpack_device": (C.c_int, [_vp, C.c_int] + [_vp] * 4 + [_vp]),
    "fmpc_solve_once": (C.c_int, [C.c_int] * 4 + [_vp] * 23 + [C.c_int, C.c_double, C.c_int, _vp, _vp]),
    "fmpc_solve_once_cache_clear": (C.c_int, []),
    "fmpc_last_dispatch": (C.c_int, [_vp, _ip, _ip]),
    "fmpc_set_dense_form": (C.c_int, [_vp, C.c_int, C.c_int]),
    "fmpc_last_dual_form": (C.c_int, [_vp]),
    "fmpc_last_tiled_wavefronts": (C.c_int, [_vp]),
    "fmpc_set_small_batch_kernel": (C.c_int, [_vp, C.c_int]),
    "fmpc_set_z_ld": (C.c_int, [_vp, C.c_int]),
    "fmpc_alloc_generation": (C.c_ulonglong, []),
    "fmpc_loop_inputs_device": (C.c_int, [_vp, C.c_int] + [_vp] * 7 + [_vp]),
    "fmpc_loop_step_device": (C.c_int, [_vp, C.c_int] + [_vp] * 8 + [C.c_int, C.c_double] + [_vp] * 6 + [_vp]),
    "fmpc_ao_step_device": (C.c_int, [_vp, C.c_int] + [_vp] * 6 + [C.c_int, C.c_double] + [_vp] * 6 + [_vp]),
    "fmpc_loop_run_device": (C.c_int, [_vp, C.c_int, C.c_int] + [_vp] * 4 + [C.c_int, C.c_int, C.c_double] + [_vp] * 7 + [_vp]),
    "fmpc_solve_u0_device": (C.c_int, [_vp, C.c_int] + [_vp] * 5 + [C.c_int, C.c_double] + [_vp] * 6 + [_vp]),
    "fmpc_solve_u0_device_ld": (C.c_int, [_vp, C.c_int] + [_vp] * 5 + [C.c_int, C.c_double] + [_vp] * 6 + [C.c_int, _vp]),
    "fmpc_solve_u0": (C.c_int, [_vp, C.c_int] + [_vp] * 5 + [C.c_int, C.c_double] + [_vp] * 4),
    "fmpc_set_ramp": (C.c_int, [_vp, _vp, _vp]),
    "fmpc_set_precision": (C.c_int, [_vp, C.c_int]),
    "fmpc_var_identify_device": (C.c_int, [C.c_int] * 4 + [_vp] * 5),
    "fmpc_solve_ramp": (C.c_int, [_vp, C.c_int] + [_vp] * 6 + [C.c_int, C.c_double] + [_vp] * 5),
    "fmpc_solve_ramp_device": (C.c_int, [_vp, C.c_int] + [_vp] * 6 + [C.c_int, C.c_double] + [_vp] * 5 + [_vp]),
    "fmpc_solve_ramp_u0_device": (C.c_int, [_vp, C.c_int] + [_vp] * 6 + [C.c_int, C.c_double] + [_vp] * 6 + [_vp]),
    "fmpc_phase_residual_device": (C.c_int, [_vp, C.c_int, C.c_longlong] + [_vp] * 4 + [_vp]),
    "fmpc_est_create": (C.c_int, [C.POINTER(_vp)] + [C.c_int] * 4 + [_vp, _vp, C.c_double, _vp, _vp, C.c_int, C.c_int, C.c_int]),
    "fmpc_est_destroy": (C.c_int, [_vp]),
    "fmpc_est_dims": (C.c_int, [_vp] + [_ip] * 6),
    "fmpc_est_apply": (C.c_int, [_vp, C.c_int] + [_vp] * 4),
    "fmpc_est_apply_device": (C.c_int, [_vp, C.c_int] + [_vp] * 4 + [_vp]),
}

_lib = None


class FastMPCError(RuntimeError):
    def __init__(self, code, where=""):
        self.code = int(code)
        msg = strerror(code)
        super().__init__(f"{where}: [{self.code}] {msg}" if where else f"[{self.code}] {msg}")


def load():
    """Load the shared library once; fail loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build the HIP extension first "
            "(`python -c 'import __graft_entry__ as g; g.build()'` or `make -C mpc-sensorlessao_amd/csrc`). "
            "This package has no CPU fallback.")
    # PyTorch-ROCm bundles its own HIP runtime (torch/lib/libamdhip64.so).  Whichever runtime is loaded FIRST serves the
    # process: with this library first (it resolves /opt/rocm's), a later `import torch` brings a second runtime and the
    # library's hipGetDeviceCount then finds no device.  Let torch, when installed, go first; plain C / ctypes clients
    # without torch are unaffected.
    # FMPC_NO_TORCH_PRELOAD=1 opts out (a plain ctypes / numpy client that never imports torch saves its import time and
    # keeps /opt/rocm's runtime); a torch that fails to import is reported, not hidden.
    if "torch" not in sys.modules and os.environ.get("FMPC_NO_TORCH_PRELOAD", "0") != "1":
        try:
            import torch  # noqa: F401
        except ImportError:
            pass                                           # no torch installed: nothing to order
        except Exception as e:                             # a broken install: say so, the library may then see no device
            import warnings
            warnings.warn(f"mpc-sensorlessao_amd: `import torch` failed ({e!r}); loading {LIB_PATH} with the system HIP runtime. "
                          "If a later torch import brings its own runtime the library will report FMPC_E_NO_DEVICE.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if a declared symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def strerror(code):
    try:
        return load().fmpc_strerror(int(code)).decode()
    except Exception:   # library absent: still give the number
        return f"fastmpc status {code}"
