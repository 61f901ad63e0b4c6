"""ctypes binding of libmisti_hip.so (C ABI: include/misti_hip.h).

The library is the only compute path.  If it is missing or no HIP device is
usable, every compute entry point raises - there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os

from . import build as _build

CPFIT, TRUE_EPS, SMOOTH, UNFOLDED = 1, 2, 4, 8
HINT_INTEGER_SPLITS = 1
STATUS_TEXT = {
    0: "ok",
    1: "Hit negative value of migration rate",
    2: "Lambda correction failed",
    3: "Infinite coalescent time (two populations in the last interval)",
    4: "Split time / band / pulse structure invalid for this candidate",
    5: "Non-finite intermediate or iteration cap",
    6: "Stiff interval: the contour solver (rate x length > 96) did not converge",
}
MAX_BANDS, MAX_PULSES, MAX_PARAMS, MAX_NUMT = 8, 8, 16, 255
LANE_ANY, MAX_LANES = -1, 64
ABI_VERSION = 6
TRACE_MAX_CAND, TRACE_MAX_ITER = 64, 200


class Band(C.Structure):
    _fields_ = [("pop", C.c_int32), ("start", C.c_int32), ("end", C.c_int32), ("param", C.c_int32), ("value", C.c_double)]


class Pulse(C.Structure):
    _fields_ = [("pop", C.c_int32), ("time", C.c_int32), ("param", C.c_int32), ("_pad", C.c_int32), ("value", C.c_double)]


class Model(C.Structure):
    _fields_ = [("numT", C.c_int32), ("sample_date", C.c_int32), ("flags", C.c_uint32), ("n_band", C.c_int32),
                ("n_pulse", C.c_int32), ("n_param", C.c_int32), ("mixture_th", C.c_double),
                ("times", C.POINTER(C.c_double)), ("lh", C.POINTER(C.c_double)),
                ("bands", C.POINTER(Band)), ("pulses", C.POINTER(Pulse))]


class MistiError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libmisti_hip error %d: %s" % (code, msg))
        self.code = code


_PD, _PI = C.POINTER(C.c_double), C.POINTER(C.c_int32)
# every symbol include/misti_hip.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "misti_abi_version": (C.c_int, []),
    "misti_last_error": (C.c_char_p, []),
    "misti_build_id": (C.c_char_p, []),
    "misti_device_count": (C.c_int, []),
    "misti_create": (C.c_int, [C.POINTER(Model), C.c_int, C.POINTER(C.c_void_p)]),
    "misti_destroy": (C.c_int, [C.c_void_p]),
    "misti_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "misti_get_stream": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "misti_sync": (C.c_int, [C.c_void_p]),
    "misti_set_hints": (C.c_int, [C.c_void_p, C.c_uint32]),
    "misti_eval_batch": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "misti_eval_batch_dev": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "misti_nm_solve": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_double, C.c_void_p, C.c_double, C.c_double, C.c_int32,
                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "misti_nm_last_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    "misti_nm_last_spec_iterations": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    "misti_basinhopping": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_double, C.c_void_p, C.c_int32, C.c_double, C.c_double, C.c_int32,
                                     C.c_double, C.c_double, C.c_double, C.c_double, C.c_int32, C.c_int64, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "misti_enable_solver_trace": (C.c_int, [C.c_void_p, C.c_int]),
    "misti_last_solver_trace": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]),
    "misti_llk_dev": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "misti_forward_rates": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "misti_forward_rates_dev": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "misti_argmax_dev": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "misti_last_diag": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p]),
    "misti_enable_timing": (C.c_int, [C.c_void_p, C.c_int]),
    "misti_kernel_times": (C.c_int, [C.c_void_p, _PD, C.POINTER(C.c_int64), C.c_int]),
    "misti_tables": (C.c_int, [_PI, _PI]),
    "misti_create_lanes": (C.c_int, [C.POINTER(Model), C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "misti_destroy_lanes": (C.c_int, [C.c_void_p]),
    "misti_lanes_size": (C.c_int, [C.c_void_p]),
    "misti_lanes_context": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]),
    "misti_lanes_set_hints": (C.c_int, [C.c_void_p, C.c_uint32]),
    "misti_lanes_eval_batch_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]),
    "misti_lanes_wait": (C.c_int, [C.c_void_p, C.c_int]),
    "misti_lanes_sync": (C.c_int, [C.c_void_p]),
    "misti_lanes_busy": (C.c_int, [C.c_void_p, C.c_int]),
    "misti_create_multi": (C.c_int, [C.POINTER(Model), C.c_int, _PI, C.POINTER(C.c_void_p)]),
    "misti_destroy_multi": (C.c_int, [C.c_void_p]),
    "misti_multi_size": (C.c_int, [C.c_void_p]),
    "misti_multi_context": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int)]),
    "misti_multi_eval_batch": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "misti_multi_last_shards": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "misti_multi_last_cost": (C.c_int, [C.c_void_p, _PD]),
    "misti_multi_eval_batch_dev": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.c_int64, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                             C.c_int64, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "misti_multi_sync": (C.c_int, [C.c_void_p]),
    "misti_multi_nm_solve": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_double, C.c_void_p, C.c_double, C.c_double, C.c_int32,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "misti_multi_basinhopping": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_double, C.c_void_p, C.c_int32, C.c_double, C.c_double, C.c_int32,
                                           C.c_double, C.c_double, C.c_double, C.c_double, C.c_int32, C.c_int64, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
}

_lib = None


def lib_path():
    return _build.LIB


def load(build_if_missing=True):
    """Load (building first if needed) the shared library and bind every symbol."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if os.environ.get("MISTI_LIB"):
        path = os.environ["MISTI_LIB"]          # an explicitly chosen build (kernel experiments); never rebuilt
    elif build_if_missing and _build.have_hipcc():
        _build.build()                           # no-op when the library is newer than every source and header
    elif not os.path.exists(path):
        raise MistiError(-3, "libmisti_hip.so is not built (%s) and hipcc is not available; run misti_amd.build.build()" % path)
    elif _build.stale():
        import warnings
        warnings.warn("libmisti_hip.so is older than its sources and hipcc is not available to rebuild it", RuntimeWarning)
    # One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64 / libhsa-runtime64; if this
    # library is loaded first it binds the system ROCm runtime and a later `import torch` brings a second one into
    # the process (observed: torch.cuda then intermittently reports "No HIP GPUs are available", and stream or
    # memory handles cross runtimes).  Loaded after torch, the NEEDED libamdhip64.so.7 resolves to the runtime
    # torch already mapped.  Without torch installed the system runtime is the only one.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(path)
    # MISTI_LIB alone keeps every check: an explicitly chosen build must still be THIS ABI (ctypes signatures of another version mean
    # memory corruption, not an error).  Only the A/B tools (tools/ab_compare.py, tools/stamp_run.py), which load older builds on
    # purpose and call nothing whose signature changed, set MISTI_LIB_AB=1 to relax them.
    experimental = bool(os.environ.get("MISTI_LIB")) and os.environ.get("MISTI_LIB_AB") == "1"
    for name, (res, args) in SYMBOLS.items():
        try:
            fn = getattr(lib, name)      # AttributeError if the .so lacks a declared symbol
        except AttributeError:
            if experimental:             # an older build loaded for an A/B measurement: its missing entry points simply cannot be called
                continue
            raise
        fn.restype = res
        fn.argtypes = args
    if lib.misti_abi_version() != ABI_VERSION and not experimental:
        raise MistiError(-1, "ABI version mismatch: %s is version %d, this binding expects %d (rebuild: python -m misti_amd.build --force)"
                         % (path, lib.misti_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def build_id():
    """The loaded library's own record of the sources it was built from (``misti_build_id``)."""
    return load().misti_build_id().decode()


def check(code):
    if code != 0:
        raise MistiError(code, load().misti_last_error().decode("utf-8", "replace"))
