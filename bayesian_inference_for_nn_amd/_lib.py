"""ctypes binding of include/pyz.h.  There is no CPU fallback: if libpyz.so cannot
be loaded every call raises."""

from __future__ import annotations

import ctypes as C
import os

from . import _build

ACT = {"linear": 0, "relu": 1, "tanh": 2, "sigmoid": 3, "softmax": 4}
LOSS = {"scce": 0, "mse": 1}
SWEEP = {"gauss_seidel": 0, "jacobi": 1}
GAMMA_MEDIAN = -1.0       # PYZ_SVGD_GAMMA_MEDIAN

STREAM_SGLD, STREAM_BBB, STREAM_HMC, STREAM_INIT, STREAM_PREDICT = 0, 1, 2, 3, 4


class PyzError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"pyz error {code}: {msg}")
        self.code = code


_p = C.c_void_p
_i32, _i64, _u32, _u64, _f = C.c_int32, C.c_int64, C.c_uint32, C.c_uint64, C.c_float

# name -> (restype, argtypes); mirrors include/pyz.h exactly
SIGNATURES = {
    "pyz_version": (C.c_int, []),
    "pyz_last_error": (C.c_char_p, []),
    "pyz_device_count": (C.c_int, []),
    "pyz_mlp_create": (C.c_int, [C.c_int, C.POINTER(_i32), C.POINTER(_i32), C.c_int, C.c_int, C.c_int, C.POINTER(_p)]),
    "pyz_mlp_destroy": (C.c_int, [_p]),
    "pyz_mlp_param_count": (_i64, [_p]),
    "pyz_mlp_workspace_bytes": (_i64, [_p]),
    "pyz_mlp_forward": (C.c_int, [_p, _p, C.c_int, _p, _p, C.c_int, _p, _p]),
    "pyz_mlp_loss_grad": (C.c_int, [_p, _p, C.c_int, _p, _p, _p, C.c_int, _p, _p, _p]),
    "pyz_sgd_step": (C.c_int, [_p, _p, _p, _p, _p, C.c_int, _f, _p, _p]),
    "pyz_swag_step": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _p, C.c_int, _f, _i64, C.c_int, _p, _p]),
    "pyz_sgld_step": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, C.c_int, _f, _i64, _u64, _p, _p, _p]),
    "pyz_sgld_run": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, C.POINTER(_i32), C.POINTER(_f), C.c_int, _i64, _i64, _u64,
                               _p, C.c_int, _p]),
    "pyz_sgd_run": (C.c_int, [_p, _p, _p, _p, _p, C.POINTER(_i32), C.POINTER(_f), C.c_int, _i64, _p, C.c_int, _p]),
    "pyz_swag_run": (C.c_int, [_p, _p, _p, _p, _p, C.c_int, C.c_int, _p, _p, _p, C.POINTER(_i32), C.POINTER(_f), C.c_int,
                               _i64, _i64, _p, C.c_int, _p]),
    "pyz_last_run_info": (C.c_int, [_p, C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i32)]),
    "pyz_check_finite": (C.c_int, [_p, _p]),
    "pyz_probe_begin": (C.c_int, [C.c_int]),
    "pyz_probe_end": (C.c_int, [_p, C.POINTER(_f), C.c_char_p, C.c_int, C.POINTER(C.c_int)]),
    "pyz_bbb_step": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, C.c_int, _f, _f, _f, _f, _p, _p, _i64, _u64, _p, _p, _p]),
    "pyz_bbb_run": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, C.POINTER(_i32), C.POINTER(_f), C.c_int, _f, _f, _f, _p, _p, _i64,
                              _i64, _u64, _p, _p, _p, _p, C.c_int, _p, C.c_int, _p]),
    "pyz_hmc_step": (C.c_int, [_p, _p, C.c_int, _p, _p, C.c_int, C.c_int, _f, _f, _f, _f, _p, _p, C.c_int, C.POINTER(_f),
                               _i64, _u64, _p, _p, _p]),
    "pyz_svgd_step": (C.c_int, [_p, _p, C.c_int, _p, C.c_int, C.c_int, _p, _p, _p, _p, _p, C.c_int, _f, _f, _i64,
                                C.c_int, _p, _p]),
    "pyz_svgd_gradients": (C.c_int, [_p, _p, C.c_int, _p, _p, _p, C.c_int, _p]),
    "pyz_svgd_sweep": (C.c_int, [_p, _p, C.c_int, _p, C.c_int, C.c_int, _p, _p, _f, _f, _i64, C.c_int, _p, _p]),
    "pyz_svgd_kernel_matrix": (C.c_int, [_p, _p, C.c_int, C.c_int, C.c_int, _f, _p]),
    "pyz_svgd_gram_groups": (C.c_int, [_p, _p, C.c_int, C.c_int, C.c_int, _p, _p]),
    "pyz_svgd_kernel_matrix_groups": (C.c_int, [_p, _p, _p, C.c_int, C.c_int, C.c_int, _f, _p]),
    "pyz_svgd_combine": (C.c_int, [_p, _p, C.c_int, _p, C.c_int, C.c_int, _p, _p, _f, _f, _i64, _p, _p]),
    "pyz_predict": (C.c_int, [_p, _p, C.c_int, _p, C.c_int, _p, _p, _p]),
    "pyz_sample_normal_rows": (C.c_int, [_p, _i64, _i64, _i64, _i64, _p, _p, _u64, _u32, _u32, _p]),
    "pyz_fill_normal": (C.c_int, [_p, _i64, _u64, _u32, _u32, _f, _f, _p]),
    "pyz_debug_stamps": (C.c_int, [C.POINTER(C.c_uint64), _i64]),
    "pyz_debug_mfma_f64_layout": (C.c_int, [C.POINTER(C.c_int32)]),
    "pyz_bench_dense_kernel": (C.c_int, [_p, C.c_int, C.c_int, _p, C.c_int, _p, _p, C.c_int, _p, C.c_int, _p]),
    "pyz_malloc": (C.c_int, [C.c_size_t, C.POINTER(_p)]),
    "pyz_free": (C.c_int, [_p]),
    "pyz_upload": (C.c_int, [_p, _p, C.c_size_t, _p]),
    "pyz_download": (C.c_int, [_p, _p, C.c_size_t, _p]),
    "pyz_sync": (C.c_int, [_p]),
    "pyz_wait_flags": (C.c_int, [_p, C.c_int, C.c_uint64, C.c_int, _p, _p]),
}

E_NAN = -6


def header_version() -> int:
    """PYZ_VERSION of include/pyz.h: a library built from another header is refused."""
    import re
    text = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "include", "pyz.h")).read()
    return int(re.search(r"#define\s+PYZ_VERSION\s+(\d+)", text).group(1))

_lib = None


def lib_path() -> str:
    return _build.LIB


def load():
    """Load (building in-tree first if the sources are newer) and type the library."""
    global _lib
    if _lib is not None:
        return _lib
    # PYZ_LIB_OVERRIDE: diagnostics only (A/B runs of an alternative build of the SAME library)
    path = os.environ.get("PYZ_LIB_OVERRIDE")
    if not path:
        try:
            path = _build.build()
        except _build.NoCompiler as e:
            # a box without hipcc can only use the library that travelled with the tree; a compile ERROR is
            # never papered over with an older build (its signatures may differ from the header's)
            if not os.path.exists(_build.LIB):
                raise RuntimeError(
                    "bayesian_inference_for_nn_amd: the HIP library csrc/libpyz.so is missing and could not be built "
                    f"({e}); there is no CPU fallback") from e
            path = _build.LIB
    # torch must be loaded first: libpyz.so then binds to the SAME libamdhip64 instance torch
    # uses, so that stream handles and device pointers are interchangeable.
    import torch  # noqa: F401
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)       # AttributeError if the header and the library disagree
        fn.restype = res
        fn.argtypes = args
    if lib.pyz_version() != header_version():
        raise RuntimeError(f"bayesian_inference_for_nn_amd: {path} reports version {lib.pyz_version()}, include/pyz.h "
                           f"declares {header_version()}: stale build, refusing to call it")
    _lib = lib
    return lib


def check(rc: int):
    if rc != 0:
        msg = load().pyz_last_error()
        raise PyzError(rc, msg.decode() if msg else "")


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())
