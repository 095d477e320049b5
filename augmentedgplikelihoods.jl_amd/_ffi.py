"""ctypes binding of libagpl.so (include/agpl.h).  No fallback: if the HIP library is missing or fails to
load, every operator raises -- there is no CPU path in the product."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libagpl.so")
CSRC = os.path.join(_HERE, "csrc")

AGPL_OK = 0
ERR_INVALID_ARGUMENT, ERR_DOMAIN, ERR_UNSUPPORTED, ERR_HIP, ERR_NOT_POSDEF, ERR_OOM = -1, -2, -3, -4, -5, -6
F32, F64 = 0, 1

# exported symbols of include/agpl.h (tests/test_abi.py checks this list against the header and the .so)
SYMBOLS = [
    "agpl_version", "agpl_ctx_create", "agpl_ctx_destroy", "agpl_ctx_set_stream", "agpl_ctx_set_seed", "agpl_ctx_set_point_offset",
    "agpl_ctx_synchronize", "agpl_last_error",
    # the operator surface of src/AugmentedGPLikelihoods.jl:18-30
    "agpl_aux_sample", "agpl_rand_polyagamma", "agpl_potential_precision", "agpl_aux_posterior",
    "agpl_expected_potential_precision", "agpl_logtilt", "agpl_aux_prior_logpdf", "agpl_aug_loglik", "agpl_expected_logtilt",
    "agpl_aux_kldivergence", "agpl_expected_aug_loglik",
    # the sweep at SURVEY.md 8(d)'s float32-input arithmetic
    "agpl_marginals", "agpl_accumulate", "agpl_gaussian_update", "agpl_cavi_pass", "agpl_gibbs_pass", "agpl_gibbs_draw_v",
    # the shipped sweep: factor-form update + the plan (split-float16 images)
    "agpl_gaussian_factor", "agpl_feature_residual", "agpl_plan_bytes", "agpl_plan_create", "agpl_plan_destroy", "agpl_plan_info",
    "agpl_plan_state", "agpl_cavi_pass_plan", "agpl_plan_update", "agpl_marginals_plan", "agpl_gibbs_pass_plan",
    # full-rank Gibbs step, multi-GPU exchange, features / synthetic data, diagnostics
    "agpl_dense_cholesky", "agpl_dense_gibbs_step", "agpl_allreduce_nat", "agpl_se_features", "agpl_transform_features",
    "agpl_synth_xy", "agpl_probe_mfma", "agpl_timing", "agpl_debug_force_factor_rescue",
]


class LikDesc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("nlatent", C.c_int32), ("p", C.c_double * 4),
                ("logtheta", C.POINTER(C.c_double))]


class AGPLError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libagpl status {code}: {msg}")
        self.code = code


class ArgumentError(AGPLError, ValueError):  # Julia ArgumentError
    pass


class DomainError(AGPLError, ValueError):  # Julia DomainError
    pass


class PosDefException(AGPLError, ArithmeticError):  # LinearAlgebra.PosDefException
    pass


_ERR_TYPES = {ERR_INVALID_ARGUMENT: ArgumentError, ERR_DOMAIN: DomainError, ERR_NOT_POSDEF: PosDefException}


def build(force: bool = False) -> str:
    """Compile libagpl.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))]
    srcs.append(os.path.join(os.path.dirname(_HERE), "include", "agpl.h"))
    stale = not os.path.exists(LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", CSRC, "-s", "-j4"])
    return LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: the HIP extension has not been built "
                "(run `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")
        # torch bundles its own libamdhip64 / librocblas: load torch first so that libagpl.so's NEEDED
        # entries resolve (by SONAME) to the runtime already in the process.  Two HIP runtimes in one
        # process do not share a device context (hipGetDeviceCount fails in the second one).
        import torch  # noqa: F401

        _lib = C.CDLL(LIB_PATH)
        _lib.agpl_last_error.restype = C.c_char_p
        _lib.agpl_last_error.argtypes = [C.c_void_p]
        _lib.agpl_plan_bytes.restype = C.c_int64
        for s in SYMBOLS:
            getattr(_lib, s)  # raises AttributeError if the library does not export the ABI
    return _lib


def check(ctx_handle, rc):
    if rc != AGPL_OK:
        msg = lib().agpl_last_error(ctx_handle)
        msg = msg.decode() if msg else ""
        raise _ERR_TYPES.get(rc, AGPLError)(rc, msg)
