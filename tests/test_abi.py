"""CPU-side checks of the drop-in boundary: libagpl.so builds for gfx950, loads, and exports exactly the
symbols include/agpl.h declares (no compute is launched here -- there is no GPU in the build container)."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g

    g.build()
    import agpl_amd

    return agpl_amd


def _header_symbols():
    src = open(os.path.join(ROOT, "include", "agpl.h")).read()
    return sorted(set(re.findall(r"^AGPL_API [\w \*]+?\b(agpl_\w+)\(", src, flags=re.M)))


def test_header_declares_the_bound_symbols(built):
    from agpl_amd import _ffi

    assert _header_symbols() == sorted(_ffi.SYMBOLS)


def test_library_exports_every_declared_symbol(built):
    from agpl_amd import _ffi

    lib = ctypes.CDLL(_ffi.LIB_PATH)
    for s in _header_symbols():
        assert hasattr(lib, s), s
    out = subprocess.check_output(["nm", "-D", "--defined-only", _ffi.LIB_PATH]).decode()
    exported = sorted(set(re.findall(r" T (agpl_\w+)", out)))
    assert exported == _header_symbols()  # nothing else leaks out of the library
    assert lib.agpl_version() == 121


def test_library_contains_gfx950_code_object(built):
    from agpl_amd import _ffi

    blob = open(_ffi.LIB_PATH, "rb").read()
    assert b"gfx950" in blob
    assert b"gfx942" not in blob and b"sm_" not in blob  # single target, no dual paths


def test_lik_desc_layout_matches_oracle_mirror(built, oracle):
    # the oracle keeps an independent mirror of agpl_lik_desc; both must agree on layout and enum values
    from agpl_amd import _ffi, likelihoods as LK
    from oracle.oracle import _Lik

    assert ctypes.sizeof(_ffi.LikDesc) == ctypes.sizeof(_Lik) == 48
    for f in ("kind", "nlatent", "p", "logtheta"):
        assert getattr(_ffi.LikDesc, f).offset == getattr(_Lik, f).offset
    assert (LK.KIND_BERNOULLI, LK.KIND_NEGBINOMIAL, LK.KIND_STUDENTT, LK.KIND_CATEGORICAL, LK.KIND_CATEGORICAL_BIJ,
            LK.KIND_POISSON, LK.KIND_LAPLACE, LK.KIND_HETEROGAUSS) == (
        oracle.BERNOULLI, oracle.NEGBINOMIAL, oracle.STUDENTT, oracle.CATEGORICAL, oracle.CATEGORICAL_BIJ,
        oracle.POISSON, oracle.LAPLACE, oracle.HETEROGAUSS)
    hdr = open(os.path.join(ROOT, "include", "agpl.h")).read()
    for name, val in (("BERNOULLI_LOGISTIC", 0), ("NEGBINOMIAL", 1), ("STUDENTT", 2), ("CATEGORICAL", 3),
                      ("CATEGORICAL_BIJ", 4), ("POISSON", 5), ("LAPLACE", 6), ("HETEROGAUSS", 7)):
        assert re.search(rf"AGPL_LIK_{name} = {val}\b", hdr)


def test_no_gpu_means_loud_failure(built):
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import agpl_amd as A

    with pytest.raises(RuntimeError):
        A.Context(0)
    h = ctypes.c_void_p()
    from agpl_amd import _ffi

    assert _ffi.lib().agpl_ctx_create(ctypes.byref(h), 0, ctypes.c_uint64(0)) != 0  # no device: error code, no crash
    tf = ctypes.c_double()
    for probe in (lambda: _ffi.lib().agpl_probe_mfma(None, 1, 16, 0, 1, ctypes.byref(tf), None),
                  lambda: _ffi.lib().agpl_probe_mfma(None, 0, 16, 0, 1, ctypes.byref(tf), None)):
        assert probe() == -1  # AGPL_ERR_INVALID_ARGUMENT on a null context, before anything touches a device


def test_product_never_touches_the_oracle():
    pkg = os.path.join(ROOT, "augmentedgplikelihoods.jl_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp")) or f == "Makefile":
                assert "oracle" not in open(os.path.join(dp, f)).read().lower(), (dp, f)


def test_plan_takes_any_feature_count(built):
    """agpl_plan_bytes is host arithmetic (no device call): a plan exists for every feature count M >= 1 -- the library pads to the next
    multiple of 256 itself (round 6) -- and a padded count costs exactly the M-sized staging of (G, g, eta0, v) on top of the plan
    of the padded size."""
    import ctypes as C

    from agpl_amd import _ffi

    lib = _ffi.lib()
    nbytes = lambda N, M, L, fl=0: lib.agpl_plan_bytes(C.c_int64(N), C.c_int32(M), C.c_int32(L), C.c_uint32(fl))
    for M in (1, 37, 64, 200, 256, 1000, 1024, 1280, 2048):
        b = nbytes(100_000, M, 1)
        Mp = (M + 255) // 256 * 256
        assert b > 0, M
        stage = 8 * (Mp * Mp + 3 * Mp)
        assert b == nbytes(100_000, Mp, 1) + (0 if M == Mp else (stage + 255) // 256 * 256), M
    assert nbytes(100_000, 0, 1) == 0 and nbytes(0, 256, 1) == 0 and nbytes(100_000, 256, 65) == 0 and nbytes(100_000, 256, 1, 2) == 0
    assert 0 < nbytes(100_000, 200, 3, 1) < nbytes(100_000, 200, 3, 0)  # AGPL_PLAN_NO_MARGINALS
