"""The device sampler decides the branch test `r > u` of sample_pg1 (polyagamma.jl:238) through a bracket of
r(z) = mass_texpon(z, pi^2/8 + z^2/2): a degree-24 Chebyshev fit on z in [0, 8] widened by kPgMassSlack = 1e-8 (agpl_random.h).
The decisions equal the exact formula's as long as |fit - r| < slack: checked here against a 50-digit evaluation of the
reference's formula (polyagamma.jl:179-192) with the coefficients parsed from the header, in the device's evaluation order."""
import os
import re

import numpy as np
import pytest

mp = pytest.importorskip("mpmath")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "augmentedgplikelihoods.jl_amd", "csrc", "agpl_random.h")


def header_constants():
    src = open(HDR).read()
    # the coefficients are ONE macro (AGPL_PG_MASS_CHEB) that initialises both tables: the literal one and the constant-memory one
    body = re.search(r"#define AGPL_PG_MASS_CHEB(.*?)\nconstexpr double kPgMassChebLit\[25\] = \{AGPL_PG_MASS_CHEB\};", src, flags=re.S).group(1)
    assert "__constant__ double kPgMassChebMem[25] = {AGPL_PG_MASS_CHEB};" in src
    coef = np.array([float(v) for v in re.findall(r"[-+]?\d\.\d+(?:e[-+]?\d+)?", body)])
    assert coef.size == 25
    slack = float(re.search(r"constexpr double kPgMassSlack = ([0-9.e-]+);", src).group(1))
    return coef, slack


def device_fit(z, c):  # pg_mass_fit: Clenshaw on x = z / 4 - 1
    x = z * 0.25 - 1.0
    x2 = 2.0 * x
    b1 = np.zeros_like(z)
    b2 = np.zeros_like(z)
    for j in range(24, 0, -1):
        b1, b2 = x2 * b1 + (c[j] - b2), b1
    return x * b1 + (c[0] - b2)


def r_exact(z):
    mp.mp.dps = 50
    t = mp.mpf("0.64")
    z = mp.mpf(float(z))
    K = mp.pi**2 / 8 + z * z / 2
    b = mp.sqrt(1 / t) * (t * z - 1)
    a = -mp.sqrt(1 / t) * (t * z + 1)
    x0 = mp.log(K) + K * t
    q = 4 / mp.pi * (mp.exp(x0 - z + mp.log(mp.ncdf(b))) + mp.exp(x0 + z + mp.log(mp.ncdf(a))))
    return 1 / (1 + q)


def test_bracket_contains_the_exact_mass():
    c, slack = header_constants()
    assert len(c) == 25 and slack == 1e-8
    rng = np.random.default_rng(5)
    z = np.concatenate([np.linspace(0.0, 8.0, 1601), rng.uniform(0.0, 8.0, 400), [1e-12, 7.999999999]])
    fit = device_fit(z, c)
    ref = np.array([float(r_exact(v)) for v in z])
    err = np.abs(fit - ref).max()
    assert err < slack / 50, err  # 8e-11 by construction (tools/fit_pg_mass.py): two orders below the bracket's half-width
    # the reference's hard-coded r(0) (polyagamma.jl:231) sits inside the bracket at z = 0 as well
    assert abs(device_fit(np.array([0.0]), c)[0] - 0.5776972428360435) < slack / 50
