"""Chebyshev fit of r(z) = mass_texpon(z, pi^2/8 + z^2/2) (polyagamma.jl:179-192, 226-236) on z in [0, 8], used by the device
sampler ONLY to bracket r: a draw whose branch uniform u is within +-1e-8 of the fit is decided by the exact formula
(Pg1Params::set), every other draw by the fit -- the decisions are those of the exact formula as long as the fit's error is
below the bracket.  This script prints the coefficients pasted into agpl_random.h (the AGPL_PG_MASS_CHEB macro behind kPgMassChebLit / kPgMassChebMem) and the error of the fit against
a 60-digit mpmath evaluation on a dense grid."""
import mpmath as mp
import numpy as np
from numpy.polynomial import chebyshev as Ch

mp.mp.dps = 60
T = mp.mpf("0.64")
LO, HI, DEG = 0.0, 8.0, 24


def r_exact(z):
    z = mp.mpf(z)
    K = mp.pi**2 / 8 + z * z / 2
    b = mp.sqrt(1 / T) * (T * z - 1)
    a = -mp.sqrt(1 / T) * (T * z + 1)
    q = 4 / mp.pi * K * mp.exp(K * T) * (mp.exp(-z) * mp.ncdf(b) + mp.exp(z) * mp.ncdf(a))
    return 1 / (1 + q)


k = np.arange(DEG + 1)
x = np.cos(np.pi * (k + 0.5) / (DEG + 1))
z = (x + 1) / 2 * (HI - LO) + LO
c = Ch.chebfit(x, np.array([float(r_exact(v)) for v in z]), DEG)
zz = np.linspace(LO, HI, 40001)
ref = np.array([float(r_exact(v)) for v in zz])


def clenshaw(zv):  # the device's evaluation order
    xv = (zv - LO) * (2.0 / (HI - LO)) - 1.0
    b1 = np.zeros_like(xv)
    b2 = np.zeros_like(xv)
    for j in range(DEG, 0, -1):
        b1, b2 = c[j] + 2.0 * xv * b1 - b2, b1
    return c[0] + xv * b1 - b2


err = np.abs(clenshaw(zz) - ref).max()
print("// degree", DEG, "on [0, 8]; max |fit - exact| on 40001 points (exact: mpmath, 60 digits) =", f"{err:.3e}")
print("// r(0) exact =", mp.nstr(r_exact(0), 20), " reference's constant 0.5776972428360435")
print("constexpr double kPgMassCheb[%d] = {" % (DEG + 1))
for i in range(0, DEG + 1, 3):
    print("    " + ", ".join(repr(float(v)) for v in c[i:i + 3]) + ",")
print("};")
