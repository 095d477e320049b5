"""ctypes front-end of the CPU oracle (oracle/agpl_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by the product package.  See agpl_oracle.c for the reference
file:line each C function follows; the M x M Gaussian update (examples/bernoulli/script.jl:35-36
in the sparse form of docs/src/index.md:154-163) is restated here in numpy/LAPACK float64.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass, field

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libagpl_oracle.so")

BERNOULLI, NEGBINOMIAL, STUDENTT, CATEGORICAL, CATEGORICAL_BIJ, POISSON, LAPLACE, HETEROGAUSS = range(8)


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "agpl_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "libagpl_oracle.so"])
    return _SO


class _Lik(C.Structure):
    _fields_ = [("kind", C.c_int32), ("nlatent", C.c_int32), ("p", C.c_double * 4),
                ("logtheta", C.POINTER(C.c_double))]


@dataclass
class Lik:
    kind: int
    nlatent: int = 1
    p: tuple = (0.0, 0.0, 0.0, 0.0)
    logtheta: np.ndarray | None = None
    _keep: object = field(default=None, repr=False)

    def c(self) -> _Lik:
        s = _Lik()
        s.kind, s.nlatent = self.kind, self.nlatent
        pp = list(self.p) + [0.0] * (4 - len(self.p))
        for i in range(4):
            s.p[i] = float(pp[i])
        if self.logtheta is not None:
            self._keep = np.ascontiguousarray(self.logtheta, dtype=np.float64)
            s.logtheta = self._keep.ctypes.data_as(C.POINTER(C.c_double))
        return s

    def ydtype(self):
        if self.kind in (BERNOULLI, CATEGORICAL, CATEGORICAL_BIJ):
            return np.uint8
        if self.kind in (NEGBINOMIAL, POISSON):
            return np.int32
        return np.float64


def bernoulli():
    return Lik(BERNOULLI)


def negbinomial(r):
    return Lik(NEGBINOMIAL, 1, (float(r),))


def studentt(nu, sigma):
    return Lik(STUDENTT, 1, (float(nu), float(sigma)))


def categorical(logtheta, bijective=False):
    lt = np.asarray(logtheta, dtype=np.float64)
    return Lik(CATEGORICAL_BIJ if bijective else CATEGORICAL, len(lt) - (1 if bijective else 0), (), lt)


def poisson(lam):
    return Lik(POISSON, 1, (float(lam),))


def laplace(beta):
    return Lik(LAPLACE, 1, (float(beta),))


def heterogauss(lam):
    return Lik(HETEROGAUSS, 2, (float(lam),))


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        d = C.c_double
        for name, res, args in [
            ("agplo_normlogcdf", d, [d]),
            ("agplo_pg_mean", d, [d, d]),
            ("agplo_pg_logtilt", d, [d, d, d]),
            ("agplo_pg_kl", d, [d, d]),
            ("agplo_pg_a", d, [C.c_int, d]),
            ("agplo_pg_mass_texpon", d, [d, d]),
            ("agplo_pg_logpdf", d, [d, d, d]),
            ("agplo_approx_expected_logistic", d, [d, d]),
            ("agplo_approx_expected_logistic_f32", C.c_float, [C.c_float, C.c_float]),
            ("agplo_synth_fstar", d, [d]),
            ("agplo_num_threads", C.c_int, []),
            ("agplo_set_point_offset", None, [C.c_int64]),
        ]:
            f = getattr(_lib, name)
            f.restype, f.argtypes = res, args
        for name in ("agplo_logtilt", "agplo_expected_logtilt", "agplo_aux_kl", "agplo_aux_prior_logpdf",
                     "agplo_aug_loglik", "agplo_full_conditional_logpdf", "agplo_expected_aug_loglik"):
            getattr(_lib, name).restype = d
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f64(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float64)


# ---------------------------------------------------------------- scalar helpers
def pg_mean(b, c):
    return lib().agplo_pg_mean(float(b), float(c))


def pg_a(n, x):
    return lib().agplo_pg_a(int(n), float(x))


def pg_mass_texpon(z):
    K = np.pi ** 2 / 8 + z * z / 2
    return lib().agplo_pg_mass_texpon(float(z), float(K))


def pg_logpdf(b, c, x):
    return lib().agplo_pg_logpdf(float(b), float(c), float(x))


def pg_kl(b, c):
    return lib().agplo_pg_kl(float(b), float(c))


def normlogcdf(z):
    return lib().agplo_normlogcdf(float(z))


def approx_expected_logistic(mu, c, f32=False):
    if f32:
        return lib().agplo_approx_expected_logistic_f32(float(mu), float(c))
    return lib().agplo_approx_expected_logistic(float(mu), float(c))


def philox(ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    lib().agplo_philox4x32_10(c, k, o)
    return list(o)


def uniforms(seed, stream, sweep, n):
    out = np.empty(n)
    lib().agplo_uniforms(C.c_uint64(seed), C.c_uint64(stream), C.c_uint32(sweep), C.c_int64(n), _p(out))
    return out


def rand_pg(b, c, n, seed, stats=False):
    out = np.empty(n)
    nu = np.empty(n, dtype=np.uint32)
    nt = np.empty(n, dtype=np.uint32)
    lib().agplo_rand_pg_many(C.c_double(b), C.c_double(c), C.c_int64(n), C.c_uint64(seed), _p(out), _p(nu), _p(nt))
    return (out, nu, nt) if stats else out


def rand_gamma(shape, scale, n, seed):
    out = np.empty(n)
    lib().agplo_rand_gamma_many(C.c_double(shape), C.c_double(scale), C.c_int64(n), C.c_uint64(seed), _p(out))
    return out


def rand_poisson(mu, n, seed):
    out = np.empty(n, dtype=np.int64)
    lib().agplo_rand_poisson_many(C.c_double(mu), C.c_int64(n), C.c_uint64(seed), _p(out))
    return out


def rand_invgaussian(mu, lam, n, seed):
    out = np.empty(n)
    lib().agplo_rand_invgaussian_many(C.c_double(mu), C.c_double(lam), C.c_int64(n), C.c_uint64(seed), _p(out))
    return out


# ---------------------------------------------------------------- operator surface (vector level)
def _ycast(lik: Lik, y):
    return np.ascontiguousarray(y, dtype=lik.ydtype())


def aux_sample(lik: Lik, y, f, seed, sweep=0, stats=False, i0=0):
    """aux_sample! src/generic.jl:5-12.  Returns dict(omega=..., n=...).  ``i0``: global index of point 0 (the
    per-point streams are keyed on i0 + i, as agpl_ctx_set_point_offset)."""
    f = _f64(f)
    y = _ycast(lik, y)
    n = f.size // lik.nlatent
    omega = np.empty(n if lik.kind == HETEROGAUSS else f.shape)
    nn = None
    if lik.kind in (CATEGORICAL, CATEGORICAL_BIJ):
        nn = np.zeros(f.shape, dtype=np.int64)
    elif lik.kind in (POISSON, HETEROGAUSS):
        nn = np.zeros(n, dtype=np.int64)
    nu = np.zeros(n, dtype=np.uint32)
    nt = np.zeros(n, dtype=np.uint32)
    lc = lik.c()
    lib().agplo_set_point_offset(C.c_int64(i0))
    rc = lib().agplo_aux_sample(C.byref(lc), C.c_int64(n), _p(y), _p(f), _p(omega), _p(nn),
                                C.c_uint64(seed), C.c_uint32(sweep), _p(nu), _p(nt))
    lib().agplo_set_point_offset(C.c_int64(0))
    if rc != 0:
        raise ValueError(f"oracle aux_sample failed rc={rc}")
    out = {"omega": omega}
    if nn is not None:
        out["n"] = nn
    if stats:
        out["nuni"], out["nterms"] = nu, nt
    return out


def aux_posterior(lik: Lik, y, mu, var):
    """aux_posterior! -- returns (out1, out2, out3) per agplo_aux_posterior."""
    mu, var = _f64(mu), _f64(var)
    y = _ycast(lik, y)
    n = mu.size // lik.nlatent
    out1 = np.empty(n if lik.kind == HETEROGAUSS else mu.shape)
    out2 = np.empty_like(out1) if lik.kind in (CATEGORICAL, CATEGORICAL_BIJ, POISSON, HETEROGAUSS) else None
    out3 = np.empty(n) if lik.kind == HETEROGAUSS else None
    lc = lik.c()
    lib().agplo_aux_posterior(C.byref(lc), C.c_int64(n), _p(y), _p(mu), _p(var), _p(out1), _p(out2), _p(out3))
    return out1, out2, out3


def expected_potential_precision(lik: Lik, y, q1, q2=None, mu_g=None):
    y = _ycast(lik, y)
    q1, q2, mu_g = _f64(q1), _f64(q2), _f64(mu_g)
    n = q1.size if lik.kind == HETEROGAUSS else q1.size // lik.nlatent
    beta = np.empty((lik.nlatent, n))
    gamma = np.empty((lik.nlatent, n))
    lc = lik.c()
    lib().agplo_expected_potential_precision(C.byref(lc), C.c_int64(n), _p(y), _p(q1), _p(q2), _p(mu_g),
                                             _p(beta), _p(gamma))
    return beta, gamma


def potential_precision(lik: Lik, y, omega, nn=None, fg=None):
    y = _ycast(lik, y)
    omega, fg = _f64(omega), _f64(fg)
    nn = None if nn is None else np.ascontiguousarray(nn, dtype=np.int64)
    n = omega.size if lik.kind == HETEROGAUSS else omega.size // lik.nlatent
    beta = np.empty((lik.nlatent, n))
    gamma = np.empty((lik.nlatent, n))
    lc = lik.c()
    lib().agplo_potential_precision(C.byref(lc), C.c_int64(n), _p(y), _p(omega), _p(nn), _p(fg), _p(beta), _p(gamma))
    return beta, gamma


def logtilt(lik: Lik, y, omega, f, nn=None):
    y = _ycast(lik, y)
    omega, f = _f64(omega), _f64(f)
    nn = None if nn is None else np.ascontiguousarray(nn, dtype=np.int64)
    lc = lik.c()
    return lib().agplo_logtilt(C.byref(lc), C.c_int64(f.size // lik.nlatent), _p(y), _p(omega), _p(nn), _p(f))


def expected_logtilt(lik: Lik, y, q1, q2, mu, var):
    y = _ycast(lik, y)
    q1, q2, mu, var = _f64(q1), _f64(q2), _f64(mu), _f64(var)
    lc = lik.c()
    return lib().agplo_expected_logtilt(C.byref(lc), C.c_int64(mu.size // lik.nlatent), _p(y), _p(q1), _p(q2),
                                        _p(mu), _p(var))


def aux_kl(lik: Lik, y, q1, q2=None):
    y = _ycast(lik, y)
    q1, q2 = _f64(q1), _f64(q2)
    lc = lik.c()
    return lib().agplo_aux_kl(C.byref(lc), C.c_int64(q1.size // lik.nlatent), _p(y), _p(q1), _p(q2))


def _i64(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.int64)


def aux_prior_logpdf(lik: Lik, y, omega, nn=None):
    """logdensity_def(aux_prior(lik, y), Omega): the second half of aug_loglik, src/generic.jl:48-50."""
    y = _ycast(lik, y)
    omega, nn = _f64(omega), _i64(nn)
    lc = lik.c()
    return lib().agplo_aux_prior_logpdf(C.byref(lc), C.c_int64(omega.size), _p(y), _p(omega), _p(nn))


def aug_loglik(lik: Lik, y, omega, f, nn=None):
    """aug_loglik src/generic.jl:48-50; heteroscedastic: heteroscedasticgaussian.jl:106-128 (f = fg [N,2])."""
    y = _ycast(lik, y)
    omega, f, nn = _f64(omega), _f64(f), _i64(nn)
    lc = lik.c()
    return lib().agplo_aug_loglik(C.byref(lc), C.c_int64(omega.size), _p(y), _p(omega), _p(nn), _p(f))


def full_conditional_logpdf(lik: Lik, y, f, omega, nn=None):
    """logdensity_def(aux_full_conditional(lik, y, f), Omega) -- src/TestUtils.jl:110-114."""
    y = _ycast(lik, y)
    omega, f, nn = _f64(omega), _f64(f), _i64(nn)
    lc = lik.c()
    return lib().agplo_full_conditional_logpdf(C.byref(lc), C.c_int64(omega.size), _p(y), _p(f), _p(omega), _p(nn))


def expected_aug_loglik(lik: Lik, y, q1, q2, mu, var):
    """expected_aug_loglik src/generic.jl:52-54; heteroscedastic: heteroscedasticgaussian.jl:130-145."""
    y = _ycast(lik, y)
    q1, q2, mu, var = _f64(q1), _f64(q2), _f64(mu), _f64(var)
    lc = lik.c()
    return lib().agplo_expected_aug_loglik(C.byref(lc), C.c_int64(q1.size // (1 if lik.kind == HETEROGAUSS else lik.nlatent)),
                                           _p(y), _p(q1), _p(q2), _p(mu), _p(var))


# ---------------------------------------------------------------- sparse sweep
def cavi_pass(lik: Lik, Phi, kdiag, y, W, alpha, mu0=None, want_points=False):
    """One pass over the N points: marginals -> aux_posterior! -> expected potential/precision ->
    G = Phi diag(gamma) Phi', g = Phi beta.  Phi: float32 [N, M] C-order (= [M,N] column-major)."""
    Phi = np.ascontiguousarray(Phi, dtype=np.float32)
    N, M = Phi.shape
    L = lik.nlatent
    W = _f64(W).reshape(L, M, M)
    alpha = _f64(alpha).reshape(L, M)
    kdiag = _f64(kdiag)
    y = _ycast(lik, y)
    G = np.empty((L, M, M))
    g = np.empty((L, M))
    mu = np.empty((N, L)) if want_points else None
    var = np.empty((N, L)) if want_points else None
    beta = np.empty((L, N)) if want_points else None
    gamma = np.empty((L, N)) if want_points else None
    lc = lik.c()
    rc = lib().agplo_cavi_pass(C.byref(lc), C.c_int64(N), C.c_int(M), _p(Phi), _p(kdiag), _p(_f64(mu0)), _p(y),
                               _p(W), _p(alpha), _p(G), _p(g), _p(mu), _p(var), _p(beta), _p(gamma))
    if rc != 0:
        raise MemoryError("oracle cavi_pass")
    if want_points:
        return G, g, dict(mu=mu, var=var, beta=beta, gamma=gamma)
    return G, g


def accumulate(Phi, beta, gamma):
    Phi = np.ascontiguousarray(Phi, dtype=np.float32)
    N, M = Phi.shape
    beta, gamma = _f64(np.atleast_2d(beta)), _f64(np.atleast_2d(gamma))
    L = beta.shape[0]
    G = np.empty((L, M, M))
    g = np.empty((L, M))
    lib().agplo_accumulate(C.c_int64(N), C.c_int(M), C.c_int(L), _p(Phi), _p(beta), _p(gamma), _p(G), _p(g))
    return G, g


def gaussian_update(G, g, eta0=None):
    """Whitened Gaussian update (a12): Lambda_v = I + G, eta_v = g + eta0,
    S_v = Lambda_v^-1, m_v = S_v eta_v.  Returns (S_v, m_v) per latent.  numpy/LAPACK float64.
    Dense form: examples/bernoulli/script.jl:35-36; sparse form docs/src/index.md:154-163 with
    kappa = K_Z^-1 K_ZX written in the whitened basis Phi = L^-1 K_ZX (K_Z = L L')."""
    G = np.atleast_3d(np.asarray(G, dtype=np.float64))
    if G.shape[-1] != G.shape[-2]:
        G = G.reshape(-1, G.shape[0], G.shape[0])
    g = np.asarray(g, dtype=np.float64).reshape(G.shape[0], -1)
    S, m = [], []
    for l in range(G.shape[0]):
        Lam = np.eye(G.shape[1]) + G[l]
        eta = g[l] + (0.0 if eta0 is None else eta0)
        cf = np.linalg.cholesky(Lam)
        Sl = np.linalg.solve(cf.T, np.linalg.solve(cf, np.eye(G.shape[1])))
        S.append((Sl + Sl.T) / 2)
        m.append(Sl @ eta)
    return np.stack(S), np.stack(m)


def randn(seed, stream0, sweep, n):
    out = np.empty(n)
    lib().agplo_randn_many(C.c_uint64(seed), C.c_uint64(stream0), C.c_uint32(sweep), C.c_int64(n), _p(out))
    return out


def gibbs_pass(lik: Lik, Phi, kdiag, y, v, seed, sweep, mu0=None, i0=0):
    """Per-point half of a sparse Gibbs sweep + accumulation.  Returns G, g and the per-point draws.
    ``i0``: global index of point 0 (stream key = i0 + i)."""
    Phi = np.ascontiguousarray(Phi, dtype=np.float32)
    N, M = Phi.shape
    Lf = lik.nlatent
    Lo = 1 if lik.kind == HETEROGAUSS else Lf
    v = _f64(v).reshape(Lf, M)
    y = _ycast(lik, y)
    f = np.empty((N, Lf))
    omega = np.empty((N, Lo))
    nn = np.zeros((N, Lo), dtype=np.int64)
    nuni = np.zeros(N, dtype=np.uint32)
    beta = np.empty((Lf, N), dtype=np.float32)
    gamma = np.empty((Lf, N), dtype=np.float32)
    lc = lik.c()
    lib().agplo_set_point_offset(C.c_int64(i0))
    rc = lib().agplo_gibbs_points(C.byref(lc), C.c_int64(N), C.c_int(M), _p(Phi), _p(_f64(kdiag)), _p(_f64(mu0)),
                                  _p(y), _p(v), C.c_uint64(seed), C.c_uint32(sweep), _p(f), _p(omega), _p(nn),
                                  _p(nuni), _p(beta), _p(gamma))
    lib().agplo_set_point_offset(C.c_int64(0))
    if rc != 0:
        raise ValueError(f"oracle gibbs_points failed rc={rc}")
    G, g = accumulate(Phi, beta.astype(np.float64), gamma.astype(np.float64))
    return G, g, dict(f=f, omega=omega, n=nn, nuni=nuni, beta=beta, gamma=gamma)


def gibbs_draw_v(G, g, seed, sweep, eta0=None):
    """v ~ N(m, S), S = (I + G)^-1, m = S (g + eta0); z from Philox streams (seed, l*M + a, sweep | 2^31).
    numpy/LAPACK float64 (examples/bernoulli/script.jl:82-84 for the inducing coordinates)."""
    import scipy.linalg as sla

    G = np.asarray(G, dtype=np.float64)
    L, M = G.shape[0], G.shape[1]
    g = np.asarray(g, dtype=np.float64).reshape(L, M)
    z = randn(seed, 0, (sweep | 0x80000000) & 0xFFFFFFFF, L * M).reshape(L, M)
    v, m = np.empty((L, M)), np.empty((L, M))
    for l in range(L):
        Cf = np.linalg.cholesky(np.eye(M) + G[l])
        rhs = g[l] + (0.0 if eta0 is None else np.asarray(eta0).reshape(L, M)[l])
        m[l] = sla.cho_solve((Cf, True), rhs)
        v[l] = m[l] + sla.solve_triangular(Cf.T, z[l], lower=False)
    return v, m


def dense_gibbs_step(lik: Lik, K, Lk, y, f, seed, sweep, mu0=None):
    """One full-rank Gibbs step (examples/bernoulli/script.jl:81-84) with the device's evaluation order:
    B = I + D^1/2 K D^1/2, f = f0 + K D^1/2 B^-1 (beta / sqrt(gamma) - sqrt(gamma) f0 - z2), f0 = mu0 + L_K z1;
    z from Philox streams (seed, 0..2N-1, sweep | 2^31).  numpy/LAPACK float64.  Returns (f_new, draw dict)."""
    import scipy.linalg as sla

    K = np.asarray(K, dtype=np.float64)
    N = K.shape[0]
    d = aux_sample(lik, y, f, seed=seed, sweep=sweep)
    beta, gamma = potential_precision(lik, y, d["omega"], d.get("n"))
    beta, gamma = beta[0], gamma[0]
    z = randn(seed, 0, (sweep | 0x80000000) & 0xFFFFFFFF, 2 * N)
    f0 = Lk @ z[:N] + (0.0 if mu0 is None else mu0)
    sg = np.sqrt(gamma)
    # gamma_i = 0 (Poisson: y_i = 0 and a drawn n_i = 0 give omega_i = PG(0, c) = 0) implies beta_i = 0: row i of B is
    # e_i and D^1/2 zeroes the component on output, so 0 is the exact value of beta / sqrt(gamma) there
    with np.errstate(divide="ignore", invalid="ignore"):
        r = np.where(sg > 0, beta / sg, 0.0) - sg * f0 - z[N:]
    B = np.eye(N) + sg[:, None] * K * sg[None, :]
    s = sla.cho_solve(sla.cho_factor(B, lower=True), r)
    return f0 + K @ (sg * s), d


def dense_conditional(K, beta, gamma, mu0=None):
    """The reference's literal formulas (examples/bernoulli/script.jl:82-83): Sigma = inv(inv(K) + Diag(gamma)),
    mu = Sigma (beta + K \\ mu0)."""
    K = np.asarray(K, dtype=np.float64)
    Sigma = np.linalg.inv(np.linalg.inv(K) + np.diag(gamma))
    rhs = beta + (0.0 if mu0 is None else np.linalg.solve(K, mu0))
    return Sigma @ rhs, (Sigma + Sigma.T) / 2


# ---------------------------------------------------------------- synthetic workload
def synth_x(seed, i0, n):
    x = np.empty(n)
    lib().agplo_synth_x(C.c_uint64(seed), C.c_int64(i0), C.c_int64(n), _p(x))
    return x


def synth_y(lik: Lik, seed, i0, n):
    shape = (n, lik.nlatent) if lik.kind in (CATEGORICAL, CATEGORICAL_BIJ) else n
    y = np.empty(shape, dtype=lik.ydtype())
    lc = lik.c()
    lib().agplo_synth_y(C.byref(lc), C.c_uint64(seed), C.c_int64(i0), C.c_int64(n), _p(y))
    return y


def se_kernel_f32(x, z, ell):
    x, z = _f64(x), _f64(z)
    out = np.empty((x.size, z.size), dtype=np.float32)
    lib().agplo_se_kernel_f32(C.c_int64(x.size), C.c_int(z.size), _p(x), _p(z), C.c_double(ell), _p(out))
    return out


def num_threads():
    return lib().agplo_num_threads()
