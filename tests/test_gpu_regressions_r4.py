"""Regression tests of the round-4 review findings.

1. The ELBO reductions (run_reduction, agpl_ops.hip) keep their partial sums in the first 8 KB of the context's small
   scratch; bytes 8192.. of it are the marginal kernel's item queues and the factor kernel's hand-off flags, which must
   read zero between launches.  With more than 1016 partials (n above ~260 000) the reduction wrote into the queues and
   the NEXT sweep was silently wrong.  Test: elbo() at N = 320 000 between sweeps changes nothing, bit for bit, and the
   natural parameters still meet the 1e-5 bar against the oracle (docs/src/index.md:154-163).
2. The blocked Cholesky's tile routine stages rows in pairs; with an odd matrix order the pair that straddles the last row
   put row N - 2 into row N - 1's slot (the `_chol_cov(fz)` of examples/bernoulli/script.jl:77 for odd N >= 8192).
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

SEED = 20240807
NAT_TOL = 1e-5  # north_star: posterior natural parameters within 1e-5 relative


@pytest.fixture(scope="module")
def A():
    import agpl_amd

    return agpl_amd


@pytest.fixture(scope="module")
def oracle():
    from oracle import oracle as O

    return O


def host(t):
    return t.detach().cpu().numpy()


def relmax(a, b):
    return float(np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(np.asarray(b)).max(), 1e-300))


def _svgp(A, ctx, lik, N, M, pad=256):
    x, y = A.synth_xy(lik, SEED, 0, N, ctx=ctx)
    z = np.linspace(-10, 10, M)
    ell = 1.5 * (z[1] - z[0])
    Kzz = np.exp(-0.5 * ((z[:, None] - z[None, :]) / ell) ** 2)
    _, Linv = A.sparse.whitening_matrix(Kzz, 1e-8)
    Kzx = A.se_features(x, torch.from_numpy(z).cuda(), ell, ctx=ctx)
    Phi = A.whiten_features(Kzx, Linv, ctx=ctx)
    kd = A.sparse.nystrom_residual(Phi, torch.ones(N, device="cuda"), ctx=ctx)
    if Phi.shape[1] % pad:
        Phi = torch.nn.functional.pad(Phi, (0, pad - Phi.shape[1] % pad)).contiguous()
    return x, y, Phi, kd


@pytest.mark.timeout(900)
def test_elbo_between_sweeps_leaves_the_next_sweep_alone(A, oracle):
    O = oracle
    lik, olik = A.BernoulliLikelihood(), O.bernoulli()
    N, M = 320_000, 96
    ctx_a, ctx_b = A.Context(0, seed=5), A.Context(0, seed=5)
    _, y, Phi, kd = _svgp(A, ctx_a, lik, N, M)
    a = A.SparseCAVI(lik, Phi, kd, y, ctx=ctx_a)  # the shipped path (factor marginals on the item queues, image accumulation)
    b = A.SparseCAVI(lik, Phi, kd, y, ctx=ctx_b)
    vals = []
    for it in range(6):
        a.sweep()
        b.sweep()
        if it in (1, 2, 4):
            vals.append(a.elbo())  # three reductions over N > 260 000 points each
            assert np.isfinite(vals[-1])
    a.check()
    b.check()
    assert torch.equal(a.G, b.G) and torch.equal(a.g, b.g), "an ELBO evaluation changed the following sweep"
    assert vals[0] <= vals[1] + 1e-6 * abs(vals[1]) and vals[1] <= vals[2] + 1e-6 * abs(vals[2]), vals
    Phi_h, kd_h, y_h = host(Phi), host(kd).astype(np.float64), host(y)
    Mp = Phi_h.shape[1]
    S, m = np.eye(Mp)[None], np.zeros((1, Mp))
    for it in range(6):
        G, g = O.cavi_pass(olik, Phi_h, kd_h, y_h, -S, m)
        S, m = O.gaussian_update(G, g)
    assert relmax(host(a.G), G) < NAT_TOL, relmax(host(a.G), G)
    assert relmax(host(a.g), g) < NAT_TOL, relmax(host(a.g), g)


@pytest.mark.timeout(600)
@pytest.mark.parametrize("N", [8192 + 1001, 8192 + 2048 + 1])
def test_dense_cholesky_blocked_route_with_an_odd_order(A, N):
    ctx = A.Context(0, seed=3)
    gen = torch.Generator(device="cuda").manual_seed(3)
    x = torch.sort(torch.rand(N, dtype=torch.float64, device="cuda", generator=gen) * 40 - 20).values
    K = torch.exp(-0.5 * ((x[:, None] - x[None, :]) / 0.05) ** 2)
    K.diagonal().add_(1e-3)
    Lk = torch.empty_like(K)
    ctx.call("agpl_dense_cholesky", C.c_int64(N), C.c_void_p(K.data_ptr()), C.c_void_p(Lk.data_ptr()))
    ref = torch.linalg.cholesky(K)
    got = torch.triu(Lk).T  # column-major lower triangle = row-major upper
    err = (got - ref).abs()
    assert err.max().item() < 1e-11, (err.max().item(), int(err.argmax()) // N, int(err.argmax()) % N)
    # and it is still a factor of K in its last rows (where the odd tail lives)
    tail = slice(N - 300, N)
    rec = got[tail] @ got.T
    assert (rec - K[tail]).abs().max().item() < 1e-12
