"""BASELINE.json configs C3 and C4 at their own shapes (run with -m gpu on an MI355X), on the path bench.py ships
(factor-form split-float16 marginals + split-float16 accumulation):

  C3  NegativeBinomialLikelihood(r = 15), M = 1024: the one-launch factor pipeline (factor_pipe_kernel) inside a CAVI sweep
      (/root/reference/src/likelihoods/negativebinomial.jl:20-49) -- 10 sweeps against the oracle at a size it
      finishes in seconds; the two-rank sharded variant lives in test_gpu_distributed.py.
  C4  CategoricalLikelihood(LogisticSoftMaxLink(zeros(10))), K = 10 latent GPs, M = 256
      (/root/reference/src/likelihoods/categorical.jl:72-136) -- 10 sweeps against the oracle, Gibbs counts bit-exact,
      and the size-independent properties at the configured N = 1e6.
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

SEED = 20240807
NAT_TOL = 1e-5


@pytest.fixture(scope="module")
def A():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import __graft_entry__ as g

    g.build()
    import agpl_amd

    return agpl_amd


@pytest.fixture(scope="module")
def ctx(A):
    return A.Context(0, seed=SEED)


def host(t):
    return t.detach().cpu().numpy()


def relmax(a, b):
    return np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(np.asarray(b)).max(), 1e-300)


def setup_svgp(A, ctx, lik, N, M, i0=0):
    """The synthetic workload of SURVEY.md 8(d) / bench.py on the device: whitened SE features, Nystrom residual."""
    x, y = A.synth_xy(lik, SEED, i0, N, ctx=ctx)
    z = np.linspace(-10, 10, M)
    ell = 1.5 * (z[1] - z[0])
    Kzz = np.exp(-0.5 * ((z[:, None] - z[None, :]) / ell) ** 2)
    _, Linv = A.sparse.whitening_matrix(Kzz, 1e-8)
    Kzx = A.se_features(x, torch.from_numpy(z).cuda(), ell, ctx=ctx)
    Phi = A.whiten_features(Kzx, Linv, ctx=ctx)
    del Kzx
    kd = A.sparse.nystrom_residual(Phi, torch.ones(N, device="cuda"), ctx=ctx)
    return x, y, Phi, kd


def shipped(A, lik, Phi, kd, y, ctx, **kw):
    return A.SparseCAVI(lik, Phi, kd, y, ctx=ctx, marginal_precision="f16x2-factor", accumulate_precision="f16x2", **kw)


def sampled_marginals_and_operators_match_float64(A, cavi, Phi, lik_name, r=None):
    """the marginal kernel at full size with a REAL posterior factor.  After >= 2 sweeps: U, v from
    agpl_plan_state, a sample of >= 1e4 points over every per-XCD item queue and the last (ragged) tile, mu / var of
    agpl_marginals_plan against float64 from the float32 feature rows (2e-5 of max, the bar of test_gpu_random_shapes),
    then gamma, beta of the next pass at those points against the float64 operators on the float64 marginals."""
    import bench

    assert cavi.nsweeps >= 2 and cavi.plan is not None
    chk = bench.full_size_marginal_check(cavi, Phi)
    assert chk["sampled_points"] >= 10_000 and chk["tile_residues_mod_8"] == list(range(8)), chk
    assert chk["max_abs_offdiag_U"] > 1e-3, chk  # a real factor, not the identity
    assert chk["max_rel_d_mu"] < 2e-5 and chk["max_rel_d_var"] < 2e-5, chk
    # gamma, beta of the next pass at the sampled points, from float64 marginals
    idx, _ = bench.marginal_sample_indices(cavi.N)
    P = Phi[idx].double()
    T = P @ torch.triu(cavi.plan.U_colmajor[0])
    mu = T @ cavi.plan.v[0]
    var = cavi.kdiag[idx].double().clamp_min(0.0) + (T * T).sum(1)
    c = torch.sqrt(mu * mu + var)
    cavi.accumulate()  # exports gamma, beta (keep_points=True) for the q(v) just checked
    y = cavi.y[idx].double()
    b = 1.0 if lik_name == "bernoulli" else y + r  # bernoulli.jl:35-45 / negativebinomial.jl:43-49
    gref = b * torch.tanh(c / 2) / (2 * c)
    bref = (y - 0.5) if lik_name == "bernoulli" else (y - r) / 2
    assert ((cavi.gamma[0][idx].double() - gref).abs().max() / gref.abs().max()).item() < 2e-5
    assert torch.equal(cavi.beta[0][idx].double(), bref)
    return chk


def ten_sweeps_against_oracle(A, ctx, O, lik, olik, N, M):
    x, y, Phi, kd = setup_svgp(A, ctx, lik, N, M)
    cavi = shipped(A, lik, Phi, kd, y, ctx)
    Phi_h, kd_h, y_h = host(Phi), host(kd).astype(np.float64), host(y)
    Mp, L = Phi_h.shape[1], olik.nlatent
    S, m = np.tile(np.eye(Mp), (L, 1, 1)), np.zeros((L, Mp))
    try:
        for it in range(10):
            cavi.sweep()
            G, g = O.cavi_pass(olik, Phi_h, kd_h, y_h, -S, m)
            S, m = O.gaussian_update(G, g)
            if it == 0:
                assert relmax(host(cavi.G), G) < NAT_TOL and relmax(host(cavi.g), g) < NAT_TOL
        cavi.check()
    finally:
        pass
    dG, dg = relmax(host(cavi.G), G), relmax(host(cavi.g), g)
    assert dG < NAT_TOL and dg < NAT_TOL, (dG, dg)
    Lam, eta = cavi.natural_parameters()
    assert relmax(host(Lam), np.eye(Mp) + G) < NAT_TOL and relmax(host(eta), g) < NAT_TOL
    kappa = max(np.linalg.cond(np.eye(Mp) + G[l]) for l in range(L))
    assert relmax(host(cavi.m), m) < max(1e-4, NAT_TOL * kappa)
    return cavi, (G, g, S, m)


@pytest.mark.timeout(900)
def test_m1280_two_block_factor_route_ten_sweeps_match_oracle(A, ctx, oracle):
    """M = 1280 (> 1024: since round 6 the M x M update is two block rows of the one-launch factorisation with four products on the
    float64 tile routine -- agpl_factor_two_block, no library call; the image sweep stays: 5 x 5 panels of 256) -- ten Bernoulli
    sweeps against the oracle.  Then M = 2304 (> 2048: the rocSOLVER route), on which the sweep's bad-gamma word is NOT forwarded by
    the update: it must be dropped at the next sweep, not reported to a later problem on the same context."""
    O = oracle
    lik, olik = A.BernoulliLikelihood(), O.bernoulli()
    cavi, _ = ten_sweeps_against_oracle(A, ctx, O, lik, olik, 6_000, 1280)
    assert cavi.plan is not None and cavi.M == 1280
    # a problem with a NaN observation-derived gamma on the library route ...
    bad = A.Context(0, seed=5)
    slik = A.StudentTLikelihood(3.0, 1.0)
    x, y, Phi, kd = setup_svgp(A, bad, slik, 3_000, 2304)
    y[77] = float("nan")
    c1 = shipped(A, slik, Phi, kd, y, bad)
    try:
        c1.sweep()
        c1.sweep()
        bad.synchronize()
    except (A.DomainError, A.PosDefException, A.AGPLError):
        pass  # (whatever the library route reports for the NaN matrix)
    del c1
    # ... leaves nothing behind for a healthy problem on the same context (M <= 1024: the forwarding route)
    x, y2, Phi2, kd2 = setup_svgp(A, bad, lik, 3_000, 512)
    c2 = shipped(A, lik, Phi2, kd2, y2, bad)
    for _ in range(3):
        c2.sweep()
    c2.check()
    assert torch.isfinite(c2.G).all()


@pytest.mark.timeout(900)
def test_c2_bernoulli_m512_ten_sweeps_match_oracle(A, ctx, oracle):
    """C2's likelihood and M (Bernoulli-logistic, M = 512: the one-launch factor kernel, the 2 x 2-panel image accumulation):
    ten plan sweeps (agpl_cavi_pass_plan + agpl_plan_update) against the oracle, at a size it finishes in seconds."""
    O = oracle
    cavi, _ = ten_sweeps_against_oracle(A, ctx, O, A.BernoulliLikelihood(), O.bernoulli(), 20_000, 512)
    assert cavi.plan is not None and cavi.M == 512


@pytest.mark.timeout(900)
def test_c3_negbin_m1024_ten_sweeps_match_oracle(A, ctx, oracle):
    """C3's likelihood and M: NegBin r = 15 (examples/negativebinomial/script.jl:17), M = 1024 -> the M x M update is ONE launch of
    factor_pipe_kernel (agpl_factor.hip, since round 5; the two-block route of round 4 is gone) ten times inside the sweep loop."""
    O = oracle
    lik, olik = A.NegativeBinomialLikelihood(15.0), O.negbinomial(15.0)
    ten_sweeps_against_oracle(A, ctx, O, lik, olik, 20_000, 1024)


@pytest.mark.timeout(900)
def test_c4_categorical_k10_m256_ten_sweeps_match_oracle(A, ctx, oracle):
    """C4's likelihood and M: non-bijective LogisticSoftMaxLink(zeros(10)) -> 10 latent GPs sharing one K_ZX
    (examples/categorical/script.jl:65), M = 256: ten 256 x 256 factorisations per update in one launch."""
    O = oracle
    lik, olik = A.CategoricalLikelihood(np.zeros(10)), O.categorical(np.zeros(10))
    assert A.nlatent(lik) == 10
    ten_sweeps_against_oracle(A, ctx, O, lik, olik, 20_000, 256)


def test_c4_categorical_k10_gibbs_counts_bit_exact(A, ctx, oracle):
    """The Gibbs half of C4: negative-multinomial counts and uniforms consumed bit-exact for K = 10
    (categorical.jl:72-78, negativemultinomial.jl:35-45), draws to 1e-9, accumulated (G, g) to 5e-6."""
    O = oracle
    lik, olik = A.CategoricalLikelihood(np.zeros(10)), O.categorical(np.zeros(10))
    N, M, L = 6_000, 256, 10
    x, y, Phi, kd = setup_svgp(A, ctx, lik, N, M)
    rng = np.random.default_rng(5)
    v = rng.normal(size=(L, M)) * 0.7
    dv = torch.from_numpy(v).cuda()
    G = torch.empty((L, M, M), dtype=torch.float64, device="cuda")
    g = torch.empty((L, M), dtype=torch.float64, device="cuda")
    f = torch.empty((N, L), dtype=torch.float64, device="cuda")
    om = torch.empty((N, L), dtype=torch.float64, device="cuda")
    nn = torch.zeros((N, L), dtype=torch.int64, device="cuda")
    nuni = torch.zeros(N, dtype=torch.int32, device="cuda")
    d = lik.desc()
    p = lambda t: C.c_void_p(t.data_ptr())
    ctx.call("agpl_gibbs_pass", C.byref(d), C.c_int64(N), C.c_int32(M), p(Phi), p(kd), C.c_void_p(0), p(y), p(dv),
             C.c_uint32(3), p(G), p(g), p(f), p(om), p(nn), p(nuni))
    Gr, gr, pts = O.gibbs_pass(olik, host(Phi), host(kd).astype(np.float64), host(y), v, seed=SEED, sweep=3)
    assert np.array_equal(host(nn), pts["n"])
    assert np.array_equal(host(nuni).astype(np.uint32), pts["nuni"])
    assert np.allclose(host(om), pts["omega"], rtol=1e-9, atol=0)
    assert relmax(host(G), Gr) < 5e-6 and relmax(host(g), gr) < 5e-6


@pytest.mark.timeout(600)
def test_c3_per_rank_full_size_properties(A, ctx):
    """C3 at the size ONE rank of its 8-GPU configuration holds (NegBin r = 15, N = 1e7 / 8 = 1.25e6, M = 1024): sizes the
    oracle cannot reach, so size-independent properties of the shipped path -- exact symmetry, bitwise reproducibility,
    tr G = sum_n gamma_n |phi_n|^2 and g = Phi beta against float64 reductions, gamma = (y + r) tanh(c / 2) / (2 c) in
    (0, (y + r) / 4], additivity over N (the sharding identity), a full sweep through the one-launch M x M factor pipeline, and
    one Gibbs point pass with finite positive draws (negativebinomial.jl:20-49)."""
    N, M, r = 1_250_000, 1024, 15.0
    lik = A.NegativeBinomialLikelihood(r)
    x, y, Phi, kd = setup_svgp(A, ctx, lik, N, M)

    def make(sl=slice(None)):
        return shipped(A, lik, Phi[sl], kd[sl], y[sl], ctx, keep_points=True)

    try:
        cavi = make()
        assert cavi.plan is not None  # the plan path: image accumulation (M % 256 == 0)
        mu, var = cavi.marginals()
        assert mu.abs().max().item() == 0.0 and (var - 1.0).abs().max().item() < 4e-5
        cavi.accumulate()
        G1, g1 = cavi.G.clone(), cavi.g.clone()
        assert torch.equal(G1, G1.transpose(1, 2))
        gam, bet = cavi.gamma[0], cavi.beta[0]
        assert torch.isfinite(gam).all() and (gam > 0).all() and bool((gam <= (y.float() + r) / 4 * (1 + 1e-6)).all())
        assert torch.equal(bet, ((y.float() - r) / 2))
        tr = 0.0
        gref = torch.zeros(M, dtype=torch.float64, device="cuda")
        for i0 in range(0, N, 250_000):
            P = Phi[i0:i0 + 250_000].double()
            tr += (gam[i0:i0 + 250_000].double() * (P * P).sum(1)).sum().item()
            gref += P.T @ bet[i0:i0 + 250_000].double()
        assert torch.diagonal(G1[0]).sum().item() == pytest.approx(tr, rel=2e-6)
        assert relmax(host(g1[0]), host(gref)) < 2e-6
        import bench  # v'Gv = sum_n gamma_n (phi_n . v)^2 for 4 random v: every tile of G (ten 256 x 256 ones here) enters

        chk = bench.full_size_quadratic_check(Phi, cavi.gamma, cavi.beta, G1, g1)
        assert chk["max_rel_d_vGv"] < 2e-6, chk
        cavi.accumulate()
        assert torch.equal(cavi.G, G1) and torch.equal(cavi.g, g1)
        for _ in range(3):  # the M = 1024 factor route inside whole sweeps
            cavi.sweep()
        cavi.check()
        assert torch.isfinite(cavi.G).all() and torch.isfinite(cavi.v).all()
        # N = 1.25e6 = 9765 full tiles + a ragged one of 80 points; M = 1024: four 256-row blocks of U per tile
        chk = sampled_marginals_and_operators_match_float64(A, cavi, Phi, "negbin", r=r)
        assert chk["last_tile_points"] == 80
        del cavi
        h = N // 2
        acc = None
        for sl in (slice(0, h), slice(h, N)):
            c = make(sl)
            c.accumulate()
            acc = (c.G.clone(), c.g.clone()) if acc is None else (acc[0] + c.G, acc[1] + c.g)
            del c
        assert relmax(host(acc[0]), host(G1)) < 2e-6 and relmax(host(acc[1]), host(g1)) < 2e-6
        gib = A.SparseGibbs(lik, Phi, kd, y, ctx=A.Context(0, seed=SEED + 1), keep_points=True)
        gib.sweep()
        assert torch.isfinite(gib.omega).all() and (gib.omega > 0).all() and torch.isfinite(gib.v).all()
        assert torch.equal(gib.G, gib.G.transpose(1, 2))
    finally:
        pass


def test_c4_full_size_properties(A, ctx):
    """C4 at its configured size (K = 10, N = 1e6, M = 256): sizes the oracle cannot reach, so size-independent
    properties of one accumulation of the shipped path, per latent: exact symmetry, bitwise reproducibility,
    tr G_l = sum_n gamma_ln |phi_n|^2 and g_l = Phi beta_l against float64 reductions, additivity over N (the sharding
    identity), first-sweep marginals in closed form, and a full ten-latent update staying finite."""
    N, M, L = 1_000_000, 256, 10
    lik = A.CategoricalLikelihood(np.zeros(L))
    x, y, Phi, kd = setup_svgp(A, ctx, lik, N, M)
    assert tuple(y.shape) == (N, L) and int(y.sum(1).max().item()) <= 1  # one-hot rows (or all zero: class K+1 absent)

    def make(sl=slice(None)):
        return shipped(A, lik, Phi[sl], kd[sl], y[sl], ctx, keep_points=True)

    try:
        cavi = make()
        mu, var = cavi.marginals()
        assert mu.abs().max().item() == 0.0 and (var - 1.0).abs().max().item() < 2e-5
        cavi.accumulate()
        G1, g1 = cavi.G.clone(), cavi.g.clone()
        assert torch.equal(G1, G1.transpose(1, 2))
        assert torch.isfinite(cavi.gamma).all() and (cavi.gamma > 0).all()
        P = Phi.double()
        n2 = (P * P).sum(1)
        for l in range(L):
            tr = (cavi.gamma[l].double() * n2).sum().item()
            assert torch.diagonal(G1[l]).sum().item() == pytest.approx(tr, rel=2e-6)
            assert relmax(host(g1[l]), host(P.T @ cavi.beta[l].double())) < 2e-6
        del P, n2
        import bench  # v'G_l v = sum_n gamma_ln (phi_n . v)^2 for 4 random v, every latent

        chk = bench.full_size_quadratic_check(Phi, cavi.gamma, cavi.beta, G1, g1)
        assert chk["max_rel_d_vGv"] < 2e-6, chk
        cavi.accumulate()
        assert torch.equal(cavi.G, G1) and torch.equal(cavi.g, g1)
        # one more full sweep through the ten-latent factor launch: finite, and S = U'U symmetric positive
        cavi.update()
        cavi.check()
        assert torch.isfinite(cavi.v).all()
        del cavi
        h = N // 2
        acc = None
        for sl in (slice(0, h), slice(h, N)):
            c = make(sl)
            c.accumulate()
            acc = (c.G.clone(), c.g.clone()) if acc is None else (acc[0] + c.G, acc[1] + c.g)
            del c
        assert relmax(host(acc[0]), host(G1)) < 2e-6 and relmax(host(acc[1]), host(g1)) < 2e-6
    finally:
        pass


def test_dense_gibbs_poisson_with_zero_counts(A, ctx, oracle):
    """Poisson full-rank Gibbs with y_i = 0 points (poisson.jl:26-28): a drawn n_i = 0 gives omega_i = PG(0, c) = 0,
    hence gamma_i = beta_i = 0 -- the step must stay finite (beta / sqrt(gamma) := 0 there, exactly) and match the
    numpy chain on the same streams."""
    O = oracle
    lik, olik = A.PoissonLikelihood(3.0), O.poisson(3.0)
    N = 500
    rng = np.random.default_rng(11)
    xs = np.sort(rng.uniform(-10, 10, size=N))
    K = np.exp(-0.5 * ((xs[:, None] - xs[None, :]) / 2.0) ** 2) + 1e-6 * np.eye(N)
    y = rng.poisson(0.4, size=N).astype(np.int32)
    assert (y == 0).sum() > N // 2
    dctx = A.Context(0, seed=606)
    dg = A.DenseGibbs(lik, torch.from_numpy(K).cuda(), torch.from_numpy(y).cuda(), ctx=dctx)
    Lk = np.linalg.cholesky(K)
    f = np.zeros(N)
    saw_zero = False
    for sweep in range(4):
        dg.sweep()
        f, d = O.dense_gibbs_step(olik, K, Lk, y, f, seed=606, sweep=sweep)
        saw_zero |= bool((d["omega"] == 0).any())
        assert np.array_equal(host(dg.n), d["n"])
        assert np.isfinite(host(dg.f)).all()
        assert np.allclose(host(dg.omega), d["omega"], rtol=1e-6)
        assert np.abs(host(dg.f) - f).max() < 1e-6 * max(1.0, np.abs(f).max())
    assert saw_zero  # the case the guard exists for was exercised
