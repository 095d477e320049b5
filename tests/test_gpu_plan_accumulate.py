"""GPU parity of the plan's split-float16 accumulation from the static point-major image (agpl_syrk.hip syrk_strip_kernel behind
agpl_cavi_pass_plan / agpl_gibbs_pass_plan) against float64: G = Phi Diag(gamma) Phi', g = Phi beta (docs/src/index.md:154-163 in
the whitened basis) from the gamma, beta the pass itself exports -- whatever the per-point kernel produced, the accumulation must
sum exactly that.  Tolerance: 5e-6 of max|ref| per array (north_star: 1e-5 on the natural parameters); symmetry and bitwise
repeat exact.  (v110 drove the same kernel through agpl_accumulate_split with arbitrary gamma; the likelihoods below span the
same cases: one point, ragged stages / slices, several latents, every panel count, gamma over nine decades in one launch.)"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def A():
    import agpl_amd as A

    return A


@pytest.fixture(scope="module")
def ctx(A):
    return A.Context(0, seed=11)


@pytest.fixture(scope="module")
def oracle():
    from oracle import oracle as O

    return O


def relmax(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def _pass(A, ctx, lik, Phi, y, resid=None, sweeps_before=0):
    """One plan pass (after `sweeps_before` whole sweeps) with gamma, beta exported; returns G, g, gamma, beta as numpy."""
    N, M = Phi.shape
    dPhi = torch.from_numpy(Phi).cuda()
    kd = torch.ones(N, device="cuda") if resid is None else torch.from_numpy(resid).cuda()
    cavi = A.SparseCAVI(lik, dPhi, kd, torch.from_numpy(y).cuda(), ctx=ctx, keep_points=True)
    assert cavi.plan is not None
    for _ in range(sweeps_before):
        cavi.sweep()
    cavi.accumulate()
    G1, g1 = cavi.G.cpu().numpy().copy(), cavi.g.cpu().numpy().copy()
    cavi.accumulate()
    assert np.array_equal(cavi.G.cpu().numpy(), G1) and np.array_equal(cavi.g.cpu().numpy(), g1), "not bitwise reproducible"
    cavi.check()
    return G1, g1, cavi.gamma.cpu().numpy(), cavi.beta.cpu().numpy()


def _float64_sums(Phi, gamma, beta):
    P = Phi.astype(np.float64)
    G = np.stack([(P * gamma[l].astype(np.float64)[:, None]).T @ P for l in range(gamma.shape[0])])
    g = np.stack([P.T @ beta[l].astype(np.float64) for l in range(beta.shape[0])])
    return G, g


# edge cases the reference's sweep can produce: one point, a ragged last stage / slice, several slices, L > 1, every panel
# count the plan accepts (M = 256, 512, 768, 1024)
@pytest.mark.parametrize("N,M,L", [(1, 256, 1), (31, 256, 1), (33, 512, 1), (4096, 256, 2), (4097, 512, 1),
                                   (20011, 512, 2), (9011, 768, 1), (9011, 1024, 1), (70001, 256, 3), (300007, 512, 1),
                                   # round 6, agpl_slice_plan: exactly one round of big slices (all of them cut fine), one point more
                                   # (one big slice + a ragged fine tail), and a long M = 256 launch (no fine tail)
                                   (352256, 512, 1), (352257, 512, 1), (430001, 256, 10)])
def test_plan_accumulation_against_float64(A, ctx, oracle, N, M, L):
    rng = np.random.default_rng(N + 7 * M + L)
    Phi = (rng.standard_normal((N, M)) * 0.3).astype(np.float32)
    if L == 1:
        lik = A.BernoulliLikelihood()
        y = (rng.uniform(size=N) < 0.5).astype(np.uint8)
    else:
        lik = A.CategoricalLikelihood(np.zeros(L))
        lab = rng.integers(0, L + 1, size=N)
        y = (lab[:, None] == np.arange(L)[None, :]).astype(np.uint8)
    G, g, gamma, beta = _pass(A, ctx, lik, Phi, y, sweeps_before=1 if N > 1000 else 0)
    assert gamma.shape == (L, N) and np.all(gamma > 0)
    Gr, gr = _float64_sums(Phi, gamma, beta)
    assert relmax(G, Gr) < 5e-6, relmax(G, Gr)
    assert relmax(g, gr) < 5e-6, relmax(g, gr)
    assert np.array_equal(G, G.transpose(0, 2, 1))
    G2, g2 = oracle.accumulate(Phi, beta, gamma)  # the oracle's own accumulation of the same records
    assert relmax(G, G2) < 5e-6 and relmax(g, g2) < 5e-6


@pytest.mark.parametrize("phi_scale,r", [(1e-6, 1.0), (3e4, 1e-4), (1.0, 4e4), (1e3, 1e-8), (1e-3, 1.2e4)])
def test_plan_accumulation_scales_itself(A, ctx, phi_scale, r):
    """The images carry 2^e phi (e from max |Phi|) and the kernel 2^e_B gamma (e_B from max gamma, per launch): the float16 operand
    range is never the caller's problem.  Negative binomial with r failures: gamma = (y + r) tanh(c/2) / (2c) spans r/4 .. (y + r)/4."""
    rng = np.random.default_rng(5)
    N, M = 6007, 256
    Phi = (rng.standard_normal((N, M)) * phi_scale).astype(np.float32)
    y = rng.poisson(3.0, size=N).astype(np.int32)
    resid = np.full(N, np.float32(max(phi_scale ** 2, 1e-12)))  # q(f_i) at the start: var_i = d_i + |phi_i|^2
    G, g, gamma, beta = _pass(A, ctx, A.NegativeBinomialLikelihood(r), Phi, y, resid=resid)
    Gr, gr = _float64_sums(Phi, gamma, beta)
    assert relmax(G, Gr) < 5e-6, relmax(G, Gr)
    assert relmax(g, gr) < 5e-6, relmax(g, gr)


def test_plan_accumulation_wide_dynamic_range_within_a_launch(A, ctx):
    """A few huge gamma beside many small ones (one heavy NegBin count): the launch scales by the maximum; the small terms
    keep an absolute floor of 2^-22 of the largest term, far below the 1e-5 bar on max|G|."""
    rng = np.random.default_rng(6)
    N, M = 9000, 256
    Phi = (rng.standard_normal((N, M)) * 0.3).astype(np.float32)
    y = rng.poisson(0.5, size=N).astype(np.int32)
    y[rng.integers(0, N, 5)] = 1_000_000
    G, g, gamma, beta = _pass(A, ctx, A.NegativeBinomialLikelihood(1e-3), Phi, y)
    assert gamma.max() / gamma[gamma > 0].min() > 1e8
    Gr, gr = _float64_sums(Phi, gamma, beta)
    assert relmax(G, Gr) < 5e-6 and relmax(g, gr) < 5e-6


def test_plan_rejects_non_finite_features_with_their_index(A, ctx):
    Phi = torch.zeros((1000, 256), dtype=torch.float32, device="cuda")
    kd, y = torch.ones(1000, device="cuda"), torch.zeros(1000, dtype=torch.uint8, device="cuda")
    for bad in (float("nan"), float("inf")):
        Phi[617, 33] = bad
        with pytest.raises(A.DomainError, match=r"point 617, feature 33"):
            A.SparseCAVI(A.BernoulliLikelihood(), Phi, kd, y, ctx=ctx)
    Phi[617, 33] = -7.0e4  # beyond the float16 range unscaled: the images scale themselves
    A.SparseCAVI(A.BernoulliLikelihood(), Phi, kd, y, ctx=ctx).sweep()


def test_plan_refuses_a_residual_that_is_not_one(A, ctx):
    """round-off below zero in d_i = k_ii - |phi_i|^2 is clamped, anything more negative is an error with its index."""
    rng = np.random.default_rng(3)
    Phi = torch.from_numpy((rng.standard_normal((2000, 256)) * 0.05).astype(np.float32)).cuda()
    y = torch.zeros(2000, dtype=torch.uint8, device="cuda")
    kd = torch.ones(2000, device="cuda")
    kd[77] = -1e-8  # float32 round-off of k_ii - |phi|^2 with |phi|^2 ~ 0.64: clamped
    c = A.SparseCAVI(A.BernoulliLikelihood(), Phi, kd, y, ctx=ctx)
    mu, var = c.marginals()
    assert torch.isfinite(var).all() and (var > 0).all()
    kd[1503] = -0.25  # a wrong array / sign
    with pytest.raises(A.DomainError, match=r"resid\[1503\]"):
        A.SparseCAVI(A.BernoulliLikelihood(), Phi, kd, y, ctx=ctx)


def test_sweep_reports_a_non_finite_expected_precision(A, oracle):
    """A NaN observation makes gamma NaN (StudentT: w_i from (y_i - f_i)^2): the per-point kernel flags it and the outcome of
    the update behind it carries AGPL_ERR_DOMAIN with the flat index -- not a silent NaN posterior."""
    import bench

    ctx = A.Context(0, seed=3)
    lik = A.StudentTLikelihood(3.0, 1.0)
    y, Phi, kd = bench.build_workload(A, ctx, lik, 0, 5_000, 256)
    y[1234] = float("nan")
    cavi = A.SparseCAVI(lik, Phi, kd, y, ctx=ctx)
    assert cavi.plan is not None  # the shipped path: the per-point kernel of the image sweep
    with pytest.raises(A.DomainError, match=r"flat index 1234"):
        for _ in range(3):
            cavi.sweep()
        cavi.check()


def test_plan_pads_the_feature_count_itself(A, ctx):
    """Round 6: a feature count that is not a multiple of 256 no longer falls to the float32-input kernels (4 x slower) or asks the
    caller to zero-pad: the plan pads its images, G / g / U stay M-sized for the caller."""
    Phi = torch.zeros((10, 384), dtype=torch.float32, device="cuda")
    kd, y = torch.ones(10, device="cuda"), torch.zeros(10, dtype=torch.uint8, device="cuda")
    c = A.SparseCAVI(A.BernoulliLikelihood(), Phi, kd, y, ctx=ctx)  # "auto": the plan path at any feature count
    assert c.plan is not None and c.marginal_precision == "f16x2-factor" and c.plan.Mp == 512 and c.plan.M == 384
    c.sweep()
    c.check()
    assert tuple(c.G.shape) == (1, 384, 384) and tuple(c.S.shape) == (1, 384, 384) and tuple(c.m.shape) == (1, 384)
    f = A.SparseCAVI(A.BernoulliLikelihood(), Phi, kd, y, ctx=ctx, marginal_precision="f32", accumulate_precision="f32")
    assert f.plan is None  # the float32-input pair on request (feature count a multiple of 128)
    with pytest.raises(A.ArgumentError, match="multiple of 128"):
        A.SparseCAVI(A.BernoulliLikelihood(), Phi[:, :200].contiguous(), kd, y, ctx=ctx, marginal_precision="f32", accumulate_precision="f32")


def test_sparse_cavi_default_path_uses_the_plan(A, ctx, oracle):
    """SparseCAVI's defaults (= what bench.py times): 10 sweeps against the oracle's float64 sweep on the synthetic Bernoulli
    workload (examples/bernoulli/script.jl:29-39)."""
    import bench

    lik = A.BernoulliLikelihood()
    y, Phi, kd = bench.build_workload(A, ctx, lik, 0, 12_000, 256)
    cavi = A.SparseCAVI(lik, Phi, kd, y, ctx=ctx)
    assert cavi.factor and cavi.plan is not None
    olik = oracle.bernoulli()
    Ph, kh, yh = Phi.cpu().numpy(), kd.cpu().numpy().astype(np.float64), y.cpu().numpy()
    S, m = np.eye(256)[None], np.zeros((1, 256))
    for _ in range(10):
        cavi.sweep()
        G, g = oracle.cavi_pass(olik, Ph, kh, yh, -S, m)
        S, m = oracle.gaussian_update(G, g)
    cavi.check()
    assert relmax(cavi.G.cpu().numpy(), G) < 1e-5
    assert relmax(cavi.g.cpu().numpy(), g) < 1e-5


def test_gibbs_pass_on_a_plan_follows_the_float32_pass(A, ctx):
    """The plan's Gibbs point pass projects phi_i' v from the accumulate image (features to 2^-22 relative) where the float32
    pass reads Phi itself: the draws of f agree to that precision, (G, g) within the split-float16 bound of each other.  (Exact
    parity of the plan's pass -- f, omega, counts, uniforms consumed -- is against the oracle on the image's own features:
    tests/test_gpu_plan.py.)"""
    import bench

    lik = A.NegativeBinomialLikelihood(15.0)
    y, Phi, kd = bench.build_workload(A, ctx, lik, 0, 9000, 256)
    ga = A.SparseGibbs(lik, Phi, kd, y, ctx=A.Context(0, seed=3), keep_points=True, accumulate_precision="f16x2")
    gb = A.SparseGibbs(lik, Phi, kd, y, ctx=A.Context(0, seed=3), keep_points=True, accumulate_precision="f32")
    assert ga.plan is not None and gb.plan is None
    ga.accumulate()
    gb.accumulate()  # (one pass from the same v: later sweeps draw different v from slightly different G)
    assert (ga.f - gb.f).abs().max().item() < 2e-6 * max(1.0, gb.f.abs().max().item())
    same = (ga.omega - gb.omega).abs() < 1e-5 * gb.omega.abs().clamp_min(1e-30)
    assert same.float().mean().item() > 0.999  # (a 1e-7 change of |f| moves a PG draw continuously, bar a rare accept / reject flip)
    assert relmax(ga.G.cpu().numpy(), gb.G.cpu().numpy()) < 1e-4
