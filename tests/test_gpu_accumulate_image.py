"""GPU parity of the split-float16 accumulation from the static point-major image (agpl_syrk.hip: agpl_accumulate_image,
agpl_accumulate_split, agpl_cavi_pass_factor_image, agpl_gibbs_pass_image) against the float64 oracle
(oracle.accumulate: docs/src/index.md:154-163 in the whitened basis, G = Phi Diag(gamma) Phi', g = Phi beta).
Tolerance: 5e-6 of max|ref| per array (north_star: 1e-5 on the natural parameters); symmetry and bitwise repeat exact."""
import ctypes as C

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


def _p(t):
    return C.c_void_p(0 if t is None else t.data_ptr())


def _accumulate(A, ctx, Phi, beta, gamma, image=True, phi_arg=True):
    N, M = Phi.shape
    L = beta.shape[0]
    dPhi = torch.from_numpy(Phi).cuda()
    img = A.sparse.accumulate_image(dPhi, ctx) if image else None
    db, dg = torch.from_numpy(beta).cuda(), torch.from_numpy(gamma).cuda()
    G = torch.empty((L, M, M), dtype=torch.float64, device="cuda")
    g = torch.empty((L, M), dtype=torch.float64, device="cuda")
    args = (C.c_int64(N), C.c_int32(M), C.c_int32(L), _p(dPhi if phi_arg else None), _p(img), _p(db), _p(dg), _p(G), _p(g))
    ctx.call("agpl_accumulate_split", *args)
    G1, g1 = G.cpu().numpy().copy(), g.cpu().numpy().copy()
    ctx.call("agpl_accumulate_split", *args)
    assert np.array_equal(G.cpu().numpy(), G1) and np.array_equal(g.cpu().numpy(), g1), "not bitwise reproducible"
    return G1, g1


# edge cases the reference's sweep can produce: one point, a ragged last stage / slice, several slices, L > 1, every panel
# count the factor form accepts (M = 256, 512, 768, 1024)
@pytest.mark.parametrize("N,M,L", [(1, 256, 1), (31, 256, 1), (33, 512, 1), (4096, 256, 2), (4097, 512, 1),
                                   (20011, 512, 2), (9011, 768, 1), (9011, 1024, 1), (70001, 256, 3), (300007, 512, 1)])
def test_image_accumulate_against_oracle(A, ctx, oracle, N, M, L):
    rng = np.random.default_rng(N + 7 * M + L)
    Phi = (rng.standard_normal((N, M)) * 0.3).astype(np.float32)
    gamma = rng.uniform(0.0, 0.25, size=(L, N)).astype(np.float32)
    beta = rng.choice([-0.5, 0.5], size=(L, N)).astype(np.float32)
    G, g = _accumulate(A, ctx, Phi, beta, gamma, phi_arg=False)  # the float32 features are not an argument of this path
    Gr, gr = oracle.accumulate(Phi, beta, gamma)
    assert relmax(G, Gr) < 5e-6, relmax(G, Gr)
    assert relmax(g, gr) < 5e-6, relmax(g, gr)
    assert np.array_equal(G, G.transpose(0, 2, 1))


@pytest.mark.parametrize("phi_scale,gamma_scale", [(1e-6, 0.25), (3e4, 1e-5), (1.0, 1e4), (1e3, 1e-9), (1e-3, 3e3)])
def test_image_accumulate_scales_itself(A, ctx, oracle, phi_scale, gamma_scale):
    """The image carries 2^e_A phi (e_A from max |Phi|) and the kernel 2^e_B gamma (e_B from max gamma, per launch): the
    float16 operand range is never the caller's problem (the float32-staged split kernel needs |sqrt(gamma) phi| < 6e4)."""
    rng = np.random.default_rng(5)
    N, M = 6007, 256
    Phi = (rng.standard_normal((N, M)) * phi_scale).astype(np.float32)
    gamma = (rng.uniform(0.0, 1.0, size=(1, N)) * gamma_scale).astype(np.float32)
    gamma[0, ::7] = 0.0  # (Poisson: y = 0 and n = 0 give omega = PG(0, c) = 0)
    beta = rng.standard_normal((1, N)).astype(np.float32)
    G, g = _accumulate(A, ctx, Phi, beta, gamma)
    Gr, gr = oracle.accumulate(Phi, beta, gamma)
    assert relmax(G, Gr) < 5e-6, relmax(G, Gr)
    assert relmax(g, gr) < 5e-6, relmax(g, gr)


def test_image_accumulate_wide_dynamic_range_within_a_launch(A, ctx, oracle):
    """A few huge gamma beside many small ones (one heavy NegBin count): the launch scales by the maximum; the small terms
    keep an absolute floor of 2^-22 of the largest term, far below the 1e-5 bar on max|G|."""
    rng = np.random.default_rng(6)
    N, M = 9000, 256
    Phi = (rng.standard_normal((N, M)) * 0.3).astype(np.float32)
    gamma = rng.uniform(0.0, 0.25, size=(1, N)).astype(np.float32)
    gamma[0, rng.integers(0, N, 5)] = 250.0
    beta = rng.standard_normal((1, N)).astype(np.float32)
    G, g = _accumulate(A, ctx, Phi, beta, gamma)
    Gr, gr = oracle.accumulate(Phi, beta, gamma)
    assert relmax(G, Gr) < 5e-6 and relmax(g, gr) < 5e-6


def test_image_rejects_non_finite_features_with_their_index(A, ctx):
    Phi = torch.zeros((1000, 256), dtype=torch.float32, device="cuda")
    Phi[617, 33] = float("nan")
    with pytest.raises(A.DomainError, match=r"point 617, feature 33"):
        A.sparse.accumulate_image(Phi, ctx)
    Phi[617, 33] = float("inf")
    with pytest.raises(A.DomainError, match=r"point 617, feature 33"):
        A.sparse.accumulate_image(Phi, ctx)


def test_marginal_image_refuses_features_beyond_the_float16_range(A, ctx):
    """The marginal image is unscaled float16 hi / lo: |x| >= 65504 (or a non-finite value) would turn into inf and every
    marginal into NaN without a word -- agpl_split_features reports AGPL_ERR_DOMAIN with the position instead."""
    N, M = 1000, 256
    Phi = torch.zeros((N, M), dtype=torch.float32, device="cuda")
    nh = A._ffi.lib().agpl_split_features_bytes(C.c_int64(N), C.c_int32(M)) // 2
    hi, lo = (torch.empty(nh, dtype=torch.float16, device="cuda") for _ in range(2))
    args = (C.c_int64(N), C.c_int32(M), _p(Phi), _p(hi), _p(lo))
    Phi[5, 7] = 65000.0  # in range
    ctx.call("agpl_split_features", *args)
    for bad in (65504.0, -7.0e4, float("inf"), float("nan")):
        Phi[411, 200] = bad
        with pytest.raises(A.DomainError, match=r"point 411, feature 200"):
            ctx.call("agpl_split_features", *args)
    # ... while the accumulate image scales itself and takes the same magnitudes
    Phi[411, 200] = -7.0e4
    A.sparse.accumulate_image(Phi, ctx)


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


def test_image_needs_a_padded_feature_count(A, ctx):
    with pytest.raises(A.ArgumentError):
        A.sparse.accumulate_image(torch.zeros((10, 100), dtype=torch.float32, device="cuda"), ctx)


def test_m_not_multiple_of_256_falls_back_to_the_float32_staged_kernel(A, ctx, oracle):
    rng = np.random.default_rng(8)
    N, M = 5003, 384
    Phi = (rng.standard_normal((N, M)) * 0.3).astype(np.float32)
    gamma = rng.uniform(0.0, 0.25, size=(1, N)).astype(np.float32)
    beta = rng.standard_normal((1, N)).astype(np.float32)
    G, g = _accumulate(A, ctx, Phi, beta, gamma)  # image given, but the image kernel tiles by 256: Phi is read instead
    Gr, gr = oracle.accumulate(Phi, beta, gamma)
    assert relmax(G, Gr) < 5e-6 and relmax(g, gr) < 5e-6


def test_sparse_cavi_default_path_uses_the_image(A, ctx, oracle):
    """SparseCAVI's defaults (factor-form marginals + image accumulation = what bench.py times): 10 sweeps against the
    oracle's float64 sweep on the synthetic Bernoulli workload (examples/bernoulli/script.jl:29-39)."""
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


def test_image_of_another_problem_is_refused(A, ctx):
    """The accumulation checks the image's header (magic, N, M) once per image: a buffer that is not an image of THIS problem
    would otherwise be read out of range."""
    N, M = 4000, 256
    Phi = torch.randn((N, M), device="cuda")
    img = A.sparse.accumulate_image(Phi, ctx)
    gam = torch.rand((1, N), device="cuda")
    bet = torch.randn((1, N), device="cuda")
    G = torch.empty((1, M, M), dtype=torch.float64, device="cuda")
    g = torch.empty((1, M), dtype=torch.float64, device="cuda")
    ok = (C.c_int64(N), C.c_int32(M), C.c_int32(1), _p(None), _p(img), _p(bet), _p(gam), _p(G), _p(g))
    ctx.call("agpl_accumulate_split", *ok)
    with pytest.raises(A.ArgumentError, match="not an accumulate image of this problem"):
        ctx.call("agpl_accumulate_split", C.c_int64(N - 64), C.c_int32(M), C.c_int32(1), _p(None), _p(img), _p(bet[:, :N - 64].contiguous()),
                 _p(gam[:, :N - 64].contiguous()), _p(G), _p(g))
    junk = torch.zeros_like(img)
    with pytest.raises(A.ArgumentError, match="not an accumulate image of this problem"):
        ctx.call("agpl_accumulate_split", C.c_int64(N), C.c_int32(M), C.c_int32(1), _p(None), _p(junk), _p(bet), _p(gam), _p(G), _p(g))
    ctx.call("agpl_accumulate_split", *ok)  # the context is still usable


@pytest.mark.parametrize("badval", [float("nan"), -0.5, float("inf")])
def test_stand_alone_image_accumulation_reports_a_bad_gamma(A, ctx, badval):
    """agpl_accumulate_split from an image takes gamma without a square root, so a negative one gives plausible
    sums; the record kernel's flag word is read back by the stand-alone call (a sweep gets the same report with its update)."""
    N, M = 3000, 256
    Phi = torch.randn((N, M), device="cuda")
    img = A.sparse.accumulate_image(Phi, ctx)
    gam = torch.rand((1, N), device="cuda")
    bet = torch.randn((1, N), device="cuda")
    G = torch.empty((1, M, M), dtype=torch.float64, device="cuda")
    g = torch.empty((1, M), dtype=torch.float64, device="cuda")
    gam[0, 1234] = badval
    with pytest.raises(A.DomainError, match="flat index 1234"):
        ctx.call("agpl_accumulate_split", C.c_int64(N), C.c_int32(M), C.c_int32(1), _p(None), _p(img), _p(bet), _p(gam), _p(G), _p(g))
    gam[0, 1234] = 0.25
    ctx.call("agpl_accumulate_split", C.c_int64(N), C.c_int32(M), C.c_int32(1), _p(None), _p(img), _p(bet), _p(gam), _p(G), _p(g))
    ref = (Phi.double().T * gam[0].double()) @ Phi.double()
    assert (G[0] - ref).abs().max().item() < 1e-5 * ref.abs().max().item()
