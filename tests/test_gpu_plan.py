"""GPU tests of the plan API (include/agpl.h: agpl_plan_create, agpl_cavi_pass_plan, agpl_plan_update, agpl_marginals_plan,
agpl_gibbs_pass_plan) -- the shipped sweep path -- against the float64 oracle:

* 10-sweep natural parameters (BASELINE north_star: 1e-5 relative) for feature matrices at three scales: both images of a plan
  carry 2^e Phi with one e from max |Phi| (the unscaled marginal image of rounds 1-3 lost precision below 2^-14 and refused
  |x| >= 65504);
* the ELBO riding the sweep (examples/bernoulli/script.jl:65-70): the per-point terms from the pass's one
  per-point kernel and the Gaussian KL from the update equal the separate-pass evaluation to 1e-9 relative, the oracle's value to
  2e-6, and do not decrease over sweeps;
* argument errors and the domain error of the images.
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


def _liks(A, O):
    return {"bernoulli": (A.BernoulliLikelihood(), O.bernoulli()),
            "negbin": (A.NegativeBinomialLikelihood(15.0), O.negbinomial(15.0)),
            "studentt": (A.StudentTLikelihood(3.5, 2.0), O.studentt(3.5, 2.0)),
            "catbij": (A.CategoricalLikelihood(np.array([0.1, -0.2, 0.3, 0.0]), bijective=True),
                       O.categorical([0.1, -0.2, 0.3, 0.0], bijective=True))}


@pytest.mark.parametrize("scale", [1.0, 2.0 ** -20, 2.0 ** -6])
def test_plan_sweeps_match_oracle_at_any_feature_scale(A, oracle, scale):
    """The same SVGP problem with Phi multiplied by `scale` (and the residual by scale^2): the plan's images scale themselves, so
    the 1e-5 bar on the natural parameters after ten sweeps holds at 2^-20 and 2^-6 as it does at 1 (the unscaled marginal image
    of rounds 1-3 carried 2^-20 Phi in float16 subnormals)."""
    O = oracle
    lik, olik = A.BernoulliLikelihood(), O.bernoulli()
    ctx = A.Context(0, seed=1)
    N, M = 9000, 200
    _, y, Phi, kd = _svgp(A, ctx, lik, N, M)
    Phi = (Phi * scale).contiguous()
    kd = (kd.clamp_min(0) * scale * scale).contiguous()  # (the plan clamps a residual that float32 rounded below zero; so must the oracle's input)
    cavi = A.SparseCAVI(lik, Phi, kd, y, ctx=ctx)
    assert cavi.plan is not None
    Phi_h, kd_h, y_h = host(Phi), host(kd).astype(np.float64), host(y)
    Mp = Phi_h.shape[1]
    S, m = np.eye(Mp)[None], np.zeros((1, Mp))
    for _ in range(10):
        cavi.sweep()
        G, g = O.cavi_pass(olik, Phi_h, kd_h, y_h, -S, m)
        S, m = O.gaussian_update(G, g)
    cavi.check()
    assert relmax(host(cavi.G), G) < NAT_TOL, relmax(host(cavi.G), G)
    assert relmax(host(cavi.g), g) < NAT_TOL, relmax(host(cavi.g), g)
    mu, var = cavi.marginals()
    _, _, pts = O.cavi_pass(olik, Phi_h, kd_h, y_h, -S, m, want_points=True)
    assert np.abs(host(mu)[0] - pts["mu"][:, 0]).max() < 2e-5 * max(1.0, np.abs(pts["mu"]).max())
    assert np.abs(host(var)[0] - pts["var"][:, 0]).max() < 2e-5 * max(1.0, np.abs(pts["var"]).max())


def test_power_of_two_feature_scales_give_the_same_bits(A):
    """Both images of a plan carry 2^e Phi with e taken from max |Phi|: multiplying the features by 2^k changes e and nothing
    else, so the first-sweep marginals (U = I, v = 0: var = d + |phi|^2) and the accumulators of a pass scale by exact powers
    of two -- bit for bit, over the whole documented range of the images."""
    lik = A.BernoulliLikelihood()
    ctx = A.Context(0, seed=1)
    N, M = 5000, 256
    _, y, Phi, kd = _svgp(A, ctx, lik, N, M)
    kd = kd.clamp_min(0)
    ref = None
    for k in (0, -16, 10, 30):  # e = 14 - k stays inside the unclamped range -30 .. 30 of the image exponent
        s = 2.0 ** k
        cavi = A.SparseCAVI(lik, (Phi * s).contiguous(), (kd * s * s).contiguous(), y, ctx=ctx, keep_points=True)
        assert cavi.plan.scale_exp == A.SparseCAVI(lik, Phi, kd, y, ctx=ctx).plan.scale_exp - k
        mu, var = cavi.marginals()
        out = (var / (s * s)).clone()
        if ref is None:
            ref = out
        assert torch.equal(out, ref), k
        assert mu.abs().max().item() == 0.0


@pytest.mark.parametrize("scale", [2.0 ** -23, 1.0, 3.0e4, 2.0 ** 40])
def test_plan_pass_and_update_at_extreme_feature_scales(A, oracle, scale):
    """One pass + update + pass at feature scales over 63 octaves (the documented domain of the images is max |Phi| in
    2^-24 .. 2^44), each against the oracle on the same inputs: at 3e4 and 2^40 the posterior is so tight that U = chol(I + G)^-1
    has entries of 1e-5 .. 1e-14 -- float16 subnormals / zeros in an unscaled image; the plan's U images carry 2^15 U.  (Ten sweeps
    are not compared at these scales: the fixed point itself is ill-conditioned there.)"""
    O = oracle
    lik, olik = A.BernoulliLikelihood(), O.bernoulli()
    ctx = A.Context(0, seed=1)
    N, M = 6000, 200
    _, y, Phi, kd = _svgp(A, ctx, lik, N, M)
    Phi = (Phi.double() * scale).float().contiguous()
    kd = (kd.clamp_min(0).double() * scale * scale).float().contiguous()
    cavi = A.SparseCAVI(lik, Phi, kd, y, ctx=ctx)
    Phi_h, kd_h, y_h = host(Phi), host(kd).astype(np.float64), host(y)
    Mp = Phi_h.shape[1]
    S, m = np.eye(Mp)[None], np.zeros((1, Mp))
    G1, g1 = O.cavi_pass(olik, Phi_h, kd_h, y_h, -S, m)
    cavi.accumulate()
    assert relmax(host(cavi.G), G1) < 5e-6 and relmax(host(cavi.g), g1) < 5e-6
    cavi.update()
    cavi.check()
    # the second pass from the ORACLE's q(v) of the device's own (G, g): isolates the marginal pass from the M x M solve
    S, m = O.gaussian_update(host(cavi.G), host(cavi.g))
    _, _, pts = O.cavi_pass(olik, Phi_h, kd_h, y_h, -S, m, want_points=True)
    mu, var = cavi.marginals()
    assert np.abs(host(mu)[0] - pts["mu"][:, 0]).max() < 1e-4 * max(1e-300, np.abs(pts["mu"]).max())
    assert np.abs(host(var)[0] - pts["var"][:, 0]).max() < 1e-4 * max(1e-300, np.abs(pts["var"]).max())


@pytest.mark.parametrize("name,N,M", [("bernoulli", 20_000, 200), ("negbin", 6_000, 256), ("studentt", 5_000, 256),
                                      ("catbij", 3_000, 250)])
def test_elbo_rides_the_sweep(A, oracle, name, N, M):
    O = oracle
    lik, olik = _liks(A, O)[name]
    ctx_a, ctx_b = A.Context(0, seed=2), A.Context(0, seed=2)
    _, y, Phi, kd = _svgp(A, ctx_a, lik, N, M)
    a = A.SparseCAVI(lik, Phi, kd, y, ctx=ctx_a, track_elbo=True)  # ELBO terms ride the pass, the KL the update
    b = A.SparseCAVI(lik, Phi, kd, y, ctx=ctx_b)                   # evaluates the ELBO by separate passes
    sep, rode = [b.elbo()], []                                    # ELBO of the initial q(v) = N(0, I)
    for k in range(6):
        a.sweep()
        rode.append(a.elbo_entering())  # the q(v) that entered sweep k + 1 = after k updates
        b.sweep()
        sep.append(b.elbo())
    assert torch.equal(a.G, b.G) and torch.equal(a.g, b.g)  # tracking changes nothing of the sweep
    for k in range(6):
        assert rode[k] == pytest.approx(sep[k], rel=1e-9), (k, rode[k], sep[k])
    if name != "catbij":  # (the bijective categorical update uses approx_expected_logistic, utils.jl:11-14: not an exact ascent step)
        assert all(y2 >= y1 - 1e-6 * abs(y1) for y1, y2 in zip(rode, rode[1:])), rode
    # the oracle's aug_elbo for the q(v) that entered the last sweep (after 5 updates)
    Phi_h, kd_h, y_h = host(Phi), host(kd).astype(np.float64), host(y)
    if lik.ykind == "real":
        y_h = y_h.astype(np.float64)
    Mp, L = Phi_h.shape[1], olik.nlatent
    S, m = np.tile(np.eye(Mp), (L, 1, 1)), np.zeros((L, Mp))
    G = np.zeros((L, Mp, Mp))
    for _ in range(5):
        G, g = O.cavi_pass(olik, Phi_h, kd_h, y_h, -S, m)
        S, m = O.gaussian_update(G, g)
    _, _, pts = O.cavi_pass(olik, Phi_h, kd_h, y_h, -S, m, want_points=True)
    mu, var = pts["mu"], pts["var"]
    if L == 1:
        mu, var = mu[:, 0], var[:, 0]
    q1, q2, _ = O.aux_posterior(olik, y_h, mu, var)
    kl_v = sum(0.5 * (np.trace(S[l]) + m[l] @ m[l] - Mp + np.linalg.slogdet(np.eye(Mp) + G[l])[1]) for l in range(L))
    ref = O.expected_logtilt(olik, y_h, q1, q2, mu, var) - O.aux_kl(olik, y_h, q1, q2) - kl_v
    assert rode[5] == pytest.approx(ref, rel=2e-6), (rode[5], ref)


def test_elbo_tracking_is_refused_where_the_reference_defines_no_terms(A):
    ctx = A.Context(0, seed=3)
    lik = A.CategoricalLikelihood(np.zeros(4))  # non-bijective link: aux_kldivergence errors (categorical.jl:165-170)
    _, y, Phi, kd = _svgp(A, ctx, lik, 2000, 128)
    cavi = A.SparseCAVI(lik, Phi, kd, y, ctx=ctx, track_elbo=True)
    with pytest.raises(A.AGPLError, match="ELBO terms are not defined"):
        cavi.sweep()


def test_plan_argument_and_domain_errors(A):
    ctx = A.Context(0, seed=4)
    lib = A._ffi.lib()
    # any feature count (round 6): the plan pads to the next multiple of 256 itself -- a padded count costs only the staging copies
    b384, b512 = (lib.agpl_plan_bytes(C.c_int64(1000), C.c_int32(m), C.c_int32(1), C.c_uint32(0)) for m in (384, 512))
    assert b512 < b384 <= b512 + 8 * (512 * 512 + 3 * 512) + 1024
    assert lib.agpl_plan_bytes(C.c_int64(1000), C.c_int32(0), C.c_int32(1), C.c_uint32(0)) == 0
    assert lib.agpl_plan_bytes(C.c_int64(1000), C.c_int32(256), C.c_int32(65), C.c_uint32(0)) == 0
    assert 0 < lib.agpl_plan_bytes(C.c_int64(1000), C.c_int32(256), C.c_int32(1), C.c_uint32(1)) < lib.agpl_plan_bytes(
        C.c_int64(1000), C.c_int32(256), C.c_int32(1), C.c_uint32(0))  # AGPL_PLAN_NO_MARGINALS
    N, M = 1000, 256
    Phi = torch.zeros((N, M), dtype=torch.float32, device="cuda")
    resid = torch.ones(N, dtype=torch.float32, device="cuda")
    h = C.c_void_p()
    p = lambda t: C.c_void_p(t.data_ptr())
    with pytest.raises(A.ArgumentError, match="bad sizes"):
        ctx.call("agpl_plan_create", C.c_int64(N), C.c_int32(0), C.c_int32(1), p(Phi), p(resid), C.c_uint32(0), C.c_void_p(0), C.byref(h))
    Phi[617, 33] = float("nan")
    with pytest.raises(A.DomainError, match=r"point 617, feature 33"):
        ctx.call("agpl_plan_create", C.c_int64(N), C.c_int32(M), C.c_int32(1), p(Phi), p(resid), C.c_uint32(0), C.c_void_p(0), C.byref(h))
    Phi[617, 33] = 1e30  # >= 2^44: cannot be scaled into float16 with an exponent the unscaling can undo
    with pytest.raises(A.DomainError, match="outside the range"):
        ctx.call("agpl_plan_create", C.c_int64(N), C.c_int32(M), C.c_int32(1), p(Phi), p(resid), C.c_uint32(0), C.c_void_p(0), C.byref(h))
    Phi[617, 33] = 0.5
    # library-owned storage (storage = NULL), info, destroy
    ctx.call("agpl_plan_create", C.c_int64(N), C.c_int32(M), C.c_int32(1), p(Phi), p(resid), C.c_uint32(0), C.c_void_p(0), C.byref(h))
    n, m, l, e, b = C.c_int64(), C.c_int32(), C.c_int32(), C.c_int32(), C.c_int64()
    assert lib.agpl_plan_info(h, C.byref(n), C.byref(m), C.byref(l), C.byref(e), C.byref(b)) == 0
    assert (n.value, m.value, l.value) == (N, M, 1) and e.value == 14  # 2^14 x 0.5 = 2^13
    assert b.value == lib.agpl_plan_bytes(C.c_int64(N), C.c_int32(M), C.c_int32(1), C.c_uint32(0))
    lik = A.CategoricalLikelihood(np.zeros(4)).desc()
    G = torch.empty((1, M, M), dtype=torch.float64, device="cuda")
    g = torch.empty((1, M), dtype=torch.float64, device="cuda")
    y = torch.zeros((N, 4), dtype=torch.uint8, device="cuda")
    rc = lib.agpl_cavi_pass_plan(h, C.byref(lik), C.c_void_p(0), p(y), p(G), p(g), None, None, None, None)
    assert rc == A._ffi.ERR_INVALID_ARGUMENT  # 4 latents against a plan for 1
    # lifetime rule: the context refuses to go while a plan created on it is alive, and destroys nothing
    assert lib.agpl_ctx_destroy(ctx._h) == A._ffi.ERR_INVALID_ARGUMENT and b"still alive" in lib.agpl_last_error(ctx._h)
    assert lib.agpl_plan_destroy(h) == 0
    ctx.synchronize()  # the context is intact


def image_features(Phi):
    """The features the plan's images hold: x' = (hi + lo) 2^-e with hi = f16(2^e x), lo = f16(2^e x - hi), e from max |Phi|
    (agpl_syrk.hip agpl_image_scale_exp / accumulate_image_kernel), restated in numpy."""
    Phi = np.asarray(Phi, dtype=np.float32)
    mx = float(np.abs(Phi).max())
    e = min(13 - int(np.floor(np.log2(mx))), 30) if mx > 0 else 0
    xs = Phi * np.float32(2.0 ** e)
    hi = xs.astype(np.float16)
    lo = (xs - hi.astype(np.float32)).astype(np.float16)
    return ((hi.astype(np.float32) + lo.astype(np.float32)).astype(np.float64) * 2.0 ** -e), e


@pytest.mark.parametrize("name,M", [("bernoulli", 256), ("negbin", 512), ("cat", 256), ("studentt", 256)])
def test_gibbs_pass_plan_matches_oracle_on_the_image_features(A, oracle, name, M):
    """agpl_gibbs_pass_plan: projection from the accumulate image + per-point Philox streams + aux_sample! + accumulation
    (examples/bernoulli/script.jl:81-84 in sparse form) against the oracle's pass on the SAME features (the image's): f to 1e-12,
    omega to 1e-9, counts and uniforms consumed bit for bit."""
    O = oracle
    liks = {"bernoulli": (A.BernoulliLikelihood(), O.bernoulli()), "negbin": (A.NegativeBinomialLikelihood(15.0), O.negbinomial(15.0)),
            "cat": (A.CategoricalLikelihood(np.array([0.1, -0.2, 0.3, 0.0])), O.categorical([0.1, -0.2, 0.3, 0.0])),
            "studentt": (A.StudentTLikelihood(3.5, 2.0), O.studentt(3.5, 2.0))}
    lik, olik = liks[name]
    rng = np.random.default_rng(17)
    N, L = 3001, olik.nlatent
    Phi = (rng.standard_normal((N, M)) * 0.15).astype(np.float32)
    kd = rng.uniform(0.0, 0.3, size=N).astype(np.float32)
    if name == "bernoulli":
        y = (rng.uniform(size=N) < 0.5).astype(np.uint8)
    elif name == "negbin":
        y = rng.poisson(4.0, size=N).astype(np.int32)
    elif name == "cat":
        y = (rng.integers(0, L, size=N)[:, None] == np.arange(L)[None, :]).astype(np.uint8)
    else:
        y = rng.normal(size=N)
    ctx = A.Context(0, seed=SEED)
    gib = A.SparseGibbs(lik, torch.from_numpy(Phi).cuda(), torch.from_numpy(kd).cuda(), torch.from_numpy(y).cuda(), ctx=ctx,
                        keep_points=True)
    assert gib.plan is not None
    v = rng.normal(size=(L, M))
    gib.v.copy_(torch.from_numpy(v).cuda())
    gib.accumulate()
    Phi_img, e = image_features(Phi)
    assert gib.plan.scale_exp == e
    Gr, gr, pts = O.gibbs_pass(olik, Phi_img.astype(np.float32), kd.astype(np.float64), y, v, seed=SEED, sweep=gib.sweep_index)
    f = host(gib.f)
    assert np.abs(f - pts["f"]).max() < 1e-12 * max(1.0, np.abs(pts["f"]).max())
    assert np.allclose(host(gib.omega), pts["omega"], rtol=1e-9, atol=0)
    if name == "cat":
        assert np.array_equal(host(gib.n), pts["n"])
    assert relmax(host(gib.G), Gr) < 5e-6 and relmax(host(gib.g), gr) < 5e-6


# ---- round 6: one fast path for every feature count -------------------------------------------------------
def _svgp_raw(A, ctx, lik, N, M):
    """The SVGP workload at its OWN feature count: no padding on the host (se_features / whiten_features work on a multiple of
    128; the first M columns are the features)."""
    _, y, Phi, kd = _svgp(A, ctx, lik, N, M, pad=128)
    return y, Phi[:, :M].contiguous(), kd


@pytest.mark.parametrize("name,N,M", [("bernoulli", 10_000, 64), ("bernoulli", 9_000, 200), ("negbin", 7_000, 200), ("bernoulli", 5_003, 37),
                                      ("bernoulli", 3_000, 1280), ("bernoulli", 2_000, 2048), ("catbij", 4_000, 100)])
def test_plan_path_at_any_feature_count_ten_sweeps(A, oracle, name, N, M):
    """`S .= inv(Symmetric(inv(K) + Diagonal(λ)))`, `m .= S * (h + K \\ mean(fz))` (examples/bernoulli/script.jl:35-36;
    docs/src/index.md:154-163) has no shape restriction.  Rounds 3-5 sent M % 256 != 0 to the float32-input kernels (4 x slower) unless
    the caller zero-padded; now agpl_plan_create pads the images itself and G, g, U, v come back at the caller's M: ten CAVI sweeps
    at M = 64 (BASELINE C1's count), 200, 37 (not a multiple of 4: ragged rows), 1280 and 2048 (beyond the one-launch
    factorisation: two block rows of it) against the oracle's float64 sweep on the same features."""
    O = oracle
    lik, olik = _liks(A, O)[name]
    ctx = A.Context(0, seed=5)
    y, Phi, kd = _svgp_raw(A, ctx, lik, N, M)
    assert Phi.shape[1] == M
    cavi = A.SparseCAVI(lik, Phi, kd, y, ctx=ctx)
    assert cavi.plan is not None and cavi.plan.Mp % 256 == 0 and cavi.plan.Mp - M < 256
    L = olik.nlatent
    assert tuple(cavi.G.shape) == (L, M, M) and tuple(cavi.g.shape) == (L, M)
    Phi_h, kd_h, y_h = host(Phi), host(kd.clamp_min(0)).astype(np.float64), host(y)
    S, m = np.tile(np.eye(M), (L, 1, 1)), np.zeros((L, M))
    for it in range(10):
        cavi.sweep()
        G, g = O.cavi_pass(olik, Phi_h, kd_h, y_h, -S, m)
        S, m = O.gaussian_update(G, g)
    cavi.check()
    assert relmax(host(cavi.G), G) < NAT_TOL, relmax(host(cavi.G), G)
    assert relmax(host(cavi.g), g) < NAT_TOL, relmax(host(cavi.g), g)
    # q(v) in the caller's size: S = U'U, m = U'v from the leading blocks of the plan's state; beyond M the state is the identity
    # (the M x M solve amplifies the natural-parameter difference by cond(I + G): the bar of test_cavi_factor_form_matches_oracle)
    kappa = max(np.linalg.cond(np.eye(M) + G[l]) for l in range(L))
    assert relmax(host(cavi.S), S) < max(1e-4, NAT_TOL * kappa) and relmax(host(cavi.m), m) < max(1e-4, NAT_TOL * kappa)
    Mp = cavi.plan.Mp
    if Mp != M:
        Ufull = host(torch.triu(cavi.plan.U_colmajor))
        assert np.array_equal(Ufull[:, M:, M:], np.tile(np.eye(Mp - M), (L, 1, 1))) and not Ufull[:, :M, M:].any()
        assert not host(cavi.plan.v)[:, M:].any()
    mu, var = cavi.marginals()
    _, _, pts = O.cavi_pass(olik, Phi_h, kd_h, y_h, -S, m, want_points=True)
    assert np.abs(host(mu).T.reshape(pts["mu"].shape) - pts["mu"]).max() < 2e-5 * max(1.0, np.abs(pts["mu"]).max())
    assert np.abs(host(var).T.reshape(pts["var"].shape) - pts["var"]).max() < 2e-5 * max(1.0, np.abs(pts["var"]).max())


def test_gibbs_pass_plan_at_a_ragged_feature_count(A, oracle):
    """The Gibbs pass on a plan with M = 200 (padded to 256 inside): the caller's v, G, g are M-sized."""
    O = oracle
    lik, olik = A.NegativeBinomialLikelihood(15.0), O.negbinomial(15.0)
    rng = np.random.default_rng(23)
    N, M = 3001, 200
    Phi = (rng.standard_normal((N, M)) * 0.15).astype(np.float32)
    kd = rng.uniform(0.0, 0.3, size=N).astype(np.float32)
    y = rng.poisson(4.0, size=N).astype(np.int32)
    ctx = A.Context(0, seed=SEED)
    gib = A.SparseGibbs(lik, torch.from_numpy(Phi).cuda(), torch.from_numpy(kd).cuda(), torch.from_numpy(y).cuda(), ctx=ctx,
                        keep_points=True)
    assert gib.plan is not None and gib.plan.Mp == 256 and tuple(gib.G.shape) == (1, M, M)
    v = rng.normal(size=(1, M))
    gib.v.copy_(torch.from_numpy(v).cuda())
    gib.accumulate()
    Phi_img, e = image_features(Phi)
    Gr, gr, pts = O.gibbs_pass(olik, Phi_img.astype(np.float32), kd.astype(np.float64), y, v, seed=SEED, sweep=gib.sweep_index)
    assert np.abs(host(gib.f) - pts["f"]).max() < 1e-12 * max(1.0, np.abs(pts["f"]).max())
    assert np.allclose(host(gib.omega), pts["omega"], rtol=1e-9, atol=0)
    assert relmax(host(gib.G), Gr) < 5e-6 and relmax(host(gib.g), gr) < 5e-6
    gib.exchange()
    gib.draw()  # v ~ N(m, S) at the caller's M
    assert tuple(gib.v.shape) == (1, M) and torch.isfinite(gib.v).all()
