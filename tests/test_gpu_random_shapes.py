"""Seeded random shapes through the two shipped contractions (run with -m gpu): the fixed parametrisations of
test_gpu_parity.py pin the shapes that broke a kernel once; this file walks shapes nobody picked -- point counts on
and around every granularity the kernels have (16-point stages, 128 / 256-point tiles, 4096-point slices, the
1024-workgroup round), feature counts 256..1024, 1-3 latents -- against float64 numpy.

  * factor-form marginals   agpl_gaussian_factor + agpl_marginals_factor_split   (marginal_factor_queue_kernel)
  * split accumulation      agpl_accumulate, precision 1                         (syrk_split_kernel)
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


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
    return A.Context(0, seed=99)


def p(t):
    return C.c_void_p(t.data_ptr())


def shapes(kind):
    rng = np.random.default_rng({"factor": 1, "accumulate": 2}[kind])
    edges = [1, 2, 15, 16, 17, 63, 64, 65, 127, 128, 129, 255, 256, 257, 511, 513, 4095, 4096, 4097, 8191, 8193, 12289]
    out = []
    for k in range(14):
        N = int(rng.choice(edges)) if k % 2 == 0 else int(rng.integers(300, 60_000))
        M = int(rng.choice([256, 512, 768, 1024] if kind == "factor" else [128, 256, 384, 512, 640, 1024]))
        L = int(rng.choice([1, 1, 2, 3]))
        if N * M * L > 40_000_000:  # keep the float64 reference in seconds
            N = 40_000_000 // (M * L)
        out.append((N, M, L))
    return out


@pytest.mark.parametrize("N,M,L", shapes("factor"))
def test_factor_marginals_random_shape(A, ctx, N, M, L):
    from agpl_amd import _ffi

    rng = np.random.default_rng(N * 7 + M + L)
    Phi = (rng.normal(size=(N, M)) * 0.3).astype(np.float32)
    P = Phi.astype(np.float64)
    kd = (np.sum(P ** 2, axis=1) + rng.uniform(0.01, 0.5, size=N)).astype(np.float32)
    B = rng.normal(size=(L, M, 2 * M)) / np.sqrt(2 * M)
    G = np.einsum("lik,ljk->lij", B, B) * 3.0
    g = rng.normal(size=(L, M))
    mu0 = rng.normal(size=(L, N)).astype(np.float32)
    dPhi, dkd, dmu0 = (torch.from_numpy(a).cuda() for a in (Phi, kd, mu0))
    dG, dg = torch.from_numpy(G).cuda(), torch.from_numpy(g).cuda()
    nh = _ffi.lib().agpl_split_features_bytes(C.c_int64(N), C.c_int32(M)) // 2
    Ph = torch.empty(nh, dtype=torch.float16, device="cuda")
    Pl = torch.empty(nh, dtype=torch.float16, device="cuda")
    ctx.call("agpl_split_features", C.c_int64(N), C.c_int32(M), p(dPhi), p(Ph), p(Pl))
    resid = torch.empty(N, dtype=torch.float32, device="cuda")
    ctx.call("agpl_feature_residual", C.c_int64(N), C.c_int32(M), p(dPhi), p(dkd), p(resid))
    Aw = torch.empty((L, M, M), dtype=torch.float64, device="cuda")
    v = torch.empty((L, M), dtype=torch.float64, device="cuda")
    v32 = torch.empty((L, M), dtype=torch.float32, device="cuda")
    Uh = torch.empty(L * M * M, dtype=torch.float16, device="cuda")
    Ul = torch.empty(L * M * M, dtype=torch.float16, device="cuda")
    ctx.call("agpl_gaussian_factor", C.c_int32(M), C.c_int32(L), p(dG), p(dg), C.c_void_p(0), p(Aw), p(v), p(v32), p(Uh),
             p(Ul), C.c_void_p(0))
    # canaries behind the outputs: a kernel that writes past row N shows up here, not in someone else's tensor
    mu = torch.full((L * N + 64,), -7.0, dtype=torch.float32, device="cuda")
    var = torch.full((L * N + 64,), -7.0, dtype=torch.float32, device="cuda")
    ctx.call("agpl_marginals_factor_split", C.c_int64(N), C.c_int32(M), C.c_int32(L), p(Ph), p(Pl), p(resid), p(dmu0),
             p(Uh), p(Ul), p(v32), p(mu), p(var))
    ctx.synchronize()
    muh, varh = mu.cpu().numpy(), var.cpu().numpy()
    assert np.all(muh[L * N:] == -7.0) and np.all(varh[L * N:] == -7.0)
    for l in range(L):
        S = np.linalg.inv(np.eye(M) + G[l])
        m = S @ g[l]
        ref_mu = mu0[l].astype(np.float64) + P @ m
        ref_var = kd.astype(np.float64) - np.einsum("ia,ab,ib->i", P, np.eye(M) - S, P)
        assert np.abs(muh[l * N:(l + 1) * N] - ref_mu).max() < 2e-6 * np.abs(P).sum(1).max() * np.abs(m).max() + 1e-6
        assert np.abs(varh[l * N:(l + 1) * N] - ref_var).max() < 3e-6 * max(1.0, np.abs(ref_var).max())
    # same inputs, same bits
    mu2, var2 = torch.empty_like(mu), torch.empty_like(var)
    ctx.call("agpl_marginals_factor_split", C.c_int64(N), C.c_int32(M), C.c_int32(L), p(Ph), p(Pl), p(resid), p(dmu0),
             p(Uh), p(Ul), p(v32), p(mu2), p(var2))
    ctx.synchronize()
    assert torch.equal(mu2[: L * N], mu[: L * N]) and torch.equal(var2[: L * N], var[: L * N])


@pytest.mark.parametrize("N,M,L", shapes("accumulate"))
def test_split_accumulate_random_shape(A, ctx, N, M, L):
    rng = np.random.default_rng(N * 5 + M + L)
    Phi = (rng.normal(size=(N, M)) * 0.3).astype(np.float32)
    gamma = rng.uniform(0.0, 0.25, size=(L, N)).astype(np.float32)
    gamma[:, rng.integers(0, N, size=max(1, N // 50))] = 0.0  # zero precisions happen (Poisson)
    beta = rng.normal(size=(L, N)).astype(np.float32)
    dPhi, dbeta, dgamma = (torch.from_numpy(a).cuda() for a in (Phi, beta, gamma))
    G = torch.empty((L, M, M), dtype=torch.float64, device="cuda")
    g = torch.empty((L, M), dtype=torch.float64, device="cuda")
    args = (C.c_int64(N), C.c_int32(M), C.c_int32(L), p(dPhi), C.c_void_p(0), p(dbeta), p(dgamma), p(G), p(g))  # (no image)
    ctx.call("agpl_accumulate_split", *args)
    ctx.synchronize()
    G1, g1 = G.cpu().numpy().copy(), g.cpu().numpy().copy()
    ctx.call("agpl_accumulate_split", *args)
    ctx.synchronize()
    P = Phi.astype(np.float64)
    for l in range(L):
        Gr = (P * gamma[l].astype(np.float64)[:, None]).T @ P
        gr = P.T @ beta[l].astype(np.float64)
        assert np.abs(G1[l] - Gr).max() <= 5e-6 * max(np.abs(Gr).max(), 1e-30)
        assert np.abs(g1[l] - gr).max() <= 5e-6 * max(np.abs(gr).max(), 1e-30)
    assert np.array_equal(G1, G1.transpose(0, 2, 1))
    assert np.array_equal(G.cpu().numpy(), G1) and np.array_equal(g.cpu().numpy(), g1)


def plan_shapes():
    rng = np.random.default_rng(3)
    edges = [1, 2, 15, 16, 17, 31, 33, 127, 129, 255, 256, 257, 4095, 4097, 8193]
    out = []
    for k in range(12):
        N = int(rng.choice(edges)) if k % 2 == 0 else int(rng.integers(300, 40_000))
        M = int(rng.choice([256, 512, 768, 1024]))
        L = int(rng.choice([1, 1, 2, 3]))
        if N * M * L > 30_000_000:
            N = 30_000_000 // (M * L)
        out.append((N, M, L))
    return out


@pytest.mark.parametrize("N,M,L", plan_shapes())
def test_plan_update_marginals_and_pass_random_shape(A, ctx, N, M, L):
    """The plan API (agpl_plan_create / _update / agpl_marginals_plan / agpl_cavi_pass_plan) at ragged point counts (a single
    point, one short of / one past every tile size of the two images) for all feature counts it accepts and several latents,
    against float64 references built from the SAME float32 inputs: q(v) after an update with a random G, g (U U' = (I + G)^-1,
    v = U (g)), the marginals mu = Phi' U' v, var = d + |U phi|^2 (docs/src/index.md:154-163), and the accumulators of a pass
    from the gamma, beta it exports."""
    gen = torch.Generator(device="cuda").manual_seed(1000 * N + M + L)
    Phi = (torch.randn((N, M), device="cuda", generator=gen) * 0.2).contiguous()
    kd = torch.rand(N, device="cuda", generator=gen) * 0.5
    lik = A.BernoulliLikelihood() if L == 1 else A.CategoricalLikelihood(np.zeros(L + 1), bijective=True)
    assert A.nlatent(lik) == L
    plan = A.sparse.Plan(Phi, kd, L, ctx)
    # a random posterior: G = B B' (rank 40) scaled, g random
    B = torch.randn((L, M, 40), dtype=torch.float64, device="cuda", generator=gen)
    G = (B @ B.transpose(1, 2) * 3.0).contiguous()
    g = torch.randn((L, M), dtype=torch.float64, device="cuda", generator=gen).contiguous()
    plan.call("agpl_plan_update", p(G), p(g), C.c_void_p(0), C.c_void_p(0))
    ctx.synchronize()
    U = torch.tril(plan.U_colmajor.transpose(1, 2))  # column-major lower triangle -> U[a][b]
    eye = torch.eye(M, dtype=torch.float64, device="cuda")
    S_ref = torch.linalg.inv(eye[None] + G)
    S = U.transpose(1, 2) @ U
    assert (S - S_ref).abs().max().item() < 1e-10 * S_ref.abs().max().item()
    m_ref = torch.einsum("lab,lb->la", S_ref, g)
    m = torch.einsum("lba,lb->la", U, plan.v)
    assert (m - m_ref).abs().max().item() < 1e-9 * max(1.0, m_ref.abs().max().item())
    # marginals from the images
    mu = torch.empty((L, N), dtype=torch.float32, device="cuda")
    var = torch.empty((L, N), dtype=torch.float32, device="cuda")
    plan.call("agpl_marginals_plan", C.c_void_p(0), p(mu), p(var))
    ctx.synchronize()
    P64 = Phi.double()
    T = torch.einsum("lab,nb->lan", U, P64)
    var_ref = kd.double()[None] + (T * T).sum(1)
    mu_ref = torch.einsum("la,lan->ln", plan.v, T)
    assert (var.double() - var_ref).abs().max().item() < 2e-5 * var_ref.abs().max().item()
    assert (mu.double() - mu_ref).abs().max().item() < 2e-5 * max(1.0, mu_ref.abs().max().item())
    # a CAVI pass on the plan: accumulators against the exported gamma, beta
    if L == 1:
        y = (torch.rand(N, device="cuda", generator=gen) < 0.5).to(torch.uint8)
    else:
        idx = torch.randint(0, L + 1, (N,), device="cuda", generator=gen)
        y = (idx[:, None] == torch.arange(L, device="cuda")[None, :]).to(torch.uint8).contiguous()
    Gp = torch.empty((L, M, M), dtype=torch.float64, device="cuda")
    gp = torch.empty((L, M), dtype=torch.float64, device="cuda")
    cc = torch.empty((L, N), dtype=torch.float32, device="cuda")
    gam = torch.empty((L, N), dtype=torch.float32, device="cuda")
    bet = torch.empty((L, N), dtype=torch.float32, device="cuda")
    d = lik.desc()
    plan.call("agpl_cavi_pass_plan", C.byref(d), C.c_void_p(0), p(y), p(Gp), p(gp), p(cc), p(gam), p(bet), C.c_void_p(0))
    ctx.synchronize()
    assert torch.isfinite(gam).all() and (gam >= 0).all()
    G_ref = torch.einsum("ln,na,nb->lab", gam.double(), P64, P64)
    g_ref = torch.einsum("ln,na->la", bet.double(), P64)
    # (5e-6: the bar of the stand-alone accumulation test above -- float32 accumulation over 4096-point slices)
    assert (Gp - G_ref).abs().max().item() < 5e-6 * max(G_ref.abs().max().item(), 1e-30)
    assert (gp - g_ref).abs().max().item() < 5e-6 * max(g_ref.abs().max().item(), 1e-30)
