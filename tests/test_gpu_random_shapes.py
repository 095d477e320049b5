"""Seeded random shapes through the two shipped contractions (run with -m gpu): the fixed parametrisations of
test_gpu_parity.py pin the shapes that broke a kernel once; this file walks shapes nobody picked -- point counts on
and around every granularity the kernels have (16-point stages, 128 / 256-point tiles, 4096-point slices, the
1024-workgroup round), feature counts 256..1024, 1-3 latents -- against float64 references, through the plan API:

  * factor-form marginals   agpl_plan_update + agpl_marginals_plan   (factor kernel, marginal_factor_queue_kernel)
  * image accumulation      agpl_cavi_pass_plan                      (agpl_fused_point_kernel, syrk_strip_kernel)
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


def plan_shapes():
    rng = np.random.default_rng(3)
    edges = [1, 2, 15, 16, 17, 31, 33, 63, 64, 65, 127, 128, 129, 255, 256, 257, 511, 513, 4095, 4096, 4097, 8191, 8193, 12289]
    out = []
    for k in range(24):
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
    # marginals from the images, with a prior mean; canaries behind the outputs: a kernel that writes past point N shows up here,
    # not in someone else's tensor
    mu0 = torch.randn((L, N), device="cuda", generator=gen)
    mu_buf = torch.full((L * N + 64,), -7.0, dtype=torch.float32, device="cuda")
    var_buf = torch.full((L * N + 64,), -7.0, dtype=torch.float32, device="cuda")
    plan.call("agpl_marginals_plan", p(mu0), p(mu_buf), p(var_buf))
    ctx.synchronize()
    assert bool((mu_buf[L * N:] == -7.0).all()) and bool((var_buf[L * N:] == -7.0).all())
    mu, var = mu_buf[: L * N].view(L, N), var_buf[: L * N].view(L, N)
    P64 = Phi.double()
    T = torch.einsum("lab,nb->lan", U, P64)
    var_ref = kd.double()[None] + (T * T).sum(1)
    mu_ref = mu0.double() + torch.einsum("la,lan->ln", plan.v, T)
    assert (var.double() - var_ref).abs().max().item() < 2e-5 * var_ref.abs().max().item()
    assert (mu.double() - mu_ref).abs().max().item() < 2e-5 * max(1.0, mu_ref.abs().max().item())
    mu2, var2 = torch.empty_like(mu_buf), torch.empty_like(var_buf)  # same inputs, same bits
    plan.call("agpl_marginals_plan", p(mu0), p(mu2), p(var2))
    ctx.synchronize()
    assert torch.equal(mu2[: L * N], mu_buf[: L * N]) and torch.equal(var2[: L * N], var_buf[: L * N])
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
