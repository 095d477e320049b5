"""Bitwise run-to-run repeatability of every hot kernel at occupancies that matter (the
g = Phi beta sums of the float32-staged split kernel once differed from run to run with compiler-formed packed-float32 FMAs,
only when four workgroups shared every CU -- DESIGN 4.4d item 8).  30 launches per case on the same inputs, every output
compared bit for bit with the first: the image accumulation (shipped), the float32-staged split tile kernel (fallback for
M % 256 != 0; run here at 4 workgroups per CU), the f32 kernel, the factor update, the marginal pass and the Gibbs point
pass, at three shapes (C2-like M = 512, north-star M = 1024, multi-latent M = 256)."""
import ctypes as C

import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
REPEATS = 30


@pytest.fixture(scope="module")
def A():
    import agpl_amd as A

    return A


def _count(run, snapshot):
    run()
    torch.cuda.synchronize()
    ref = [t.clone() for t in snapshot() if t is not None]
    bad = 0
    for _ in range(REPEATS):
        run()
        torch.cuda.synchronize()
        if not all(torch.equal(a, b) for a, b in zip([t for t in snapshot() if t is not None], ref)):
            bad += 1
    return bad


@pytest.mark.timeout(900)
@pytest.mark.parametrize("likname,N,M", [("bernoulli", 3_000_000, 512), ("negbin", 1_000_000, 1024),
                                         ("categorical", 400_000, 256)])
def test_every_hot_kernel_repeats_bit_for_bit(A, likname, N, M):
    import bench

    ctx = A.Context(0, seed=bench.SEED)
    lik = bench.make_lik(A, likname)
    y, Phi, kd = bench.build_workload(A, ctx, lik, 0, N, M)
    cavi = A.SparseCAVI(lik, Phi, kd, y, ctx=ctx, keep_points=True)  # the shipped path: factor marginals + image accumulation
    assert cavi.factor and cavi.plan is not None
    cavi.sweep()
    cavi.sweep()
    cavi.check()
    assert _count(cavi.accumulate, lambda: (cavi.G, cavi.g, cavi.gamma, cavi.beta, cavi.c)) == 0
    assert _count(cavi.update, lambda: (cavi.A_work, cavi.v, cavi.plan.v32, cavi.plan.U_hi, cavi.plan.U_lo, cavi.plan.logdet)) == 0
    mv = {}

    def marg():
        mv["m"] = cavi.marginals()

    assert _count(marg, lambda: mv["m"]) == 0
    # the float32-fed accumulation kernel on the same (gamma, beta)
    L = A.nlatent(lik)
    G = torch.empty((L, M, M), dtype=torch.float64, device="cuda")
    g = torch.empty((L, M), dtype=torch.float64, device="cuda")
    p = lambda t: C.c_void_p(t.data_ptr())
    run = lambda: ctx.call("agpl_accumulate", C.c_int64(N), C.c_int32(M), C.c_int32(L), p(Phi), p(cavi.beta), p(cavi.gamma), p(G), p(g))
    assert _count(run, lambda: (G, g)) == 0, "agpl_accumulate (float32 MFMA)"
    del cavi
    gib = A.SparseGibbs(lik, Phi, kd, y, ctx=ctx, keep_points=True)
    gib.sweep()

    def gpass():
        ctx.sweep = 7  # the same Philox streams every time
        gib.accumulate()

    assert _count(gpass, lambda: (gib.G, gib.g, gib.f, gib.omega)) == 0


@pytest.mark.timeout(600)
@pytest.mark.parametrize("N,M,L", [(1_000_000, 256, 2), (2_000_000, 512, 1), (700_001, 384, 1)])
def test_slab_reduction_second_level_inside_the_launch(A, N, M, L):
    """Round 6: the second level of the slab reduction (groups of 64 slices -> G, g) runs inside reduce_slab_kernel, by the workgroup
    that arrives last at a segment's counter.  60 launches into NaN-filled outputs: every element written (a skipped segment would
    keep its NaN), symmetric, equal to float64 within the float32 kernel's bar, and the same bits every time (the sum runs over the
    groups in index order whoever performs it; a partial read before it was visible would differ).  tools/soak_reduce.py is the
    long form (profiles/r06_soak_reduce.json)."""
    ctx = A.Context(0, seed=5)
    gen = torch.Generator(device="cuda").manual_seed(17)
    Phi = (torch.randn((N, M), dtype=torch.float32, device="cuda", generator=gen) / M ** 0.5).contiguous()
    gamma = torch.rand((L, N), dtype=torch.float32, device="cuda", generator=gen) * 0.25
    beta = torch.randn((L, N), dtype=torch.float32, device="cuda", generator=gen)
    G = torch.empty((L, M, M), dtype=torch.float64, device="cuda")
    g = torch.empty((L, M), dtype=torch.float64, device="cuda")
    p = lambda t: C.c_void_p(t.data_ptr())
    ref = None
    for i in range(60):
        G.fill_(float("nan"))
        g.fill_(float("nan"))
        ctx.call("agpl_accumulate", C.c_int64(N), C.c_int32(M), C.c_int32(L), p(Phi), p(beta), p(gamma), p(G), p(g))
        torch.cuda.synchronize()
        if ref is None:
            assert bool(torch.isfinite(G).all()) and bool(torch.isfinite(g).all())
            assert torch.equal(G, G.transpose(1, 2))
            for l in range(L):
                Gr = torch.zeros((M, M), dtype=torch.float64, device="cuda")
                for s in range(0, N, 250_000):
                    P = Phi[s:s + 250_000].double()
                    Gr += (P * gamma[l, s:s + 250_000].double().unsqueeze(1)).T @ P
                assert float(((G[l] - Gr).abs().max() / Gr.abs().max()).item()) < 1e-5
                gr = Phi.double().T @ beta[l].double()
                assert float(((g[l] - gr).abs().max() / gr.abs().max()).item()) < 1e-5
            ref = (G.clone(), g.clone())
        else:
            assert torch.equal(G, ref[0]) and torch.equal(g, ref[1]), f"launch {i} differs"
