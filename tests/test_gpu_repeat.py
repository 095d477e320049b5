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
