"""Two ranks on ONE GPU (gloo exchange) running the device sweep on their shards, against the single-process
device result: exercises shard_range + SparseCAVI(group=...) + exchange_natural_parameters end to end on the HIP
path.  (RCCL refuses two ranks on one device, so the single-GPU box uses gloo for the all-reduce; the 8-GPU
runs of the driver use backend "nccl" = RCCL through the same code.)"""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SEED = 20240807


LIKS = {"bernoulli": lambda A: A.BernoulliLikelihood(), "negbin": lambda A: A.NegativeBinomialLikelihood(15.0)}


def _setup(A, ctx, lik, i0, n, M):
    x, y = A.synth_xy(lik, SEED, i0, n, ctx=ctx)
    z = np.linspace(-10, 10, M)
    ell = 1.5 * (z[1] - z[0])
    Kzz = np.exp(-0.5 * ((z[:, None] - z[None, :]) / ell) ** 2)
    _, Linv = A.sparse.whitening_matrix(Kzz, 1e-8)
    Kzx = A.se_features(x, torch.from_numpy(z).cuda(), ell, ctx=ctx)
    Phi = A.whiten_features(Kzx, Linv, ctx=ctx)
    kd = A.sparse.nystrom_residual(Phi, torch.ones(n, device="cuda"), ctx=ctx)
    if Phi.shape[1] % 256:  # the factor form runs on 256-row blocks; zero columns change nothing
        Phi = torch.nn.functional.pad(Phi, (0, 256 - Phi.shape[1] % 256)).contiguous()
    return Phi, kd, y


def _worker(rank, world, port, N, M, nsweeps, q, kw, likname="bernoulli"):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist

    import agpl_amd as A

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ctx = A.Context(0, seed=SEED)
        lik = LIKS[likname](A)
        i0, i1 = A.shard_range(N, rank, world)
        Phi, kd, y = _setup(A, ctx, lik, i0, i1 - i0, M)
        cavi = A.SparseCAVI(lik, Phi, kd, y, ctx=ctx, group=dist.group.WORLD, **kw)
        cavi.run(nsweeps)
        torch.cuda.synchronize()
        q.put((rank, cavi.G.cpu().numpy(), cavi.g.cpu().numpy(), cavi.m.cpu().numpy()))
    finally:
        dist.destroy_process_group()


SHIPPED = {"marginal_precision": "f16x2-factor", "accumulate_precision": "f16x2"}


@pytest.mark.timeout(600)
@pytest.mark.parametrize("likname,N,M,kw,world", [
    ("bernoulli", 30_001, 64, {"marginal_precision": "f32", "accumulate_precision": "f32"}, 2),
    ("bernoulli", 30_001, 200, SHIPPED, 2),
    # BASELINE config C3's likelihood and M, sharded: NegBin r = 15, M = 1024 (the M = 1024 factor route on every rank)
    ("negbin", 20_000, 1024, SHIPPED, 2),
    # ... and at C3's own rank count: eight ranks, ragged shards (20 003 = 8 x 2500 + 3)
    ("negbin", 20_003, 512, SHIPPED, 8)])
def test_ranks_on_one_gpu_match_single_process(likname, N, M, kw, world):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import torch.multiprocessing as mp

    import __graft_entry__ as g

    g.build()
    import agpl_amd as A

    nsweeps = 4
    port = 29600 + (os.getpid() % 1000)
    mpctx = mp.get_context("spawn")
    q = mpctx.Queue()
    procs = [mpctx.Process(target=_worker, args=(r, world, port, N, M, nsweeps, q, kw, likname)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=500) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ctx = A.Context(0, seed=SEED)
    lik = LIKS[likname](A)
    Phi, kd, y = _setup(A, ctx, lik, 0, N, M)
    ref = A.SparseCAVI(lik, Phi, kd, y, ctx=ctx, **kw)
    ref.run(nsweeps)
    G, gg, m = ref.G.cpu().numpy(), ref.g.cpu().numpy(), ref.m.cpu().numpy()
    # every rank holds the identical reduced natural parameters and hence the identical update
    for r in res[1:]:
        assert np.array_equal(res[0][1], r[1]) and np.array_equal(res[0][2], r[2]) and np.array_equal(res[0][3], r[3])
    # sharded == unsharded up to the float32 slab partition (different 4096-point slices per rank)
    assert np.abs(res[0][1] - G).max() / np.abs(G).max() < 1e-5
    assert np.abs(res[0][2] - gg).max() / np.abs(gg).max() < 1e-5


def _gibbs_worker(rank, world, port, N, M, nsweeps, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist

    import agpl_amd as A

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ctx = A.Context(0, seed=SEED)  # the SAME seed on every rank: identical v, disjoint per-point streams
        lik = A.NegativeBinomialLikelihood(15.0)
        i0, i1 = A.shard_range(N, rank, world)
        Phi, kd, y = _setup(A, ctx, lik, i0, i1 - i0, M)
        gib = A.SparseGibbs(lik, Phi, kd, y, ctx=ctx, group=dist.group.WORLD, keep_points=True, point_offset=i0)
        vs, first = [gib.v.cpu().numpy().copy()], None
        for s in range(nsweeps):
            vs.append(gib.sweep().cpu().numpy().copy())
            if s == 0:  # the per-point draws of the first sweep, where every input is identical to the unsharded run
                first = (gib.f.cpu().numpy().copy(), gib.omega.cpu().numpy().copy(), gib.G.cpu().numpy().copy())
        torch.cuda.synchronize()
        q.put((rank, i0, i1, np.stack(vs)) + first)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", [2, 8])
def test_sharded_sparse_gibbs_is_the_single_process_chain(world):
    """N-sharded Gibbs (SparseGibbs(group=..., point_offset=...)): the per-point Philox streams are keyed on the GLOBAL
    point index and every rank draws the identical v, so the sharded chain IS the single-process chain: in the first
    sweep (identical v in) f and omega of every point are bit-identical to the unsharded run; (G, v) agree up to the
    float32 slab partition of the accumulation from then on."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import torch.multiprocessing as mp

    import __graft_entry__ as g

    g.build()
    import agpl_amd as A

    N, M, nsweeps = 12_001, 128, 3
    port = 29650 + (os.getpid() % 1000)
    mpctx = mp.get_context("spawn")
    q = mpctx.Queue()
    procs = [mpctx.Process(target=_gibbs_worker, args=(r, world, port, N, M, nsweeps, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=500) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ctx = A.Context(0, seed=SEED)
    lik = A.NegativeBinomialLikelihood(15.0)
    Phi, kd, y = _setup(A, ctx, lik, 0, N, M)
    ref = A.SparseGibbs(lik, Phi, kd, y, ctx=ctx, keep_points=True)
    vref = [ref.v.cpu().numpy().copy()]
    for s in range(nsweeps):
        vref.append(ref.sweep().cpu().numpy().copy())
        if s == 0:
            f1, om1, G1 = ref.f.cpu().numpy().copy(), ref.omega.cpu().numpy().copy(), ref.G.cpu().numpy().copy()
    vref = np.stack(vref)
    assert res[0][1] == 0 and res[-1][2] == N and all(res[i][2] == res[i + 1][1] for i in range(world - 1))
    for r in res[1:]:
        assert np.array_equal(res[0][3], r[3])     # identical v on all ranks at every sweep, bit for bit
    assert np.array_equal(res[0][3][0], vref[0])   # the prior draw
    f_sh, om_sh = np.concatenate([r[4] for r in res]), np.concatenate([r[5] for r in res])
    assert np.array_equal(f_sh, f1)    # f_i = phi_i' v + sqrt(d_i) eps_i on the global stream: bit-identical
    assert np.array_equal(om_sh, om1)  # and so is every PG draw
    # ranks do NOT replay each other's streams (the round-1 defect: local indices as stream keys)
    assert not np.array_equal(res[0][5][:256], res[1][5][:256])
    for r in res[1:]:
        assert np.array_equal(res[0][6], r[6])  # the reduced G is identical on every rank
    assert np.abs(res[0][6] - G1).max() / np.abs(G1).max() < 1e-5
    kappa = np.linalg.cond(np.eye(G1.shape[1]) + G1[0])
    assert np.abs(res[0][3][1] - vref[1]).max() < 1e-5 * kappa * max(1.0, np.abs(vref[1]).max())
