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


def _worker(rank, world, port, N, M, nsweeps, q, kw):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist

    import agpl_amd as A

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ctx = A.Context(0, seed=SEED)
        lik = A.BernoulliLikelihood()
        i0, i1 = A.shard_range(N, rank, world)
        Phi, kd, y = _setup(A, ctx, lik, i0, i1 - i0, M)
        cavi = A.SparseCAVI(lik, Phi, kd, y, ctx=ctx, group=dist.group.WORLD, **kw)
        cavi.run(nsweeps)
        torch.cuda.synchronize()
        q.put((rank, cavi.G.cpu().numpy(), cavi.g.cpu().numpy(), cavi.m.cpu().numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("M,kw", [(64, {}),
                                  (200, {"marginal_precision": "f16x2-factor", "accumulate_precision": "f16x2"})])
def test_two_ranks_one_gpu_match_single_process(M, kw):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import torch.multiprocessing as mp

    import __graft_entry__ as g

    g.build()
    import agpl_amd as A

    N, nsweeps, world = 30_001, 4, 2
    port = 29600 + (os.getpid() % 1000)
    mpctx = mp.get_context("spawn")
    q = mpctx.Queue()
    procs = [mpctx.Process(target=_worker, args=(r, world, port, N, M, nsweeps, q, kw)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ctx = A.Context(0, seed=SEED)
    lik = A.BernoulliLikelihood()
    Phi, kd, y = _setup(A, ctx, lik, 0, N, M)
    ref = A.SparseCAVI(lik, Phi, kd, y, ctx=ctx, **kw)
    ref.run(nsweeps)
    G, gg, m = ref.G.cpu().numpy(), ref.g.cpu().numpy(), ref.m.cpu().numpy()
    # every rank holds the identical reduced natural parameters and hence the identical update
    assert np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][2], res[1][2])
    assert np.array_equal(res[0][3], res[1][3])
    # sharded == unsharded up to the float32 slab partition (different 4096-point slices per rank)
    assert np.abs(res[0][1] - G).max() / np.abs(G).max() < 1e-5
    assert np.abs(res[0][2] - gg).max() / np.abs(gg).max() < 1e-5
