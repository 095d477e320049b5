"""world_size-2 test of the N-sharded path on CPU (gloo): the product's shard_range() and
exchange_natural_parameters() with the oracle standing in for the per-rank device pass.

Checks: shards tile [0, N) exactly; the synthetic inputs of a shard are the slice of the
global inputs (pure function of (seed, index)); sum over ranks of the per-shard (G, g) equals the
single-process result to float64 round-off; every rank ends with identical natural parameters and hence the
identical M x M update (no broadcast needed)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, N, M, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist

    import agpl_amd as A
    from oracle import oracle as O

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        seed = 20240807
        olik = O.bernoulli()
        i0, i1 = A.shard_range(N, rank, world)
        x = O.synth_x(seed, i0, i1 - i0)
        y = O.synth_y(olik, seed, i0, i1 - i0)
        z = np.linspace(-10, 10, M)
        ell = 1.5 * (z[1] - z[0])
        Phi = O.se_kernel_f32(x, z, ell)
        kd = np.ones(i1 - i0)
        S, m = np.eye(M)[None] * 0.5, np.full((1, M), 0.1)
        G, g = O.cavi_pass(olik, Phi, kd, y, -S, m)
        Gt, gt = torch.from_numpy(G.copy()), torch.from_numpy(g.copy())
        A.exchange_natural_parameters(Gt, gt, dist.group.WORLD)
        # the single-collective form the sweep drivers use: G and g as views of one flat buffer
        flat, Gf, gf = A.sparse.natural_parameter_buffers(1, M, "cpu")
        Gf.copy_(torch.from_numpy(G)); gf.copy_(torch.from_numpy(g))
        A.exchange_natural_parameters(Gf, gf, dist.group.WORLD, flat=flat)
        assert torch.equal(Gf, Gt) and torch.equal(gf, gt)
        Sn, mn = O.gaussian_update(Gt.numpy(), gt.numpy())
        q.put((rank, i0, i1, x[:3].tolist(), Gt.numpy(), gt.numpy(), Sn, mn))
    finally:
        dist.destroy_process_group()


def test_shard_range_tiles_exactly():
    sys.path.insert(0, ROOT)
    import agpl_amd as A

    for N in (1, 7, 8, 1000, 10_000_000, 10_000_019):
        for world in (1, 2, 3, 8):
            rs = [A.shard_range(N, r, world) for r in range(world)]
            assert rs[0][0] == 0 and rs[-1][1] == N
            assert all(rs[i][1] == rs[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in rs]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.timeout(300)
def test_two_rank_sharded_sweep_matches_single_process(oracle):
    import torch.multiprocessing as mp

    O = oracle
    N, M, world = 6001, 32, 2
    port = 29500 + (os.getpid() % 2000)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, N, M, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process reference
    seed = 20240807
    olik = O.bernoulli()
    x = O.synth_x(seed, 0, N)
    y = O.synth_y(olik, seed, 0, N)
    z = np.linspace(-10, 10, M)
    Phi = O.se_kernel_f32(x, z, 1.5 * (z[1] - z[0]))
    S, m = np.eye(M)[None] * 0.5, np.full((1, M), 0.1)
    G, g = O.cavi_pass(olik, Phi, np.ones(N), y, -S, m)
    assert res[0][1] == 0 and res[1][2] == N and res[0][2] == res[1][1]
    assert res[1][3] == x[res[1][1]:res[1][1] + 3].tolist()  # shard inputs are slices of the global inputs
    for r in res:
        assert np.allclose(r[4], G, rtol=1e-12, atol=1e-12)
        assert np.allclose(r[5], g, rtol=1e-12, atol=1e-12)
    # identical on every rank, bit for bit -> identical update, no broadcast
    assert np.array_equal(res[0][4], res[1][4]) and np.array_equal(res[0][5], res[1][5])
    assert np.array_equal(res[0][6], res[1][6]) and np.array_equal(res[0][7], res[1][7])


def _gibbs_worker(rank, world, port, N, M, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist

    import agpl_amd as A
    from oracle import oracle as O

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        seed = 20240807
        olik = O.negbinomial(15.0)
        i0, i1 = A.shard_range(N, rank, world)
        x = O.synth_x(seed, i0, i1 - i0)
        y = O.synth_y(olik, seed, i0, i1 - i0)
        z = np.linspace(-10, 10, M)
        Phi = O.se_kernel_f32(x, z, 1.5 * (z[1] - z[0]))
        kd = np.full(i1 - i0, 0.3)
        v, _ = O.gibbs_draw_v(np.zeros((1, M, M)), np.zeros((1, M)), seed=seed, sweep=0)  # same key on every rank
        # the shard's point pass on the GLOBAL point streams (agpl_ctx_set_point_offset / SparseGibbs(point_offset=i0))
        G, g, pts = O.gibbs_pass(olik, Phi, kd, y, v, seed=seed, sweep=1, i0=i0)
        flat, Gf, gf = A.sparse.natural_parameter_buffers(1, M, "cpu")
        Gf.copy_(torch.from_numpy(G)); gf.copy_(torch.from_numpy(g))
        A.exchange_natural_parameters(Gf, gf, dist.group.WORLD, flat=flat)
        v1, _ = O.gibbs_draw_v(Gf.numpy(), gf.numpy(), seed=seed, sweep=1)
        q.put((rank, i0, i1, pts["f"], pts["omega"], pts["nuni"], Gf.numpy().copy(), v, v1))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_sharded_gibbs_sweep_is_the_single_process_sweep(oracle):
    """The N-sharded Gibbs sweep: per-point streams keyed on the global point index, the inducing draw
    on the same key everywhere -- the shards' f, omega and uniforms-consumed concatenate to the single-process arrays
    bit for bit, the reduced G matches to float64 round-off, and every rank draws the identical v."""
    import torch.multiprocessing as mp

    O = oracle
    N, M, world = 3001, 32, 2
    port = 29400 + (os.getpid() % 2000)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gibbs_worker, args=(r, world, port, N, M, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    seed = 20240807
    olik = O.negbinomial(15.0)
    x, y = O.synth_x(seed, 0, N), O.synth_y(olik, seed, 0, N)
    z = np.linspace(-10, 10, M)
    Phi = O.se_kernel_f32(x, z, 1.5 * (z[1] - z[0]))
    v, _ = O.gibbs_draw_v(np.zeros((1, M, M)), np.zeros((1, M)), seed=seed, sweep=0)
    G, g, pts = O.gibbs_pass(olik, Phi, np.full(N, 0.3), y, v, seed=seed, sweep=1)
    assert np.array_equal(res[0][7], v) and np.array_equal(res[1][7], v)
    assert np.array_equal(np.concatenate([res[0][3], res[1][3]]), pts["f"])
    assert np.array_equal(np.concatenate([res[0][4], res[1][4]]), pts["omega"])
    assert np.array_equal(np.concatenate([res[0][5], res[1][5]]), pts["nuni"])  # integer bookkeeping: bit-exact
    # a rank keyed on LOCAL indices (the round-1 defect) would replay rank 0's streams on rank 1
    n1 = res[1][2] - res[1][1]
    assert not np.array_equal(res[0][5][:n1], res[1][5])
    assert np.allclose(res[0][6], G, rtol=1e-12, atol=1e-12)
    assert np.array_equal(res[0][6], res[1][6]) and np.array_equal(res[0][8], res[1][8])


def _agree_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import types

    import torch
    import torch.distributed as dist

    import agpl_amd as A

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        out = []
        for seed, sweep in ((7, 3), (7 + rank, 3), (2**40 + 5, 3 + rank)):
            me = types.SimpleNamespace(group=dist.group.WORLD, Phi=torch.zeros(2, 2), ctx=types.SimpleNamespace(seed=seed),
                                       sweep_index=sweep)
            try:
                A.SparseGibbs._check_ranks_agree(me)
                out.append("ok")
            except A.ArgumentError:
                out.append("refused")
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_sparse_gibbs_refuses_ranks_with_different_seed_or_draw_counter():
    """Every rank draws v itself (no broadcast): SparseGibbs(group=...) checks at construction that all ranks hold the same
    Philox key and draw counter (one MIN / MAX all-reduce of three words)."""
    import torch.multiprocessing as mp

    world = 2
    port = 31400 + (os.getpid() % 2000)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_agree_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(60)
    assert res[0] == res[1] == ["ok", "refused", "refused"]


# ---------------------------------------------------------------------------------------------------------------------
# the real rank count of BASELINE config C3 (8 ranks), whole sweeps: CAVI (pass -> one all-reduce -> the identical update on
# every rank) and Gibbs (global-index Philox streams, identical v everywhere), the oracle standing in for the device pass
# ---------------------------------------------------------------------------------------------------------------------
def _sweeps_worker(rank, world, port, N, M, nsweeps, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      OMP_NUM_THREADS="1")
    import torch
    import torch.distributed as dist

    torch.set_num_threads(1)
    import agpl_amd as A
    from oracle import oracle as O

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        seed = 20240807
        olik = O.negbinomial(15.0)  # C3's likelihood
        i0, i1 = A.shard_range(N, rank, world)
        x = O.synth_x(seed, i0, i1 - i0)
        y = O.synth_y(olik, seed, i0, i1 - i0)
        z = np.linspace(-10, 10, M)
        Phi = O.se_kernel_f32(x, z, 1.5 * (z[1] - z[0]))
        kd = np.full(i1 - i0, 0.3)
        # ---- CAVI: examples/bernoulli/script.jl:29-39 sharded over N
        S, m = np.eye(M)[None].copy(), np.zeros((1, M))
        flat, Gf, gf = A.sparse.natural_parameter_buffers(1, M, "cpu")
        for _ in range(nsweeps):
            G, g = O.cavi_pass(olik, Phi, kd, y, -S, m)
            Gf.copy_(torch.from_numpy(G)); gf.copy_(torch.from_numpy(g))
            A.exchange_natural_parameters(Gf, gf, dist.group.WORLD, flat=flat)  # ONE collective of M^2 + M doubles
            S, m = O.gaussian_update(Gf.numpy(), gf.numpy())  # every rank, identical inputs -> identical (S, m): no broadcast
        cavi = (Gf.numpy().copy(), gf.numpy().copy(), S.copy(), m.copy())
        # ---- Gibbs: script.jl:76-87 sharded; streams keyed on the GLOBAL point index, v on the same key everywhere
        v, _ = O.gibbs_draw_v(np.zeros((1, M, M)), np.zeros((1, M)), seed=seed, sweep=0)
        fs, oms, nunis = [], [], []
        for sw in range(1, nsweeps + 1):
            G, g, pts = O.gibbs_pass(olik, Phi, kd, y, v, seed=seed, sweep=sw, i0=i0)
            Gf.copy_(torch.from_numpy(G)); gf.copy_(torch.from_numpy(g))
            A.exchange_natural_parameters(Gf, gf, dist.group.WORLD, flat=flat)
            v, _ = O.gibbs_draw_v(Gf.numpy(), gf.numpy(), seed=seed, sweep=sw)
            fs.append(pts["f"]); oms.append(pts["omega"]); nunis.append(pts["nuni"])
        q.put((rank, i0, i1, cavi, v.copy(), fs[-1], oms[-1], nunis[-1]))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", [8])
def test_eight_rank_sharded_cavi_and_gibbs_sweeps_are_the_single_process_sweeps(oracle, world):
    """BASELINE config C3's rank count on CPU (gloo, world size 8): shard_range at the real
    world size (ragged: N = 4003 is not a multiple of 8), three whole CAVI sweeps and three whole Gibbs sweeps.  Every rank ends
    every sweep with bit-identical (G, g) and hence the bit-identical M x M update / inducing draw; the sharded result equals
    the single-process one (sums to float64 round-off, the per-point draws of the last Gibbs sweep bit for bit once
    concatenated in rank order -- which they can only be if the chains agreed in every earlier sweep to the last bit of v...
    they agree to round-off of the reduced G, so f is compared to 1e-9 and the integer bookkeeping exactly)."""
    import torch.multiprocessing as mp

    O = oracle
    N, M, nsweeps = 4003, 32, 3
    port = 30100 + (os.getpid() % 2000)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sweeps_worker, args=(r, world, port, N, M, nsweeps, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=500) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == 0 and res[-1][2] == N and all(res[i][2] == res[i + 1][1] for i in range(world - 1))
    assert {r[2] - r[1] for r in res} == {N // world, N // world + 1}
    # identical on every rank, bit for bit: the reduced accumulators, the update, the inducing draw
    for r in res[1:]:
        for a, b in zip(r[3], res[0][3]):
            assert np.array_equal(a, b)
        assert np.array_equal(r[4], res[0][4])
    # single-process reference
    seed = 20240807
    olik = O.negbinomial(15.0)
    x, y = O.synth_x(seed, 0, N), O.synth_y(olik, seed, 0, N)
    z = np.linspace(-10, 10, M)
    Phi = O.se_kernel_f32(x, z, 1.5 * (z[1] - z[0]))
    kd = np.full(N, 0.3)
    S, m = np.eye(M)[None].copy(), np.zeros((1, M))
    for _ in range(nsweeps):
        G, g = O.cavi_pass(olik, Phi, kd, y, -S, m)
        S, m = O.gaussian_update(G, g)
    assert np.allclose(res[0][3][0], G, rtol=1e-10, atol=1e-12) and np.allclose(res[0][3][1], g, rtol=1e-10, atol=1e-12)
    rel = lambda a, b: np.abs(a - b).max() / np.abs(b).max()
    assert rel(res[0][3][2], S) < 1e-7 and rel(res[0][3][3], m) < 1e-7  # (round-off of the reduced G times cond(I + G))
    v, _ = O.gibbs_draw_v(np.zeros((1, M, M)), np.zeros((1, M)), seed=seed, sweep=0)
    for sw in range(1, nsweeps + 1):
        G, g, pts = O.gibbs_pass(olik, Phi, kd, y, v, seed=seed, sweep=sw)
        v, _ = O.gibbs_draw_v(G, g, seed=seed, sweep=sw)
    assert rel(res[0][4], v) < 1e-6
    f_all = np.concatenate([r[5] for r in res])
    assert rel(f_all, pts["f"]) < 1e-6
    # integer bookkeeping of the last sweep: uniforms consumed per point (the chains have not branched apart)
    nuni_all = np.concatenate([r[7] for r in res])
    assert (nuni_all == pts["nuni"]).mean() > 0.99
    # first sweep is exact by construction (same v): covered bit for bit by the two-rank test above
