"""bench.py --gpus N without a launcher: the parent refuses -- exit 2, no JSON line -- when fewer than N
devices are visible, before anything touches a GPU.  Runs on the CPU box (0 devices visible)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_refuses_to_measure_fewer_gpus_than_asked():
    import torch

    if torch.cuda.device_count() >= 2:
        import pytest

        pytest.skip("two devices are visible here")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "AGPL_BENCH_SINGLE_DEVICE")}
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "1", "--warmup", "0"], cwd=ROOT,
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 2, (r.returncode, r.stderr[-500:])
    assert "refusing" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


# ---- the host logic of the N > 1 legs and of the line's summary: no GPU needed ------------------------
def _gather_worker(rank, world, port, q):
    import sys

    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist

    import bench

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mine = {"rank": rank, "points": 1000 + rank, "ms_per_step": 2.0 + 0.1 * rank, "allreduce_ms_avg": 0.05 * (rank + 1)}
        per_rank, dt = bench.gather_over_ranks(mine, 0.02 + 0.001 * rank, dist.group.WORLD, "cpu")
        none_rank, dt2 = bench.gather_over_ranks(None, float(rank), dist.group.WORLD, "cpu")
        q.put((rank, per_rank, dt, none_rank, dt2))
    finally:
        dist.destroy_process_group()


def _run_gather(world):
    import torch.multiprocessing as mp

    port = 29300 + (os.getpid() % 1500) + world
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gather_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def _check_gather(world):
    res = _run_gather(world)
    for rank, per_rank, dt, none_rank, dt2 in res:
        # every rank sees every rank's record, in rank order, and the MAX of the timed regions
        assert [r["rank"] for r in per_rank] == list(range(world))
        assert [r["points"] for r in per_rank] == [1000 + r for r in range(world)]
        assert dt == 0.02 + 0.001 * (world - 1)
        assert none_rank == [None] * world and dt2 == float(world - 1)


def test_gather_over_ranks_two_gloo_ranks():
    _check_gather(2)


def test_gather_over_ranks_eight_gloo_ranks():
    _check_gather(8)


def _fake_leg(ms, frac, par=None, **kw):
    o = {"ms_per_step": ms, "value": 1e3 / ms, "roofline": {"frac": frac, "kernels": [{"avg_ms": 0.4 * ms}, {"avg_ms": 0.5 * ms}]}}
    if par is not None:
        o["parity"] = {"max_rel_dG": par, "max_rel_dg": par / 3}
    o.update(kw)
    return o


def test_summary_is_the_last_key_material_and_stays_small():
    """`summary` carries every configuration's number in <= 1.5 KB (the driver keeps the last ~2 KB of the line)."""
    import json
    import sys

    sys.path.insert(0, ROOT)
    import bench

    out = _fake_leg(13.26, 0.1593, 4.3e-7, n_gpus=1)
    out["m1024"] = _fake_leg(42.353, 0.1945, 2.8e-7, c3_full_one_gpu=_fake_leg(42.1, 0.19))
    out["n8"] = _fake_leg(1.97, 0.1437, 6.3e-7, projected_scaling_8=6.748)
    out["n8_m1024"] = _fake_leg(5.946, 0.1858, 3.9e-7, projected_scaling_8=7.123)
    out["c3r"] = _fake_leg(5.924, 0.1861, 3.6e-7, projected_scaling_8=7.108, gibbs_ms_per_sweep=6.124)
    out["c4"] = _fake_leg(4.217, 0.1251, 1.4e-6, gibbs_ms_per_sweep=3.7)
    out["c5"] = {"ms_per_step": 1610.4, "value": 0.621, "roofline": {"frac": 0.88}}
    out["f32_contract"] = _fake_leg(52.142, 0.7245, 4.2e-7)
    out["gibbs"] = {"ms_per_sweep": 11.525, "sampler": {"pg1_draws_per_s": 1.84e10}, "sampler_negbin": {"pg1_draws_per_s": 2.64e10}}
    out["cpu_baseline"] = {"value": 0.0361, "cores": 128, "gpu_over_cpu": 2082.6}
    out["full_size_check"] = {"pass": True}
    sm = bench.summarize(out)
    assert len(json.dumps(sm)) <= 1536, len(json.dumps(sm))
    assert sm["c2"]["ms"] == 13.26 and sm["c2"]["frac"] == 0.1593 and sm["c2"]["par"] == 4.3e-7
    assert sm["m1024"]["ms"] == 42.353 and sm["c3_full_one_gpu"]["ms"] == 42.1
    assert sm["n8"]["proj8"] == 6.748 and sm["c3r"]["proj8"] == 7.108 and sm["c3r"]["gibbs_ms"] == 6.124
    assert sm["c4"]["gibbs_ms"] == 3.7 and sm["c5"]["ms"] == 1610.4 and sm["f32_contract"]["frac"] == 0.7245
    assert sm["gibbs"]["pg1_per_s_negbin"] == 2.64e10 and sm["cpu"]["cores"] == 128 and sm["full_size_pass"] is True
    # an N > 1 line: the sharded legs with their exchange time and rank count; a leg that failed keeps its error
    multi = _fake_leg(2.1, 0.14, n_gpus=8)
    multi["c3"] = _fake_leg(6.2, 0.18, 3.7e-7, allreduce_ms=0.21, world=8, gibbs_ms_per_sweep=6.4)
    multi["m1024"] = {"error": "RuntimeError: out of memory " + "x" * 200}
    sm8 = bench.summarize(multi)
    assert sm8["c3"]["ar_ms"] == 0.21 and sm8["c3"]["ranks"] == 8 and sm8["c3"]["gibbs_ms"] == 6.4 and sm8["n_gpus"] == 8
    assert len(sm8["m1024"]["error"]) <= 60
