"""bench.py's output contract on a small workload: one JSON line with the driver's keys, the roofline and
cpu_baseline objects, the oracle parity slice and the full-size self-check -- single process, and two ranks sharing
the one GPU (gloo hook) through the same launcher line the driver uses."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
        "vs_baseline", "dtype", "data", "config", "roofline"}


def _last_json(stdout):
    lines = [ln for ln in stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.timeout(600)
def test_bench_line_single_gpu_small_workload():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    r = subprocess.run([sys.executable, "bench.py", "--points", "300000", "--inducing", "256", "--steps", "3",
                        "--warmup", "1", "--cpu-sample", "20000"], cwd=ROOT, capture_output=True, text=True, timeout=550)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _last_json(r.stdout)
    assert KEYS <= set(d), sorted(KEYS - set(d))
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["unit"] == "sweeps/s" and d["vs_baseline"] is None and d["data"].startswith("synthetic")
    assert d["value"] == pytest.approx(1e3 / d["ms_per_step"], rel=2e-3)
    rf = d["roofline"]
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(rf)
    assert rf["frac"] == pytest.approx(rf["achieved"] / rf["peak"], abs=1e-4) and 0 < rf["frac"] < 1  # 4 decimals
    # the measured ceiling next to the data-sheet peak: a sustained float16 MFMA rate below the peak and above half of it,
    # and every split kernel's executed rate below that ceiling
    sus = rf["sustained_mfma_f16"]
    assert 0.5 * rf["peak"] < sus["tflops"] < rf["peak"] and "agpl_probe_mfma" in sus["probe"]
    sus32 = rf["sustained_mfma_f16_16x16x32"]  # the same for the instruction shape the shipped kernels issue
    assert 0.4 * rf["peak"] < sus32["tflops"] < rf["peak"] and "mode 3" in sus32["probe"]
    for k in rf["kernels"]:
        assert 0 < k["executed_frac_of_sustained"] < 1.0, k
    cb = d["cpu_baseline"]
    assert {"value", "unit", "cores", "kind", "sample"} <= set(cb) and cb["kind"] == "port" and cb["value"] > 0
    assert d["parity"]["pass"] and d["parity"]["max_rel_dG"] < 1e-5 and d["parity"]["max_rel_dg"] < 1e-5
    fc = d["full_size_check"]
    assert fc["pass"] and fc["points"] == 300000 and fc["G_symmetric"]
    assert fc["max_rel_d_vGv"] < 2e-6 and fc["quadratic_forms"] == 4  # the check that sees the off-diagonal tiles
    # the sweep at the contract's own arithmetic (float32-input MFMA), priced by what it executes against the 157.3 TFLOP/s roof
    f32 = d["f32_contract"]
    assert f32["dtype"] == "f32" and f32["ms_per_step"] > 0 and 0 < f32["roofline"]["frac"] < 1
    assert f32["parity"]["pass"]
    # aug_elbo riding the sweep: the last values do not decrease
    assert d["elbo"]["non_decreasing"] and len(d["elbo"]["elbo_entering_last_sweeps"]) == 3
    assert d["hbm_gb"]["plan"] > 0
    # the compact summary is the LAST key of the line (the driver's record keeps the tail) and repeats the headline
    assert list(d)[-1] == "summary" and len(json.dumps(d["summary"])) <= 1536
    sm = d["summary"]
    assert sm["c2"]["ms"] == d["ms_per_step"] and sm["c2"]["frac"] == rf["frac"] and sm["c2"]["par"] < 1e-5
    assert sm["f32_contract"]["ms"] == f32["ms_per_step"] and sm["gibbs"]["ms"] == d["gibbs"]["ms_per_sweep"]
    assert sm["cpu"]["cores"] == cb["cores"] and sm["full_size_pass"] is True


@pytest.mark.timeout(600)
def test_bench_line_two_ranks_through_the_driver_launcher():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = dict(os.environ, AGPL_BENCH_SINGLE_DEVICE="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29617", "bench.py", "--gpus", "2",
                        "--points", "300000", "--inducing", "256", "--steps", "3", "--warmup", "1", "--sharded-legs", "on"],
                       cwd=ROOT, capture_output=True, text=True, timeout=550, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _last_json(r.stdout)
    assert KEYS <= set(d)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0
    # BASELINE C3 itself (NegBin r = 15, M = 1024, CAVI + Gibbs) and the Bernoulli M = 1024 north-star target on the same two ranks
    for name, lik in (("c3", "NegBin"), ("m1024", "bernoulli")):
        o = d[name]
        assert o["world"] == 2 and o["config"]["M"] == 1024 and o["config"]["N"] == 300000 and lik in o["config"]["workload"]
        assert o["ms_per_step"] > 0 and o["allreduce_ms"] > 0 and o["allreduce_bytes"] == 8 * (1024 * 1024 + 1024)
        assert len(o["per_rank"]) == 2 and all(r["points"] == 150000 for r in o["per_rank"])
        assert o["ms_per_step_min_rank"] <= o["ms_per_step_max_rank"] <= o["ms_per_step"] * 1.001
        assert o["parity"]["pass"] and o["parity"]["sweeps"] == 10
    assert d["c3"]["gibbs_ms_per_sweep"] > 0 and "gibbs_ms_per_sweep" not in d["m1024"]
    assert list(d)[-1] == "summary"
    sm = d["summary"]
    assert sm["n_gpus"] == 2 and sm["c3"]["ranks"] == 2 and sm["c3"]["ms"] == d["c3"]["ms_per_step"]
    assert sm["m1024"]["ar_ms"] == d["m1024"]["allreduce_ms"] and sm["c3"]["gibbs_ms"] == d["c3"]["gibbs_ms_per_sweep"]
    assert "cpu_baseline" not in d  # rank 0 at N = 1 only
    # the diagnosis a scaling run needs: the exchange step as the stream saw it, and every rank's own numbers
    assert d["allreduce_ms"] > 0 and d["allreduce_bytes"] == 8 * (256 * 256 + 256)
    assert len(d["per_rank"]) == 2 and {r["rank"] for r in d["per_rank"]} == {0, 1}
    assert all(r["points"] == 150000 and r["ms_per_step"] > 0 and r["allreduce_ms_avg"] > 0 for r in d["per_rank"])
    assert d["ms_per_step_min_rank"] <= d["ms_per_step_max_rank"] <= d["ms_per_step"] * 1.001
    assert len(d["per_rank_kernels"]) == 2 and all(k["accumulate_kernel_ms"] > 0 for k in d["per_rank_kernels"])


@pytest.mark.timeout(600)
def test_bench_gpus_2_without_a_launcher_starts_two_ranks_itself():
    """`python bench.py --gpus 2` (no torchrun): the script starts its own ranks as a child process and the line says
    n_gpus = 2 and rccl_ranks = 2 -- a run that asks for N GPUs never silently measures one."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["AGPL_BENCH_SINGLE_DEVICE"] = "1"
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--points", "300000", "--inducing", "256", "--steps",
                        "3", "--warmup", "1"], cwd=ROOT, capture_output=True, text=True, timeout=550, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _last_json(r.stdout)
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["value"] > 0


def test_bench_refuses_more_gpus_than_are_visible():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "AGPL_BENCH_SINGLE_DEVICE")}
    n = torch.cuda.device_count() + 1
    r = subprocess.run([sys.executable, "bench.py", "--gpus", str(n), "--steps", "1", "--warmup", "0"], cwd=ROOT,
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 2 and "refusing" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
