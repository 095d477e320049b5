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
