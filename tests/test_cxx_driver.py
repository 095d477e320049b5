"""bench/agpl_bench.cpp: the sweep driven from C++ through include/agpl.h alone (caller (2)).

CPU: the driver builds with -Wall -Wextra -Werror against the header, links libagpl.so and starts (argument errors are
reported before anything touches a GPU).  GPU: its natural parameters after three sweeps equal the Python host's on the
same seeded workload -- the Python layer adds no arithmetic -- and two runs are byte-identical."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "bench", "agpl_bench")
ENV = dict(os.environ, LD_LIBRARY_PATH="/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))


def ensure_built():
    if not os.path.exists(EXE):
        sys.path.insert(0, ROOT)
        import __graft_entry__ as g

        g.build()
    assert os.path.exists(EXE)


def test_cxx_driver_builds_links_and_rejects_bad_arguments():
    ensure_built()
    r = subprocess.run([EXE, "--bogus"], capture_output=True, text=True, timeout=120, env=ENV)
    assert r.returncode == 1 and "unknown argument" in r.stderr, (r.returncode, r.stderr[-500:])
    r = subprocess.run([EXE, "--m", "100"], capture_output=True, text=True, timeout=120, env=ENV)
    assert r.returncode == 1 and "multiple of 256" in r.stderr, (r.returncode, r.stderr[-500:])
    src = open(os.path.join(ROOT, "bench", "agpl_bench.cpp")).read()
    assert "torch" not in src.replace("no torch", "") and "Python.h" not in src
    # every entry point of the shipped sweep is called through the header
    for sym in ("agpl_synth_xy", "agpl_se_features", "agpl_transform_features", "agpl_plan_create", "agpl_cavi_pass_plan",
                "agpl_plan_update", "agpl_plan_destroy", "agpl_ctx_synchronize"):
        assert sym + "(" in src, sym


@pytest.mark.gpu
def test_cxx_driver_matches_the_python_host(tmp_path):
    import torch

    sys.path.insert(0, ROOT)
    import agpl_amd as A
    import bench

    ensure_built()
    N, M, sweeps = 200_000, 256, 3
    outs = []
    for k in range(2):
        dump = tmp_path / f"nat{k}.bin"
        r = subprocess.run([EXE, "--n", str(N), "--m", str(M), "--sweeps", str(sweeps), "--warmup", "0", "--dump",
                            str(dump)], capture_output=True, text=True, timeout=600, env=ENV)
        assert r.returncode == 0, r.stderr[-2000:]
        line = json.loads(r.stdout.strip().splitlines()[-1])
        assert line["N"] == N and line["M"] == M and line["value"] > 0
        outs.append(np.fromfile(dump, dtype=np.float64))
        assert outs[-1].size == M * M + M
    assert np.array_equal(outs[0], outs[1]), "two runs of the C++ driver differ"
    Gc, gc_ = outs[0][: M * M].reshape(M, M), outs[0][M * M:]

    ctx = A.Context(0, seed=bench.SEED)
    lik = A.BernoulliLikelihood()
    y, Phi, kd = bench.build_workload(A, ctx, lik, 0, N, M)
    cavi = A.SparseCAVI(lik, Phi, kd, y, ctx=ctx, marginal_precision="f16x2-factor", accumulate_precision="f16x2")
    cavi.run(sweeps)
    torch.cuda.synchronize()
    G, g = cavi.G.cpu().numpy()[0], cavi.g.cpu().numpy()[0]
    # the only difference between the two hosts is the M x M host Cholesky of the whitening (LAPACK there, three loops
    # here): float64 rounding of L^-1 before it is cast to float32
    dG = np.abs(Gc - G).max() / np.abs(G).max()
    dg = np.abs(gc_ - g).max() / np.abs(g).max()
    assert dG < 1e-6 and dg < 1e-6, (dG, dg)
    assert np.array_equal(Gc, Gc.T)
