"""Checkpoint / resume of the sweep loops: after k sweeps the state -- q(v) in the plan's factor
form (agpl_plan_state), the reduced (G, g), the context's Philox key and draw counter; for the Gibbs chain the
inducing draw v -- is written to disk by one process, restored by a FRESH process that rebuilds the static images from the same
features, and the next CAVI sweep and the next Gibbs sweeps are bit for bit those of the uninterrupted run.
(The reference keeps its state in user scope: (m, S, qΩ) examples/bernoulli/script.jl:41-43, (f, Ω) :89-90.)"""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SEED = 20240807


def _workload(A, ctx, lik, N, M):
    x, y = A.synth_xy(lik, SEED, 0, N, ctx=ctx)
    z = np.linspace(-10, 10, M)
    ell = 1.5 * (z[1] - z[0])
    Kzz = np.exp(-0.5 * ((z[:, None] - z[None, :]) / ell) ** 2)
    _, Linv = A.sparse.whitening_matrix(Kzz, 1e-8)
    Kzx = A.se_features(x, torch.from_numpy(z).cuda(), ell, ctx=ctx)
    Phi = A.whiten_features(Kzx, Linv, ctx=ctx)
    kd = A.sparse.nystrom_residual(Phi, torch.ones(N, device="cuda"), ctx=ctx)
    return Phi, kd, y


def _run(mode, path, likname, N, M):
    sys.path.insert(0, ROOT)
    import agpl_amd as A

    lik = A.BernoulliLikelihood() if likname == "bernoulli" else A.NegativeBinomialLikelihood(15.0)
    ctx = A.Context(0, seed=SEED)
    Phi, kd, y = _workload(A, ctx, lik, N, M)
    cavi = A.SparseCAVI(lik, Phi, kd, y, ctx=ctx, track_elbo=True)
    gctx = A.Context(0, seed=SEED + 1)
    gib = A.SparseGibbs(lik, Phi, kd, y, ctx=gctx, keep_points=True)
    if mode == "save":
        cavi.run(3)
        for _ in range(3):
            gib.sweep()
        torch.save({"cavi": cavi.state_dict(), "gibbs": gib.state_dict()}, path + ".ckpt")
    else:
        st = torch.load(path + ".ckpt", weights_only=False)
        cavi.load_state_dict(st["cavi"])
        gib.load_state_dict(st["gibbs"])
    # the continuation: one CAVI sweep (+ the ELBO that rode it), two Gibbs sweeps
    cavi.sweep()
    elbo = cavi.elbo_entering()
    cavi.sweep()
    cavi.check()
    vs = [gib.sweep().cpu().clone() for _ in range(2)]
    out = {"G": cavi.G.cpu(), "g": cavi.g.cpu(), "U": torch.triu(cavi.plan.U_colmajor).cpu(), "v": cavi.plan.v.cpu(),
           "elbo": elbo, "elbo2": cavi.elbo_entering(), "gibbs_v": torch.stack(vs), "omega": gib.omega.cpu(), "f": gib.f.cpu(),
           "nsweeps": cavi.nsweeps}
    torch.save(out, path + "." + mode)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("likname,N,M", [("bernoulli", 40_001, 512), ("negbin", 20_000, 1024)])
def test_checkpoint_restore_in_a_fresh_process_continues_bit_for_bit(tmp_path, likname, N, M):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import __graft_entry__ as g

    g.build()
    path = str(tmp_path / "state")
    for mode in ("save", "restore"):  # two processes, one after the other; this one never touches the state
        r = subprocess.run([sys.executable, os.path.abspath(__file__), mode, path, likname, str(N), str(M)],
                           capture_output=True, text=True, timeout=800)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    a = torch.load(path + ".save", weights_only=False)
    b = torch.load(path + ".restore", weights_only=False)
    assert a["nsweeps"] == b["nsweeps"] == 5
    for k in ("G", "g", "U", "v", "gibbs_v", "omega", "f"):
        assert torch.equal(a[k], b[k]), k
    assert a["elbo"] == b["elbo"] and a["elbo2"] == b["elbo2"]
    assert np.isfinite(a["elbo"]) and torch.isfinite(a["gibbs_v"]).all()
    # (and the state is not trivially constant: the continuation moved q(v))
    assert not torch.equal(a["gibbs_v"][0], a["gibbs_v"][1])


if __name__ == "__main__":
    _run(sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5]))
