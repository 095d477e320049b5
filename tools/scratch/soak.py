import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import agpl_amd as A
N, M = 2_000_000, 512
ctx = A.Context(0, seed=20240807)
lik = A.BernoulliLikelihood()
x, y = A.synth_xy(lik, 20240807, 0, N, ctx=ctx)
z = np.linspace(-10, 10, M); ell = 1.5 * (z[1] - z[0])
Kzz = np.exp(-0.5 * ((z[:, None] - z[None, :]) / ell) ** 2)
_, Linv = A.sparse.whitening_matrix(Kzz, 1e-8)
Phi = A.whiten_features(A.se_features(x, torch.from_numpy(z).cuda(), ell, ctx=ctx), Linv, ctx=ctx)
kd = A.sparse.nystrom_residual(Phi, torch.ones(N, device="cuda"), ctx=ctx)
cavi = A.SparseCAVI(lik, Phi, kd, y, ctx=ctx, marginal_precision="f16x2-factor", accumulate_precision="f16x2")
mem0 = torch.cuda.memory_allocated()
prev = -np.inf
t = time.time()
for it in range(300):
    cavi.sweep()
    if it % 50 == 49 or it < 3:
        e = cavi.elbo()
        print(it + 1, f"elbo {e:.6f}", "mono" if e >= prev - 1e-6 * abs(e) else "DECREASED", flush=True)
        prev = e
torch.cuda.synchronize()
print("300 sweeps", round(time.time() - t, 2), "s; finite", bool(torch.isfinite(cavi.G).all()), "mem growth", torch.cuda.memory_allocated() - mem0)
