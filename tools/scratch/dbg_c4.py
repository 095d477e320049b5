import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import agpl_amd as A
import bench as B
from agpl_amd import _ffi
ctx = A.Context(0, seed=1)
lik = B.make_lik(A, "categorical")
N, M = 1000000, 256
x, y = A.synth_xy(lik, 20240807, 0, N, ctx=ctx)
g = torch.Generator(device="cuda").manual_seed(0)
Phi = torch.randn((N, M), device="cuda", generator=g) * 0.1
kd = torch.ones(N, device="cuda")
cavi = A.SparseCAVI(lik, Phi, kd, y, ctx=ctx, marginal_precision="f16x2-factor", accumulate_precision="f16x2")
for _ in range(2): cavi.sweep()
torch.cuda.synchronize()
_ffi.lib().agpl_timing_enable(ctx.bind(), 1)
for i in range(8):
    t = time.perf_counter(); cavi.sweep(); torch.cuda.synchronize(); print(i, round(1e3*(time.perf_counter()-t), 2))
