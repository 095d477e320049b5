#!/usr/bin/env python3
"""Wall time of the parts of a sweep (host + device), for hunting host-side stalls. Usage: lik N M"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import agpl_amd as A
import bench as B

likname, N, M = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
ctx = A.Context(0, seed=1)
lik = B.make_lik(A, likname)
x, y = A.synth_xy(lik, 20240807, 0, N, ctx=ctx)
g = torch.Generator(device="cuda").manual_seed(0)
Phi = torch.randn((N, M), device="cuda", generator=g) * 0.1
kd = torch.ones(N, device="cuda")
cavi = A.SparseCAVI(lik, Phi, kd, y, ctx=ctx, marginal_precision="f16x2-factor", accumulate_precision="f16x2")
for _ in range(2):
    cavi.sweep()
torch.cuda.synchronize()
for name, fn in (("accumulate", cavi.accumulate), ("update", cavi.update), ("sweep", cavi.sweep)):
    t = time.time()
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    print(f"{name:10s} {1e3 * (time.time() - t) / 5:8.3f} ms")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(5):
    cavi.sweep()
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumtime").print_stats(8)
