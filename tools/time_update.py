#!/usr/bin/env python3
"""Time the M x M update (agpl_gaussian_update vs agpl_gaussian_factor) alone. Usage: time_update.py M [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import agpl_amd as A
from agpl_amd import _ffi
if os.environ.get("AGPL_LIB_AB"):
    _ffi.LIB_PATH = os.environ["AGPL_LIB_AB"]

M = int(sys.argv[1]); reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
ctx = A.Context(0, seed=1)
N = 4096
g = torch.Generator(device="cuda").manual_seed(0)
Phi = torch.randn((N, M), device="cuda", generator=g) * 0.1
kd = torch.ones(N, device="cuda")
y = (torch.rand(N, device="cuda", generator=g) < 0.5).to(torch.uint8)
for mode in ("f32", "f16x2-factor"):
    cavi = A.SparseCAVI(A.BernoulliLikelihood(), Phi, kd, y, ctx=ctx, marginal_precision=mode)
    cavi.sweep(); cavi.sweep()
    torch.cuda.synchronize()
    t = time.time()
    for _ in range(reps):
        cavi.update()
    torch.cuda.synchronize()
    print(f"{mode:14s} M={M} update {1e3 * (time.time() - t) / reps:7.3f} ms")
