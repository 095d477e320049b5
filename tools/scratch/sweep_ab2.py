#!/usr/bin/env python3
"""Whole-sweep A/B of two builds on the same box: AGPL_LIB_AB=<lib> python tools/scratch/sweep_ab2.py [N M]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import agpl_amd as A
from agpl_amd import _ffi
if os.environ.get("AGPL_LIB_AB"):
    _ffi.LIB_PATH = os.environ["AGPL_LIB_AB"]
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
M = int(sys.argv[2]) if len(sys.argv) > 2 else 512
ctx = A.Context(0, seed=1)
g = torch.Generator(device="cuda").manual_seed(0)
Phi = torch.randn((N, M), device="cuda", generator=g) * 0.05
kd = torch.ones(N, device="cuda") * 0.1
y = (torch.rand(N, device="cuda", generator=g) < 0.5).to(torch.uint8)
cavi = A.SparseCAVI(A.BernoulliLikelihood(), Phi, kd, y, ctx=ctx, marginal_precision="f16x2-factor", accumulate_precision="f16x2")
for _ in range(3): cavi.sweep()
cavi.check(); torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(20): cavi.sweep()
cavi.check(); torch.cuda.synchronize()
print(f"sweep {(time.perf_counter() - t) / 20 * 1e3:.3f} ms  lib={os.environ.get('AGPL_LIB_AB', 'default')}  checksum {cavi.g.sum().item():.9e}")
