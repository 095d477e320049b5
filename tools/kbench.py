#!/usr/bin/env python3
"""Micro-benchmark of the two MFMA kernels (in-library hipEvent timing). Usage: kbench.py N M [reps]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import agpl_amd as A
from agpl_amd import _ffi

if os.environ.get("AGPL_LIB_AB"):  # A/B a second build of libagpl.so in the same gpurun call
    _ffi.LIB_PATH = os.environ["AGPL_LIB_AB"]

N, M = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
ctx = A.Context(0, seed=1)
g = torch.Generator(device="cuda").manual_seed(0)
Phi = torch.randn((N, M), device="cuda", generator=g) * 0.1
kd = torch.ones(N, device="cuda")
y = (torch.rand(N, device="cuda", generator=g) < 0.5).to(torch.uint8)
cavi = A.SparseCAVI(A.BernoulliLikelihood(), Phi, kd, y, ctx=ctx, marginal_precision=os.environ.get("AGPL_PREC", "auto"))  # AGPL_PREC=f32: the float32-input pair
cavi.sweep()
torch.cuda.synchronize()
_ffi.lib().agpl_timing(ctx.bind(), C.c_int32(-1), None, None)
for _ in range(reps):
    cavi.accumulate()
torch.cuda.synchronize()
for which, nm in ((0, "marginal"), (1, "syrk")):
    ms, cnt = C.c_double(), C.c_int64()
    _ffi.lib().agpl_timing(ctx.bind(), which, C.byref(ms), C.byref(cnt))
    print(f"{nm:9s} N={N} M={M} avg {ms.value / cnt.value:8.3f} ms  prec={cavi.marginal_precision}")
