#!/usr/bin/env python3
"""Same-box A/B of two builds of libagpl.so on the two contraction kernels: one process per build (AGPL_LIB_AB=<path of the .so>),
the bench workload's features (SE kernel, whitened) at (N, M), `reps` sweeps timed by the in-library hipEvents, and SHA-256 of the
natural parameters after the sweeps -- a variant whose arithmetic is unchanged must print the same hashes.
Usage: ab_lib.py N M [reps] [lik]"""
import ctypes as C
import hashlib
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import agpl_amd as A
from agpl_amd import _ffi

if os.environ.get("AGPL_LIB_AB"):
    _ffi.LIB_PATH = os.environ["AGPL_LIB_AB"]
import bench

N, M = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 8
likname = sys.argv[4] if len(sys.argv) > 4 else "bernoulli"
ctx = A.Context(0, seed=bench.SEED)
lik = bench.make_lik(A, likname)
y, Phi, kd = bench.build_workload(A, ctx, lik, 0, N, M)
cavi = A.SparseCAVI(lik, Phi, kd, y, ctx=ctx)
ACC_ONLY = os.environ.get("AGPL_AB_ACC_ONLY") == "1"  # probe builds (wrong sums): time the pass alone, never factorise its G
for _ in range(3):
    cavi.accumulate() if ACC_ONLY else cavi.sweep()
cavi.check()
_ffi.lib().agpl_timing(ctx.bind(), C.c_int32(-1), None, None)
bench.read_timing(ctx, 0)
bench.read_timing(ctx, 1)
torch.cuda.synchronize()
import time

t0 = time.perf_counter()
for _ in range(reps):
    cavi.accumulate() if ACC_ONLY else cavi.sweep()
cavi.check()
dt = (time.perf_counter() - t0) / reps * 1e3
km = [bench.read_timing(ctx, w) for w in (0, 1)]
h = lambda t: hashlib.sha256(t.detach().cpu().numpy().tobytes()).hexdigest()[:16]
print(json.dumps({"lib": os.path.basename(_ffi.LIB_PATH), "N": N, "M": M, "lik": likname, "ms_per_sweep": round(dt, 3),
                  "marginal_ms": round(km[0][0] / max(km[0][1], 1), 4), "accumulate_ms": round(km[1][0] / max(km[1][1], 1), 4),
                  "sha_G": h(cavi.G), "sha_g": h(cavi.g)}))
