#!/usr/bin/env python3
"""BASELINE config C5: StudentTLikelihood Gibbs path, full-rank N x N conditional solve on one GPU.
    python tools/bench_c5.py [--n 65536] [--steps 3]
Prints one JSON line (Gibbs sweeps/s; float64 Cholesky rate of the dominant rocSOLVER potrf)."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import agpl_amd as A

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=65536)
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--warmup", type=int, default=1)
args = ap.parse_args()
N = args.n
ctx = A.Context(0, seed=20240807)
lik = A.StudentTLikelihood(3.5, 2.0)  # examples/studentt/script.jl:17-19
x, y32 = A.synth_xy(lik, 20240807, 0, N, ctx=ctx)
x, order = torch.sort(x)  # sorted inputs as in examples/studentt/script.jl:14 (plotting order); y follows x
y = y32.to(torch.float64)[order].contiguous()
t0 = time.time()
K = torch.empty((N, N), dtype=torch.float64, device="cuda")
ell = 2.0  # with_lengthscale(SqExponentialKernel(), 2.0), examples/studentt/script.jl:15
for r0 in range(0, N, 4096):  # row blocks: no N x N temporaries
    r1 = min(N, r0 + 4096)
    blk = K[r0:r1]
    torch.sub(x[r0:r1, None], x[None, :], out=blk)
    blk.div_(ell).pow_(2).mul_(-0.5).exp_()
K.diagonal().add_(1e-6)  # LatentGP(gp, lik, 1e-6), script.jl:18
torch.cuda.synchronize()
t_k = time.time() - t0
t0 = time.time()
dg = A.DenseGibbs(lik, K, y, ctx=ctx)
torch.cuda.synchronize()
t_chol = time.time() - t0
for _ in range(args.warmup):
    dg.sweep()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(args.steps):
    dg.sweep()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / args.steps
f = dg.f
print(json.dumps({
    "metric": "Gibbs sweeps/sec (full-rank N x N conditional solve)", "value": round(1.0 / dt, 4), "unit": "sweeps/s",
    "ms_per_step": round(dt * 1e3, 1), "n_gpus": 1, "dtype": "f64", "data": "synthetic",
    "config": {"workload": f"StudentT(3.5, 2.0) full-rank Gibbs, N={N}, SE kernel lengthscale 2.0, jitter 1e-6"},
    "potrf_tflops_f64": round(N ** 3 / 3 / dt / 1e12, 2), "setup_K_s": round(t_k, 2), "setup_chol_s": round(t_chol, 2),
    "f_finite": bool(torch.isfinite(f).all().item()),
    "rmse_f_vs_truth": float((f - (2.0 * torch.sin(0.7 * x) + torch.cos(0.23 * x))).pow(2).mean().sqrt().item()),
    "hbm_gb": round(torch.cuda.max_memory_allocated() / 1e9, 1)}))
