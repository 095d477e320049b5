#!/usr/bin/env python3
"""Same-box A/B of the factor-form marginal kernels (AGPL_MARGINAL_STAGE is read per call):
    22  marginal_split256_kernel<2, factor, 2>     one workgroup per tile, 32x32x16 MFMA (round-1 shipped)
    16  marginal_factor16_kernel                   one workgroup per tile, 16x16x32 MFMA
    132 marginal_factor_persist_kernel<false>      persistent workgroups,  32x32x16 MFMA
    116 marginal_factor_persist_kernel<true>       persistent workgroups,  16x16x32 MFMA
python tools/ab_marginal.py [--lik bernoulli --n 10000000 --m 512] ...  prints one JSON line per configuration."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import agpl_amd as A  # noqa: E402
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--lik", default="bernoulli")
ap.add_argument("--n", type=int, default=10_000_000)
ap.add_argument("--m", type=int, default=512)
ap.add_argument("--reps", type=int, default=8)
ap.add_argument("--cfgs", default="22,16,132,116")
ap.add_argument("--envs", default="", help="';'-separated VAR=VALUE variants of the shipped kernel (stage 200) to compare instead of --cfgs")
args = ap.parse_args()

ctx = A.Context(0, seed=bench.SEED)
lik = bench.make_lik(A, args.lik)
y, Phi, kd = bench.build_workload(A, ctx, lik, 0, args.n, args.m)
os.environ["AGPL_MARGINAL_STAGE"] = "22"
cavi = A.SparseCAVI(lik, Phi, kd, y, ctx=ctx, marginal_precision="f16x2-factor", accumulate_precision="f16x2")
for _ in range(3):
    cavi.sweep()
cavi.check()
ref = None
out = {"lik": args.lik, "N": args.n, "M": args.m, "L": A.nlatent(lik), "results": {}}
for rnd in range(2):  # two rounds: order effects / clock drift show up as a difference between them
    for cfg in (args.envs.split(";") if args.envs else args.cfgs.split(",")):
        if args.envs:
            os.environ["AGPL_MARGINAL_STAGE"] = "200"
            k_, v_ = cfg.split("=")
            os.environ[k_] = v_
        else:
            os.environ["AGPL_MARGINAL_STAGE"] = cfg
        mu, var = cavi.marginals()
        torch.cuda.synchronize()
        bench._ffi = None
        from agpl_amd import _ffi
        _ffi.lib().agpl_timing_enable(ctx.bind(), 1)
        for _ in range(args.reps):
            mu, var = cavi.marginals()
        ms, cnt = bench.read_timing(ctx, 0)
        _ffi.lib().agpl_timing_enable(ctx.bind(), 0)
        if ref is None:
            ref = (mu.clone(), var.clone())
        dmu = float((mu - ref[0]).abs().max() / ref[0].abs().max().clamp_min(1e-30))
        dvar = float((var - ref[1]).abs().max() / ref[1].abs().max())
        r = out["results"].setdefault(cfg, {"avg_ms": [], "rel_dmu_vs_22": dmu, "rel_dvar_vs_22": dvar})
        r["avg_ms"].append(round(ms / cnt, 4))
        r["rel_dmu_vs_22"], r["rel_dvar_vs_22"] = max(r["rel_dmu_vs_22"], dmu), max(r["rel_dvar_vs_22"], dvar)
        r["finite"] = bool(torch.isfinite(mu).all() and torch.isfinite(var).all())
print(json.dumps(out), flush=True)
