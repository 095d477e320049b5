#!/usr/bin/env python3
"""Same-box A/B of the split-float16 accumulation kernels (AGPL_SYRK is read per call):
    tile   syrk_split_kernel  one 128 x 128 tile per 4-wave workgroup, 4 workgroups per CU (round-1 shipped)
    strip  syrk_strip_kernel  <= 16 sub-tiles over <= 4 staged panels per 16-wave workgroup, 32-point stages
python tools/ab_syrk.py [--lik bernoulli --n 10000000 --m 512]  prints one JSON line."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import agpl_amd as A  # noqa: E402
import bench  # noqa: E402
from agpl_amd import _ffi  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--lik", default="bernoulli")
ap.add_argument("--n", type=int, default=10_000_000)
ap.add_argument("--m", type=int, default=512)
ap.add_argument("--reps", type=int, default=6)
ap.add_argument("--forms", default="tile,strip")
ap.add_argument("--envs", default="", help="';'-separated VAR=VALUE variants of the tile form to compare instead of forms")
args = ap.parse_args()

ctx = A.Context(0, seed=bench.SEED)
lik = bench.make_lik(A, args.lik)
y, Phi, kd = bench.build_workload(A, ctx, lik, 0, args.n, args.m)
cavi = A.SparseCAVI(lik, Phi, kd, y, ctx=ctx, marginal_precision="f16x2-factor", accumulate_precision="f16x2")
os.environ["AGPL_SYRK"] = "tile"
for _ in range(2):
    cavi.sweep()
cavi.check()
ref = None
out = {"lik": args.lik, "N": args.n, "M": args.m, "L": A.nlatent(lik), "results": {}}
for rnd in range(2):
    for form in (args.envs.split(";") if args.envs else args.forms.split(",")):
        if args.envs:
            k, v = form.split("=")
            os.environ[k] = v
        else:
            os.environ["AGPL_SYRK"] = form
        cavi.accumulate()
        torch.cuda.synchronize()
        _ffi.lib().agpl_timing_enable(ctx.bind(), 1)
        for _ in range(args.reps):
            cavi.accumulate()
        bench.read_timing(ctx, 0)
        ms, cnt = bench.read_timing(ctx, 1)
        _ffi.lib().agpl_timing_enable(ctx.bind(), 0)
        G, g = cavi.G.clone(), cavi.g.clone()
        if ref is None:
            ref = (G, g)
        r = out["results"].setdefault(form, {"avg_ms": []})
        r["avg_ms"].append(round(ms / cnt, 4))
        r["rel_dG_vs_first"] = float((G - ref[0]).abs().max() / ref[0].abs().max())
        r["rel_dg_vs_first"] = float((g - ref[1]).abs().max() / ref[1].abs().max())
        r["symmetric"] = bool(torch.equal(G, G.transpose(1, 2)))
        cavi.accumulate()
        r["bitwise_reproducible"] = bool(torch.equal(cavi.G, G) and torch.equal(cavi.g, g))
print(json.dumps(out), flush=True)
