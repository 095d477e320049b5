#!/usr/bin/env python3
"""Soak test of bitwise run-to-run reproducibility on one GPU: every hot kernel launched REPEATS times on the same inputs,
counting launches whose outputs differ from the first (DESIGN 4.4d item 8 is why this exists).
python3 tools/soak_repro.py [repeats]   -> one JSON line"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import agpl_amd as A  # noqa: E402
import bench  # noqa: E402

REPEATS = int(sys.argv[1]) if len(sys.argv) > 1 else 100
ctx = A.Context(0, seed=bench.SEED)
out = {"repeats": REPEATS, "cases": {}}


def count(run, snapshot):
    run()
    torch.cuda.synchronize()
    ref = [t.clone() for t in snapshot() if t is not None]
    bad = 0
    for _ in range(REPEATS):
        run()
        torch.cuda.synchronize()
        if not all(torch.equal(a, b) for a, b in zip([t for t in snapshot() if t is not None], ref)):
            bad += 1
    return bad


for likname, N, M in (("bernoulli", 3_000_000, 512), ("negbin", 1_000_000, 1024), ("categorical", 500_000, 256)):
    lik = bench.make_lik(A, likname)
    y, Phi, kd = bench.build_workload(A, ctx, lik, 0, N, M)
    cavi = A.SparseCAVI(lik, Phi, kd, y, ctx=ctx, marginal_precision="f16x2-factor", accumulate_precision="f16x2",
                        keep_points=True)
    cavi.sweep()
    cavi.sweep()
    cavi.check()
    case = {}
    # the fused pass: marginals -> aux -> accumulation (G, g, per-point c / gamma / beta)
    case["cavi_pass_differs"] = count(cavi.accumulate, lambda: (cavi.G, cavi.g, cavi.gamma, cavi.beta, cavi.c))
    # the M x M update from the same (G, g)
    case["factor_update_differs"] = count(cavi.update, lambda: (cavi.A_work, cavi.v, cavi.plan.v32, cavi.plan.U_hi, cavi.plan.U_lo))
    mv = {}

    def marg():
        mv["m"] = cavi.marginals()

    case["marginals_differs"] = count(marg, lambda: mv["m"])
    del cavi
    # the Gibbs point pass + accumulation at a fixed sweep index
    gib = A.SparseGibbs(lik, Phi, kd, y, ctx=ctx)
    gib.sweep()

    def gpass():
        ctx.sweep = 7  # same Philox streams every time
        gib.accumulate()

    case["gibbs_pass_differs"] = count(gpass, lambda: (gib.G, gib.g, gib.f, gib.omega))
    del gib, Phi, kd, y
    torch.cuda.empty_cache()
    out["cases"][f"{likname},N={N},M={M}"] = case
print(json.dumps(out))
