"""Run-to-run reproducibility of the accumulation's (G, g) on the resident workload: 30 repeats of the fused pass at two
sizes, counting launches whose G or g differ bitwise from the first (profiles/r02_g_reproducibility_variants.jsonl: the
AGPL_G_VARIANT investigation builds of agpl_mfma.hip, `make GVAR=n`).  python3 tools/g_repro_check.py"""
import os, sys, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, agpl_amd as A, bench
ctx = A.Context(0, seed=bench.SEED)
lik = bench.make_lik(A, "bernoulli")
out = {}
for (N, M) in ((2_000_000, 512), (1_000_000, 256)):
    y, Phi, kd = bench.build_workload(A, ctx, lik, 0, N, M)
    cavi = A.SparseCAVI(lik, Phi, kd, y, ctx=ctx, marginal_precision="f16x2-factor", accumulate_precision="f16x2")
    cavi.sweep(); cavi.check()
    cavi.accumulate(); torch.cuda.synchronize()
    G0, g0 = cavi.G.clone(), cavi.g.clone()
    bad_g = bad_G = 0
    worst = 0.0
    for r in range(30):
        cavi.accumulate(); torch.cuda.synchronize()
        if not torch.equal(cavi.G, G0): bad_G += 1
        if not torch.equal(cavi.g, g0):
            bad_g += 1
            worst = max(worst, float((cavi.g - g0).abs().max() / g0.abs().max()))
    out[f"N={N},M={M}"] = {"repeats": 30, "G_differs": bad_G, "g_differs": bad_g, "worst_rel_dg": worst}
    del cavi, Phi, kd, y
print(json.dumps(out))
