#!/usr/bin/env python3
"""Soak of the slab reduction's in-launch second level (reduce_slab_kernel: the workgroup that arrives last at a segment's counter sums
the group partials other workgroups stored write-through): for every (N, M, L) the first G, g checked against float64, then REPEATS
more launches on the same input that must reproduce them bit for bit -- a partial read before it was visible, or a counter left
non-zero, shows as different bits.  python3 tools/soak_reduce.py [repeats]  -> one JSON line"""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import agpl_amd as A  # noqa: E402

REPEATS = int(sys.argv[1]) if len(sys.argv) > 1 else 300
ctx = A.Context(0, seed=3)
p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
out = {"repeats": REPEATS, "cases": {}}
gen = torch.Generator(device="cuda").manual_seed(11)
# (slices per case = ceil(N / agpl_slice_points): 64 per group -- one group, two, several, and a ragged last group)
for N, M, L in ((100_000, 128, 1), (600_000, 128, 3), (1_000_000, 256, 1), (1_000_000, 256, 2), (2_000_000, 512, 1), (700_001, 384, 2),
                (1_250_000, 1024, 1), (3_000_000, 256, 1)):
    Phi = (torch.randn((N, M), dtype=torch.float32, device="cuda", generator=gen) / M ** 0.5).contiguous()
    gamma = torch.rand((L, N), dtype=torch.float32, device="cuda", generator=gen) * 0.25
    beta = torch.randn((L, N), dtype=torch.float32, device="cuda", generator=gen)
    G = torch.zeros((L, M, M), dtype=torch.float64, device="cuda")
    g = torch.zeros((L, M), dtype=torch.float64, device="cuda")

    def run():
        ctx.call("agpl_accumulate", C.c_int64(N), C.c_int32(M), C.c_int32(L), p(Phi), p(beta), p(gamma), p(G), p(g))

    run()
    ctx.synchronize()
    dG = dg = 0.0
    for l in range(L):  # float64 reference in slices (memory)
        Gr = torch.zeros((M, M), dtype=torch.float64, device="cuda")
        gr = torch.zeros((M,), dtype=torch.float64, device="cuda")
        for s in range(0, N, 250_000):
            P = Phi[s:s + 250_000].double()
            Gr += (P * gamma[l, s:s + 250_000].double().unsqueeze(1)).T @ P
            gr += P.T @ beta[l, s:s + 250_000].double()
        dG = max(dG, float(((G[l] - Gr).abs().max() / Gr.abs().max()).item()))
        dg = max(dg, float(((g[l] - gr).abs().max() / gr.abs().max()).item()))
    sym = bool(torch.equal(G, G.transpose(1, 2)))
    ref = (G.clone(), g.clone())
    bad = 0
    for i in range(REPEATS):
        if i % 20 == 0:  # (dirty outputs: a launch that skipped a segment cannot hide behind the previous result)
            G.fill_(float("nan"))
            g.fill_(float("nan"))
        run()
        if i % 5 == 4 or i == REPEATS - 1:
            ctx.synchronize()
            if not (torch.equal(G, ref[0]) and torch.equal(g, ref[1])):
                bad += 1
    out["cases"][f"N={N},M={M},L={L}"] = {"rel_dG": dG, "rel_dg": dg, "symmetric": sym, "checks_differing": bad}
    del Phi, gamma, beta
out["pass"] = all(c["checks_differing"] == 0 and c["symmetric"] and c["rel_dG"] < 1e-5 and c["rel_dg"] < 1e-5 for c in out["cases"].values())
print(json.dumps(out))
