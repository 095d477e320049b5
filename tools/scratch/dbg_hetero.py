import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import agpl_amd as A
from oracle import oracle as O
SEED = 20240807
ctx = A.Context(0, seed=SEED)
lik, olik = A.HeteroscedasticGaussianLikelihood(5.0), O.heterogauss(5.0)
rng = np.random.default_rng(5)
n, L = 3000, 2
f = rng.normal(size=(n, L)) * 2.0
y = rng.normal(size=n)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
for sweep in (7, 8):
    Om = A.aux_sample_(A.init_aux_variables(lik, n, ctx=ctx), lik, dev(y), dev(f), ctx=ctx, sweep=sweep)
    om, nn = Om.ω.cpu().numpy(), Om.n.cpu().numpy()
    ref = O.aux_sample(olik, y, f, seed=SEED, sweep=sweep)
    print("sweep", sweep, "n equal", np.array_equal(nn, ref["n"]), "max rel omega", np.abs(om / ref["omega"] - 1).max())
    al = A.aug_loglik(lik, Om, dev(y), dev(f), ctx=ctx)
    ral = O.aug_loglik(olik, y, om, f, nn)
    print("  device", al, "oracle(on device draw)", ral, "diff", al - ral)
    # bisect over chunks
    for c0 in range(0, n, 500):
        sl = slice(c0, c0 + 500)
        Oc = A.TupleVector(ω=Om.ω[sl].contiguous(), n=Om.n[sl].contiguous())
        a = A.aug_loglik(lik, Oc, dev(y[sl]), dev(f[sl]), ctx=ctx)
        r = O.aug_loglik(olik, y[sl], om[sl], f[sl], nn[sl])
        if abs(a - r) > 1e-8 * abs(r):
            print("   chunk", c0, a - r)
            for i in range(c0, min(n, c0 + 500)):
                Oi = A.TupleVector(ω=Om.ω[i:i + 1].contiguous(), n=Om.n[i:i + 1].contiguous())
                ai = A.aug_loglik(lik, Oi, dev(y[i:i + 1]), dev(f[i:i + 1]), ctx=ctx)
                ri = O.aug_loglik(olik, y[i:i + 1], om[i:i + 1], f[i:i + 1], nn[i:i + 1])
                if abs(ai - ri) > 1e-8 * max(1, abs(ri)):
                    print("      point", i, "omega", om[i], "n", nn[i], "f", f[i], "y", y[i], "dev", ai, "orc", ri)
    print("  identity constant:", al - O.full_conditional_logpdf(olik, y, f, om, nn), " oracle-only:", ral - O.full_conditional_logpdf(olik, y, f, om, nn))
