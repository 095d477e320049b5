import os, sys, ctypes as C
sys.path.insert(0, os.getcwd())
import torch, agpl_amd as A, bench
from agpl_amd import _ffi
ctx = A.Context(0, seed=bench.SEED)
N, M = 10_000_000, 1024
likb = bench.make_lik(A, "bernoulli"); likn = bench.make_lik(A, "negbin")
yb, Phi, kd = bench.build_workload(A, ctx, likb, 0, N, M)
_, yn = A.synth_xy(likn, bench.SEED, 0, N, ctx=ctx, want_x=False)
def timing(which):
    ms, cnt = C.c_double(), C.c_int64()
    _ffi.lib().agpl_timing_read(ctx.bind(), which, C.byref(ms), C.byref(cnt))
    return ms.value / max(cnt.value, 1)
_ffi.lib().agpl_timing_enable(ctx.bind(), 1)
for name, lik, y in (("bernoulli", likb, yb), ("negbin", likn, yn), ("bernoulli", likb, yb), ("negbin", likn, yn)):
    cavi = A.SparseCAVI(lik, Phi, kd, y, ctx=ctx)
    out = []
    for it in range(8):
        cavi.sweep(); ctx.synchronize()
        out.append((round(timing(0), 2), round(timing(1), 2)))
    Uh = cavi.U_hi.float() if hasattr(cavi, "U_hi") else None
    info = {}
    if Uh is not None:
        Ul = cavi.U_lo.float()
        info = {"U_hi_absmean": float(Uh.abs().mean()), "U_lo_nonzero_frac": float((Ul != 0).float().mean()),
                "U_lo_denormal_frac": float(((Ul != 0) & (Ul.abs() < 6.1e-5)).float().mean()),
                "U_hi_denormal_frac": float(((Uh != 0) & (Uh.abs() < 6.1e-5)).float().mean())}
    print(name, "marginal/strip ms per sweep:", out, info, flush=True)
    del cavi
