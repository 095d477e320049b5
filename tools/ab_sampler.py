"""A/B of the PG sampler kernel (aux_sample_kernel) between builds of the library: python tools/ab_sampler.py libagpl.so libagpl_v1.so ...
Each library runs in its own process; prints ms per launch for Bernoulli (1e7 points) and NegBin r = 15 (4e6 points) and a checksum
of the draws (identical across builds = the same draws).  A library built with `make PGTRACE=1 OUT=../libagpl_pgt.so` also reports the
elapsed cycles per wave of the general engine's phases (negative binomial; the Bernoulli kernel is not instrumented)."""
import ctypes as C
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child():
    import torch

    import agpl_amd as A
    from agpl_amd import _ffi

    if os.environ.get("AGPL_LIB_AB"):
        _ffi.LIB_PATH = os.path.join(os.path.dirname(_ffi.LIB_PATH), os.environ["AGPL_LIB_AB"])
    ctx = A.Context(0, seed=5)
    out = {"lib": os.path.basename(_ffi.LIB_PATH)}
    _ffi.lib().agpl_timing(ctx.bind(), C.c_int32(-1), None, None)
    import numpy as np

    for name, lik, n in (("bernoulli", A.BernoulliLikelihood(), 10_000_000), ("negbin", A.NegativeBinomialLikelihood(15.0), 4_000_000),
                         ("categorical", A.CategoricalLikelihood(np.zeros(10)), 1_000_000)):
        _, y = A.synth_xy(lik, 20240807, 0, n, ctx=ctx, want_x=False)
        g = torch.Generator(device="cuda").manual_seed(1)
        L = A.nlatent(lik)
        f = torch.randn((n,) if L == 1 else (n, L), dtype=torch.float64, device="cuda", generator=g) * 1.5
        Om = A.aux_sample(lik, y, f, ctx=ctx, sweep=1)
        ms, cnt = C.c_double(), C.c_int64()
        _ffi.lib().agpl_timing(ctx.bind(), 3, C.byref(ms), C.byref(cnt))
        best = 1e9
        for _ in range(4):
            A.aux_sample_(Om, lik, y, f, ctx=ctx, sweep=1)
            _ffi.lib().agpl_timing(ctx.bind(), 3, C.byref(ms), C.byref(cnt))
            best = min(best, ms.value / max(cnt.value, 1))
        if hasattr(_ffi.lib(), "agpl_debug_pgtrace"):
            tr = (C.c_ulonglong * 8)()
            _ffi.lib().agpl_debug_pgtrace(tr, 1)
            A.aux_sample_(Om, lik, y, f, ctx=ctx, sweep=1)
            ctx.synchronize()
            _ffi.lib().agpl_debug_pgtrace(tr, 1)
            nw = (n + 63) // 64
            out[name + "_cycles_per_wave"] = dict(zip(["setup", "sync0+sum", "A", "waitA", "B1", "waitB1", "B2+C"], [round(v / nw) for v in tr][:7]))
        om = Om.ω if hasattr(Om, "ω") else Om[0]
        out[name] = {"ms": round(best, 4), "checksum": float(om.sum().item()), "max": float(om.max().item())}
    print(json.dumps(out))


if __name__ == "__main__":
    if os.environ.get("AGPL_AB_CHILD"):
        child()
    else:
        for lib in sys.argv[1:] or ["libagpl.so"]:
            env = dict(os.environ, AGPL_AB_CHILD="1", AGPL_LIB_AB=lib)
            r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            print(line[-1] if line else ("FAILED " + lib + " " + r.stderr[-400:]))
