#!/usr/bin/env python3
"""Development: per-wave cycle breakdown of the marginal kernel's stage loop (marginal_factor_queue_kernel).  Needs the diagnostic
build: tools/build_variant.sh mtrace agpl_split.hip "-DAGPL_MTRACE".  python tools/mtrace.py N M  (AGPL_LIB_AB=libagpl_mtrace.so)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import agpl_amd as A
from agpl_amd import _ffi
if os.environ.get("AGPL_LIB_AB"):
    _ffi.LIB_PATH = os.path.join(os.path.dirname(_ffi.LIB_PATH), os.environ["AGPL_LIB_AB"])
import bench
N, M = int(sys.argv[1]), int(sys.argv[2])
ctx = A.Context(0, seed=1); lib = _ffi.lib()
lik = A.BernoulliLikelihood()
y, Phi, kd = bench.build_workload(A, ctx, lik, 0, N, M)
cavi = A.SparseCAVI(lik, Phi, kd, y, ctx=ctx)
for _ in range(4):
    cavi.sweep()
cavi.check()
buf = (C.c_ulonglong * (256 * 16 * 4))()
lib.agpl_debug_mtrace(buf)
a = np.frombuffer(buf, dtype=np.uint64).reshape(256, 16, 4).astype(np.float64)
st = a[:, :, 3].clip(min=1)
print(f"N={N} M={M}: per stage and wave, mean over {int((a[:, :, 3] > 0).sum())} waves: total {np.mean(a[:, :, 0] / st):.0f} cycles, "
      f"waiting for the stage's DMA {np.mean(a[:, :, 1] / st):.0f}, at the barrier {np.mean(a[:, :, 2] / st):.0f}; stages per wave {st.mean():.0f}")
for wr in range(4):
    sel = a[:, wr * 4:(wr + 1) * 4, :]
    s2 = sel[:, :, 3].clip(min=1)
    print(f"  row group {wr}: total {np.mean(sel[:, :, 0] / s2):.0f}  dma wait {np.mean(sel[:, :, 1] / s2):.0f}  barrier {np.mean(sel[:, :, 2] / s2):.0f}")
buf2 = (C.c_ulonglong * (256 * 16 * 18))()
lib.agpl_debug_mtrace_phase(buf2)
ph = np.frombuffer(buf2, dtype=np.uint64).reshape(256, 16, 9, 2).astype(np.float64)
tot, cnt = ph[:, :, :, 0].sum((0, 1)), ph[:, :, :, 1].sum((0, 1)).clip(min=1)
print("  stage length by kind (barrier exit to barrier exit), cycles: full", round(tot[0] / cnt[0]), " diagonal 1..8:",
      [int(round(tot[k] / cnt[k])) for k in range(1, 9)], " share of stages: full %.2f" % (cnt[0] / cnt.sum()))
