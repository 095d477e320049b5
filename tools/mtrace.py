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
buf2 = (C.c_ulonglong * (256 * 16 * 24))()
lib.agpl_debug_mtrace_phase(buf2)
ph = np.frombuffer(buf2, dtype=np.uint64).reshape(256, 16, 3, 8).astype(np.float64)
rt = ph[:, :, 0, 6]
print(f"  stage loop of a workgroup, median: {np.median(rt[rt > 0]) / 100:.1f} us = {np.median(a[:, :, 0][rt > 0]) / 1e6:.3f} M cycles")
print(f"  in-kernel clock (s_memtime / s_memrealtime x 100 MHz), median over waves: {np.median(a[:, :, 0][rt > 0] / rt[rt > 0]) * 0.1:.3f} GHz")
if a[:, :, 3].max() <= 1:  # the clock-only build (-DAGPL_MCLOCK): stage lengths by kind from one stamp per stage
    raw = np.frombuffer(buf2, dtype=np.uint64).reshape(256, 16, 24).astype(np.float64)
    for wr in (None, 0, 3):
        sel = raw if wr is None else raw[:, wr * 4:(wr + 1) * 4]
        sums = [sel[:, :, q].sum() for q in (0, 1, 2, 3, 4, 5, 18, 7, 8)]
        cnts = [max(sel[:, :, 9 + q].sum(), 1) for q in range(9)]
        print(f"  stage length by kind, barrier exit to barrier exit ({'all waves' if wr is None else 'row group %d' % wr}): full {sums[0] / cnts[0]:.0f}  diagonal 1..8: "
              + " ".join(f"{sums[q] / cnts[q]:.0f}" for q in range(1, 9)) + f"   share of full stages {cnts[0] / sum(cnts):.2f}")
    sys.exit(0)
SEG = ["dma wait", "barrier", "to issue", "dma issue", "mfma rest", "item end"]
KIND = ["full stage", "diagonal stage 1..6", "diagonal stage 7, 8"]
print("  cycles per stage by stage kind, row group and segment (barrier exit -> DMA issue = bookkeeping + first fragment reads + first MFMAs):")
for k in range(3):
    cnt = ph[:, :, k, 7]
    print(f"   {KIND[k]} (share of stages {cnt.sum() / ph[:, :, :, 7].sum():.2f})")
    for wr in range(4):
        sel = ph[:, wr * 4:(wr + 1) * 4, k, :]
        c = sel[:, :, 7].sum().clip(min=1)
        segs = [sel[:, :, q].sum() / c for q in range(6)]
        print(f"     row group {wr}: " + "  ".join(f"{SEG[q]} {segs[q]:.0f}" for q in range(6)) + f"  = {sum(segs):.0f}")
