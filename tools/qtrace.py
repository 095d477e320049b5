#!/usr/bin/env python3
"""Development: per-wave cycle breakdown of the accumulation kernel's step loop.  Needs the diagnostic build:
make -C augmentedgplikelihoods.jl_amd/csrc QTRACE=1 (touch agpl_syrk.hip first).  python tools/qtrace.py N M"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import agpl_amd as A
from agpl_amd import _ffi
if os.environ.get("AGPL_LIB_AB"):
    _ffi.LIB_PATH = os.path.join(os.path.dirname(_ffi.LIB_PATH), os.environ["AGPL_LIB_AB"])
N, M = int(sys.argv[1]), int(sys.argv[2])
ctx = A.Context(0, seed=1); lib = _ffi.lib()
g = torch.Generator(device="cuda").manual_seed(0)
Phi = torch.randn((N, M), device="cuda", generator=g) * 0.1
gam = torch.rand((1, N), device="cuda", generator=g) * 0.25
bet = torch.randn((1, N), device="cuda", generator=g)
img = torch.empty(lib.agpl_accumulate_image_bytes(C.c_int64(N), C.c_int32(M)), dtype=torch.uint8, device="cuda")
p = lambda t: C.c_void_p(t.data_ptr())
ctx.call("agpl_accumulate_image", C.c_int64(N), C.c_int32(M), p(Phi), p(img))

G = torch.empty((1, M, M), dtype=torch.float64, device="cuda"); gg = torch.empty((1, M), dtype=torch.float64, device="cuda")
for _ in range(4):
    ctx.call("agpl_accumulate_split", C.c_int64(N), C.c_int32(M), C.c_int32(1), p(Phi), p(img), p(bet), p(gam), p(G), p(gg))
torch.cuda.synchronize()
buf = (C.c_ulonglong * (64 * 16 * 8))()
lib.agpl_debug_qtrace(buf)
a = np.frombuffer(buf, dtype=np.uint64).reshape(64, 16, 8)

for b in (0, 9):
    ns = int(a[b, 0, 6]) & 0xffffffff
    if ns == 0:
        continue
    dg = (int(a[b, 0, 6]) >> 32) & 1
    clk = float(a[b, 0, 1]) / max(float(a[b, 0, 5]), 1) * 100.0
    print(f"block {b} diag {dg} steps {ns} clock {clk:.0f} MHz; per step and wave (strip: total / memory wait / barrier wait):")
    print("   " + "  ".join(f"c{(int(a[b,w,6])>>40)&15}: {a[b,w,1]/ns:.0f}/{a[b,w,0]/ns:.0f}/{a[b,w,2]/ns:.0f}" for w in range(8)))
