#!/usr/bin/env python3
"""Development: per-wave cycle breakdown of syrk_image_kernel's stage loop (AGPL_SYRKQ_TRACE=1)."""
import ctypes as C, os, sys
pass  # needs the diagnostic build: make -C augmentedgplikelihoods.jl_amd/csrc QTRACE=1
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import agpl_amd as A
from agpl_amd import _ffi
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
a = np.frombuffer(buf, dtype=np.uint64).reshape(64, 16, 8).astype(np.float64)
for b in (0, 14):
    ns = int(a[b, 0, 6]) & 0xffffffff
    print("block", b, "per wave (simd = w & 3): wait / head / rest / total per stage")
    for w in range(16 if a[b,8,7] > 0 else 8):
        r = np.frombuffer(buf, dtype=np.uint64).reshape(64, 16, 8)
        h = [int(r[b,w,0]) & 0xffffffff, int(r[b,w,0]) >> 32, int(r[b,w,3]) & 0xffffffff, int(r[b,w,3]) >> 32]
        print(f"  w{w:2d} simd {w&3} act {(int(a[b,w,6])>>33)&1}: wait {a[b,w,2]/ns:6.0f} body {a[b,w,4]/ns:6.0f} total {a[b,w,1]/ns:6.0f}  hooks 0/2/4/6 at " + " ".join(f"{x/ns:6.0f}" for x in h))
for b in range(0, 64, 21):
    ns = int(a[b, 0, 6]) & 0xffffffff; dg = (int(a[b, 0, 6]) >> 32) & 1
    act = [(int(a[b, w, 6]) >> 33) & 1 for w in range(16)]
    clk = a[b, :, 7].mean() / a[b, :, 5].mean() * 100.0
    w = [i for i in range(16) if act[i]]
    print(f"block {b} diag {dg} nstage {ns} clock {clk:.0f} MHz  prologue {a[b,:,0].mean():.0f} cyc; per stage (active waves): total {a[b,w,1].mean()/ns:.0f} wait+barrier {a[b,w,2].mean()/ns:.0f} head {a[b,w,3].mean()/ns:.0f} rest {a[b,w,4].mean()/ns:.0f}"
          f" | inactive: wait {np.mean([a[b,i,2] for i in range(16) if not act[i]] or [0])/ns:.0f}")
