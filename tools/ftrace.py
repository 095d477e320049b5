#!/usr/bin/env python3
"""Development aid: wall-clock stamps of the roles of factor_pipe_kernel per block step (diagnostic build:
tools/build_variant.sh ftrace agpl_factor.hip "-DAGPL_FTRACE"; AGPL_LIB_AB=.../libagpl_ftrace.so python tools/ftrace.py M).
Rows: workgroup (0 = F, 1.. = P, then T), columns: the stamp ids of that role; values in microseconds since the launch's first stamp."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import agpl_amd as A
from agpl_amd import _ffi

if os.environ.get("AGPL_LIB_AB"):
    _ffi.LIB_PATH = os.environ["AGPL_LIB_AB"]
M = int(sys.argv[1])
ctx = A.Context(0, seed=1)
g = torch.Generator(device="cuda").manual_seed(0)
B = torch.randn((1, M, 2 * M), dtype=torch.float64, device="cuda", generator=g) / (2 * M) ** 0.5
G = (B @ B.transpose(1, 2) * 3.0).contiguous()
gv = torch.randn((1, M), dtype=torch.float64, device="cuda", generator=g)
Aw = torch.empty((1, M, M), dtype=torch.float64, device="cuda")
v = torch.empty((1, M), dtype=torch.float64, device="cuda")
p = lambda t: C.c_void_p(t.data_ptr())
for _ in range(5):
    ctx.call("agpl_gaussian_factor", C.c_int32(M), C.c_int32(1), p(G), p(gv), C.c_void_p(0), p(Aw), p(v), C.c_void_p(0))
ctx.synchronize()
buf = (C.c_ulonglong * (16 * 32 * 8))()
assert _ffi.lib().agpl_debug_ftrace(buf) == 0
t = np.array(buf[:], dtype=np.float64).reshape(16, 32, 8) / 100.0  # wall_clock64: 100 MHz -> us
t0 = t[t > 0].min()
nb = M // 32
names = {0: "F  [start, factored, W flagged, hand seen, T1/T2 in LDS, P0 flagged]",
         1: "P0 [start, W seen, W in LDS, panel+diag done & ready, p0/crit seen, P0 in LDS, own update done]",
         2: "T0 [start, ready seen, staged, crit, done]"}
for wg in range(16):
    if not (t[wg] > 0).any():
        continue
    print(f"--- workgroup {wg}  {names.get(min(wg, 2), '')}")
    for k in range(min(nb, 32)):
        row = t[wg, k]
        if not (row > 0).any():
            continue
        print(f"  k={k:2d} " + " ".join(f"{(x - t0):8.2f}" if x > 0 else "       -" for x in row))
print("total us:", t.max() - t0)
