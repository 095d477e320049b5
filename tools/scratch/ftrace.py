#!/usr/bin/env python3
"""Phase stamps of the factor kernel (debug build: make -C csrc EXTRA_ALL=-DAGPL_FTRACE OUT=../../ab/libagpl_trace.so).
Usage: AGPL_LIB_AB=ab/libagpl_trace.so python tools/scratch/ftrace.py [M]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import agpl_amd as A
from agpl_amd import _ffi
_ffi.LIB_PATH = os.environ["AGPL_LIB_AB"]
M = int(sys.argv[1]) if len(sys.argv) > 1 else 512
ctx = A.Context(0, seed=1)
rng = np.random.default_rng(0)
B = rng.normal(size=(1, M, 2 * M)) / np.sqrt(2 * M)
G = torch.tensor(np.einsum("lik,ljk->lij", B, B) * 3.0, device="cuda"); g = torch.zeros((1, M), dtype=torch.float64, device="cuda")
Aw = torch.empty((1, M, M), dtype=torch.float64, device="cuda"); v = torch.empty((1, M), dtype=torch.float64, device="cuda")
def run():
    ctx.call("agpl_gaussian_factor", C.c_int32(M), C.c_int32(1), C.c_void_p(G.data_ptr()), C.c_void_p(g.data_ptr()), C.c_void_p(0),
             C.c_void_p(Aw.data_ptr()), C.c_void_p(v.data_ptr()), C.c_void_p(0), C.c_void_p(0), C.c_void_p(0), C.c_void_p(0))
for _ in range(5):
    try: run()
    except Exception as e: err = e
torch.cuda.synchronize()
buf = (C.c_ulonglong * (8 * 16 * 8))()
assert _ffi.lib().agpl_debug_ftrace(buf) == 0
t = np.array(buf, dtype=np.float64).reshape(8, 16, 8) * 0.01  # 100 MHz -> us
t0 = t[0, 0, 0]
nb = M // 32
print("spine: k  wait  stage factor panel publish critwait lookahead | total   (us)")
for k in range(nb):
    r = t[0, k]
    e = r[7] if k < nb - 1 else r[5]
    cw, la = (r[6]-r[5], r[7]-r[6]) if k < nb - 1 else (0.0, 0.0)
    print(f"  {k:2d} {r[1]-r[0]:6.2f} {r[2]-r[1]:6.2f} {r[3]-r[2]:6.2f} {r[4]-r[3]:6.2f} {r[5]-r[4]:6.2f} {cw:6.2f} {la:6.2f} | {e-r[0]:6.2f}   start {r[0]-t0:7.2f}")
for w in (1,):
    print(f"tile wg {w}: k  spin  load  crit  rest | total")
    for k in range(nb - 2):
        r = t[w, k]
        print(f"  {k:2d} {r[1]-r[0]:6.2f} {r[2]-r[1]:6.2f} {r[3]-r[2]:6.2f} {r[4]-r[3]:6.2f} | {r[4]-r[0]:6.2f}   start {r[0]-t0:7.2f}")
print("kernel span (first stamp -> last spine publish):", t[0, nb - 1, 5] - t0)
