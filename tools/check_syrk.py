#!/usr/bin/env python3
"""Quick correctness check of agpl_accumulate against torch float64 (debug aid for kernel variants)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import agpl_amd as A
from agpl_amd import _ffi
if os.environ.get("AGPL_LIB_AB"):
    _ffi.LIB_PATH = os.environ["AGPL_LIB_AB"]
N, M = int(sys.argv[1]), int(sys.argv[2])
ctx = A.Context(0, seed=1)
g = torch.Generator(device="cuda").manual_seed(0)
Phi = torch.randn((N, M), device="cuda", generator=g) * 0.1
gam = torch.rand((1, N), device="cuda", generator=g) * 0.25
bet = torch.randn((1, N), device="cuda", generator=g)
G = torch.empty((1, M, M), dtype=torch.float64, device="cuda"); gg = torch.empty((1, M), dtype=torch.float64, device="cuda")
ctx.call("agpl_accumulate", C.c_int64(N), C.c_int32(M), C.c_int32(1), C.c_void_p(Phi.data_ptr()), C.c_void_p(bet.data_ptr()),
         C.c_void_p(gam.data_ptr()), C.c_void_p(G.data_ptr()), C.c_void_p(gg.data_ptr()))
Gr = torch.zeros((M, M), dtype=torch.float64, device="cuda"); gr = torch.zeros(M, dtype=torch.float64, device="cuda")
for i0 in range(0, N, 1000000):
    P = Phi[i0:i0 + 1000000].double()
    Gr += (P * gam[0, i0:i0 + 1000000].double()[:, None]).t() @ P
    gr += P.t() @ bet[0, i0:i0 + 1000000].double()
dg = (gg[0] - gr).abs()
top = torch.topk(dg, 5)
print("top |dg|:", [(int(i), float(v)) for v, i in zip(top.values, top.indices)], "max|g|", gr.abs().max().item())
print("abl", os.environ.get("AGPL_ABL", "0"), "relG", ((G[0] - Gr).abs().max() / Gr.abs().max()).item(), "relg", ((gg[0] - gr).abs().max() / gr.abs().max()).item())
if os.environ.get("AGPL_DUMP"):
    E = (G[0] - Gr).abs()
    blk = E.reshape(M // 32, 32, M // 32, 32).amax(dim=(1, 3))
    torch.set_printoptions(precision=2, linewidth=200)
    print((blk / Gr.abs().max()).cpu())
