import os, sys, time
sys.path.insert(0, "/root/repo")
import torch
import agpl_amd as A
from agpl_amd import _ffi
if os.environ.get("AGPL_LIB_AB"):
    _ffi.LIB_PATH = os.environ["AGPL_LIB_AB"]
N, M = int(sys.argv[1]), 512
ctx = A.Context(0, seed=1)
g = torch.Generator(device="cuda").manual_seed(0)
Phi = torch.randn((N, M), device="cuda", generator=g) * 0.1
kd = torch.ones(N, device="cuda")
y = (torch.rand(N, device="cuda", generator=g) < 0.5).to(torch.uint8)
cavi = A.SparseCAVI(A.BernoulliLikelihood(), Phi, kd, y, ctx=ctx, marginal_precision="f16x2-factor", accumulate_precision="f16x2")
for _ in range(3): cavi.sweep()
torch.cuda.synchronize(); t = time.time()
for _ in range(10): cavi.sweep()
torch.cuda.synchronize()
print(f"N={N} sweep {(time.time()-t)*100:.3f} ms  G00 {cavi.G[0,0,0].item():.12f}")
