import os, sys, time
sys.path.insert(0, "/root/repo")
import torch
import agpl_amd as A
from agpl_amd import _ffi
if os.environ.get("AGPL_LIB_AB"):
    _ffi.LIB_PATH = os.environ["AGPL_LIB_AB"]
import ctypes as C
N, M = 10_000_000, 512
ctx = A.Context(0, seed=1)
lik = A.BernoulliLikelihood()
g = torch.Generator(device="cuda").manual_seed(0)
Phi = torch.randn((N, M), device="cuda", generator=g) * 0.1
kd = torch.ones(N, device="cuda") * 0.1
y = (torch.rand(N, device="cuda", generator=g) < 0.5).to(torch.uint8)
gib = A.SparseGibbs(lik, Phi, kd, y, ctx=ctx, accumulate_precision="f16x2")
for _ in range(2): gib.sweep()
torch.cuda.synchronize()
_ffi.lib().agpl_timing_enable(ctx.bind(), 1)
t = time.time()
for _ in range(4): gib.sweep()
torch.cuda.synchronize()
dt = (time.time() - t) / 4
ms, cnt = C.c_double(), C.c_int64()
_ffi.lib().agpl_timing_read(ctx.bind(), 2, C.byref(ms), C.byref(cnt))
f = torch.linspace(-3, 3, N, dtype=torch.float64, device="cuda")
for _ in range(2): A.aux_sample(lik, y, f, ctx=ctx, sweep=3)
_ffi.lib().agpl_timing_enable(ctx.bind(), 0)
torch.cuda.synchronize(); t = time.time()
for _ in range(5): om = A.aux_sample(lik, y, f, ctx=ctx, sweep=3)
torch.cuda.synchronize(); ts = (time.time() - t) / 5
print(f"gibbs sweep {dt*1e3:.2f} ms  point pass {ms.value/cnt.value:.3f} ms  aux_sample {ts*1e3:.3f} ms  checksum {om.ω.sum().item():.6f}")
