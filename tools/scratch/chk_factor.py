import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import agpl_amd as A
from agpl_amd import _ffi
if os.environ.get("AGPL_LIB_AB"): _ffi.LIB_PATH = os.environ["AGPL_LIB_AB"]
ctx = A.Context(0, seed=1)
p = lambda t: C.c_void_p(t.data_ptr())
for M, L in ((128, 1), (256, 1), (512, 1), (1024, 1), (512, 9)):
    rng = np.random.default_rng(M + L)
    B = rng.normal(size=(L, M, M + 7)) / np.sqrt(M)
    G = np.einsum("lik,ljk->lij", B, B) * 5.0
    g = rng.normal(size=(L, M))
    dG, dg = torch.from_numpy(G).cuda(), torch.from_numpy(g).cuda()
    Aw = torch.zeros((L, M, M), dtype=torch.float64, device="cuda"); v = torch.empty((L, M), dtype=torch.float64, device="cuda")
    errs = []
    for rep in range(3):
        ctx.call("agpl_gaussian_factor", C.c_int32(M), C.c_int32(L), p(dG), p(dg), C.c_void_p(0), p(Aw), p(v), C.c_void_p(0))
        ctx.synchronize()
        Ut = np.triu(Aw.cpu().numpy()[0]); S = np.linalg.inv(np.eye(M) + G[0])
        E = np.abs(Ut @ Ut.T - S); errs.append(E.max() / np.abs(S).max())
    # where is U wrong? compare with numpy U
    R = np.linalg.cholesky(np.eye(M) + G[0]); Uref = np.linalg.inv(R)
    D = np.abs(Ut.T - Uref)
    bi = np.argwhere(D > 1e-9 * np.abs(Uref).max())
    print(M, L, "relerr", ["%.1e" % e for e in errs], "bad entries", len(bi), "blocks", sorted({(int(a) // 32, int(b) // 32) for a, b in bi})[:12])
