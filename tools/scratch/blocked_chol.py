import sys, time, torch
N = int(sys.argv[1]); nb = int(sys.argv[2])
torch.manual_seed(0)
x = torch.sort(torch.rand(N, dtype=torch.float64, device="cuda") * 20 - 10).values
K = torch.exp(-0.5 * ((x[:, None] - x[None, :]) / 2.0) ** 2)
g = torch.rand(N, dtype=torch.float64, device="cuda") * 0.5 + 0.1
B = torch.eye(N, dtype=torch.float64, device="cuda") + g.sqrt()[:, None] * K * g.sqrt()[None, :]
del K
torch.cuda.synchronize()
A0 = B.clone()
t = time.time(); L0 = torch.linalg.cholesky(A0); torch.cuda.synchronize(); t_lib = time.time() - t
print(f"rocSOLVER potrf N={N}: {t_lib*1e3:.1f} ms  {N**3/3/t_lib/1e12:.1f} TF")
def blocked(A, nb):
    n = A.shape[0]
    for k in range(0, n, nb):
        e = min(n, k + nb)
        A[k:e, k:e] = torch.linalg.cholesky(A[k:e, k:e])
        if e < n:
            # L21 = A21 L11^-T
            A[e:, k:e] = torch.linalg.solve_triangular(A[k:e, k:e], A[e:, k:e].T, upper=False).T
            # trailing lower block-triangle, one block column at a time
            for j in range(e, n, nb):
                je = min(n, j + nb)
                A[j:, j:je].addmm_(A[j:, k:e], A[j:je, k:e].T, alpha=-1.0)
    return A
A1 = B.clone()
torch.cuda.synchronize(); t = time.time(); blocked(A1, nb); torch.cuda.synchronize(); t_b = time.time() - t
A1 = B.clone()
torch.cuda.synchronize(); t = time.time(); blocked(A1, nb); torch.cuda.synchronize(); t_b = time.time() - t
print(f"blocked nb={nb}: {t_b*1e3:.1f} ms  {N**3/3/t_b/1e12:.1f} TF  maxdiff {(torch.tril(A1) - L0).abs().max().item():.2e}")
