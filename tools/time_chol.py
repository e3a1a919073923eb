"""times beatamd_chol_inverse_batch against torch.linalg (rocSOLVER) on a stack of Toeplitz covariances"""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import beat_amd

nd, n = int(sys.argv[1]) if len(sys.argv) > 1 else 64, int(sys.argv[2]) if len(sys.argv) > 2 else 4096
ctx = beat_amd.get_context(0)
ctx.use_torch_stream()
t = torch.arange(n, device="cuda:0", dtype=torch.float64)
base = torch.exp(-(t[:, None] - t[None, :]).abs() / 10.0) + 1e-3 * torch.eye(n, device="cuda:0", dtype=torch.float64)
C = torch.stack([base * (0.5 + 0.01 * i) for i in range(nd)])
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    W, lp = ctx.chol_inverse_batch(C)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    L = torch.linalg.cholesky(C)
    lp_t = 2.0 * torch.log(torch.diagonal(L, dim1=1, dim2=2)).sum(1)
    K = torch.linalg.cholesky(torch.cholesky_inverse(L))
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print("nd %d n %d: hand-written %.1f ms (%.1f TF on 2/3 n^3), torch.linalg %.1f ms" % (
        nd, n, (t1 - t0) * 1e3, nd * 2 / 3 * n ** 3 / (t1 - t0) / 1e12, (t2 - t1) * 1e3))
Wt = K.transpose(1, 2)
print("max |W - W_torch| / max|W|:", float((W - Wt).abs().max() / Wt.abs().max()), " logdet diff:", float((lp - lp_t).abs().max()))
I = W[0].T @ W[0] @ C[0]
print("||W^T W C - I||_max:", float((I - torch.eye(n, device="cuda:0", dtype=torch.float64)).abs().max()))
