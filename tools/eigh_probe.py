"""How long do the dense steps of one Rayleigh-Ritz take ON THE DEVICE (rocSOLVER behind torch.linalg)?  Decides whether
the <= 248 x 248 generalised problem can move off the host."""
import time
import numpy as np
import scipy.linalg as sla
import torch

dev = torch.device("cuda:0")
for n in (88, 168, 248):
    rng = np.random.default_rng(0)
    A = rng.standard_normal((n, n)); A = A + A.T
    B = rng.standard_normal((n, 2 * n)); B = B @ B.T / n + np.eye(n)
    t0 = time.time()
    for _ in range(20):
        w, v = sla.eigh(A, B, driver="gvd")
    host = (time.time() - t0) / 20
    Ad, Bd = torch.from_numpy(A).to(dev), torch.from_numpy(B).to(dev)

    def devsolve():
        L = torch.linalg.cholesky(Bd)
        Li = torch.linalg.solve_triangular(L, torch.eye(n, dtype=torch.float64, device=dev), upper=False)
        C = Li @ Ad @ Li.T
        w, Z = torch.linalg.eigh(C)
        return w, Li.T @ Z

    for _ in range(3):
        devsolve()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(20):
        devsolve()
    torch.cuda.synchronize()
    d = (time.time() - t0) / 20
    t0 = time.time()
    for _ in range(20):
        torch.linalg.eigh(Ad)
    torch.cuda.synchronize()
    e = (time.time() - t0) / 20
    print(f"n={n}: host scipy eigh(A,B) {host*1e3:.2f} ms; device chol+trsm+gemm+eigh {d*1e3:.2f} ms (eigh alone {e*1e3:.2f} ms)", flush=True)
