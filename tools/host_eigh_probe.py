#!/usr/bin/env python3
"""Host LAPACK timings for the Rayleigh-Ritz problem of the eigensolver (symmetric n x n, n = 240 / 160 / 80), one thread:
SciPy's dsyevd / dsyevr (all / lowest third) / ssyevd, torch.linalg.eigh (MKL), and the dgemm rate.  python tools/host_eigh_probe.py"""
import time

import numpy as np
import scipy.linalg.lapack as la
import torch
from threadpoolctl import threadpool_limits

rng = np.random.default_rng(0)


def best(fn, reps=30):
    fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return 1e3 * min(ts), 1e3 * float(np.median(ts))


with threadpool_limits(limits=1):
    torch.set_num_threads(1)
    for n in (240, 176, 160, 80):
        A = rng.standard_normal((n, n))
        A = np.asfortranarray(A + A.T + np.diag(np.arange(n) * 3.0))
        A32 = A.astype(np.float32)
        At = torch.from_numpy(np.ascontiguousarray(A))
        k = n // 3
        print(f"n = {n}:")
        print("  scipy dsyevd (all vectors)      min / median ms: %.3f / %.3f" % best(lambda: la.dsyevd(A, compute_v=1, lower=1)))
        print("  scipy ssyevd (fp32)                               %.3f / %.3f" % best(lambda: la.ssyevd(A32, compute_v=1, lower=1)))
        print("  scipy dsyevr (all)                                %.3f / %.3f" % best(lambda: la.dsyevr(A, compute_v=1, lower=1)))
        print("  scipy dsyevr (lowest n/3, range='I')              %.3f / %.3f" % best(lambda: la.dsyevr(A, compute_v=1, lower=1, range="I", il=1, iu=k)))
        print("  scipy dsyevx (lowest n/3)                         %.3f / %.3f" % best(lambda: la.dsyevx(A, compute_v=1, lower=1, range="I", il=1, iu=k)))
        print("  scipy dsyev                                       %.3f / %.3f" % best(lambda: la.dsyev(A, compute_v=1, lower=1)))
        print("  torch.linalg.eigh (fp64, MKL)                     %.3f / %.3f" % best(lambda: torch.linalg.eigh(At)))
        print("  torch.linalg.eigh (fp32)                          %.3f / %.3f" % best(lambda: torch.linalg.eigh(At.float())))
    B, C = rng.standard_normal((256, 256)), rng.standard_normal((256, 240))
    t = best(lambda: B @ C)[0]
    print(f"dgemm 256 x 256 x 240 (numpy/OpenBLAS): {t:.3f} ms = {2 * 256 * 256 * 240 / t / 1e6:.1f} GF/s")
    Bt, Ct = torch.from_numpy(B), torch.from_numpy(C)
    t = best(lambda: Bt @ Ct)[0]
    print(f"dgemm 256 x 256 x 240 (torch/MKL):      {t:.3f} ms = {2 * 256 * 256 * 240 / t / 1e6:.1f} GF/s")

# more than one BLAS thread for the one problem (a single hypothesis at a time leaves the host's other cores idle)
for nthr in (1, 2, 4, 8):
    with threadpool_limits(limits=nthr):
        torch.set_num_threads(nthr)
        for n in (240, 160):
            A = rng.standard_normal((n, n))
            A = np.asfortranarray(A + A.T + np.diag(np.arange(n) * 3.0))
            At = torch.from_numpy(np.ascontiguousarray(A))
            print(f"{nthr} threads, n = {n}: scipy dsyevd %.3f / %.3f ms;" % best(lambda: la.dsyevd(A, compute_v=1, lower=1)),
                  "torch.linalg.eigh (MKL) %.3f / %.3f ms" % best(lambda: torch.linalg.eigh(At)))
