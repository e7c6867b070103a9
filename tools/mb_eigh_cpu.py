"""Host-side Rayleigh-Ritz cost: full eigh vs LAPACK range-selecting drivers (single thread)."""
import time, numpy as np, scipy.linalg as sl, torch
torch.set_num_threads(1)
try:
    from threadpoolctl import threadpool_limits
    threadpool_limits(1)
except Exception:
    pass
rng = np.random.default_rng(0)
for m, k in ((224, 72), (240, 80), (152, 48)):
    A = rng.standard_normal((m, m)); A = A @ A.T + np.diag(np.arange(m) * 10.0)
    At = torch.from_numpy(A)
    def tm(f, n=20):
        f(); t = time.time()
        for _ in range(n): f()
        return (time.time() - t) / n * 1e3
    print(m, k, "torch eigh %.2f ms" % tm(lambda: torch.linalg.eigh(At)),
          "scipy evd %.2f" % tm(lambda: sl.eigh(A, driver="evd")),
          "scipy evr subset %.2f" % tm(lambda: sl.eigh(A, subset_by_index=[0, k - 1], driver="evr")),
          "scipy evx subset %.2f" % tm(lambda: sl.eigh(A, subset_by_index=[0, k - 1], driver="evx")),
          "evr overwrite %.2f" % tm(lambda: sl.eigh(A.copy(), subset_by_index=[0, k - 1], driver="evr", overwrite_a=True, check_finite=False)))
