"""CPU prototype: two-level (P2 -> P1) V-cycle preconditioner vs plain Chebyshev block-Jacobi, iteration counts."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp, torch
from diffsound_amd import meshgen
from diffsound_amd.lobpcg.modal_solver import ModalSolver, SolverConfig
from oracle import fem
from oracle.ops_cpu import CpuModalOps

cells = int(os.environ.get("CELLS", 10)); k = int(os.environ.get("K", 32)); blk = int(os.environ.get("BLOCK", 40))
v, t = meshgen.kuhn_box(cells)
v, t = fem.to_high_order(torch.from_numpy(v), torch.from_numpy(t).long(), 2)
d = fem.OracleDeform(v, t, 2)
Kl, Km = fem.assemble_stiffness(d, 1.0, 0.0), fem.assemble_stiffness(d, 0.0, 1.0)
M3, _ = fem.assemble_mass(v, t, 2, 2700.0)
lam, mu = fem.lame(5e10, 0.25)
ops = CpuModalOps(Kl, Km, M3, v.numpy(), lam, mu, dtype=torch.float32)
K = (lam * Kl + mu * Km).tocsr()
n = K.shape[0]; nv = n // 3
tn = t.numpy()
corners = np.unique(tn[:, [0, 2, 4, 9]])
cid = -np.ones(nv, dtype=np.int64); cid[corners] = np.arange(len(corners))
pa = cid.copy(); pb = cid.copy()
for slot, (p, q) in {1: (0, 2), 3: (2, 4), 5: (4, 0), 6: (0, 9), 7: (2, 9), 8: (4, 9)}.items():
    pa[tn[:, slot]] = cid[tn[:, p]]; pb[tn[:, slot]] = cid[tn[:, q]]
assert (pa >= 0).all() and (pb >= 0).all()
rows = np.repeat(np.arange(nv), 2); cols = np.stack([pa, pb], 1).ravel()
Pn = sp.csr_matrix((np.full(2 * nv, 0.5), (rows, cols)), shape=(nv, len(corners)))
P = sp.kron(Pn, sp.identity(3)).tocsr()
Kc = (P.T @ K @ P).tocsr()
print("fine n", n, "coarse n", Kc.shape[0], "nnz", K.nnz, Kc.nnz, flush=True)

def block_jacobi_inv(A):
    nb = A.shape[0] // 3
    D = np.zeros((nb, 3, 3))
    A = A.tobsr((3, 3))
    for i in range(nb):
        for kk in range(A.indptr[i], A.indptr[i + 1]):
            if A.indices[kk] == i: D[i] = A.data[kk]
    Di = np.linalg.inv(D)
    return sp.block_diag([Di[i] for i in range(nb)], format="csr") if nb < 20000 else sp.bsr_matrix((Di, np.arange(nb), np.arange(nb + 1)), shape=A.shape).tocsr()

def lmax_of(A, T):
    x = np.random.default_rng(0).standard_normal((A.shape[0], 4))
    for _ in range(40):
        x = T @ (A @ x); nr = np.linalg.norm(x, axis=0); x /= nr
    return 1.2 * nr.max()

class Cheb:
    def __init__(self, A, ratio, degree):
        self.A, self.T = A, block_jacobi_inv(A)
        self.lmax = min(lmax_of(A, self.T), 10.0); self.lmin = self.lmax / ratio; self.degree = degree
        self.count = 0
    def iterate(self, R, W0=None):
        theta = 0.5 * (self.lmax + self.lmin); delta = 0.5 * (self.lmax - self.lmin)
        A, T = self.A, self.T
        sigma1 = theta / delta; rho = 1 / sigma1
        if W0 is None:
            Wp = np.zeros_like(R); W = T @ R / theta; start = 1
        else:
            self.count += 1
            Wp = W0; W = W0 + T @ (R - A @ W0) / theta; start = 1
        for kk in range(start, self.degree):
            rho_new = 1 / (2 * sigma1 - rho)
            self.count += 1
            Wn = W + rho_new * rho * (W - Wp) + 2 * rho_new / delta * (T @ (R - A @ W))
            Wp, W = W, Wn; rho = rho_new
        return W

mode = os.environ.get("MODE", "two")
deg = int(os.environ.get("DEG", 48)); ratio = float(os.environ.get("RATIO", 800))
nu = int(os.environ.get("NU", 3)); alpha = float(os.environ.get("ALPHA", 20)); dc = int(os.environ.get("DC", 48)); rc = float(os.environ.get("RC", 800))
K32 = K.astype(np.float32); Kc32 = Kc.astype(np.float32); P32 = P.astype(np.float32)
if mode == "cheb":
    ch = Cheb(K32, ratio, deg)
    def precond(R, W):
        W.copy_(torch.from_numpy(ch.iterate(R.numpy()).astype(np.float32)))
    fine_count = lambda: ch.count; coarse_count = lambda: 0
else:
    sm = Cheb(K32, alpha, nu + 1); co = Cheb(Kc32, rc, dc)
    print("lmax fine", sm.lmax, "coarse", co.lmax)
    def precond(R, W):
        R_ = R.numpy()
        W1 = sm.iterate(R_)
        sm.count += 1
        W2 = W1 + P32 @ co.iterate(P32.T @ (R_ - K32 @ W1))
        W3 = sm.iterate(R_, W0=W2)
        W.copy_(torch.from_numpy(W3.astype(np.float32)))
    fine_count = lambda: sm.count; coarse_count = lambda: co.count
t0 = time.time()
res = ModalSolver(ops, SolverConfig(block=blk, lmax_cap=10.0), precond=precond).solve(k)
print(mode, "iterations", res.iterations, "fine spmm", fine_count(), "coarse spmm", coarse_count(), "time %.1f" % (time.time() - t0))
print("hist", [f"{h[1]:.1e}" for h in res.history] if hasattr(res, "history") else "")
