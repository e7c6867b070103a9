"""Experiment (round 6, CPU only, no product code): would a GROUP-block Jacobi (the inverse of the 48 x 48 diagonal block of every
16-node group of the matrix-core kernels' tiles) let the corner-node level's Chebyshev polynomial drop terms against the node-block
(3 x 3) Jacobi it uses now?  P1 elasticity on a jittered Kuhn box, nodes in Morton order, groups of 16 consecutive nodes;
scipy's LOBPCG on the pencil (K, M) with B = p(T K) T as preconditioner, block 80, rigid modes as constraints: iterations to a
residual reduction, per (T, degree, ratio).      python tools/experiments/group_block_jacobi_cpu.py [cells=16]"""
import sys, time
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))))
from diffsound_amd import meshgen

cells = int(sys.argv[1]) if len(sys.argv) > 1 else 16
v, t = meshgen.kuhn_box(cells)
v = v.astype(np.float64)
nv = len(v)


def morton(v):
    q = ((v - v.min(0)) / (v.max(0) - v.min(0) + 1e-30) * 1023).astype(np.uint64)
    def spread(x):
        x = (x | (x << 16)) & 0x030000FF
        x = (x | (x << 8)) & 0x0300F00F
        x = (x | (x << 4)) & 0x030C30C3
        x = (x | (x << 2)) & 0x09249249
        return x
    return spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)


perm = np.argsort(morton(v), kind="stable")
inv = np.empty(nv, np.int64); inv[perm] = np.arange(nv)
v = v[perm]; t = inv[t]


def assemble(E, nu, rho=2700.0):
    lam = E * nu / ((1 + nu) * (1 - 2 * nu)); mu = E / (2 * (1 + nu))
    X = v[t]                                   # (T, 4, 3)
    J = X[:, 1:] - X[:, :1]                    # rows: edges
    vol = np.abs(np.linalg.det(J)) / 6.0
    Ji = np.linalg.inv(J)                      # (T,3,3): columns = grad of L1..L3
    G = np.concatenate([-Ji.sum(2, keepdims=True), Ji], 2).transpose(0, 2, 1)  # (T, 4 nodes, 3) gradients
    # K_ab^{kl} = vol (lam g_a^k g_b^l + mu g_a^l g_b^k + mu delta_kl g_a . g_b)
    Kab = (lam * np.einsum("tak,tbl->tabkl", G, G) + mu * np.einsum("tal,tbk->tabkl", G, G)
           + mu * np.einsum("tam,tbm->tab", G, G)[..., None, None] * np.eye(3)) * vol[:, None, None, None, None]
    rows = (3 * t[:, :, None, None, None] + np.arange(3)[None, None, None, :, None]) + 0 * t[:, None, :, None, None]
    cols = (3 * t[:, None, :, None, None] + np.arange(3)[None, None, None, None, :]) + 0 * t[:, :, None, None, None]
    K = sp.coo_matrix((Kab.ravel(), (np.broadcast_to(rows, Kab.shape).ravel(), np.broadcast_to(cols, Kab.shape).ravel())), shape=(3 * nv, 3 * nv)).tocsr()
    Mab = (np.ones((4, 4)) + np.eye(4)) / 20.0
    Mv = rho * vol[:, None, None] * Mab[None]
    Ms = sp.coo_matrix((Mv.ravel(), (np.repeat(t, 4, 1).ravel(), np.tile(t, (1, 4)).ravel())), shape=(nv, nv)).tocsr()
    return K, sp.kron(Ms, sp.eye(3)).tocsr()


def block_inverse(K, size):
    n = K.shape[0]
    blocks = []
    for s in range(0, n, size):
        e = min(n, s + size)
        blocks.append(np.linalg.inv(K[s:e, s:e].toarray()))
    return sp.block_diag(blocks).tocsr()


def power_lmax(TK, it=40):
    x = np.random.default_rng(0).standard_normal((TK.shape[0], 8))
    for _ in range(it):
        y = TK @ x
        l = np.linalg.norm(y, axis=0).max() / np.linalg.norm(x, axis=0).max()
        x = y / np.linalg.norm(y, axis=0)
    return float(np.max(np.linalg.norm(TK @ x, axis=0)))


def chebyshev(K, T, degree, ratio, lmax):
    """W = p(T K) T R: degree terms of the Chebyshev iteration for T K on [lmax / ratio, lmax]."""
    a, b = lmax / ratio, lmax
    theta, delta = (b + a) / 2, (b - a) / 2

    def apply(R):
        R = R.reshape(K.shape[0], -1)
        z = T @ R
        x = z / theta
        d = x.copy()
        sigma = theta / delta
        rho_old = 1.0 / sigma
        for _ in range(degree - 1):
            r = z - T @ (K @ x)
            rho_new = 1.0 / (2 * sigma - rho_old)
            d = rho_new * rho_old * d + 2 * rho_new / delta * r
            x = x + d
            rho_old = rho_new
        return x
    return apply


def rigid(v):
    Y = np.zeros((3 * nv, 6))
    c = v - v.mean(0)
    for k in range(3):
        Y[k::3, k] = 1
    Y[1::3, 3], Y[2::3, 3] = -c[:, 2], c[:, 1]
    Y[0::3, 4], Y[2::3, 4] = c[:, 2], -c[:, 0]
    Y[0::3, 5], Y[1::3, 5] = -c[:, 1], c[:, 0]
    return Y


print(f"P1 Kuhn box {cells}^3: {nv} nodes, {len(t)} tets, n = {3 * nv}")
for E, nu in ((5e10, 0.25), (7.1e10, 0.40))[:int(sys.argv[2]) if len(sys.argv) > 2 else 2]:
    K, M = assemble(E, nu)
    Y = rigid(v)
    X0 = np.random.default_rng(1).standard_normal((3 * nv, 80))
    for name, size in ((("node 3x3", 3), ("group 48x48", 48)) if len(sys.argv) < 4 else ((f"group of {int(sys.argv[3])} nodes", 3 * int(sys.argv[3])),)):
        T = block_inverse(K, size)
        lmax = 1.05 * power_lmax(T @ K)
        for degree, ratio in ((22, 350.0), (16, 350.0), (16, 200.0), (14, 150.0), (12, 100.0), (10, 60.0)):
            B = spla.LinearOperator(K.shape, matvec=chebyshev(K, T, degree, ratio, lmax), matmat=chebyshev(K, T, degree, ratio, lmax), dtype=np.float64)
            t0 = time.time()
            w, X, hist = spla.lobpcg(K, X0.copy(), B=M, M=B, Y=Y, tol=None, maxiter=12, largest=False, retResidualNormsHistory=True, verbosityLevel=0)
            hist = np.array(hist)  # (its + 1, 80) residual norms
            rel = hist[:, :64].max(1) / hist[0, :64].max()
            its = next((i for i, r in enumerate(rel) if r < 1e-3), None)
            print(f"nu={nu:.2f} {name:12s} lmax {lmax:6.3f} degree {degree:2d} ratio {ratio:5.0f}: iterations to 1e-3 of the first residual (64 pairs): {its}, "
                  f"history {' '.join(f'{r:.1e}' for r in rel[:9])}  ({time.time() - t0:.0f} s)", flush=True)
