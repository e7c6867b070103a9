"""CPU stand-in for the HIP ``ModalOps`` backend (TEST INFRASTRUCTURE ONLY).

Implements the ``ops`` protocol of ``diffsound_amd/lobpcg/modal_solver.py`` with plain
SciPy / PyTorch-CPU so that (a) the host-side solver logic can be tested without a GPU and
(b) every HIP kernel has a same-signature oracle to be compared with on the GPU box.
Never imported by the product package.
"""
import numpy as np
import scipy.sparse as sp
import torch


def rigid_body_basis(verts):
    """Translations + infinitesimal rotations x cross e_c about the centroid, (3nv, 6) fp64."""
    v = np.asarray(verts, dtype=np.float64)
    c = v - v.mean(0)
    Y = np.zeros((3 * len(v), 6))
    for a in range(3):
        Y[a::3, a] = 1
    Y[0::3, 3], Y[1::3, 3] = -c[:, 1], c[:, 0]
    Y[1::3, 4], Y[2::3, 4] = -c[:, 2], c[:, 1]
    Y[2::3, 5], Y[0::3, 5] = -c[:, 0], c[:, 2]
    return Y


def corner_embedding(tets, nv):
    """(nv x nvc) node-level matrix of the P1-in-P2 embedding of an ord-2 mesh: a corner node copies its
    coarse value, a mid-edge node (local slots of reference mesh.py:139-154) averages its two end points."""
    tn = np.asarray(tets)
    corners = np.unique(tn[:, [0, 2, 4, 9]])
    cid = -np.ones(nv, dtype=np.int64)
    cid[corners] = np.arange(len(corners))
    pa, pb = cid.copy(), cid.copy()
    for slot, (p, q) in {1: (0, 2), 3: (2, 4), 5: (4, 0), 6: (0, 9), 7: (2, 9), 8: (4, 9)}.items():
        pa[tn[:, slot]] = cid[tn[:, p]]
        pb[tn[:, slot]] = cid[tn[:, q]]
    assert (pa >= 0).all() and (pb >= 0).all()
    rows = np.repeat(np.arange(nv), 2)
    return sp.csr_matrix((np.full(2 * nv, 0.5), (rows, np.stack([pa, pb], 1).ravel())), shape=(nv, len(corners)))


class _CpuLevel:
    """The K-only part of the protocol on one level (what the preconditioner touches)."""

    def __init__(self, K, dtype):
        self.n = K.shape[0]
        self.device = torch.device("cpu")
        self.dtype = dtype
        self.npdt = np.float32 if dtype == torch.float32 else np.float64
        self.Kd = K.astype(self.npdt).tocsr()
        nb = self.n // 3
        Kb = K.tobsr((3, 3))
        rows = np.repeat(np.arange(nb), np.diff(Kb.indptr))
        diag = np.zeros((nb, 3, 3))
        m = rows == Kb.indices
        diag[rows[m]] = Kb.data[m]
        self.Dinv = torch.from_numpy(np.linalg.inv(diag).astype(self.npdt))
        self.counts = dict(apply_K_cols=0, apply_M_cols=0, gram=0, mix=0)

    apply_K = lambda self, X, out: CpuModalOps.apply_K(self, X, out)
    _bj = lambda self, R: CpuModalOps._bj(self, R)
    cheb_init = lambda self, R, D, W, c: CpuModalOps.cheb_init(self, R, D, W, c)
    cheb_step = lambda self, AD, R, D, W, c1, c2: CpuModalOps.cheb_step(self, AD, R, D, W, c1, c2)
    cheb_spmm = lambda self, Wk, Wprev, R0, c1, c2, first: CpuModalOps.cheb_spmm(self, Wk, Wprev, R0, c1, c2, first)


class CpuModalOps:
    def __init__(self, Kl, Km, M3, verts, lam, mu, dtype=torch.float32, tets=None):
        """tets: the ord-2 connectivity; given, the corner-node level of the two-level preconditioner is
        built as the Galerkin product P^T K P."""
        self.Kl, self.Km, self.M = Kl.tocsr(), Km.tocsr(), M3.tocsr()
        self.lame = (float(lam), float(mu))
        self.n = Kl.shape[0]
        self.device = torch.device("cpu")
        self.dtype = dtype
        self.npdt = np.float32 if dtype == torch.float32 else np.float64
        K = (lam * self.Kl + mu * self.Km).tocsr()
        self.K64 = K
        self.Kd = K.astype(self.npdt)
        self.Md = self.M.astype(self.npdt)
        nb = self.n // 3
        Kb = K.tobsr((3, 3))
        rows = np.repeat(np.arange(nb), np.diff(Kb.indptr))
        diag = np.zeros((nb, 3, 3))
        m = rows == Kb.indices
        diag[rows[m]] = Kb.data[m]
        self.Dinv = torch.from_numpy(np.linalg.inv(diag).astype(self.npdt))
        Y = rigid_body_basis(verts)
        G = Y.T @ (self.M @ Y)
        L = np.linalg.cholesky(G)
        Y = np.linalg.solve(L, Y.T).T
        self.rigid = torch.from_numpy(Y.astype(self.npdt))
        self._rigid64 = Y
        self.counts = dict(apply_K_cols=0, apply_M_cols=0, gram=0, mix=0)
        self.coarse = None
        if tets is not None:
            P = sp.kron(corner_embedding(tets, nb), sp.identity(3), format="csr")
            self.P = P.astype(self.npdt)
            self.PT = self.P.T.tocsr()
            self.coarse = _CpuLevel((P.T @ K @ P).tocsr(), dtype)

    # -- two-level pieces ------------------------------------------------------------------
    def spmm_residual(self, X, R0, Y):
        Y.copy_(R0 - torch.from_numpy(self.Kd @ X.numpy()))
        self.counts["apply_K_cols"] += X.shape[1]

    def restrict(self, Rf, Rc):
        Rc.copy_(torch.from_numpy(self.PT @ Rf.numpy()))

    def prolong_add(self, Ec, Wf):
        Wf.add_(torch.from_numpy(self.P @ Ec.numpy()))

    # -- sparse products -----------------------------------------------------------------
    def apply_K(self, X, out):
        out.copy_(torch.from_numpy(self.Kd @ X.numpy()))
        self.counts["apply_K_cols"] += X.shape[1]

    def apply_M(self, X, out):
        out.copy_(torch.from_numpy(self.Md @ X.numpy()))
        self.counts["apply_M_cols"] += X.shape[1]

    # -- the fused forms of the HIP operators (same signatures; round 5: with them the solver's Python loop takes the
    #    Rayleigh-Ritz step on the raw basis on the CPU as well) ---------------------------------------------------------
    fused = False  # set True to offer residual_fused / apply_KM (the stand-in of ds_union_residual / ds_spmm_union_km)

    def residual_fused_ok(self, X, R):
        return self.fused

    def residual_fused(self, X, lam, R):
        Xc = np.ascontiguousarray(X.numpy())
        r = torch.from_numpy(self.Kd @ Xc) - torch.from_numpy(self.Md @ Xc) * lam.to(self.dtype)[None, :]
        R.copy_(r)
        self.counts["apply_K_cols"] += X.shape[1]
        self.counts["apply_M_cols"] += X.shape[1]
        return (r.double() ** 2).sum(0), (X.double() ** 2).sum(0)

    def apply_KM_ok(self, X, KX, MX):
        return self.fused

    def apply_KM(self, X, KX, MX):
        Xc = np.ascontiguousarray(X.numpy())
        KX.copy_(torch.from_numpy(self.Kd @ Xc))
        MX.copy_(torch.from_numpy(self.Md @ Xc))
        self.counts["apply_K_cols"] += X.shape[1]
        self.counts["apply_M_cols"] += X.shape[1]

    # -- tall-skinny dense ---------------------------------------------------------------
    def gram(self, A, B, symmetric=False, exact=False):
        self.counts["gram"] += 1
        return A.double().transpose(0, 1) @ B.double()

    def gram_blocks(self, A_blocks, B_blocks, symmetric=False):
        return torch.cat(list(A_blocks), 1).double().transpose(0, 1) @ torch.cat(list(B_blocks), 1).double()

    def mix(self, A, C, out, alpha=1.0, beta=0.0):
        self.counts["mix"] += 1
        r = (A @ C.to(self.dtype)) * alpha
        if beta != 0.0:
            r = r + beta * out
        out.copy_(r)

    def mix_inplace(self, W, T):
        W.copy_(W @ T.to(self.dtype))

    def mix64(self, blocks, C, out=None, alpha=1.0, beta=0.0):
        """alpha * [blocks] C + beta * out in fp64; an entry (block, row) names the block's first row of C."""
        acc, row = None, 0
        for blk in blocks:
            if isinstance(blk, tuple):
                blk, row = blk
            t = blk @ C[row:row + blk.shape[1]]
            acc = t if acc is None else acc + t
            row += blk.shape[1]
        acc = alpha * acc
        if out is None:
            return acc
        out.copy_(acc + beta * out if beta != 0.0 else acc)
        return out

    # -- fused elementwise ---------------------------------------------------------------
    def residual(self, R, MX, X, lam, src=None):
        if src is not None:
            R.copy_(src)
        R.sub_(MX * lam.to(self.dtype)[None, :])
        return (R.double() ** 2).sum(0), (X.double() ** 2).sum(0)

    def _bj(self, R):
        n, c = R.shape
        return torch.einsum("nij,njc->nic", self.Dinv, R.reshape(-1, 3, c)).reshape(n, c)

    def cheb_init(self, R, D, W, c):
        D.copy_(self._bj(R) * c)
        W.copy_(D)

    def cheb_step(self, AD, R, D, W, c1, c2):
        R.sub_(AD)
        D.copy_(c1 * D + c2 * self._bj(R))
        W.add_(D)

    def cheb_spmm(self, Wk, Wprev, R0, c1, c2, first):
        KW = torch.from_numpy(self.Kd @ Wk.numpy())
        new = Wk + c2 * self._bj(R0 - KW)
        new = new + c1 * (Wk if first else Wk - Wprev)  # first: W_prev = 0 and is not read
        Wprev.copy_(new)
        self.counts["apply_K_cols"] += Wk.shape[1]

    # -- fp64 iterates (refinement phase) ---------------------------------------------------
    def apply_K64(self, X, out, terms=False):
        Xd = X.numpy()
        parts = [torch.from_numpy(self.Kl @ Xd), torch.from_numpy(self.Km @ Xd)]
        out.copy_(self.lame[0] * parts[0] + self.lame[1] * parts[1])
        return parts if terms else []

    def apply_M64(self, X, out):
        out.copy_(torch.from_numpy(self.M @ X.numpy()))

    def rigid64(self):
        Y = torch.zeros((self.n, 8), dtype=torch.float64)
        Y[:, :6] = torch.from_numpy(self._rigid64)
        return Y

    # -- fp64 polish ---------------------------------------------------------------------
    def polish_products(self, X):
        Xd = X.double().numpy()
        f = lambda A: torch.from_numpy(Xd.T @ (A @ Xd))
        return [f(self.Kl), f(self.Km)], list(self.lame), f(self.M)
