"""Oracle restatement of the reference FEM building blocks (TEST INFRASTRUCTURE ONLY).

Follows, function by function:
  gauss rule            /root/reference/src/diffelastic/gauss.py:4-38
  shape functions       /root/reference/src/diffelastic/shape_func.py:3-108   (ord 1, 2 only)
  element mass table    /root/reference/src/diffelastic/mass_matrix.py:9-31
  TetMesh               /root/reference/src/diffelastic/mesh.py:58-179
  Deform                /root/reference/src/diffelastic/deform.py:35-166
  K / M assembly        /root/reference/src/diffelastic/diff_model.py:184-312
  stiff_func            /root/reference/src/diffelastic/diff_model.py:314-328

Everything runs on the CPU in NumPy / PyTorch.  Quirks of the reference that change
numbers are reproduced on purpose (fp32 quadrature constants, fp32 shape-function
gradients, fp64 |det| for M but fp32 |det| for K, |det| so inverted tets add positively).
"""
import numpy as np
import scipy.sparse as sp
import torch
from numpy.polynomial.legendre import Legendre, legroots

NODES_PER_TET = {1: 4, 2: 10}
CORNERS = {1: (0, 1, 2, 3), 2: (0, 2, 4, 9)}  # mesh.py:75-84
# ord-2 edge midpoints: local slot -> (corner a, corner b) in ord-1 numbering, mesh.py:125-154
EDGE_SLOTS = {1: (0, 1), 3: (1, 2), 5: (0, 2), 6: (0, 3), 7: (1, 3), 8: (2, 3)}


def gauss_points_weights(npts):
    """Collapsed Gauss-Legendre rule on the unit tet with npts^3 points, fp32 (gauss.py:4-38)."""
    c = np.zeros(npts + 1, dtype=np.float32)
    c[-1] = 1
    roots = legroots(c)
    dP = Legendre(c).deriv()(roots)
    w1 = 2 / ((1 - roots ** 2) * dP ** 2)
    r = (roots + 1) / 2
    pts = np.zeros((npts ** 3, 4), dtype=np.float32)
    wts = np.zeros(npts ** 3, dtype=np.float32)
    q = 0
    for i in range(npts):
        for j in range(npts):
            for k in range(npts):
                w = np.float32(r[i])
                z = np.float32(r[j] * (1 - w))
                y = np.float32(r[k] * (1 - w - z))
                x = np.float32(1 - w - z - y)
                pts[q] = (x, y, z, w)
                wts[q] = w1[i] * w1[j] * w1[k] * (1 - w) * (1 - w - z) / 8
                q += 1
    return pts, wts


def shape_functions(L, order):
    """N_a(L) for P1/P2, L = (n,4) barycentrics (shape_func.py:3-25)."""
    L = torch.as_tensor(L)
    if order == 1:
        return L
    L1, L2, L3, L4 = L.unbind(1)
    cols = [L1 * (2 * L1 - 1), 4 * L1 * L2, L2 * (2 * L2 - 1), 4 * L2 * L3, L3 * (2 * L3 - 1),
            4 * L3 * L1, 4 * L1 * L4, 4 * L2 * L4, 4 * L3 * L4, L4 * (2 * L4 - 1)]
    return torch.stack(cols, 1)


def shape_function_grads(L, order):
    """dN_a/dL_k, shape (n, N, 4) (shape_func.py:51-84)."""
    L = torch.as_tensor(L)
    n = L.shape[0]
    if order == 1:
        return torch.eye(4, dtype=L.dtype).expand(n, 4, 4).clone()
    g = torch.zeros(n, 10, 4, dtype=L.dtype)
    L1, L2, L3, L4 = L.unbind(1)
    one = torch.ones_like(L1)
    corner = {0: (0, L1), 2: (1, L2), 4: (2, L3), 9: (3, L4)}
    for a, (k, Lk) in corner.items():
        g[:, a, k] = 4 * Lk - one
    mids = {1: (0, 1), 3: (1, 2), 5: (2, 0), 6: (0, 3), 7: (1, 3), 8: (2, 3)}
    Ls = (L1, L2, L3, L4)
    for a, (p, q) in mids.items():
        g[:, a, p] = 4 * Ls[q]
        g[:, a, q] = 4 * Ls[p]
    return g


def element_mass_table(order):
    """Reference-element consistent mass  M^_ab = sum_g w_g N_a N_b, fp32 (mass_matrix.py:9-23)."""
    pts, w = gauss_points_weights(order + 2)
    N = shape_functions(torch.from_numpy(pts), order)
    w = torch.from_numpy(w)
    nn = N.shape[1]
    M = torch.zeros(nn, nn, dtype=torch.float32)
    for a in range(nn):
        for b in range(nn):
            M[a, b] = torch.sum(N[:, a] * N[:, b] * w)
    return M


def element_mass_flat(order):
    """(M^ (x) I3) flattened row-major, length (3N)^2 (mass_matrix.py:25-31)."""
    M = element_mass_table(order)
    nn = M.shape[0]
    full = M[:, :, None, None] * torch.eye(3)
    return full.transpose(1, 2).reshape(-1)


def to_high_order(verts, tets, order):
    """ord-1 -> ord-2 lifting with duplicate merge by torch.unique (mesh.py:101-179).

    Returns (vertices (nv2,3), tets (T,10)); nodes end up renumbered in lexicographic
    (x,y,z) order; representative coordinates = those of the lowest original index."""
    verts = torch.as_tensor(verts)
    tets = torch.as_tensor(tets).long()
    if order == 1:
        return verts, tets
    T, nv = tets.shape[0], verts.shape[0]
    vf = verts[tets]
    mids = []
    new_tets = torch.zeros(T, 10, dtype=torch.long)
    for slot, corner in zip((0, 2, 4, 9), range(4)):
        new_tets[:, slot] = tets[:, corner]
    for e, slot in enumerate((1, 3, 5, 6, 7, 8)):
        a, b = EDGE_SLOTS[slot]
        mids.append((vf[:, a] + vf[:, b]) / 2)
        new_tets[:, slot] = torch.arange(nv + e * T, nv + (e + 1) * T)
    allv = torch.cat([verts] + mids, 0)
    uniq, inv = torch.unique(allv, dim=0, return_inverse=True)
    first = torch.full((uniq.shape[0],), allv.shape[0], dtype=torch.long)
    first.scatter_reduce_(0, inv, torch.arange(allv.shape[0]), reduce="amin", include_self=True)
    return allv[first], inv[new_tets]


def transform_matrix(verts, tets, order):
    """A_e = [v1-v4, v2-v4, v3-v4] as columns, fp32 (mesh.py:58-99)."""
    c = CORNERS[order]
    v = [verts[tets[:, i]] for i in c]
    A = torch.stack([v[0] - v[3], v[1] - v[3], v[2] - v[3]], dim=2)
    return A.float()


class OracleDeform:
    """Gauss-point tables of one mesh (deform.py:8-147)."""

    def __init__(self, verts, tets, order):
        self.verts = torch.as_tensor(verts).float()
        self.tets = torch.as_tensor(tets).long()
        self.order = order
        pts, w = gauss_points_weights(order + 2)
        self.gp = torch.from_numpy(pts)
        self.gw = torch.from_numpy(w)
        self.G = self.gp.shape[0]
        self.N = self.tets.shape[1]
        self.T = self.tets.shape[0]
        self.A = transform_matrix(self.verts, self.tets, order)

    def shape_func_deriv(self):
        """B[t*G+g] = (dN_dL[g] @ dL_dx) @ A[t]^-1, (T*G, N, 3) fp32 (deform.py:35-68)."""
        Ainv = torch.inverse(self.A)
        dL_dx = torch.tensor([[1, 0, 0], [0, 1, 0], [0, 0, 1], [-1, -1, -1]], dtype=torch.float32)
        dN = shape_function_grads(self.gp, self.order) @ dL_dx  # (G, N, 3)
        B = dN[None] @ Ainv[:, None]  # (T, G, N, 3)
        return B.reshape(self.T * self.G, self.N, 3)

    def integration_weights(self):
        """w[t*G+g] = gw[g] * |det A[t]|, fp32 (deform.py:136-147)."""
        return (self.gw[None, :] * torch.abs(torch.det(self.A))[:, None]).reshape(-1)

    def dof_index(self):
        """global DOF ids 3*node+c per element, (T, 3N) (deform.py:113-125)."""
        return (self.tets[:, :, None] * 3 + torch.arange(3)[None, None, :]).reshape(self.T, 3 * self.N)


def lame(E, nu):
    """(lambda_L, mu) of isotropic linear elasticity (diff_model.py:35-39)."""
    return E * nu / ((1 + nu) * (1 - 2 * nu)), E / (2 * (1 + nu))


def piola_jacobian(lam, mu):
    """Constant 9x9 d vec(P)/d vec(F) of P = mu(F+F^T) + lam tr(F) I (diff_model.py:34-48)."""
    J = np.zeros((9, 9))
    for i in range(3):
        for j in range(3):
            r = 3 * i + j
            J[r, 3 * i + j] += mu
            J[r, 3 * j + i] += mu
            if i == j:
                for k in range(3):
                    J[r, 4 * k] += lam
    return J


def assemble_stiffness_faithful(deform, lam, mu, batch=20000):
    """Per-Gauss-point A^T B A accumulation, batch by batch (diff_model.py:184-220).  Returns scipy CSR f64."""
    N, T, G = deform.N, deform.T, deform.G
    SFDT = deform.shape_func_deriv().transpose(1, 2)  # (T*G, 3, N)
    w = deform.integration_weights().double()
    B9 = torch.from_numpy(piola_jacobian(lam, mu)).double()[None]
    dof = deform.dof_index().repeat_interleave(G, dim=0)  # (T*G, 3N)
    nq = T * G
    n = 3 * deform.verts.shape[0]
    bs = min(batch, nq)
    edges = torch.linspace(0, nq, nq // bs + 1).long()
    K = sp.csr_matrix((n, n), dtype=np.float64)
    for s, e in zip(edges[:-1].tolist(), edges[1:].tolist()):
        A = torch.zeros(e - s, 9, 3 * N, dtype=torch.float64)
        A[:, 0:3, 0::3] = SFDT[s:e]
        A[:, 3:6, 1::3] = SFDT[s:e]
        A[:, 6:9, 2::3] = SFDT[s:e]
        vals = (A.transpose(1, 2) @ B9 @ A) * w[s:e, None, None]
        rows = dof[s:e, :, None].expand(-1, -1, 3 * N).reshape(-1)
        cols = dof[s:e, None, :].expand(-1, 3 * N, -1).reshape(-1)
        K = K + sp.coo_matrix((vals.reshape(-1).numpy(), (rows.numpy(), cols.numpy())), shape=(n, n)).tocsr()
    return K


def element_stiffness(deform, lam, mu, chunk=2048):
    """Per-element Ke = sum_g w A_g^T B A_g (same numbers as the faithful path up to fp64
    summation order), returned as (T, 3N, 3N) f64 chunks via generator."""
    N, T, G = deform.N, deform.T, deform.G
    SFDT = deform.shape_func_deriv().transpose(1, 2).reshape(T, G, 3, N)
    w = deform.integration_weights().double().reshape(T, G)
    B9 = torch.from_numpy(piola_jacobian(lam, mu)).double()
    for s in range(0, T, chunk):
        e = min(T, s + chunk)
        A = torch.zeros(e - s, G, 9, 3 * N, dtype=torch.float64)
        A[:, :, 0:3, 0::3] = SFDT[s:e]
        A[:, :, 3:6, 1::3] = SFDT[s:e]
        A[:, :, 6:9, 2::3] = SFDT[s:e]
        Ke = torch.einsum("tgri,rs,tgsj,tg->tij", A, B9, A, w[s:e])
        yield s, e, Ke


def assemble_stiffness(deform, lam, mu):
    """K as scipy CSR f64, per-element pre-summed (memory-safe variant of diff_model.py:184-220)."""
    n = 3 * deform.verts.shape[0]
    dof = deform.dof_index().numpy()
    K = sp.csr_matrix((n, n), dtype=np.float64)
    m = dof.shape[1]
    for s, e, Ke in element_stiffness(deform, lam, mu):
        rows = np.repeat(dof[s:e], m, axis=1).reshape(-1)
        cols = np.tile(dof[s:e], (1, m)).reshape(-1)
        K = K + sp.coo_matrix((Ke.reshape(-1).numpy(), (rows, cols)), shape=(n, n)).tocsr()
    return K


def tet_abs_det_f64(verts, tets, order):
    """|det| by the explicit triple product in fp64 from fp32 coordinates (diff_model.py:233-289)."""
    c = CORNERS[order]
    v = torch.as_tensor(verts).double()
    p = [v[tets[:, i]] for i in c]
    d1, d2, d3 = p[1] - p[0], p[2] - p[0], p[3] - p[0]
    V = (d1[:, 0] * (d2[:, 1] * d3[:, 2] - d3[:, 1] * d2[:, 2])
         + d1[:, 1] * (d2[:, 2] * d3[:, 0] - d3[:, 2] * d2[:, 0])
         + d1[:, 2] * (d2[:, 0] * d3[:, 1] - d3[:, 0] * d2[:, 1]))
    return torch.abs(V)


def assemble_mass(verts, tets, order, density):
    """Consistent mass Me = rho |det| (M^ (x) I3) (diff_model.py:222-312).  scipy CSR f64, structural zeros kept
    out (the reference strips them with eliminate_zeros() before ARPACK, diff_model.py:355)."""
    tets = torch.as_tensor(tets).long()
    N = NODES_PER_TET[order]
    # element_mm (fp32 tensor) * density is evaluated in fp32 before the fp64 |det| multiplies it
    # (diff_model.py:301-303: f32 0-dim tensor * python float -> f32)
    mhat = (element_mass_table(order) * density).double().numpy()
    J = tet_abs_det_f64(verts, tets, order).numpy()
    nv = verts.shape[0]
    t = tets.numpy()
    rows = np.repeat(t, N, axis=1).reshape(-1)
    cols = np.tile(t, (1, N)).reshape(-1)
    vals = (J[:, None, None] * mhat[None]).reshape(-1)
    Ms = sp.coo_matrix((vals, (rows, cols)), shape=(nv, nv)).tocsr()
    return sp.kron(Ms, sp.identity(3), format="csr"), Ms


def stiff_func(deform, lam, mu, U):
    """Matrix-free K(theta) U: gradient_batch -> stress -> stress_to_force_batch, fp32
    (diff_model.py:314-328; deform.py:70-87,149-166).  U: (n, m) float32 torch; lam, mu may be autograd scalars."""
    x = U.transpose(0, 1).reshape(U.shape[1], -1, 3)
    B = deform.shape_func_deriv()  # (TG, N, 3)
    G = deform.G
    u = x[:, deform.tets].transpose(2, 3)  # (m, T, 3, N)
    u = u.unsqueeze(2).repeat(1, 1, G, 1, 1).reshape(x.shape[0], -1, 3, deform.N)
    F = u @ B
    tr = F.diagonal(dim1=-2, dim2=-1).sum(-1)
    P = mu * (F + F.transpose(-1, -2)) + lam * tr[..., None, None] * torch.eye(3)
    force = (P @ B.transpose(1, 2)) * deform.integration_weights()[None, :, None, None]
    force = force.transpose(2, 3).reshape(x.shape[0], -1)
    idx = deform.dof_index().repeat_interleave(G, dim=0).reshape(-1)
    out = torch.zeros(x.shape[0], 3 * deform.verts.shape[0], dtype=force.dtype)
    out.index_add_(1, idx, force)
    return out.transpose(0, 1)
