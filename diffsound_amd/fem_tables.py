"""Reference-element tables for P1 / P2 tetrahedra (host side, NumPy).

The HIP assembly kernel integrates in closed form over the *same* quadrature the reference uses
(collapsed Gauss-Legendre with (order+2)^3 points, built in fp32: reference
src/diffelastic/gauss.py:4-38), so the tables are derived from that rule and carry its fp32
rounding; everything downstream is fp64.

  dtab[a,k,b,l] = sum_g w_g dN_a/dL_k dN_b/dL_l      (stiffness, reference deform.py:47-67 +
                                                      diff_model.py:207-213 collapsed over g)
  mtab[a,b]     = sum_g w_g N_a N_b                  (mass, reference mass_matrix.py:9-23)

Local node order of the P2 element (reference src/diffelastic/mesh.py:139-154,
shape_func.py:14-24): 0 v1, 1 mid(v1,v2), 2 v2, 3 mid(v2,v3), 4 v3, 5 mid(v1,v3), 6 mid(v1,v4),
7 mid(v2,v4), 8 mid(v3,v4), 9 v4.
"""
import functools

import numpy as np
from numpy.polynomial import legendre as _leg

NODES_PER_TET = {1: 4, 2: 10}
CORNER_SLOTS = {1: (0, 1, 2, 3), 2: (0, 2, 4, 9)}
# P2 mid-edge slot -> the two barycentric indices it joins
_MID = {1: (0, 1), 3: (1, 2), 5: (2, 0), 6: (0, 3), 7: (1, 3), 8: (2, 3)}
_CORNER = {0: 0, 2: 1, 4: 2, 9: 3}


@functools.lru_cache(maxsize=None)
def gauss_rule(order):
    """(points (G,4) barycentric fp32, weights (G,) fp32) of the (order+2)^3 collapsed rule."""
    npts = order + 2
    coef = np.zeros(npts + 1, dtype=np.float32)
    coef[-1] = 1
    x = _leg.legroots(coef)
    dp = _leg.Legendre(coef).deriv()(x)
    w1 = 2 / ((1 - x ** 2) * dp ** 2)
    r = (x + 1) / 2
    pts = np.zeros((npts ** 3, 4), dtype=np.float32)
    wts = np.zeros(npts ** 3, dtype=np.float32)
    for i in range(npts):
        lw = np.float32(r[i])
        for j in range(npts):
            lz = np.float32(r[j] * (1 - lw))
            for k in range(npts):
                ly = np.float32(r[k] * (1 - lw - lz))
                q = (i * npts + j) * npts + k
                pts[q] = (np.float32(1 - lw - lz - ly), ly, lz, lw)
                wts[q] = w1[i] * w1[j] * w1[k] * (1 - lw) * (1 - lw - lz) / 8
    return pts, wts


def shape_values(L, order):
    """N (G, N) in the precision of L."""
    if order == 1:
        return L.copy()
    N = np.zeros((L.shape[0], 10), dtype=L.dtype)
    for slot, k in _CORNER.items():
        N[:, slot] = L[:, k] * (2 * L[:, k] - 1)
    for slot, (p, q) in _MID.items():
        N[:, slot] = 4 * L[:, p] * L[:, q]
    return N


def shape_gradients(L, order):
    """dN/dL (G, N, 4) in the precision of L."""
    G = L.shape[0]
    if order == 1:
        return np.broadcast_to(np.eye(4, dtype=L.dtype), (G, 4, 4)).copy()
    g = np.zeros((G, 10, 4), dtype=L.dtype)
    for slot, k in _CORNER.items():
        g[:, slot, k] = 4 * L[:, k] - 1
    for slot, (p, q) in _MID.items():
        g[:, slot, p] = 4 * L[:, q]
        g[:, slot, q] = 4 * L[:, p]
    return g


@functools.lru_cache(maxsize=None)
def stiffness_table(order):
    """dtab (N,4,N,4) fp64."""
    pts, w = gauss_rule(order)
    dN = shape_gradients(pts, order).astype(np.float64)  # fp32 values, as the reference holds them
    return np.ascontiguousarray(np.einsum("g,gak,gbl->akbl", w.astype(np.float64), dN, dN))


@functools.lru_cache(maxsize=None)
def mass_table_f32(order):
    """M^ (N,N) accumulated in fp32 exactly like the reference (mass_matrix.py:17-22: fp32 products,
    fp32 sum over Gauss points)."""
    import torch  # fp32 summation order of torch.sum is part of the constant

    pts, w = gauss_rule(order)
    N = torch.from_numpy(shape_values(pts, order))
    wt = torch.from_numpy(w)
    nn = N.shape[1]
    M = torch.zeros(nn, nn, dtype=torch.float32)
    for a in range(nn):
        for b in range(nn):
            M[a, b] = torch.sum(N[:, a] * N[:, b] * wt)
    return M.numpy()


def mass_table(order, density):
    """mtab (N,N) fp64 = fp32(M^ * density): the reference multiplies the fp32 table by the density
    in fp32 before the fp64 |det| comes in (diff_model.py:301-303)."""
    return (mass_table_f32(order) * np.float32(density)).astype(np.float64)


@functools.lru_cache(maxsize=None)
def minimal_gradient_rule(order):
    """(dN/dL at the points (ng, N, 4) fp64, weights (ng,) fp64) of the smallest rule that integrates
    grad N_a . grad N_b exactly: 1 point for P1 (constant gradients), the 4-point degree-2 rule for P2.
    Weights are scaled to the total of the reference's fp32 rule so that the element energy matches the
    assembled K to rounding."""
    total = float(gauss_rule(order)[1].astype(np.float64).sum())
    if order == 1:
        pts = np.full((1, 4), 0.25)
        w = np.array([total])
    else:
        a, b = (5 + 3 * np.sqrt(5)) / 20, (5 - np.sqrt(5)) / 20
        pts = np.full((4, 4), b)
        pts[np.arange(4), np.arange(4)] = a
        w = np.full(4, total / 4)
    return np.ascontiguousarray(shape_gradients(pts, order)), w
