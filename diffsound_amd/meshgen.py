"""Synthetic tet meshes for benchmarks and tests (host, NumPy).

Axis-aligned box of nx*ny*nz cells, each split into 6 tets by the Kuhn/Freudenthal subdivision
around the main diagonal (conforming).  Orientation is irrelevant because the path integrates with
|det| (reference src/diffelastic/deform.py:143-144).  Interior nodes are jittered so that no exact
eigenvalue degeneracies survive.  This is the generator SURVEY.md 8(d) specifies for the named
benchmark sizes: C2 = 12^3 cells (10 368 tets), C3 = 26^3 (105 456 tets), C5 = 55^3 (998 250 tets).
"""
import numpy as np

_KUHN = ((1, 2), (2, 3), (3, 7), (7, 4), (4, 5), (5, 1))


def kuhn_box(nx, ny=None, nz=None, box=(0.10, 0.08, 0.06), jitter=0.15, seed=1234):
    """Return (verts (nv,3) float32, tets (T,4) int32)."""
    ny = nx if ny is None else ny
    nz = nx if nz is None else nz
    xs = np.linspace(0, box[0], nx + 1)
    ys = np.linspace(0, box[1], ny + 1)
    zs = np.linspace(0, box[2], nz + 1)
    X, Y, Z = np.meshgrid(xs, ys, zs, indexing="ij")
    verts = np.stack([X, Y, Z], -1).reshape(-1, 3)
    if jitter > 0:
        rng = np.random.default_rng(seed)
        h = np.array([box[0] / nx, box[1] / ny, box[2] / nz])
        inner = np.ones((nx + 1, ny + 1, nz + 1), bool)
        inner[[0, -1]] = False
        inner[:, [0, -1]] = False
        inner[:, :, [0, -1]] = False
        verts = verts + rng.uniform(-jitter, jitter, size=verts.shape) * h * inner.reshape(-1, 1)
    idx = np.arange((nx + 1) * (ny + 1) * (nz + 1)).reshape(nx + 1, ny + 1, nz + 1)
    i, j, k = np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij")
    c = [idx[i, j, k], idx[i + 1, j, k], idx[i + 1, j + 1, k], idx[i, j + 1, k],
         idx[i, j, k + 1], idx[i + 1, j, k + 1], idx[i + 1, j + 1, k + 1], idx[i, j + 1, k + 1]]
    # cell-major ordering: the 6 tets of a cell are consecutive
    tets = np.stack([np.stack([c[0], c[a], c[b], c[6]], -1) for a, b in _KUHN], axis=3)
    tets = tets.reshape(-1, 6, 4).reshape(-1, 4)
    return verts.astype(np.float32), tets.astype(np.int32)


def plate(n=40, thickness_cells=1, size=(0.2, 0.2, 0.005)):
    """The 40x40x1-cell plate of SURVEY.md 8(d) C1 (9600 tets)."""
    return kuhn_box(n, n, thickness_cells, box=size)
