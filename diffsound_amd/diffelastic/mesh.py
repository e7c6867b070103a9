"""TetMesh - host-side mirror of the reference container (reference src/diffelastic/mesh.py:12-223).

Same constructor, attributes and method names, so callers (DiffSoundObj, the DMTet geometries)
keep working; the heavy consumers of the mesh (assembly, eigensolve) read ``vertices`` / ``tets``
straight into the HIP kernels.  Mesh ingestion here is plumbing on torch tensors; Gmsh 2.2 I/O is
implemented natively (the reference goes through ``meshio``, which is not a dependency here).
"""
import os
import struct

import numpy as np
import torch

from .. import fem_tables

_EDGES = ((0, 1), (1, 2), (0, 2), (0, 3), (1, 3), (2, 3))  # -> local slots 1,3,5,6,7,8
_EDGE_SLOTS = (1, 3, 5, 6, 7, 8)


def _default_device():
    if not torch.cuda.is_available():
        raise RuntimeError("diffsound_amd: no HIP device available (there is no CPU fallback)")
    return torch.device("cuda", torch.cuda.current_device())


def read_gmsh22(path):
    """Gmsh 2.2 reader (binary with 8-byte reals, or ASCII).  Returns (points f64 (nv,3), tets i64 (T,4|10))."""
    data = open(path, "rb").read()
    head = data.index(b"$MeshFormat\n") + len(b"$MeshFormat\n")
    ver, ftype, dsize = data[head:data.index(b"\n", head)].split()
    binary = int(ftype) == 1
    if not ver.startswith(b"2"):
        raise ValueError(f"{path}: only Gmsh 2.x files are supported (got {ver.decode()})")
    pos = data.index(b"$Nodes\n") + len(b"$Nodes\n")
    end = data.index(b"\n", pos)
    nv = int(data[pos:end])
    pos = end + 1
    if binary:
        rec = np.dtype([("id", "<i4"), ("xyz", "<f8", 3)])
        nodes = np.frombuffer(data, dtype=rec, count=nv, offset=pos)
        ids, pts = nodes["id"].astype(np.int64), np.array(nodes["xyz"], dtype=np.float64)
    else:
        stop = data.index(b"$EndNodes", pos)
        arr = np.array(data[pos:stop].split(), dtype=np.float64).reshape(nv, 4)
        ids, pts = arr[:, 0].astype(np.int64), arr[:, 1:].copy()
    pos = data.index(b"$Elements\n") + len(b"$Elements\n")
    end = data.index(b"\n", pos)
    ne = int(data[pos:end])
    pos = end + 1
    nnodes = {1: 2, 2: 3, 3: 4, 4: 4, 5: 8, 8: 3, 9: 6, 11: 10, 15: 1}
    tets = []
    if binary:
        done = 0
        while done < ne:
            etype, cnt, ntags = struct.unpack_from("<iii", data, pos)
            pos += 12
            width = 1 + ntags + nnodes[etype]
            block = np.frombuffer(data, dtype="<i4", count=cnt * width, offset=pos).reshape(cnt, width)
            pos += 4 * cnt * width
            if etype in (4, 11):
                tets.append(block[:, 1 + ntags:].astype(np.int64))
            done += cnt
    else:
        stop = data.index(b"$EndElements", pos)
        for line in data[pos:stop].splitlines():
            f = line.split()
            if len(f) > 2 and int(f[1]) in (4, 11):
                ntags = int(f[2])
                tets.append(np.array(f[3 + ntags:], dtype=np.int64)[None])
    if not tets:
        raise ValueError(f"{path}: no tetrahedra found")
    tets = np.concatenate(tets, axis=0)
    lut = np.full(ids.max() + 1, -1, dtype=np.int64)
    lut[ids] = np.arange(nv)
    return pts, lut[tets]


def write_gmsh22(path, vertices, tets):
    """Gmsh 2.2 binary writer (tetra = type 4, tetra10 = type 11)."""
    v = np.asarray(vertices, dtype=np.float64)
    t = np.asarray(tets, dtype=np.int32)
    etype = {4: 4, 10: 11}[t.shape[1]]
    with open(path, "wb") as f:
        f.write(b"$MeshFormat\n2.2 1 8\n" + struct.pack("<i", 1) + b"\n$EndMeshFormat\n")
        f.write(b"$Nodes\n%d\n" % len(v))
        rec = np.zeros(len(v), dtype=[("id", "<i4"), ("xyz", "<f8", 3)])
        rec["id"] = np.arange(1, len(v) + 1)
        rec["xyz"] = v
        f.write(rec.tobytes())
        f.write(b"\n$EndNodes\n$Elements\n%d\n" % len(t))
        f.write(struct.pack("<iii", etype, len(t), 0))
        body = np.concatenate([np.arange(1, len(t) + 1, dtype=np.int32)[:, None], t + 1], axis=1)
        f.write(np.ascontiguousarray(body, dtype="<i4").tobytes())
        f.write(b"\n$EndElements\n")


def largest_connected_component(vertices, tets):
    """Keep the largest node-connected component of a tet mesh: (vertices', tets') with nodes renumbered in
    their original order and every element of other components dropped (reference
    src/dmtet/geometry/dmtet_thickness.py:254-285, which round-trips through scipy.sparse.csgraph on the host;
    callers need it because rigid-mode removal assumes ONE free body).  Runs on the tensors'
    device: min-label propagation over the elements with pointer jumping, O(log diameter) sweeps."""
    nv = vertices.shape[0]
    dev = vertices.device
    t = tets.long()
    label = torch.arange(nv, device=dev)
    while True:
        m = label[t].amin(dim=1, keepdim=True).expand_as(t)  # smallest label met in each element
        new = label.scatter_reduce(0, t.reshape(-1), m.reshape(-1), reduce="amin", include_self=True)
        new = new[new]  # pointer jumping: labels are node ids, follow them
        if bool((new == label).all()):
            break
        label = new
    counts = torch.bincount(label, minlength=nv)
    if int((counts > 0).sum()) == 1:
        return vertices, tets
    keep = label == torch.argmax(counts)  # ties: the component with the lowest node id, like the reference's loop
    new_index = torch.full((nv,), -1, dtype=torch.long, device=dev)
    new_index[keep] = torch.arange(int(keep.sum()), device=dev)
    nt = new_index[t]
    return vertices[keep], nt[(nt >= 0).all(dim=1)].to(tets.dtype)


class TetMesh:
    """A tetrahedral mesh: ``vertices`` (nv,3) float tensor and ``tets`` (T,4|10) long tensor."""

    def __init__(self, vertices=None, tets=None, order=1):
        self.vertices = vertices
        self.tets = tets
        if vertices is not None:
            self.device = vertices.device
        self.order = order

    def __repr__(self):
        return "TetMesh(vertices={}, tets={}, order={})".format(self.vertices.shape, self.tets.shape, self.order)

    @staticmethod
    def from_triangle_mesh(filename, log=False):
        """Load ``<filename>_.msh`` (the pre-tetrahedralised companion the reference keeps next to each
        surface mesh, reference mesh.py:37-50).  Running fTetWild is outside this package."""
        msh = filename + "_.msh"
        if not os.path.exists(msh):
            raise FileNotFoundError(
                f"{msh} not found: tetrahedralise {filename} first (the reference shells out to FloatTetwild_bin)")
        pts, tets = read_gmsh22(msh)
        dev = _default_device()
        vertices = torch.from_numpy(pts).float().to(dev)
        tets_t = torch.from_numpy(tets[:, :4]).long().to(dev)
        print("Load tetramesh with ", len(vertices), " vertices & ", len(tets_t), " tets")
        return TetMesh(vertices, tets_t)

    # ------------------------------------------------------------------ geometry
    @property
    def transform_matrix(self):
        """(T,3,3) float32, columns v1-v4, v2-v4, v3-v4 of the corner nodes (reference mesh.py:58-99).
        Differentiable w.r.t. ``vertices``."""
        if not hasattr(self, "_transform_matrix"):
            c = fem_tables.CORNER_SLOTS[self.order]
            p = [self.vertices[self.tets[:, i]] for i in c]
            self._transform_matrix = torch.stack([p[0] - p[3], p[1] - p[3], p[2] - p[3]], dim=2).float()
        return self._transform_matrix

    def to_high_order(self, order):
        """ord-1 -> ord-2: insert the 6 edge midpoints per element into local slots 1,3,5,6,7,8 and merge
        duplicates (reference mesh.py:101-160).  Midpoints stay differentiable w.r.t. the corners."""
        assert self.order == 1
        assert order in (1, 2), "only order 1 and 2 are supported (the reference's order 3 is broken, SURVEY.md 8a-iii)"
        if order == 1:
            return TetMesh(self.vertices, self.tets, order=1)
        T, nv = self.tets.shape[0], self.vertices.shape[0]
        dev = self.vertices.device
        new_tets = torch.zeros((T, 10), dtype=self.tets.dtype, device=dev)
        for slot, corner in zip((0, 2, 4, 9), range(4)):
            new_tets[:, slot] = self.tets[:, corner]
        if self.vertices.is_cuda and self.tets.dtype == torch.int64 and T > 0:
            # device path: one midpoint per distinct edge (ds_edge_table), not one per (element, edge)
            from .. import _hip

            ea, eb, tet_edge = _hip.edge_table(self.tets, nv)
            mids = [(self.vertices[ea] + self.vertices[eb]) / 2]
            for e, slot in enumerate(_EDGE_SLOTS):
                new_tets[:, slot] = nv + tet_edge[:, e]
        else:
            vf = self.vertices[self.tets]
            mids = [(vf[:, a] + vf[:, b]) / 2 for a, b in _EDGES]
            for e, slot in enumerate(_EDGE_SLOTS):
                new_tets[:, slot] = torch.arange(nv + e * T, nv + (e + 1) * T, device=dev)
        new_vertices = torch.cat([self.vertices] + mids, dim=0)
        mesh = TetMesh(new_vertices, new_tets, order=2)
        mesh.remove_duplicate_vertices()
        return mesh

    def remove_duplicate_vertices(self):
        """Merge bit-identical coordinates; nodes end up in lexicographic (x,y,z) order and keep the
        coordinates (and autograd history) of the lowest original index (reference mesh.py:162-179)."""
        if self.vertices.is_cuda and self.vertices.dtype == torch.float32 and self.vertices.shape[0] > 0:
            from .. import _hip

            inv, first = _hip.unique_rows3(self.vertices.detach())  # radix sort on the coordinate bits, no float unique
        else:
            _, inv = torch.unique(self.vertices.detach(), dim=0, return_inverse=True)
            nu = int(inv.max()) + 1
            first = torch.full((nu,), self.vertices.shape[0], dtype=torch.long, device=self.vertices.device)
            first.scatter_reduce_(0, inv, torch.arange(self.vertices.shape[0], device=self.vertices.device),
                                  reduce="amin", include_self=True)
        self.tets = inv[self.tets]
        self.vertices = self.vertices[first]
        if hasattr(self, "_transform_matrix"):
            del self._transform_matrix

    # ------------------------------------------------------------------ I/O
    def import_from_file(self, filename):
        pts, tets = read_gmsh22(filename)
        dev = _default_device()
        self.vertices = torch.from_numpy(pts).float().to(dev)
        self.tets = torch.from_numpy(tets[:, :4]).long().to(dev)
        self.device = self.vertices.device
        self.order = 1
        self.remove_duplicate_vertices()
        print(f"Mesh loaded from file {filename}")
        return self

    def export(self, filename):
        write_gmsh22(filename, self.vertices.detach().cpu().numpy(), self.tets.detach().cpu().numpy())
        print(f"Mesh saved to file {filename}")
