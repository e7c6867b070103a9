"""Device-side FEM system and the HIP implementation of the eigensolver's ``ops`` protocol.

``TetSystem``   one mesh (topology + geometry): symbolic BSR-3 pattern, then K_lambda, K_mu, M_s
                assembled on the GPU in fp64 (reference DiffSoundObj.update_stiff_matrix /
                update_mass_matrix, src/diffelastic/diff_model.py:184-312).
``HipModalOps`` one material hypothesis (lam, mu) on a TetSystem: fp32 K = lam K_lambda + mu K_mu,
                block-Jacobi blocks, rigid-body basis, and every large operation the solver
                needs, each one a call into libdiffsound_hip.so.
No operation here has a CPU implementation; tensors must be HIP tensors.
"""
import ctypes
import os

import numpy as np
import torch

from . import _hip, fem_tables

DS_F32, DS_F64 = 0, 1
import threading

MF_BATCH = 16  # entries per LDS batch of the MFMA kernel (DS_MF_BATCH of include/diffsound_hip.h)
MF_TAIL = 2  # entries a group's last batch may take beyond MF_BATCH when no group has more than 128 (DS_MF_TAIL)
MF32_BATCH = 8  # entries per LDS batch of the fp32 MFMA kernel (DS_MF32_BATCH), groups of MF32_G = 4 nodes
MF32_G = 4
_MFMA_TABLES_LOCK = threading.Lock()
UNION_CAP = 140  # blocks per chunk of the neighbour-union tables (the kernel's LDS image; DS_UNION_CAP of the header)


def _ld(t):
    if t.dim() != 2 or t.stride(1) != 1:
        raise ValueError("block must be a 2-D row-major view (unit column stride)")
    return t.stride(0)


def _axis_buckets(x, lq):
    """Bucket index of every node along ONE axis, and the bucket count.  A structured mesh - the benchmark's Kuhn boxes,
    jittered or not, plates, voxel-derived meshes - has its nodes on PLANES: the sorted coordinates then show a knee between
    the (planes - 1) large gaps that separate the planes and the tiny gaps inside them, and the planes themselves are the
    buckets (ties and jittered clusters stay together).  Without such a knee (unstructured meshes): ``lq`` equally populated
    quantile buckets."""
    nv = x.numel()
    s, o = torch.sort(x, stable=True)
    gaps = s[1:] - s[:-1]
    if gaps.numel():
        k = min(gaps.numel(), 4 * lq + 8)
        g = torch.topk(gaps, k).values  # the largest gaps, descending
        lo, hi = max(1, lq // 4), min(k - 1, 4 * lq)
        if hi > lo:
            ratio = g[lo - 1:hi] / g[lo:hi + 1].clamp(min=1e-300)
            j = int(torch.argmax(ratio))
            planes = lo + j + 1
            if float(ratio[j]) >= 3.0:
                tau = 0.5 * (g[planes - 2] + g[planes - 1])
                cid = torch.cat([torch.zeros(1, dtype=torch.int64, device=x.device), torch.cumsum((gaps > tau).long(), 0)])
                cnt = torch.bincount(cid)
                if int(cnt.max()) <= 4 * max(int(cnt.min()), 1):  # planes of comparable population, not outliers split off
                    q = torch.empty(nv, dtype=torch.int64, device=x.device)
                    q[o] = cid
                    return q, planes
    lq = max(1, lq)
    edges = s[(torch.arange(1, lq, device=x.device) * nv) // lq]
    return torch.searchsorted(edges, x.contiguous(), right=True), lq


def morton_order(vertices):
    """Permutation (new index -> old index) sorting nodes along a 3-D Morton (Z-order) curve over per-axis BUCKET indices
    (``_axis_buckets``: the mesh's own node planes where it has them, quantile slabs where not).  Consecutive node ranges
    then form compact bricks - groups of 4 / 8 consecutive nodes are 2 x 2 x 1 / 2 x 2 x 2 bricks of the node grid on a
    structured mesh - so the rows of a group share most of their neighbours (the neighbour-union SpMM kernels walk the
    UNION of a group's rows) and the block-SpMM's gathers stay inside one XCD's 4 MiB L2.
    Round 4: until then the curve ran over the absolute coordinates quantised to 10 bits, whose cells cut the node grid at
    arbitrary offsets; on the benchmark mesh the unions of 4 / 8 consecutive rows held 0.584 / 0.430 of the rows' blocks,
    with bricks aligned to the node planes 0.461 / 0.282 (unstructured meshes: unchanged within 1 %)."""
    v = vertices.detach().double()
    nv = v.shape[0]
    ext = (v.max(0).values - v.min(0).values).clamp(min=1e-300)
    vol = float(ext.prod())
    q, bits = [], 1
    for a in range(3):
        lq = int(min(1024, max(1, round((nv * float(ext[a]) ** 3 / vol) ** (1.0 / 3.0)))))  # aspect-aware slab count
        qa, la = _axis_buckets(v[:, a], lq)
        q.append(qa)
        bits = max(bits, max(la - 1, 1).bit_length())
    # (Round 5 tried the bricks along a HILBERT curve instead - consecutive bricks always face neighbours, partial bricks behind
    # the full ones so that groups stay aligned: the unions of 64 / 512 consecutive rows shrink by 9 / 7 %, but the kernels run
    # the same times and draw 2.5 % MORE bytes from memory (445 -> 456 MB per bf16 term, 740 -> 760 MB per [K W | M W]);
    # profiles/r05_order_ab.txt.  The Morton curve stays.)
    key = torch.zeros(nv, dtype=torch.int64, device=v.device)
    for b in range(bits):
        for a in range(3):
            key |= ((q[a] >> b) & 1) << (3 * b + a)
    return torch.argsort(key, stable=True)


class TetSystem:
    def __init__(self, vertices, tets, order, density, reorder=True):
        """vertices (nv,3) float32 HIP tensor, tets (T,N) integer HIP tensor in the reference's local
        node order, N = 4 / 10.  With ``reorder`` the nodes are renumbered internally along a Morton
        curve; ``perm`` / ``inv_perm`` map between the caller's node ids and the internal ones and
        ``rows_to_external`` / ``rows_to_internal`` convert (n x c) DOF blocks."""
        _hip.require_gpu(vertices, tets)
        L = _hip.lib()
        self.order = int(order)
        self.N = fem_tables.NODES_PER_TET[self.order]
        if tets.shape[1] != self.N:
            raise ValueError(f"tets must have {self.N} columns for order {order}")
        self.device = vertices.device
        nv0 = vertices.shape[0]
        if reorder:
            self.perm = morton_order(vertices)  # internal -> external
            self.inv_perm = torch.empty_like(self.perm)
            self.inv_perm[self.perm] = torch.arange(nv0, device=self.device)
            self.vertices = vertices.detach().to(torch.float32)[self.perm].contiguous()
            self.tets = self.inv_perm[tets.long()].to(torch.int32).contiguous()
        else:
            self.perm = self.inv_perm = None
            # (a PRIVATE copy: an fp32 contiguous input would otherwise be kept by reference, and a caller that later moves its
            # coordinates in place would move this snapshot with them - assemble(vertices) could then never see a change)
            self.vertices = vertices.detach().to(torch.float32).contiguous().clone()
            self.tets = tets.to(torch.int32).contiguous()
        self.nv = self.vertices.shape[0]
        self.n = 3 * self.nv
        self.T = self.tets.shape[0]
        self.density = float(density)
        # symbolic phase on the device (ds_dpattern_build): pattern, per-slot contribution lists (the numeric phase
        # then needs neither COO, sort nor atomics) and the tables of the neighbour-union SpMM (ds_spmm_union, the
        # kernel of every product on blocks of <= 84 columns): one wavefront per 4 consecutive nodes walks the union
        # of their neighbours (with the Morton numbering 0.58 x as many panel loads as one wavefront per node).  Every
        # group is cut into chunks of whole entries that fit the kernel's LDS images (cap entries / blocks; almost
        # always ONE chunk): ctab rows (e0, e1, b0, b1), utab rows (first chunk, end chunk) per group.
        dev = self.device
        pat = _hip.DevicePattern(self.tets, self.nv, UNION_CAP)
        self.nnzb = pat.nnzb
        self.rowptr, self.colidx, self.diagidx, self.cptr, self.clist = pat.rowptr, pat.colidx, pat.diagidx, pat.cptr, pat.clist
        self.dtab = torch.from_numpy(fem_tables.stiffness_table(self.order)).to(dev)
        self.mtab = torch.from_numpy(fem_tables.mass_table(self.order, self.density)).to(dev)
        self.klam = torch.empty((self.nnzb, 9), dtype=torch.float64, device=dev)
        self.kmu = torch.empty((self.nnzb, 9), dtype=torch.float64, device=dev)
        self.ms = torch.empty((self.nnzb,), dtype=torch.float64, device=dev)
        self._tetgeo = torch.empty((self.T, 13), dtype=torch.float64, device=dev)
        self.groups = None
        if self.nv >= 8:
            self.groups = dict(ne=pat.ne, gent=pat.gent, kperm=pat.kperm, kperm64=pat.kperm.long(),
                               union=dict(utab=pat.utab, ctab=pat.ctab, capb=UNION_CAP, ngroups=pat.ngroups,
                                          single=pat.single))  # single: every group is one chunk
        self._coarse = None
        self.assemble()

    def mfma_tables(self, group_nodes=8, batch=MF_BATCH):
        """Topology tables of the MFMA forms - of the bf16 terms (ds_spmm_union16m, groups of 8 consecutive nodes, batches of
        16 entries) and of the eigensolver's fp32 products (ds_spmm_union32m, groups of 4, batches of 8):
        gptr / gcol = the sorted union of the column ids of each group's rows; gmeta per entry = presence mask of the
        group's nodes | (first block of the entry inside the group) << 8; gbase = first block of each group; kperm = BSR
        block of every position of the (group, entry, node) order.  Built once per topology with device sorts."""
        cache = self.__dict__.setdefault("_mfma_tables", {})
        with _MFMA_TABLES_LOCK:  # hypothesis lanes share the cache (with_own_values copies the dict reference)
            key_ = (int(group_nodes), int(batch))
            if key_ not in cache:
                cache[key_] = self._build_mfma_tables(*key_)
                torch.cuda.current_stream(self.device).synchronize()  # other lanes use the tables on their own streams
        return cache[key_]

    def _build_mfma_tables(self, group_nodes, mf_batch):
        G, nv, dev = int(group_nodes), self.nv, self.device
        rows = torch.repeat_interleave(torch.arange(nv, device=dev), (self.rowptr[1:] - self.rowptr[:-1]).long())
        key = ((rows // G) * nv + self.colidx.long()) * G + rows % G
        key, order = torch.sort(key)
        ekey, inv, counts = torch.unique_consecutive(key // G, return_inverse=True, return_counts=True)
        ng = (nv + G - 1) // G
        gptr = torch.searchsorted(ekey // nv, torch.arange(ng + 1, device=dev))
        goff = torch.zeros(ekey.numel() + 1, dtype=torch.int64, device=dev)
        goff[1:] = torch.cumsum(counts, 0)
        mask = torch.zeros(ekey.numel(), dtype=torch.int64, device=dev)
        mask.scatter_add_(0, inv, torch.ones_like(key) << (key % G))
        gbase = goff[gptr[:-1].clamp(max=ekey.numel())]
        within = goff[:-1] - torch.repeat_interleave(gbase, gptr[1:] - gptr[:-1])
        ne_g = gptr[1:] - gptr[:-1]
        # blocks per batch of mf_batch entries (counted from each group's first entry): sizes the kernel's LDS
        eidx = torch.arange(ekey.numel(), device=dev)
        # (a group of more than 256 entries is not served by the kernel; its tail is lumped into the last slot here)
        nslot = 256 // mf_batch
        ewithin = eidx - torch.repeat_interleave(gptr[:-1], ne_g)
        slot = (ewithin // mf_batch).clamp(max=nslot - 1)
        if int(ne_g.max()) <= 128 and mf_batch == MF_BATCH:
            # the kernel's TAIL form: a group's last batch also takes up to MF_TAIL entries beyond mf_batch (ds_spmm_union16m)
            nb_g = ((ne_g - MF_TAIL + mf_batch - 1) // mf_batch).clamp(min=1)
            slot = torch.minimum(slot, torch.repeat_interleave(nb_g, ne_g) - 1)
        batch = (ekey // nv) * nslot + slot
        per_batch = torch.zeros(ng * nslot, dtype=torch.int64, device=dev).scatter_add_(0, batch, counts)
        gcol = (ekey % nv).to(torch.int32).contiguous()
        gmeta = (mask | (within << 8)).to(torch.int32).contiguous()
        # fixed-stride record of each group's first 64 entries (ids, then meta words; zero behind the last): what a wave asks for
        # before it knows where its group's entries start
        ghead = torch.zeros((ng, 128), dtype=torch.int32, device=dev)
        sel = ewithin < 64
        ghead[(ekey // nv)[sel], ewithin[sel]] = gcol[sel]
        ghead[(ekey // nv)[sel], 64 + ewithin[sel]] = gmeta[sel]
        return dict(G=G, batch=mf_batch, ngroups=ng, max_entries=int(ne_g.max()), max_batch_blocks=int(per_batch.max()),
                    gptr=gptr.to(torch.int32), gcol=gcol, gmeta=gmeta, gbase=gbase.to(torch.int32).contiguous(), ghead=ghead,
                    kperm=order.to(torch.int32).contiguous())

    def mfma_tables_dense(self, group_nodes=8):
        """The tables of ``mfma_tables`` with EVERY (node of the group, entry of its union) position present: what the term kernel
        is handed when the level's blocks are those of T_g K (group-block Jacobi, ds_group_pack_kc).  Same gptr / gcol; gmeta = all
        presence bits of the group's real nodes | (8 x entry) << 8; gbase = 8 x the group's first entry; plain batches of MF_BATCH
        entries (128 blocks: the kernel's form without the tail)."""
        cache = self.__dict__.setdefault("_mfma_tables", {})
        mt = self.mfma_tables(group_nodes)
        with _MFMA_TABLES_LOCK:
            key_ = ("dense", int(group_nodes))
            if key_ not in cache:
                G, nv, dev = int(group_nodes), self.nv, self.device
                gptr = mt["gptr"].long()
                ne_g = gptr[1:] - gptr[:-1]
                ng = ne_g.numel()
                grp = torch.repeat_interleave(torch.arange(ng, device=dev), ne_g)
                ewithin = torch.arange(int(gptr[-1]), device=dev) - gptr[:-1][grp]
                nreal = (nv - G * torch.arange(ng, device=dev)).clamp(max=G)
                mask = ((1 << nreal) - 1)[grp]
                gmeta = (mask | ((G * ewithin) << 8)).to(torch.int32).contiguous()
                ghead = torch.zeros((ng, 128), dtype=torch.int32, device=dev)
                sel = ewithin < 64
                ghead[grp[sel], ewithin[sel]] = mt["gcol"][sel]
                ghead[grp[sel], 64 + ewithin[sel]] = gmeta[sel]
                cache[key_] = dict(G=G, batch=mt["batch"], ngroups=ng, max_entries=mt["max_entries"],
                                   max_batch_blocks=G * min(mt["batch"], mt["max_entries"]), gptr=mt["gptr"], gcol=mt["gcol"],
                                   gmeta=gmeta, gbase=(G * gptr[:-1]).to(torch.int32).contiguous(), ghead=ghead,
                                   nblocks=G * int(gptr[-1]))
                torch.cuda.current_stream(self.device).synchronize()
        return cache[key_]

    def with_own_values(self):
        """A view of this system that shares the mesh, pattern and tables but OWNS its assembled values
        (K_lambda, K_mu, M_s, per-tet geometry; 0.7 GB on the benchmark mesh): concurrent hypothesis lanes each
        run their own numeric assembly without racing on the shared arrays."""
        import copy

        o = copy.copy(self)
        o.klam, o.kmu, o.ms = torch.empty_like(self.klam), torch.empty_like(self.kmu), torch.empty_like(self.ms)
        o._tetgeo = torch.empty_like(self._tetgeo)
        if self._coarse is not None:
            o._coarse = dict(self._coarse)
            o._coarse["sys"] = self._coarse["sys"].with_own_values()
        o.assemble()
        return o

    def coarse_level(self):
        """ord-2 meshes only: the corner-node (P1) sub-mesh as an ord-1 ``TetSystem`` plus the transfer
        operators between the two levels, both in internal numbering.  P1 is a subspace of P2 on the same
        tets, so the ord-1 stiffness of the sub-mesh IS the Galerkin operator P^T K P, with P = "corner
        copies its coarse value, mid-edge node (reference mesh.py:139-154) averages its edge's end points".
        Returns None when there is no such level (ord-1 mesh, or nodes that no tet references)."""
        if self._coarse is not None or self.order != 2:
            return self._coarse
        dev = self.device
        tets = self.tets.long()
        cs = fem_tables.CORNER_SLOTS[2]
        corners = torch.unique(tets[:, list(cs)])  # ascending internal ids: the sub-mesh inherits the Morton order
        cid = torch.full((self.nv,), -1, dtype=torch.int64, device=dev)
        cid[corners] = torch.arange(corners.numel(), device=dev)
        pa, pb = cid.clone(), cid.clone()
        for slot, (p_, q_) in fem_tables._MID.items():
            pa[tets[:, slot]] = cid[tets[:, cs[p_]]]
            pb[tets[:, slot]] = cid[tets[:, cs[q_]]]
        if bool((pa < 0).any()) or bool((pb < 0).any()):
            return None
        nvc = corners.numel()
        csys = TetSystem(self.vertices[corners], cid[tets[:, list(cs)]], 1, self.density, reorder=False)
        i32 = lambda t: t.to(torch.int32).contiguous()
        fine = torch.arange(self.nv, device=dev)
        mid = pa != pb
        # restriction rows (coarse node <- itself, weight 1, and the mid-edge nodes of its edges, weight 1/2)
        rrow = torch.cat([pa, pb[mid]])
        rcol = torch.cat([fine, fine[mid]])
        rw = torch.cat([torch.where(mid, 0.5, 1.0), torch.full((int(mid.sum()),), 0.5, device=dev)])
        o = torch.argsort(rrow, stable=True)
        rptr = torch.zeros(nvc + 1, dtype=torch.int64, device=dev)
        rptr[1:] = torch.cumsum(torch.bincount(rrow, minlength=nvc), 0)
        self._coarse = dict(
            sys=csys, corners=corners,
            pptr=i32(torch.arange(0, 2 * self.nv + 1, 2, device=dev)), pcol=i32(torch.stack([pa, pb], 1).reshape(-1)),
            pw=torch.full((2 * self.nv,), 0.5, dtype=torch.float32, device=dev),
            rptr=i32(rptr), rcol=i32(rcol[o]), rw=rw[o].float().contiguous())
        return self._coarse

    def rows_to_external(self, X):
        """(n x c) block in internal DOF order -> the caller's node numbering."""
        if self.perm is None:
            return X
        return X.reshape(self.nv, 3, -1)[self.inv_perm].reshape(self.n, -1)

    def rows_to_internal(self, X):
        if self.perm is None:
            return X
        return X.reshape(self.nv, 3, -1)[self.perm].reshape(self.n, -1)

    def assemble(self, vertices=None):
        """Numeric phase only (pattern reused): refresh K_lambda, K_mu, M_s from the coordinates
        (given in the caller's node numbering)."""
        changed = False
        if vertices is not None:
            v = vertices.detach().to(torch.float32)
            # (without a permutation ``v`` may BE the caller's storage: the kept snapshot is always a copy of our own)
            v = v.contiguous().clone() if self.perm is None else v[self.perm].contiguous()
            # (what depends on the geometry only - the rigid-body basis, the solver's norm probe - is kept per GENERATION of the
            # coordinates; a caller that hands the same coordinates over again, as DiffSoundObj.eigen_decomposition does on every
            # call, stays in the generation)
            changed = v.shape != self.vertices.shape or not bool(torch.equal(v, self.vertices))
            self.vertices = v
            if changed:
                self.geometry_generation = getattr(self, "geometry_generation", 0) + 1
            elif getattr(self, "_assembled_generation", None) == getattr(self, "geometry_generation", 0):
                # The same coordinates as the last assembly (DiffSoundObj.eigen_decomposition hands them over on every call of a
                # material-fit loop): K_lambda, K_mu and M_s are functions of the geometry alone and are in place - nothing to do
                # on either level (round 6; the headline's passes call assemble() WITHOUT coordinates and always assemble: the
                # numeric assembly is part of the pass the metric defines).
                self.assemblies_skipped = getattr(self, "assemblies_skipped", 0) + 1
                return
        L = _hip.lib()
        p = _hip.ptr
        _hip.check(L.ds_assemble_kml(p(self.vertices), p(self.tets), self.T, self.N, self.nv, p(self.cptr),
                                     p(self.clist), self.nnzb, p(self.dtab), p(self.mtab), p(self._tetgeo),
                                     p(self.klam), p(self.kmu), p(self.ms), _hip.stream_ptr()), "ds_assemble_kml")
        self._assembled_generation = getattr(self, "geometry_generation", 0)
        if getattr(self, "_coarse", None) is not None:
            # (the corner-node level always receives the coordinates it is to be assembled on when the caller handed any over:
            # its own change detection decides whether its generation moves)
            self._coarse["sys"].assemble(self.vertices[self._coarse["corners"]] if vertices is not None else None)

    def geometry_grad(self, U, gk, gm, lam, mu):
        """d/dx sum_i gk_i u_i^T K u_i - gm_i u_i^T M u_i  ->  (nv, 3) fp64 in the caller's node numbering.
        U: (n, m) f32 modes in INTERNAL order (as the solver returns them); geometry = last assemble()."""
        order = self.order
        gt, gw = fem_tables.minimal_gradient_rule(order)
        dev = self.device
        gtab = torch.from_numpy(gt).to(dev)
        gwt = torch.from_numpy(gw).to(dev)
        grad = torch.zeros((self.nv, 3), dtype=torch.float64, device=dev)
        U = U.contiguous()
        p = _hip.ptr
        _hip.check(_hip.lib().ds_geometry_grad(p(self.tets), self.T, self.N, self.nv, p(self._tetgeo), p(U), U.stride(0),
                                               U.shape[1], p(gk.double().contiguous()), p(gm.double().contiguous()),
                                               float(lam), float(mu), p(gtab), p(gwt), gt.shape[0], p(self.mtab), p(grad),
                                               _hip.stream_ptr()), "ds_geometry_grad")
        return grad if self.perm is None else grad[self.inv_perm]

    # scipy views for tests / interop (host copies, in the caller's node numbering)
    def to_scipy(self, lam=None, mu=None):
        import scipy.sparse as sp

        rp = self.rowptr.cpu().numpy()
        ci = self.colidx.cpu().numpy()
        mk = lambda v: sp.bsr_matrix((v.cpu().numpy().reshape(-1, 3, 3), ci, rp), shape=(self.n, self.n)).tocsr()
        Kl, Km = mk(self.klam), mk(self.kmu)
        Ms = sp.csr_matrix((self.ms.cpu().numpy(), ci, rp), shape=(self.nv, self.nv))
        if self.perm is not None:
            ip = self.inv_perm.cpu().numpy()
            dof = (3 * ip[:, None] + np.arange(3)[None, :]).reshape(-1)
            Kl, Km, Ms = Kl[dof][:, dof], Km[dof][:, dof], Ms[ip][:, ip]
        if lam is None:
            return Kl, Km, Ms
        return (lam * Kl + mu * Km).tocsr(), sp.kron(Ms, sp.identity(3), format="csr")


class _HipBlockOps:
    """HIP implementation of the solver's ``ops`` protocol on a BSR-3 pattern (shared part).

    Subclasses provide: rowptr, colidx, nv, k32 (nnzb x 9 f32), ms32 (+ m_kind), dinv, rigid,
    lame and polish_terms()."""

    dtype = torch.float32
    m_kind = 1  # 1: M = M_s (x) I3 (one scalar per block), 0: general 3x3 blocks
    k32t = None
    kgrp = None  # transposed blocks in node-group order (neighbour-union SpMM)
    mgrp = None  # node-scalar mass values in node-group order (neighbour-union SpMM, epilogue 3)

    def _init_common(self, rowptr, colidx, nv, device):
        self.rowptr, self.colidx = rowptr, colidx
        self.nv = nv
        self.n = 3 * nv
        self.device = device
        self._L = _hip.lib()
        self._gram_ws = None
        self._native_ws = {}
        self.gram_exact = False
        self._tmp = {}
        self._nrm = torch.empty((2, 1024), dtype=torch.float64, device=device)
        self.counts = dict(apply_K_cols=0, apply_M_cols=0, gram=0, mix=0, mix64=0)

    # ------------------------------------------------------------------ sparse products
    def _spmm(self, kind, vals, X, out):
        p = _hip.ptr
        ncols = X.shape[1]
        if out.shape != X.shape:
            raise ValueError("spmm: shape mismatch")
        maxc = 256 if kind < 2 else 128
        for c0 in range(0, ncols, maxc):
            c1 = min(ncols, c0 + maxc)
            xs, os_ = X[:, c0:c1], out[:, c0:c1]
            vt = self.k32t if (kind == 0 and vals is self.k32) else None
            _hip.check(self._L.ds_spmm_bsr3(kind, p(self.rowptr), p(self.colidx), p(vals), p(vt), self.nv, p(xs),
                                            _ld(xs), p(os_), _ld(os_), c1 - c0, _hip.stream_ptr()), "ds_spmm_bsr3")

    @staticmethod
    def col_slices(c):
        """Column ranges of at most 84 columns (multiples of 4, as equal as possible) that tile a c-column block: what the
        neighbour-union kernels take per launch.  136 -> (0, 68), (68, 136); 240 -> three of 80."""
        if c <= 84:
            return [(0, c)]
        k = -(-c // 84)
        w = -(-(-(-c // k)) // 4) * 4
        return [(c0, min(c, c0 + w)) for c0 in range(0, c, w)]

    def _union_ok(self, X, *others, wide=False):
        g = getattr(getattr(self, "sys", None), "groups", None)
        if g is None or g.get("union") is None or self.kgrp is None or (X.shape[1] > 84 and not wide) or X.shape[1] % 4:
            return False
        # every operand is read / written 16 bytes at a time (blocks of 2 GB and more take the kernel's per-panel
        # descriptor variant; the dinv table and the value array stay under one descriptor: nv * 36, nnzb * 36 < 4 GB)
        if self.nv * 36 >= 0x7F000000 or self.kgrp.shape[0] * 36 >= (1 << 32):
            return False
        for T in (X,) + others:
            if T is not None:
                ld = T.stride(0)
                if ld % 4 or T.data_ptr() % 16 or T.stride(1) != 1:
                    return False
        return True

    def level_desc(self, d, degree, lmax, lmin):
        """Fill a ds_level_t with this level's neighbour-union tables (None when the level has none)."""
        g = getattr(getattr(self, "sys", None), "groups", None)
        if g is None or g.get("union") is None or self.kgrp is None:
            return None
        u = g["union"]
        d.utab, d.ctab, d.ngroups, d.cap_blocks = (None if u.get("single") else u["utab"].data_ptr()), u["ctab"].data_ptr(), u["ngroups"], u["capb"]
        d.gent, d.kgrp, d.nnzb, d.nv, d.dinv = g["gent"].data_ptr(), self.kgrp.data_ptr(), self.kgrp.shape[0], self.nv, self.dinv.data_ptr()
        d.degree, d.lmax, d.lmin = int(degree), float(lmax), float(lmin)
        d.level_tag = self._level_tag
        mt = self._mfma
        d.tgrp, d.mf_nblocks = None, 0
        if self.group_jacobi and self.tgrp is not None:
            # group-block Jacobi: the blocks of T_g K on the dense tables, an identity for dinv, T_g for the right-hand side
            md = self._mfma_dense
            d.mf_group_nodes, d.mf_max_entries, d.mf_max_batch_blocks = md["G"], md["max_entries"], md["max_batch_blocks"]
            d.mf_gptr, d.mf_gcol, d.mf_gmeta, d.mf_gbase, d.mf_ghead = (md[k].data_ptr() for k in ("gptr", "gcol", "gmeta", "gbase", "ghead"))
            d.mf_kc, d.mf_nblocks = self.kc_dense.data_ptr(), md["nblocks"]
            d.tgrp, d.dinv = self.tgrp.data_ptr(), self.dinv_id.data_ptr()
        elif mt is not None and self.kc is not None:  # the level's bf16 terms run on the matrix cores (ds_spmm_union16m)
            d.mf_group_nodes, d.mf_max_entries, d.mf_max_batch_blocks = mt["G"], mt["max_entries"], mt["max_batch_blocks"]
            d.mf_gptr, d.mf_gcol, d.mf_gmeta, d.mf_gbase, d.mf_ghead = (mt[k].data_ptr() for k in ("gptr", "gcol", "gmeta", "gbase", "ghead"))
            d.mf_kc = self.kc.data_ptr()
        else:
            d.mf_group_nodes = 0
        m4 = self._mfma32
        if m4 is not None and self.k4 is not None:  # the level's own fp32 products run on the matrix cores (ds_spmm_union32m)
            d.m32_max_entries, d.m32_max_batch_blocks = m4["max_entries"], m4["max_batch_blocks"]
            d.m32_gptr, d.m32_gcol, d.m32_gmeta, d.m32_gbase = (m4[k].data_ptr() for k in ("gptr", "gcol", "gmeta", "gbase"))
            d.m32_k = self.k4.data_ptr()
            d.m32_m = None if self.m4 is None else self.m4.data_ptr()
        else:
            d.m32_gptr = None
        return d

    _mfma = None  # tables of the MFMA form of the bf16 terms (TetSystem.mfma_tables), None: the VALU kernel
    # GROUP-block Jacobi of the level's bf16 polynomial (round 6; the corner-node level only): 8 = T is the inverse of the 24 x 24
    # diagonal block of every group of 8 nodes of the matrix-core tables, 0 = the 3 x 3 node blocks (dinv).  tgrp (ng, 24, 24) fp32,
    # kc_dense the blocks of T_g K on TetSystem.mfma_tables_dense, dinv_id an identity per node - all per material (set_material).
    group_jacobi = 0
    tgrp = kc_dense = dinv_id = _mfma_dense = None
    kc = None     # (nnzb, 3, 4) bf16: the 3x3 blocks in the order of those tables (ds_pack_kc)
    _mfma32 = None  # tables of the fp32 MFMA form of the level's own products K X / M X (groups of 4 nodes), None: VALU
    k4 = None     # (nnzb * 9 + 4,) fp32: the 3x3 blocks (row-major) in the order of those tables, 16 bytes of slack
    m4 = None     # (nnzb + 4,) fp32: the node-scalar mass values in that order
    _level_tag = 0  # 0: fine level, 1: corner-node level (selects kernel symbols, nothing else)

    def _union32_ok(self, X, out):
        return (self._mfma32 is not None and self.k4 is not None and self._union_ok(X, out)
                and 3 * self.nv * X.stride(0) * 4 < 0x7F000000)

    def _union32(self, epilogue, X, Y):
        pp = _hip.ptr
        m4 = self._mfma32
        vals = self.m4 if epilogue == 3 else self.k4
        _hip.check(self._L.ds_spmm_union32m(epilogue, self._level_tag, pp(m4["gptr"]), pp(m4["gcol"]), pp(m4["gmeta"]),
                                            pp(m4["gbase"]), pp(vals), vals.numel() * 4, self.colidx.shape[0], m4["ngroups"],
                                            m4["max_entries"], m4["max_batch_blocks"], self.nv, pp(X), _ld(X), pp(Y), _ld(Y),
                                            X.shape[1], _hip.stream_ptr()), "ds_spmm_union32m")

    def twolevel_apply(self, smooth, coarse, R, W, D, AD, Rr, Rc, Ec, Dc, ADc, Wc, R16=None):
        """The whole two-level V-cycle W = B R through the native driver (ds_twolevel_apply): one call instead of
        ~45 launches issued one by one.  ``smooth`` / ``coarse``: (degree, lmax, lmin) of the two Chebyshev operators.
        R16 given: every scratch block (D ... Wc, R16) is bf16 and the cycle runs on bf16 iterates (R, W stay fp32).
        Returns False (nothing done) when a level or a block does not qualify for the neighbour-union kernels."""
        co = self.coarse
        if co is None:
            return False
        if R16 is not None:
            blocks = (D, AD, Rr, Wc, R16, Rc, Ec, Dc, ADc)
            ok = all(t.dtype == torch.bfloat16 and t.stride(1) == 1 and t.stride(0) % 4 == 0 and t.data_ptr() % 8 == 0
                     for t in blocks) and self._union_ok(R, W) and self.kgrp is not None and co.kgrp is not None
            if not ok:
                return False
        elif not (self._union_ok(R, W, D, AD, Rr, Wc) and co._union_ok(Rc, Ec, Dc, ADc)):
            return False
        d = self._tl_desc
        if d is None:
            d = self._tl_desc = _hip.TwoLevelDesc()
            t = self._xfer
            d.rptr, d.rcol, d.rw = t["rptr"].data_ptr(), t["rcol"].data_ptr(), t["rw"].data_ptr()
            d.pptr, d.pcol, d.pw = t["pptr"].data_ptr(), t["pcol"].data_ptr(), t["pw"].data_ptr()
        if self.level_desc(d.fine, *smooth) is None or co.level_desc(d.coarse, *coarse) is None:
            return False
        if not (Rc.stride(0) == Ec.stride(0) == Dc.stride(0) == ADc.stride(0) and
                Wc.stride(0) == D.stride(0) == AD.stride(0)):
            return False
        d.R, d.ldr, d.W, d.ldw = R.data_ptr(), R.stride(0), W.data_ptr(), W.stride(0)
        d.D, d.ldd, d.AD, d.lda = D.data_ptr(), D.stride(0), AD.data_ptr(), AD.stride(0)
        d.Rr, d.ldrr = Rr.data_ptr(), Rr.stride(0)
        d.Rc, d.Ec, d.Dc, d.ADc, d.ldc = Rc.data_ptr(), Ec.data_ptr(), Dc.data_ptr(), ADc.data_ptr(), Rc.stride(0)
        d.ncols = R.shape[1]
        d.Wc, d.ldwc = Wc.data_ptr(), Wc.stride(0)
        d.storage = 0 if R16 is None else 1
        d.R16, d.ldr16 = (None, 0) if R16 is None else (R16.data_ptr(), R16.stride(0))
        _hip.check(self._L.ds_twolevel_apply(ctypes.byref(d), _hip.stream_ptr()), "ds_twolevel_apply")
        c = R.shape[1]
        self.counts["apply_K_cols"] += c * (max(smooth[0] - 1, 0) + 1 + smooth[0])
        co.counts["apply_K_cols"] += c * (coarse[0] - 1)
        return True

    _tl_desc = None

    def chebyshev_apply16(self, precond, R, W):
        """W <- p(T K) T R through the native one-level driver on bf16 iterates (ds_chebyshev_apply16: the launches the native
        iteration issues for the same preconditioner) for blocks of <= 84 columns; False when the level or block does not qualify."""
        d = _hip.LevelDesc()
        if (precond.degree < 2 or R.shape[1] > 84 or R.shape[1] % 4 or not self._union_ok(R, W)
                or self.level_desc(d, precond.degree, precond.lmax, precond.lmin) is None or self._mfma is None or self.kc is None):
            return False
        b = R.shape[1]
        scr = self._scratch("native_cheb", (3, self.n, b), torch.bfloat16)
        _hip.check(self._L.ds_chebyshev_apply16(ctypes.byref(d), R.data_ptr(), _ld(R), W.data_ptr(), _ld(W), scr[0].data_ptr(),
                                                scr[1].data_ptr(), scr[2].data_ptr(), b, b, _hip.stream_ptr()), "ds_chebyshev_apply16")
        self.counts["apply_K_cols"] += b * (precond.degree - 1)
        return True

    # ------------------------------------------------------------------ native iteration driver
    def native_lobpcg(self, precond, cfg, k, b, ny, S, S2, KS, KS2, R, MX, MW, lam, A_norm, B_norm, tol):
        """Run the eigensolver's iteration through ds_lobpcg_iterate (csrc/lobpcg.cpp).  Returns None when this
        configuration has to stay on the Python loop (block wider than the union kernels take, a mass matrix that is
        not node-scalar, a preconditioner the driver does not know), else
        (iterations, result_in_s2, lam (b,) fp64 device, rerr (b,) fp64 device, history [(it, worst backward error)])."""
        from .lobpcg.modal_solver import ChebyshevBlockJacobi, TwoLevelChebyshev

        g = getattr(getattr(self, "sys", None), "groups", None)
        if (g is None or g.get("union") is None or self.kgrp is None or self.mgrp is None or self.m_kind != 1
                or b > 160 or b % 4 or ny % 4 or not self._union_ok(R, MX, MW, S[:, ny:ny + b], KS[:, :b], wide=True)):
            return None
        dev = self.device
        d = _hip.LobpcgDesc()
        keep = []  # tensors the descriptor points into
        if isinstance(precond, TwoLevelChebyshev):
            co = self.coarse
            if co is None or precond.ops is not self:
                return None
            tl = _hip.TwoLevelDesc()
            t = self._xfer
            tl.rptr, tl.rcol, tl.rw = t["rptr"].data_ptr(), t["rcol"].data_ptr(), t["rw"].data_ptr()
            tl.pptr, tl.pcol, tl.pw = t["pptr"].data_ptr(), t["pcol"].data_ptr(), t["pw"].data_ptr()
            sm, cs = precond.smooth, precond.coarse
            if (self.level_desc(tl.fine, sm.degree, sm.lmax, sm.lmin) is None
                    or co.level_desc(tl.coarse, cs.degree, cs.lmax, cs.lmin) is None):
                return None
            bf = cfg.precond_storage == "bf16"
            if cs.group and not bf:
                return None  # (the group-block Jacobi lives on the bf16 cycle: an fp32 cycle goes through the Python loop)
            sdt = torch.bfloat16 if bf else torch.float32
            scr = self._scratch("native_tl_fine", (5, self.n, b), sdt)
            scc = co._scratch("native_tl_coarse", (4, co.n, b), sdt)
            tl.Wc, tl.D, tl.AD, tl.Rr = (scr[i].data_ptr() for i in range(4))
            tl.ldwc = tl.ldd = tl.lda = tl.ldrr = b
            tl.Rc, tl.Ec, tl.Dc, tl.ADc = (scc[i].data_ptr() for i in range(4))
            tl.ldc = b
            tl.storage = 1 if bf else 0
            tl.R16, tl.ldr16 = (scr[4].data_ptr(), b) if bf else (None, 0)
            tl.R = tl.W = 1  # (set per application by the driver; non-null for its argument check)
            d.twolevel = ctypes.pointer(tl)
            keep += [tl, scr, scc]
            self.level_desc(d.level, sm.degree, sm.lmax, sm.lmin)
        elif isinstance(precond, ChebyshevBlockJacobi):
            if precond.ops is not self or self.level_desc(d.level, precond.degree, precond.lmax, precond.lmin) is None:
                return None
            bf = cfg.precond_storage == "bf16" and precond.degree >= 2
            if precond.group and not bf:
                return None
            scr = self._scratch("native_cheb", (3, self.n, b), torch.bfloat16 if bf else torch.float32)
            d.pa, d.pb, d.ldp = scr[0].data_ptr(), scr[1].data_ptr(), b
            d.pr16 = scr[2].data_ptr() if bf else None
            keep.append(scr)
        else:
            return None
        d.n, d.nv, d.b, d.k, d.ny = self.n, self.nv, b, k, ny
        d.maxit, d.lock, d.ortho_passes, d.rr_refresh = cfg.maxit, int(cfg.lock), cfg.ortho_passes, cfg.rr_refresh
        d.gram_exact = int(bool(self.gram_exact))
        d.kx_fresh = int(bool(getattr(cfg, "kx_fresh", False)))
        d.raw_rr = int(bool(getattr(cfg, "raw_rr", False)))
        d.tol, d.ortho_tol, d.A_norm, d.B_norm = float(tol), float(cfg.ortho_tol), A_norm, B_norm
        d.S, d.S2, d.KS, d.KS2 = S.data_ptr(), S2.data_ptr(), KS.data_ptr(), KS2.data_ptr()
        d.R, d.MX, d.MW = R.data_ptr(), MX.data_ptr(), MW.data_ptr()
        d.lds, d.ldks, d.ldr = S.stride(0), KS.stride(0), R.stride(0)
        if not (S2.stride(0) == d.lds and KS2.stride(0) == d.ldks and MX.stride(0) == d.ldr and MW.stride(0) == d.ldr):
            return None
        d.mgrp = self.mgrp.data_ptr()
        d.rowptr, d.colidx, d.k32, d.k32t = (self.rowptr.data_ptr(), self.colidx.data_ptr(), self.k32.data_ptr(),
                                             self.k32t.data_ptr())
        m = ny + 3 * b
        gbuf = self._scratch("native_g", (m * 3 * b,), torch.float64)
        cbuf = self._scratch("native_c", (8 * m * 2 * b,), torch.float32)
        lam_dev = self._scratch("native_lam", (b,), torch.float64)
        key = (self.n, b, ny)
        need = self._native_ws.get(key)
        if need is None:  # the split count depends on the shape: take the largest need over every shape the driver forms
            # (q: the active width na, 2 na for [K W | M W] of the Ritz step on the raw basis, p itself for the full refresh)
            need = max(self._L.ds_gram_workspace_bytes(self.n, p_, q_)
                       for p_ in range(4, m + 1, 4) for q_ in sorted(set(range(4, 2 * b + 1, 4)) | {p_}) if q_ <= 3 * b)
            self._native_ws[key] = need
        if self._gram_ws is None or self._gram_ws.numel() < need:
            self._gram_ws = torch.empty((need,), dtype=torch.uint8, device=dev)
        d.gbuf, d.cbuf, d.nrm, d.lam_dev = gbuf.data_ptr(), cbuf.data_ptr(), self._nrm.data_ptr(), lam_dev.data_ptr()
        d.gram_work, d.gram_work_bytes = self._gram_ws.data_ptr(), self._gram_ws.numel()
        if getattr(cfg, "fused_residual", False) and getattr(cfg, "kx_fresh", False):
            rws = self._residual_ws(b)
            d.res_work, d.res_work_bytes = rws.data_ptr(), rws.numel()
        else:
            d.res_work, d.res_work_bytes = None, 0
        lam_h = (ctypes.c_double * b)(*lam.detach().double().cpu().tolist())
        rerr_h = (ctypes.c_double * b)()
        hist_h = (ctypes.c_double * (cfg.maxit + 1))()
        d.lam, d.rerr, d.history, d.history_cap = lam_h, rerr_h, hist_h, cfg.maxit + 1
        d.ritz_tol = float(getattr(cfg, "ritz_tol", 0.0))
        d.wait_mode = int(getattr(self, "host_wait_mode", -1))  # (this operator object's - i.e. this lane's - own setting; -1: the process default)
        with _hip.blas_one_thread():
            _hip.check(self._L.ds_lobpcg_iterate(ctypes.byref(d), ctypes.byref(_hip.lapack_table()), _hip.stream_ptr()),
                       "ds_lobpcg_iterate")
        it = int(d.iterations)
        lam_t = torch.tensor(list(lam_h), dtype=torch.float64, device=dev)
        rel_t = torch.tensor(list(rerr_h), dtype=torch.float64, device=dev)
        history = [(i, hist_h[i]) for i in range(min(it + 1, cfg.maxit + 1))]
        del keep
        return it, bool(d.result_in_s2), lam_t, rel_t, history

    def _union(self, epilogue, X, Y, R0=None, c1=0.0, c2=0.0, first=False, Wprev=None):
        pp = _hip.ptr
        g = self.sys.groups
        u = g["union"]
        vals = self.mgrp if epilogue == 3 else self.kgrp
        _hip.check(self._L.ds_spmm_union(epilogue, self._level_tag, None if u.get("single") else pp(u["utab"]), pp(u["ctab"]), u["ngroups"], u["capb"], pp(g["gent"]), pp(vals),
                                         vals.shape[0], self.nv, pp(X), _ld(X), pp(Y), _ld(Y), pp(R0),
                                         0 if R0 is None else _ld(R0), pp(self.dinv) if epilogue == 1 else None,
                                         X.shape[1], float(c1), float(c2), int(bool(first)), pp(Wprev),
                                         0 if Wprev is None else _ld(Wprev), _hip.stream_ptr()),
                   "ds_spmm_union")

    def _narrow(self, kind, X, Y):
        """<= 16 columns: the kernel that deals a wave's lanes over the union's entries (ds_spmm_union_narrow)."""
        pp = _hip.ptr
        g, u = self.sys.groups, self.sys.groups["union"]
        vals = self.mgrp if kind == 3 else self.kgrp
        _hip.check(self._L.ds_spmm_union_narrow(kind, self._level_tag, None if u.get("single") else pp(u["utab"]), pp(u["ctab"]),
                                                u["ngroups"], pp(g["gent"]), pp(vals), vals.shape[0], self.nv, pp(X), _ld(X), pp(Y),
                                                _ld(Y), X.shape[1], _hip.stream_ptr()), "ds_spmm_union_narrow")

    def apply_K(self, X, out):
        if self._union32_ok(X, out):
            self._union32(0, X, out)
        elif X.shape[1] <= 16 and self._level_tag == 0 and self._union_ok(X, out):
            # (fine level only: 131 -> 115 us on 8 columns at C3; the corner-node level's production launch is as short as a wave's
            # life either way, and the node-scalar product M X is faster on the production kernel: profiles/r05_mb_narrow.txt)
            self._narrow(0, X, out)
        elif self._union_ok(X, out):
            self._union(0, X, out)
        elif self._union_ok(X, out, wide=True):
            # wider than one launch takes (configs[4]'s 136-column block, the periodic refresh K [X P W]): column slices through the
            # same kernel - the wave-per-node kernel this used to fall to runs at 21 % of STREAM on the 1M-tet mesh, the slices at ~40 %
            for c0, c1 in self.col_slices(X.shape[1]):
                self._union(0, X[:, c0:c1], out[:, c0:c1])
        else:
            self._spmm(0, self.k32, X, out)
        self.counts["apply_K_cols"] += X.shape[1]

    def apply_KM_ok(self, X, KX, MX):
        return self.m_kind == 1 and self.mgrp is not None and self._union_ok(X, KX, MX, wide=True)

    def apply_KM(self, X, KX, MX):
        """KX <- K X and MX <- M X in ONE walk of the neighbour unions (ds_spmm_union_km): X is gathered once; each product
        equals what apply_K / apply_M give bit for bit."""
        pp = _hip.ptr
        g, u = self.sys.groups, self.sys.groups["union"]
        for c0, c1 in self.col_slices(X.shape[1]):
            xs, ks, ms = X[:, c0:c1], KX[:, c0:c1], MX[:, c0:c1]
            _hip.check(self._L.ds_spmm_union_km(self._level_tag, None if u.get("single") else pp(u["utab"]), pp(u["ctab"]), u["ngroups"],
                                                u["capb"], pp(g["gent"]), pp(self.kgrp), pp(self.mgrp), self.kgrp.shape[0], self.nv,
                                                pp(xs), _ld(xs), pp(ks), _ld(ks), pp(ms), _ld(ms), c1 - c0, _hip.stream_ptr()),
                       "ds_spmm_union_km")
        self.counts["apply_K_cols"] += X.shape[1]
        self.counts["apply_M_cols"] += X.shape[1]

    def apply_M(self, X, out):
        if self.m_kind == 1 and self.m4 is not None and self._union32_ok(X, out):
            self._union32(3, X, out)
        elif self.m_kind == 1 and self.mgrp is not None and self._union_ok(X, out):
            self._union(3, X, out)
        elif self.m_kind == 1 and self.mgrp is not None and self._union_ok(X, out, wide=True):
            for c0, c1 in self.col_slices(X.shape[1]):
                self._union(3, X[:, c0:c1], out[:, c0:c1])
        else:
            self._spmm(self.m_kind, self.ms32, X, out)
        self.counts["apply_M_cols"] += X.shape[1]

    # ------------------------------------------------------------------ tall-skinny dense
    def gram(self, A, B, symmetric=False, exact=False):
        """G = A^T B in fp64.  exact=False: fp32 MFMA folded into fp64 every 48 rows (~1e-9 of |A_i||B_j| at the
        benchmark's row count, ~1e-7 on a few hundred rows); ops with gram_exact set always take the fp64 MFMA."""
        exact = exact or self.gram_exact
        p, q = A.shape[1], B.shape[1]
        need = self._L.ds_gram_workspace_bytes(self.n, p, q)
        if self._gram_ws is None or self._gram_ws.numel() < need:
            self._gram_ws = torch.empty((need,), dtype=torch.uint8, device=self.device)
        G = torch.empty((p, q), dtype=torch.float64, device=self.device)
        adt = DS_F64 if A.dtype == torch.float64 else DS_F32
        bdt = DS_F64 if B.dtype == torch.float64 else DS_F32
        pp = _hip.ptr
        _hip.check(self._L.ds_gram(pp(A), adt, _ld(A), p, pp(B), bdt, _ld(B), q, self.n, int(bool(symmetric)) | (2 if exact else 0), pp(G),
                                   pp(self._gram_ws), self._gram_ws.numel(), _hip.stream_ptr()), "ds_gram")
        self.counts["gram"] += 1
        return G

    def gram_blocks(self, A_blocks, B_blocks, symmetric=False):
        """G = [A_0 | A_1 | ...]^T [B_0 | B_1 | ...] in fp64 (ds_gram64_blocks) for bases held as LISTS of (n x p) fp64
        blocks: one pass over the rows for all pairs of blocks.  ``symmetric``: B = K A with a symmetric K and the same
        widths on both sides - only the tiles on and above the diagonal are computed.  At most 4 blocks per side."""
        def table(blocks):
            arr, off = (_hip.Block64 * len(blocks))(), 0
            for d, blk in zip(arr, blocks):
                if blk.dtype != torch.float64 or blk.shape[0] != self.n or blk.stride(1) != 1:
                    raise ValueError("gram_blocks: blocks are (n x p) fp64 with unit column stride")
                d.a, d.lda, d.p, d.offset = _hip.ptr(blk), _ld(blk), blk.shape[1], off
                off += blk.shape[1]
            return arr, off
        (ta, p), (tb, q) = table(A_blocks), table(B_blocks)
        need = self._L.ds_gram_workspace_bytes(self.n, p, q)
        if self._gram_ws is None or self._gram_ws.numel() < need:
            self._gram_ws = torch.empty((need,), dtype=torch.uint8, device=self.device)
        G = torch.empty((p, q), dtype=torch.float64, device=self.device)
        _hip.check(self._L.ds_gram64_blocks(len(A_blocks), ctypes.addressof(ta), len(B_blocks), ctypes.addressof(tb), self.n,
                                            int(bool(symmetric)), _hip.ptr(G), _hip.ptr(self._gram_ws), self._gram_ws.numel(),
                                            _hip.stream_ptr()), "ds_gram64_blocks")
        self.counts["gram"] += 1
        return G

    def _scratch(self, key, shape, dtype):
        t = self._tmp.get(key)
        if t is None or t.shape != tuple(shape) or t.dtype != dtype:
            t = torch.empty(tuple(shape), dtype=dtype, device=self.device)
            self._tmp[key] = t
        return t

    def mix(self, A, C, out, alpha=1.0, beta=0.0):
        p, q = C.shape
        if A.shape[1] != p or out.shape[1] != q:
            raise ValueError("mix: shape mismatch")
        C32 = C.to(torch.float32).contiguous()
        pp = _hip.ptr
        _hip.check(self._L.ds_mix(pp(A), _ld(A), p, pp(C32), q, pp(out), _ld(out), self.n, float(alpha),
                                  float(beta), _hip.stream_ptr()), "ds_mix")
        self.counts["mix"] += 1

    def mix64(self, blocks, C, out=None, alpha=1.0, beta=0.0):
        """out <- alpha * [blocks[0] | blocks[1] | ...] C + beta * out in fp64 (ds_mix64): the basis is a LIST of (n x p_i)
        fp64 blocks - never concatenated - and C their stacked (sum p_i) x q coefficients; every block is read once and
        the result written once.  An entry of ``blocks`` is a block - its coefficients are the rows of C that follow the
        previous entry's - or a ``(block, first_row)`` tuple that addresses its rows of C explicitly (rows of C no entry
        names are skipped).  ``out`` must not share memory with a block or with C."""
        C = C.contiguous()
        if C.dtype != torch.float64:
            raise ValueError("mix64: fp64 coefficients")
        q = C.shape[1]
        items, row = [], 0
        for blk in blocks:
            if isinstance(blk, tuple):
                blk, row = blk
            if blk.dtype != torch.float64 or blk.shape[0] != self.n or blk.stride(1) != 1:
                raise ValueError("mix64: blocks are (n x p) fp64 with unit column stride")
            items.append((blk, row))
            row += blk.shape[1]
        if max(r + b.shape[1] for b, r in items) > C.shape[0]:
            raise ValueError("mix64: the blocks need more coefficient rows than C has")
        if out is None:
            out = torch.empty((self.n, q), dtype=torch.float64, device=self.device)
            if beta != 0.0:
                raise ValueError("mix64: beta != 0 needs an out")
        if out.dtype != torch.float64 or out.shape != (self.n, q) or out.stride(1) != 1:
            raise ValueError("mix64: out is (n x q) fp64 with unit column stride")
        pp = _hip.ptr
        nmax = 4  # DS_MIX64_MAX_BLOCKS
        for i0 in range(0, len(items), nmax):
            part = items[i0:i0 + nmax]
            arr = (_hip.Block64 * len(part))()
            for d, (blk, r) in zip(arr, part):
                d.a, d.lda, d.p, d.offset = pp(blk), _ld(blk), blk.shape[1], r
            _hip.check(self._L.ds_mix64(len(part), ctypes.addressof(arr), pp(C), _ld(C), q, pp(out), _ld(out), self.n,
                                        float(alpha), float(beta if i0 == 0 else 1.0), _hip.stream_ptr()), "ds_mix64")
        self.counts["mix64"] += 1
        return out

    # ------------------------------------------------------------------ fp64 refinement: fused element-wise passes
    def residual64(self, KX, MX, X, lam):
        """(||K x_j - lam_j M x_j||^2, ||x_j||^2) of every column of the fp64 blocks in ONE pass (ds_residual64_norms)."""
        b = X.shape[1]
        if b % 2 or b > 512 or any(t.dtype != torch.float64 or t.stride(1) != 1 or (t.data_ptr() | (t.stride(0) * 8)) % 16
                                   for t in (KX, MX, X)):
            R = torch.addcmul(KX, MX, lam[None, :], value=-1.0)
            return (R * R).sum(0), (X * X).sum(0)
        pp = _hip.ptr
        need = self._L.ds_residual64_workspace_doubles(b)
        ws = self._scratch("residual64_ws", (need,), torch.float64)
        out = torch.empty((2, b), dtype=torch.float64, device=self.device)
        lam = lam.to(torch.float64).contiguous()
        _hip.check(self._L.ds_residual64_norms(pp(KX), _ld(KX), pp(MX), _ld(MX), pp(X), _ld(X), pp(lam), self.n, b, pp(ws), need,
                                               pp(out[0]), pp(out[1]), _hip.stream_ptr()), "ds_residual64_norms")
        return out[0], out[1]

    def residual64_scaled(self, KX, MX, lam, scale, idx, out=None):
        """(n x len(idx)) fp32 block of the residual columns ``idx`` of the fp64 blocks, each times ``scale[col]``
        (ds_residual64_scaled): the scaled input of the fp32 preconditioner, without an fp64 residual block in between."""
        nact = int(idx.numel())
        if nact % 4:
            raise ValueError("residual64_scaled: a multiple of 4 columns")
        pp = _hip.ptr
        R = torch.empty((self.n, nact), dtype=torch.float32, device=self.device) if out is None else out
        if R.dtype != torch.float32 or R.shape != (self.n, nact) or R.stride(1) != 1:
            raise ValueError("residual64_scaled: out is (n x len(idx)) fp32 with unit column stride")
        cols = idx.to(torch.int32).contiguous()
        lam, scale = lam.to(torch.float64).contiguous(), scale.to(torch.float64).contiguous()
        _hip.check(self._L.ds_residual64_scaled(pp(KX), _ld(KX), pp(MX), _ld(MX), pp(lam), pp(scale), pp(cols), nact, pp(R), _ld(R),
                                                self.n, _hip.stream_ptr()), "ds_residual64_scaled")
        return R

    def mix_inplace(self, W, T):
        if T.shape[1] <= 160:  # ds_mix reads a row tile completely before writing it
            self.mix(W, T, W)
            return
        tmp = self._scratch("mix_inplace", W.shape, W.dtype)
        self.mix(W, T, tmp)
        W.copy_(tmp)

    # ------------------------------------------------------------------ fused elementwise
    def _residual_ws(self, ncols):
        u = self.sys.groups["union"]
        need = self._L.ds_union_residual_workspace_bytes(u["ngroups"], ncols)
        ws = self._tmp.get("residual_ws")
        if ws is None or ws.numel() < need:
            ws = self._tmp["residual_ws"] = torch.empty((need,), dtype=torch.uint8, device=self.device)
        return ws

    def residual_fused_ok(self, X, R):
        # (operand blocks of 2 GB and more - configs[4]'s basis buffer - take the kernel's per-panel descriptor variant, as every
        # other epilogue does: tests/test_hip_kernels.py::test_union_spmm_operands_beyond_2gb)
        return self.m_kind == 1 and self.mgrp is not None and self._union_ok(X, R, wide=True)

    def residual_fused(self, X, lam, R):
        """R <- K X - (M X) diag(lam) and (||R_j||^2, ||X_j||^2) in ONE walk of the neighbour unions (ds_union_residual): K X
        and M X are never written.  R equals what apply_K + apply_M + residual give bit for bit."""
        b = X.shape[1]
        lam64 = lam.to(torch.float64).contiguous()
        pp = _hip.ptr
        g, u = self.sys.groups, self.sys.groups["union"]
        slices = self.col_slices(b)  # (a block wider than one launch takes: column slices, each with its share of the norms)
        ws = self._residual_ws(max(c1 - c0 for c0, c1 in slices))
        for c0, c1 in slices:
            xs, rs = X[:, c0:c1], R[:, c0:c1]
            _hip.check(self._L.ds_union_residual(self._level_tag, None if u.get("single") else pp(u["utab"]), pp(u["ctab"]), u["ngroups"],
                                                 u["capb"], pp(g["gent"]), pp(self.kgrp), pp(self.mgrp), self.kgrp.shape[0], self.nv,
                                                 pp(xs), _ld(xs), pp(lam64[c0:]), pp(rs), _ld(rs), c1 - c0, pp(ws), ws.numel(),
                                                 pp(self._nrm[0, c0:]), pp(self._nrm[1, c0:]), _hip.stream_ptr()), "ds_union_residual")
        self.counts["apply_K_cols"] += b
        self.counts["apply_M_cols"] += b
        return self._nrm[0, :b].clone(), self._nrm[1, :b].clone()

    def residual(self, R, MX, X, lam, src=None):
        """R <- src - MX diag(lam) (src = K X; None: R holds it already), returns (||R_j||^2, ||X_j||^2) in fp64."""
        b = R.shape[1]
        lam64 = lam.to(torch.float64).contiguous()
        pp = _hip.ptr
        src = R if src is None else src
        _hip.check(self._L.ds_residual(pp(src), _ld(src), pp(R), _ld(R), pp(MX), _ld(MX), pp(X), _ld(X), pp(lam64), self.n, b,
                                       pp(self._nrm[0]), pp(self._nrm[1]), _hip.stream_ptr()), "ds_residual")
        return self._nrm[0, :b].clone(), self._nrm[1, :b].clone()

    def cheb_init(self, R, D, W, c):
        pp = _hip.ptr
        _hip.check(self._L.ds_cheb_init(pp(R), _ld(R), pp(D), _ld(D), pp(W), _ld(W), pp(self.dinv), self.nv,
                                        R.shape[1], float(c), _hip.stream_ptr()), "ds_cheb_init")

    def cheb_step(self, AD, R, D, W, c1, c2):
        pp = _hip.ptr
        _hip.check(self._L.ds_cheb_step(pp(AD), _ld(AD), pp(R), _ld(R), pp(D), _ld(D), pp(W), _ld(W), pp(self.dinv),
                                        self.nv, R.shape[1], float(c1), float(c2), _hip.stream_ptr()), "ds_cheb_step")

    def cheb_spmm(self, Wk, Wprev, R0, c1, c2, first):
        """Wprev <- Wk + c1 (Wk - Wprev) + c2 T (R0 - K Wk): one fused launch per polynomial term."""
        self._cheb_spmm_launch(Wk, Wprev, R0, c1, c2, first)

    def cheb_term_bytes(self, ncols, first=False, elem_bytes=4):
        """ALGORITHMIC bytes of one fused Chebyshev-term launch, SURVEY.md section 8(d)'s BSR-3 count: 9 values and one int32
        column id per block, the row pointers, the block-Jacobi blocks T, and the vector streams - W_k (gathered, counted
        once), R0 and W_{k-1} read (W_{k-1} = 0 is not read when ``first``), W_{k+1} written; ``elem_bytes`` = 2 for the bf16
        blocks of the production preconditioner.  The values count at the width the kernel multiplies with: 2 bytes where
        the term runs on the matrix cores (3x3 blocks rounded to bf16: 18 B of payload per block), else 4.  What the kernels
        actually fetch beyond that - the 24-byte padded block rows and the two table words per union entry of the MFMA form,
        re-gathered panels - is traffic, not algorithm: it shows in the PMC figure beside this one."""
        nnzb = self.colidx.shape[0]
        vec = (3 if first else 4) * self.n * ncols * elem_bytes
        vbytes = 2 if (elem_bytes == 2 and self._mfma is not None and self.kc is not None) else 4
        return nnzb * (9 * vbytes + 4) + (self.nv + 1) * 4 + self.nv * 36 + vec

    def cheb_spmm16(self, Wk, Wprev, R0, c1, c2, first):
        """The fused term on bf16 blocks, in place on W_prev: what the bf16 V-cycle launches (ds_spmm_union16m when the
        level carries the MFMA tables, else ds_spmm_union16)."""
        pp = _hip.ptr
        mt = self._mfma
        if mt is not None and self.kc is not None:
            _hip.check(self._L.ds_spmm_union16m(1, mt["G"], self._level_tag, pp(mt["gptr"]), pp(mt["gcol"]), pp(mt["gmeta"]), pp(mt["gbase"]), pp(mt["ghead"]),
                                                pp(self.kc), self.kc.shape[0], mt["ngroups"], mt["max_entries"], mt["max_batch_blocks"],
                                                self.nv, pp(Wk), _ld(Wk), pp(Wprev), _ld(Wprev), 0, pp(R0), _ld(R0), pp(self.dinv), Wk.shape[1],
                                                float(c1), float(c2), int(bool(first)), None, 0, _hip.stream_ptr()),
                       "ds_spmm_union16m")
            return
        g, u = self.sys.groups, self.sys.groups["union"]
        _hip.check(self._L.ds_spmm_union16(1, None if u.get("single") else pp(u["utab"]), pp(u["ctab"]), u["ngroups"], u["capb"],
                                           pp(g["gent"]), pp(self.kgrp), self.kgrp.shape[0], self.nv, pp(Wk), _ld(Wk), pp(Wprev),
                                           _ld(Wprev), 0, pp(R0), _ld(R0), pp(self.dinv), Wk.shape[1], float(c1), float(c2),
                                           int(bool(first)), None, 0, _hip.stream_ptr()), "ds_spmm_union16")

    # ------------------------------------------------------------------ two-level preconditioner pieces
    coarse = None  # ops of the corner-node level (HipModalOps on an ord-2 mesh sets it)

    def spmm_residual(self, X, R0, Y):
        """Y <- R0 - K X (<= 84 columns, one fused launch)."""
        pp = _hip.ptr
        if self._union_ok(X, Y, R0):
            self._union(2, X, Y, R0)
            self.counts["apply_K_cols"] += X.shape[1]
            return
        _hip.check(self._L.ds_spmm_residual(pp(self.rowptr), pp(self.colidx), pp(self.k32), self.nv, pp(X), _ld(X),
                                            pp(R0), _ld(R0), pp(Y), _ld(Y), X.shape[1], _hip.stream_ptr()),
                   "ds_spmm_residual")
        self.counts["apply_K_cols"] += X.shape[1]

    def _transfer(self, ptr_, col, w, nrows, X, Y, beta):
        pp = _hip.ptr
        _hip.check(self._L.ds_scalar_csr_spmm(pp(ptr_), pp(col), pp(w), nrows, pp(X), _ld(X), pp(Y), _ld(Y),
                                              X.shape[1], float(beta), _hip.stream_ptr()), "ds_scalar_csr_spmm")

    def restrict(self, Rf, Rc):
        """Rc <- P^T Rf (fine block -> corner-node level)."""
        t = self._xfer
        self._transfer(t["rptr"], t["rcol"], t["rw"], self.coarse.nv, Rf, Rc, 0.0)

    def prolong_add(self, Ec, Wf):
        """Wf <- Wf + P Ec."""
        t = self._xfer
        self._transfer(t["pptr"], t["pcol"], t["pw"], self.nv, Ec, Wf, 1.0)

    def prolong(self, Ec, Wf):
        """Wf <- P Ec (nothing of Wf is read: the nested start writes its block straight into the solver's basis buffer)."""
        t = self._xfer
        self._transfer(t["pptr"], t["pcol"], t["pw"], self.nv, Ec, Wf, 0.0)

    def _cheb_spmm_launch(self, Wk, Wprev, R0, c1, c2, first):
        if self._union_ok(Wk, Wprev, R0):
            self._union(1, Wk, Wprev, R0, c1, c2, first)
            self.counts["apply_K_cols"] += Wk.shape[1]
            return
        pp = _hip.ptr
        _hip.check(self._L.ds_cheb_spmm(pp(self.rowptr), pp(self.colidx), pp(self.k32), self.nv, pp(Wk), _ld(Wk),
                                        pp(Wprev), _ld(Wprev), pp(R0), _ld(R0), pp(self.dinv), Wk.shape[1],
                                        float(c1), float(c2), int(bool(first)), _hip.stream_ptr()), "ds_cheb_spmm")
        self.counts["apply_K_cols"] += Wk.shape[1]

    # ------------------------------------------------------------------ fp64 iterates (refinement phase)
    def _spmm64(self, kind, vals, X, out):
        """out (fp64) <- A X for an fp64 block X, in chunks of <= 80 columns (kinds 4 / 5 of ds_spmm_bsr3)."""
        p = _hip.ptr
        if X.dtype != torch.float64 or out.dtype != torch.float64 or X.shape != out.shape:
            raise ValueError("_spmm64: fp64 blocks of equal shape expected")
        for c0 in range(0, X.shape[1], 80):
            c1 = min(X.shape[1], c0 + 80)
            xs, os_ = X[:, c0:c1], out[:, c0:c1]
            _hip.check(self._L.ds_spmm_bsr3(kind + 2, p(self.rowptr), p(self.colidx), p(vals), None, self.nv, p(xs),
                                            _ld(xs), p(os_), _ld(os_), c1 - c0, _hip.stream_ptr()), "ds_spmm_bsr3(f64)")

    _k64 = None
    _k64grp = _m64grp = None  # the combined fp64 K / the fp64 mass scalars in the union tables' group order (ds_spmm_f64_union)

    def combined_k64(self, on):
        """fp64 refinement: while ``on``, ``apply_K64`` multiplies by ONE fp64 block array K = sum c_i K_i, formed on the
        first product (configs[4]: 2.7 GB), instead of one product per term - half the matrix traffic of every K W.
        The caller brackets a phase in which neither the material nor the assembled terms change, and switches it off
        afterwards (the array is released)."""
        self._k64 = False if on else None
        self._k64grp = self._m64grp = None

    def _union64_ok(self, X, out):
        g = getattr(getattr(self, "sys", None), "groups", None)
        return (g is not None and g.get("union") is not None and X.dtype == torch.float64 and out.dtype == torch.float64
                and X.shape == out.shape and X.shape[1] % 4 == 0 and X.stride(1) == 1 and out.stride(1) == 1
                and (X.data_ptr() | (X.stride(0) * 8) | out.data_ptr() | (out.stride(0) * 8)) % 16 == 0)

    def _union64(self, kind, vals_grp, X, out):
        """out (fp64) <- A X on the neighbour-union tables (ds_spmm_f64_union), values in group order; column slices of <= 84."""
        pp = _hip.ptr
        g, u = self.sys.groups, self.sys.groups["union"]
        for c0, c1 in self.col_slices(X.shape[1]):
            xs, os_ = X[:, c0:c1], out[:, c0:c1]
            _hip.check(self._L.ds_spmm_f64_union(kind, 1, None if u.get("single") else pp(u["utab"]), pp(u["ctab"]), u["ngroups"],
                                                 u["capb"], pp(g["gent"]), pp(vals_grp), vals_grp.shape[0], self.nv, pp(xs), _ld(xs),
                                                 pp(os_), _ld(os_), c1 - c0, _hip.stream_ptr()), "ds_spmm_f64_union")

    def apply_K64(self, X, out, terms=False):
        """out <- K X, all fp64 (fp64 block values).  ``terms``: also return the list of the separate K_i X."""
        kterms, _ = self.polish_terms()
        if self._k64 is not None and not terms and len({kt[0] for kt in kterms}) == 1:
            if self._k64 is False:
                k = kterms[0][1] * float(kterms[0][2])
                for _, vals, c in kterms[1:]:
                    k.add_(vals, alpha=float(c))
                if kterms[0][0] == 2 and self._union64_ok(X, out):
                    # round 5: the combined array in the union tables' group order, blocks transposed - the refinement's K W then
                    # walks the unions of 4 rows (ds_spmm_f64_union) instead of gathering every neighbour's panel once per row
                    kp = self.sys.groups["kperm64"]
                    self._k64grp = k[kp].reshape(-1, 3, 3).transpose(1, 2).reshape(-1, 9).contiguous()
                    self._m64grp = None
                    k = True  # (the BSR-order array is not kept beside it)
                self._k64 = k
            if self._k64 is True and self._union64_ok(X, out):
                self._union64(0, self._k64grp, X, out)
                return []
            if self._k64 is True:  # (a block the union kernel does not take: the BSR-order array after all)
                k = kterms[0][1] * float(kterms[0][2])
                for _, vals, c in kterms[1:]:
                    k.add_(vals, alpha=float(c))
                self._k64 = k
            self._spmm64(kterms[0][0], self._k64, X, out)
            return []
        tmp = self._scratch("k64tmp", X.shape, torch.float64)
        parts = []
        out.zero_()
        for kind, vals, c in kterms:
            self._spmm64(kind, vals, X, tmp)
            out.add_(tmp, alpha=float(c))
            if terms:
                parts.append(tmp.clone())
        return parts

    def apply_M64(self, X, out):
        _, (mkind, mvals) = self.polish_terms()
        if self._k64 is not None and self._k64 is not False and mkind == 3 and getattr(self, "_k64grp", None) is not None \
                and self._union64_ok(X, out):
            if self._m64grp is None:  # (inside a combined_k64 phase the assembled terms do not change)
                self._m64grp = mvals[self.sys.groups["kperm64"]].contiguous()
            self._union64(1, self._m64grp, X, out)
            return
        self._spmm64(mkind, mvals, X, out)

    def rigid64(self):
        return None

    # ------------------------------------------------------------------ fp64 polish
    def polish_products(self, X):
        """fp64 Gram matrices of the terms of K and of M on the block X (fp64 values, fp64
        accumulation, fp32 X): returns ([X^T K_i X], [c_i], X^T M X) with K = sum c_i K_i."""
        kterms, (mkind, mvals) = self.polish_terms()
        c = X.shape[1]
        if (len(kterms) == 2 and kterms[0][0] == 2 and kterms[1][0] == 2 and mkind == 3 and X.dtype == torch.float32
                and c % 4 == 0 and c <= 84 and X.stride(1) == 1 and (X.data_ptr() | (X.stride(0) * 4)) % 16 == 0):
            # K_lambda X, K_mu X and M_s X in one walk of the pattern (ds_spmm_f64_polish): one gather of X instead of three;
            # the three results sit side by side in ONE (n x 3c) block, so their Gram products with X are one launch that
            # reads X once (round 4; three launches before)
            # The three results are fp64 blocks.  ``polish_f32_blocks`` (round 5, OFF): the same fp64 sums stored as fp32 blocks - half
            # the bytes written here and read by the Gram launch, which then takes the fp32 matrix-core path (1.10 -> ~0.7 ms per
            # pass at the benchmark size, 1.8 % of a pass).  Built, tested (tests/test_hip_kernels.py) and NOT adopted: the polish
            # then returns eigenvalues and quadratic forms with ~3e-8 of relative noise instead of values accurate to second order
            # in the iteration error (two solves whose fp32 blocks differ in rounding agreed to 3.6e-8 instead of < 1e-9) - inside
            # the stated 1e-4, but a precision cut in the one stage whose job is precision
            f64 = not bool(getattr(self, "polish_f32_blocks", False))
            Y3 = self._scratch("polish3", (X.shape[0], 3 * c), torch.float64 if f64 else torch.float32)
            p = _hip.ptr
            ya, yb, ym = Y3[:, :c], Y3[:, c:2 * c], Y3[:, 2 * c:]
            fn = self._L.ds_spmm_f64_polish if f64 else self._L.ds_spmm_f64_polish_f32out
            _hip.check(fn(p(self.rowptr), p(self.colidx), p(kterms[0][1]), p(kterms[1][1]), p(mvals),
                          self.nv, p(X), _ld(X), p(ya), p(yb), p(ym), 3 * c, c, _hip.stream_ptr()), "ds_spmm_f64_polish")
            G3 = self.gram(X, Y3)
            return ([G3[:, :c].contiguous(), G3[:, c:2 * c].contiguous()], [kterms[0][2], kterms[1][2]],
                    G3[:, 2 * c:].contiguous())
        Y = self._scratch("polish", X.shape, torch.float64)
        GK, coef = [], []
        for kind, vals, c in kterms:
            self._spmm(kind, vals, X, Y)
            GK.append(self.gram(X, Y))
            coef.append(c)
        self._spmm(mkind, mvals, X, Y)
        return GK, coef, self.gram(X, Y)


class HipModalOps(_HipBlockOps):
    """One material hypothesis (lam, mu) on a TetSystem."""

    # nodes per wavefront of the MFMA form of the preconditioner's bf16 terms on (fine level, corner-node level): 8, or 0 =
    # the VALU kernel (ds_spmm_union16).  Both levels since round 3: with 16-entry batches on the fine level and 32-entry
    # batches on levels smaller than the device's wave slots, the corner-node level's term takes 17.7 us instead of 22.1.
    mfma_groups = (8, 8)

    # the level's own fp32 products (K W, M W, M X of the eigensolver) on the matrix cores (ds_spmm_union32m) instead of
    # the VALU neighbour-union kernel.  OFF: built, parity-green and measured in round 4 - at C3 K X takes 263 us against
    # 229 us (M X 246 against 151): v_mfma_f32_16x16x4_f32 runs at the fp32 VECTOR rate and the 16 x 4 tile of 3x3 blocks on
    # a 4-node union is 3/4 x 0.43 full, so the matrix pipe needs 125-150 us for what the VALU does in 40, on top of the
    # LDS staging the form requires (profiles/r04_mfma32_*.txt)
    mfma32 = False

    # the corner-node level's polynomial on the group-block Jacobi (``group_jacobi`` of that level's operator object): 8 or 0
    coarse_group_jacobi = 8
    # the same for the ONE-level polynomial of an ord-1 mesh's operator object (no corner-node level): 8 or 0
    one_level_group_jacobi = 0

    def __init__(self, system: TetSystem, lam, mu, two_level=None, _level=0, mfma_groups=None, mfma32=None, coarse_group_jacobi=None,
                 one_level_group_jacobi=None):
        """two_level: build the corner-node level for the two-level preconditioner (ord-2 meshes; default on)."""
        self.sys = system
        self._level_tag = min(int(_level), 1)
        if coarse_group_jacobi is not None:
            self.coarse_group_jacobi = int(coarse_group_jacobi)
        if one_level_group_jacobi is not None:
            self.one_level_group_jacobi = int(one_level_group_jacobi)
        if self.coarse_group_jacobi not in (0, 8) or self.one_level_group_jacobi not in (0, 8):
            raise ValueError("coarse_group_jacobi / one_level_group_jacobi: groups of 8 nodes, or 0 for the node blocks")
        if mfma_groups is not None:
            self.mfma_groups = tuple(mfma_groups)
        if mfma32 is not None:
            self.mfma32 = bool(mfma32)
        self._init_common(system.rowptr, system.colidx, system.nv, system.device)
        dev = self.device
        self.k32 = torch.empty((system.nnzb, 9), dtype=torch.float32, device=dev)
        self.k32t = torch.empty((system.nnzb, 9), dtype=torch.float32, device=dev)  # blocks transposed
        self.ms32 = torch.empty((system.nnzb,), dtype=torch.float32, device=dev)
        self.dinv = torch.empty((system.nv, 9), dtype=torch.float32, device=dev)
        if two_level is None:
            two_level = True
        if two_level and _level == 0 and system.order == 2:
            lvl = system.coarse_level()
            if lvl is not None:
                self._xfer = lvl
                self.coarse = HipModalOps(lvl["sys"], lam, mu, two_level=False, _level=1, mfma_groups=self.mfma_groups,
                                          mfma32=self.mfma32, coarse_group_jacobi=self.coarse_group_jacobi)
        G = self.mfma_groups[min(_level, 1)]
        if G not in (0, 8):
            raise ValueError("mfma_groups: 8 nodes per wavefront, or 0 for the VALU kernel")
        if G and system.groups is not None and system.nnzb * 24 < 0x7F000000:
            mt = system.mfma_tables(G)
            if mt["max_entries"] <= 256 and mt["max_batch_blocks"] <= MF_BATCH * G:  # what ds_spmm_union16m serves
                self._mfma = mt
                if (((_level == 1 and self.coarse_group_jacobi == G) or
                     (_level == 0 and system.order == 1 and self.one_level_group_jacobi == G)) and system.nv >= 4 * G):
                    md = system.mfma_tables_dense(G)
                    if md["nblocks"] * 24 < 0x7F000000:
                        self._mfma_dense, self.group_jacobi = md, G
        if self.mfma32 and system.groups is not None and system.nnzb * 36 + 16 < 0x7F000000:
            m4 = system.mfma_tables(MF32_G, MF32_BATCH)
            if m4["max_entries"] <= 256 and m4["max_batch_blocks"] <= MF32_BATCH * MF32_G:  # what ds_spmm_union32m serves
                self._mfma32 = m4
        self.set_material(lam, mu)
        self.rigid = self._rigid_basis() if _level == 0 else None
        self._rigid_generation = getattr(system, "geometry_generation", 0)

    def group_T(self, X):
        """T_g X with the group-block Jacobi's blocks, fp32 in and out (ds_group_apply16): the power iteration's 8 columns and the
        Python path of the polynomial; the bf16 cycle applies T_g to its right-hand side inside the native driver."""
        X = X if (X.stride(1) == 1 and (X.data_ptr() | (X.stride(0) * 4)) % 16 == 0) else X.contiguous()
        Y = torch.empty((X.shape[0], X.shape[1]), dtype=torch.float32, device=X.device)
        for c0 in range(0, X.shape[1], 256):
            c1 = min(X.shape[1], c0 + 256)
            _hip.check(self._L.ds_group_apply16(_hip.ptr(self.tgrp), self.group_jacobi, X[:, c0:c1].data_ptr(), 1, _ld(X),
                                                Y[:, c0:c1].data_ptr(), 1, _ld(Y), self.nv, c1 - c0, _hip.stream_ptr()),
                       "ds_group_apply16")
        return Y

    def probe_products(self, G0):
        """(K_lambda G0, K_mu G0, M G0) in fp64 for an fp32 probe block G0 of a multiple of 4 columns - ONE walk of the pattern
        (ds_spmm_f64_polish).  Geometry only: the solver keeps them per geometry generation and forms ||K G0|| of every material
        as ||lam K_lambda G0 + mu K_mu G0||.  None when the operator is not of that two-term form."""
        kterms, (mkind, mvals) = self.polish_terms()
        c = G0.shape[1]
        if not (len(kterms) == 2 and kterms[0][0] == 2 and kterms[1][0] == 2 and mkind == 3 and G0.dtype == torch.float32
                and c % 4 == 0 and c <= 84 and G0.stride(1) == 1 and (G0.data_ptr() | (G0.stride(0) * 4)) % 16 == 0):
            return None
        Y3 = torch.empty((G0.shape[0], 3 * c), dtype=torch.float64, device=self.device)
        p = _hip.ptr
        _hip.check(self._L.ds_spmm_f64_polish(p(self.rowptr), p(self.colidx), p(kterms[0][1]), p(kterms[1][1]), p(mvals), self.nv,
                                              p(G0), _ld(G0), p(Y3[:, :c]), p(Y3[:, c:2 * c]), p(Y3[:, 2 * c:]), 3 * c, c,
                                              _hip.stream_ptr()), "ds_spmm_f64_polish")
        return Y3[:, :c], Y3[:, c:2 * c], Y3[:, 2 * c:]

    def norm_probe_key(self):
        """What the solver's cached norm probe (random block, ||M G0|| / ||G0||) is valid for: this system's geometry."""
        return (id(self.sys), getattr(self.sys, "geometry_generation", 0))

    def set_material(self, lam, mu):
        s = self.sys
        if self.coarse is not None:
            self.coarse.set_material(lam, mu)
        # New coordinates (TetSystem.assemble(vertices)) since the rigid-body basis was formed: the rotations are fields of the
        # coordinates and the basis is M-orthonormal in the OLD mass matrix - re-form it (round 5; until then an operator object that
        # outlived a geometry update deflated the previous geometry's rotations).  The corner-node level forms its basis on demand.
        gen = getattr(s, "geometry_generation", 0)
        regen = getattr(self, "_rigid_generation", gen) != gen
        p = _hip.ptr
        self.lame = (float(lam), float(mu))
        self._k64 = self._k64grp = self._m64grp = None  # (a combined fp64 K array of the previous material must never outlive it)
        _hip.check(self._L.ds_combine_material(p(s.klam), p(s.kmu), p(s.ms), s.nnzb, p(s.diagidx), s.nv,
                                               float(lam), float(mu), p(self.k32), p(self.k32t), p(self.ms32),
                                               p(self.dinv), _hip.stream_ptr()), "ds_combine_material")
        if s.groups is not None:
            if self.kgrp is None:
                self.kgrp = torch.empty((s.nnzb, 9), dtype=torch.float32, device=self.device)
            _hip.check(self._L.ds_pack_groups(p(self.k32t), p(s.groups["kperm"]), s.nnzb, p(self.kgrp),
                                              _hip.stream_ptr()), "ds_pack_groups")
            if self.m_kind == 1:
                self.mgrp = self.ms32[s.groups["kperm64"]].contiguous()  # node-scalar mass values in group order
            if self._mfma is not None:
                if self.kc is None:
                    self.kc = torch.empty((s.nnzb, 3, 4), dtype=torch.bfloat16, device=self.device)
                _hip.check(self._L.ds_pack_kc(p(self.k32), p(self._mfma["kperm"]), s.nnzb, p(self.kc), _hip.stream_ptr()),
                           "ds_pack_kc")
            if self.group_jacobi and self._mfma is not None:
                md = self._mfma_dense
                if self.tgrp is None:
                    ng = md["ngroups"]
                    self.tgrp = torch.empty((ng, 3 * self.group_jacobi, 3 * self.group_jacobi), dtype=torch.float32, device=self.device)
                    self.kc_dense = torch.zeros((md["nblocks"], 3, 4), dtype=torch.bfloat16, device=self.device)
                    self.dinv_id = torch.eye(3, dtype=torch.float32, device=self.device).reshape(1, 9).repeat(s.nv, 1).contiguous()
                _hip.check(self._L.ds_group_inverse(p(s.rowptr), p(s.colidx), p(self.k32), s.nv, self.group_jacobi, p(self.tgrp),
                                                    _hip.stream_ptr()), "ds_group_inverse")
                mt_ = self._mfma
                _hip.check(self._L.ds_group_pack_kc(p(self.k32), p(self.tgrp), p(mt_["gptr"]), p(mt_["gmeta"]), p(mt_["gbase"]),
                                                    p(mt_["kperm"]), self.group_jacobi, s.nv, p(self.kc_dense), _hip.stream_ptr()),
                           "ds_group_pack_kc")
            if self._mfma32 is not None:
                if self.k4 is None:  # (zeros: the 16 bytes of slack behind the last block are read and must be finite)
                    self.k4 = torch.zeros((s.nnzb * 9 + 4,), dtype=torch.float32, device=self.device)
                    self._kperm4 = self._mfma32["kperm"].long()
                _hip.check(self._L.ds_pack_groups(p(self.k32), p(self._mfma32["kperm"]), s.nnzb, p(self.k4),
                                                  _hip.stream_ptr()), "ds_pack_groups")
                if self.m_kind == 1:
                    if self.m4 is None:
                        self.m4 = torch.zeros((s.nnzb + 4,), dtype=torch.float32, device=self.device)
                    torch.index_select(self.ms32, 0, self._kperm4, out=self.m4[:s.nnzb])
        if regen:  # (the new mass values are in place)
            self._rigid_generation = gen
            self.rigid = self._rigid_basis() if self._level_tag == 0 else None

    def _rigid_basis(self):
        """Translations + rotations about the centroid, M-orthonormalised in fp64; stored (n, 8) fp32 with
        two zero pad columns so every kernel sees a multiple of 4 columns."""
        v = self.sys.vertices.double()
        c = v - v.mean(0, keepdim=True)
        Y = torch.zeros((self.n, 8), dtype=torch.float64, device=self.device)
        for a in range(3):
            Y[a::3, a] = 1
        Y[0::3, 3], Y[1::3, 3] = -c[:, 1], c[:, 0]
        Y[1::3, 4], Y[2::3, 4] = -c[:, 2], c[:, 1]
        Y[2::3, 5], Y[0::3, 5] = -c[:, 0], c[:, 2]
        Y32 = Y.float()
        MY = torch.empty((self.n, 8), dtype=torch.float64, device=self.device)
        for _ in range(2):  # second pass removes the fp32 rounding of the first
            self._spmm(3, self.sys.ms, Y32, MY)
            G = self.gram(Y32, MY)[:6, :6]  # (a 6 x n by n x 6 fp64 product takes rocBLAS 24 ms at the benchmark size)
            Lc = torch.linalg.cholesky(0.5 * (G + G.T))
            Y6 = torch.linalg.solve_triangular(Lc, Y32.double()[:, :6].T, upper=False).T
            Y32 = torch.zeros_like(Y32)
            Y32[:, :6] = Y6.float()
        return Y32.contiguous()

    def polish_terms(self):
        lam, mu = self.lame
        return [(2, self.sys.klam, lam), (2, self.sys.kmu, mu)], (3, self.sys.ms)

    def rigid64(self):
        """The six rigid-body modes in fp64 (n x 8, two zero pad columns), M-orthonormal to fp64 accuracy."""
        if self.rigid is None:
            return None
        v = self.sys.vertices.double()
        c = v - v.mean(0, keepdim=True)
        Y = torch.zeros((self.n, 8), dtype=torch.float64, device=self.device)
        for a in range(3):
            Y[a::3, a] = 1
        Y[0::3, 3], Y[1::3, 3] = -c[:, 1], c[:, 0]
        Y[1::3, 4], Y[2::3, 4] = -c[:, 2], c[:, 1]
        Y[2::3, 5], Y[0::3, 5] = -c[:, 0], c[:, 2]
        MY = torch.empty_like(Y)
        for _ in range(2):
            self.apply_M64(Y, MY)
            G = self.gram(Y, MY)[:6, :6]
            Lc = torch.linalg.cholesky(0.5 * (G + G.T))
            Y[:, :6] = torch.linalg.solve_triangular(Lc, Y[:, :6].T, upper=False).T
        return Y


def _coo_to_bsr3(A, nv, pattern=None):
    """torch sparse (COO/CSR) (3nv x 3nv) -> (rowptr, colidx, blocks (nnzb,9) fp64) on A's device,
    on the union pattern ``pattern`` = (rowptr, colidx) if given."""
    A = A.to_sparse_coo().coalesce()
    idx, val = A.indices(), A.values().double()
    bkey = (idx[0] // 3) * nv + (idx[1] // 3)
    if pattern is None:
        keys = torch.unique(bkey)
    else:
        keys = pattern
    slot = torch.searchsorted(keys, bkey)
    if not bool((keys[slot.clamp(max=keys.numel() - 1)] == bkey).all()):
        raise ValueError("sparse matrix has entries outside the common block pattern")
    blocks = torch.zeros((keys.numel(), 9), dtype=torch.float64, device=val.device)
    blocks.view(-1).index_add_(0, slot * 9 + (idx[0] % 3) * 3 + (idx[1] % 3), val)
    return keys, blocks


class HipSparseOps(_HipBlockOps):
    """Generic pencil (A, B) given as torch sparse tensors on the HIP device (the ``lobpcg_func`` entry).
    Both are re-blocked into 3x3 node blocks on their common pattern; no rigid-mode deflation."""

    m_kind = 0

    def __init__(self, A, B):
        _hip.require_gpu(A, B)
        n = B.shape[-1]
        if n % 3 != 0 or tuple(A.shape) != (n, n) or tuple(B.shape) != (n, n):
            raise ValueError("HipSparseOps: A and B must be square with a row count divisible by 3 "
                             "(3 DOFs per node); got {} and {}".format(tuple(A.shape), tuple(B.shape)))
        nv = n // 3
        dev = B.device
        ka, _ = _coo_to_bsr3(A, nv)
        kb, _ = _coo_to_bsr3(B, nv)
        keys = torch.unique(torch.cat([ka, kb, torch.arange(nv, device=dev) * (nv + 1)]))
        _, self.a64 = _coo_to_bsr3(A, nv, keys)
        _, self.b64 = _coo_to_bsr3(B, nv, keys)
        rows = keys // nv
        rowptr = torch.zeros(nv + 1, dtype=torch.int64, device=dev)
        rowptr[1:] = torch.cumsum(torch.bincount(rows, minlength=nv), 0)
        self._init_common(rowptr.to(torch.int32), (keys % nv).to(torch.int32), nv, dev)
        # arbitrary pencils (no deflation, possibly -A for the largest end, no preconditioner): keep the Gram exact
        self.gram_exact = True
        self.k32 = self.a64.float().contiguous()
        self.k32t = self.k32.reshape(-1, 3, 3).transpose(1, 2).reshape(-1, 9).contiguous()
        self.ms32 = self.b64.float().contiguous()
        diag = self.a64[torch.searchsorted(keys, torch.arange(nv, device=dev) * (nv + 1))].reshape(nv, 3, 3)
        eye = torch.eye(3, dtype=torch.float64, device=dev)
        bad = torch.linalg.det(diag).abs() < 1e-300
        diag = torch.where(bad[:, None, None], eye, diag)
        dinv64 = torch.linalg.inv(diag)
        self.dinv = dinv64.float().reshape(nv, 9).contiguous()
        # rigorous bound of lambda_max(T A), T = the inverse diagonal blocks: the block-infinity norm max_i sum_j ||T_i A_ij||_F
        # (an operator norm for the vector norm max_i ||x_i||_2).  The Chebyshev polynomial of lobpcg_func takes it as the end of
        # its interval: a power-iteration estimate that falls short of the true value makes the polynomial blow up on the top of
        # the spectrum, and on pencils that are not FEM matrices 30 steps x 1.2 do fall short (round 6, tests/test_api_gpu.py)
        rows_ = torch.repeat_interleave(torch.arange(nv, device=dev), (rowptr[1:] - rowptr[:-1]))
        ta = torch.linalg.matrix_norm(dinv64[rows_] @ self.a64.reshape(-1, 3, 3))
        self.lmax_bound = float(torch.zeros(nv, dtype=torch.float64, device=dev).index_add_(0, rows_, ta).max())
        self.rigid = None
        self.lame = None

    def polish_terms(self):
        return [(2, self.a64, 1.0)], (2, self.b64)
