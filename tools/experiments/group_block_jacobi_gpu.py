"""Experiment (round 6): the group-block Jacobi of profiles/r06_group_block_jacobi_cpu.txt inside the REAL solver - the corner-node
level's polynomial (the V-cycle's corner solve and the preconditioner of the nested start's corner phase) with T = the inverse of the
24 x 24 diagonal block of every group of 8 Morton-consecutive nodes instead of the 3 x 3 node blocks.  Python loop, the polynomial as
torch code on the product's K product (fp32), iteration counts only.     python tools/experiments/group_block_jacobi_gpu.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from diffsound_amd import meshgen
from diffsound_amd.diffelastic.mesh import TetMesh
from diffsound_amd.lobpcg import modal_solver as ms
from diffsound_amd.modal_ops import HipModalOps, TetSystem
from diffsound_amd.diffelastic.diff_model import _lame


class GroupJacobiChebyshev:
    """W = p(T K) T R, T = blockdiag(K_gg^-1) over groups of G consecutive nodes (G = 1: the node blocks that ship)."""

    def __init__(self, ops, degree, ratio, G, lmax=None):
        self.ops, self.degree, self.G = ops, int(degree), int(G)
        s = ops.sys
        nv, dev = s.nv, ops.device
        ng = (nv + G - 1) // G
        rows = torch.repeat_interleave(torch.arange(nv, device=dev), (s.rowptr[1:] - s.rowptr[:-1]).long())
        cols = s.colidx.long()
        sel = (rows // G) == (cols // G)
        Kgg = torch.zeros((ng, 3 * G, 3 * G), dtype=torch.float64, device=dev)
        blk = ops.k32[sel].double().reshape(-1, 3, 3)
        g, a, b = rows[sel] // G, rows[sel] % G, cols[sel] % G
        for i in range(3):
            for j in range(3):
                Kgg[g, 3 * a + i, 3 * b + j] = blk[:, i, j]
        pad = ng * G - nv
        if pad:  # the last group's missing nodes: identity rows
            idx = torch.arange(3 * (G - pad), 3 * G, device=dev)
            Kgg[-1, idx, idx] = 1.0
        self.T = torch.linalg.inv(Kgg).float()
        self.n, self.npad, self.ng = 3 * nv, 3 * ng * G, ng
        self._buf = {}
        if lmax is None:
            x = torch.randn((self.n, 8), device=dev, generator=torch.Generator(device=dev).manual_seed(3))
            y = torch.empty_like(x)
            for _ in range(30):
                ops.apply_K(x, y)
                y = self.applyT(y)
                est = float((y.norm(dim=0) / x.norm(dim=0)).max())
                x = y / y.norm(dim=0)
                y = torch.empty_like(x)
            lmax = min(1.05 * est, 4.0)
        self.lmax, self.lmin = float(lmax), float(lmax) / float(ratio)

    def applyT(self, X):
        b = X.shape[1]
        Xp = X if self.npad == self.n else torch.cat([X, X.new_zeros((self.npad - self.n, b))], 0)
        Y = torch.bmm(self.T, Xp.reshape(self.ng, 3 * self.G, b)).reshape(self.npad, b)
        return Y[:self.n].contiguous()

    def apply(self, R, W, from_guess=False):
        assert not from_guess
        K = self.ops.apply_K
        theta, delta = (self.lmax + self.lmin) / 2, (self.lmax - self.lmin) / 2
        z = self.applyT(R.contiguous())
        x = z / theta
        d = x.clone()
        sigma = theta / delta
        rho_old = 1.0 / sigma
        kx = torch.empty_like(x)
        for _ in range(self.degree - 1):
            K(x, kx)
            r = z - self.applyT(kx)
            rho_new = 1.0 / (2 * sigma - rho_old)
            d = rho_new * rho_old * d + 2 * rho_new / delta * r
            x = x + d
            rho_old = rho_new
        W.copy_(x)


dev = torch.device("cuda:0")
v, t = meshgen.kuhn_box(26)
mesh = TetMesh(torch.from_numpy(v).to(dev), torch.from_numpy(t).long().to(dev)).to_high_order(2)
sysd = TetSystem(mesh.vertices, mesh.tets, 2, bench.MAT[0])
VARIANTS = (("shipped kernels, node blocks (22, 350)", None), ("torch polynomial, node blocks (22, 350)", (22, 350.0, 1)),
            ("groups of 8 (22, 350)", (22, 350.0, 8)), ("groups of 8 (16, 350)", (16, 350.0, 8)), ("groups of 8 (16, 200)", (16, 200.0, 8)),
            ("groups of 8 (14, 150)", (14, 150.0, 8)), ("groups of 8 (12, 100)", (12, 100.0, 8)), ("node blocks (16, 200)", (16, 200.0, 1)),
            ("groups of 16 (14, 150)", (14, 150.0, 16)))
orig_init = ms.TwoLevelChebyshev.__init__
for E, nu in ((5e10, 0.25), (7.1e10, 0.40), (2e11, 0.14)):
    lam, mu = (float(x) for x in _lame(E, nu))
    ops = HipModalOps(sysd, lam, mu)
    for name, var in VARIANTS:
        cfg = bench.solver_config()
        cfg.native = False
        ms.TwoLevelChebyshev.use_native = False
        if var is not None:
            degree, ratio, G = var
            cfg.coarse_degree = cfg.nested_cheb_degree = degree
            cfg.coarse_ratio = cfg.nested_cheb_ratio = ratio
            gj = GroupJacobiChebyshev(ops.coarse, degree, ratio, G)

            def patched(self, ops_, cfg_, gj=gj):
                orig_init(self, ops_, cfg_)
                self.coarse = gj
            ms.TwoLevelChebyshev.__init__ = patched
        else:
            ms.TwoLevelChebyshev.__init__ = orig_init
        res = ms.ModalSolver(ops, cfg).solve(64)
        print(f"nu={nu:.2f} {name:40s}: corner {res.coarse_iterations}, fine {res.iterations}, worst {float(res.rerr.max()):.1e}"
              + (f"  lmax(T K) {gj.lmax:.3f}" if var is not None else ""), flush=True)
ms.TwoLevelChebyshev.__init__ = orig_init
ms.TwoLevelChebyshev.use_native = True
