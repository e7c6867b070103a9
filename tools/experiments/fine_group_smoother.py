"""Experiment (round 6): would the FINE level's smoother drop a term on the group-block Jacobi?  The two-level cycle in its Python
form with the fine smoother's T as the inverse of the 24 x 24 diagonal blocks of the fine level's 8-node groups (torch polynomial on
the fp32 product; ds_group_inverse on the fine level's blocks), against the node blocks, per smoother degree.  Iteration counts only,
C3 mesh, 64 modes, block 80, the corner level as shipped (group blocks).    python tools/experiments/fine_group_smoother.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from diffsound_amd import meshgen, _hip
from diffsound_amd.diffelastic.mesh import TetMesh
from diffsound_amd.lobpcg import modal_solver as ms
from diffsound_amd.modal_ops import HipModalOps, TetSystem
from diffsound_amd.diffelastic.diff_model import _lame

dev = torch.device("cuda:0")
v, t = meshgen.kuhn_box(26)
mesh = TetMesh(torch.from_numpy(v).to(dev), torch.from_numpy(t).long().to(dev)).to_high_order(2)
sysd = TetSystem(mesh.vertices, mesh.tets, 2, bench.MAT[0])


class GroupSmoother(ms.ChebyshevBlockJacobi):
    """ChebyshevBlockJacobi on the fine level with T = group blocks, torch polynomial, from a zero guess or from a guess."""

    def _apply_group(self, R, W, from_guess):
        ops = self.ops
        if from_guess:  # W <- W_0 + p(T K) T (R - K W_0), p with one term more (as the shipped iteration from a guess)
            kw = torch.empty_like(W)
            ops.apply_K(W, kw)
            r0 = R - kw
            e = torch.empty_like(W)
            self.degree += 1
            try:
                super()._apply_group(r0, e, False)
            finally:
                self.degree -= 1
            W += e
            return
        super()._apply_group(R, W, False)


for E, nu in ((5e10, 0.25), (7.1e10, 0.40), (2e11, 0.14)):
    lam, mu = (float(x) for x in _lame(E, nu))
    ops = HipModalOps(sysd, lam, mu)
    s = ops.sys
    ng = (s.nv + 7) // 8
    tg = torch.empty((ng, 24, 24), dtype=torch.float32, device=dev)
    _hip.check(_hip.lib().ds_group_inverse(_hip.ptr(s.rowptr), _hip.ptr(s.colidx), _hip.ptr(ops.k32), s.nv, 8, _hip.ptr(tg), _hip.stream_ptr()), "ds_group_inverse")
    orig_init = ms.TwoLevelChebyshev.__init__
    for name, grp, sd, sr in (("node blocks, degree 3 (shipped)", 0, 3, None), ("node blocks, degree 2", 0, 2, None), ("group blocks, degree 3", 8, 3, None),
                              ("group blocks, degree 2", 8, 2, None), ("group blocks, degree 2, ratio / 2", 8, 2, 0.5), ("group blocks, degree 1", 8, 1, None)):
        cfg = bench.solver_config()
        cfg.native = False
        cfg.smooth_degree = sd
        if sr:
            cfg.smooth_ratio = cfg.smooth_ratio * sr
        ms.TwoLevelChebyshev.use_native = False
        ops.group_jacobi, ops.tgrp = grp, (tg if grp else None)

        def patched(self, ops_, cfg_):
            orig_init(self, ops_, cfg_)
            if ops_.group_jacobi:
                args = (cfg_.power_iters, cfg_.seed, cfg_.lmax_safety)
                self.smooth = GroupSmoother(ops_, cfg_.smooth_degree, cfg_.smooth_ratio, *args, cap=cfg_.lmax_cap)
                self.lmax = self.smooth.lmax
        ms.TwoLevelChebyshev.__init__ = patched
        try:
            ops._power_block = None
            res = ms.ModalSolver(ops, cfg).solve(64)
            print(f"nu={nu:.2f} {name:36s}: corner {res.coarse_iterations}, fine {res.iterations}, worst {float(res.rerr.max()):.1e}", flush=True)
        except Exception as e:
            print(f"nu={nu:.2f} {name:36s}: {type(e).__name__}: {e}", flush=True)
        finally:
            ms.TwoLevelChebyshev.__init__ = orig_init
            ops.group_jacobi, ops.tgrp = 0, None
ms.TwoLevelChebyshev.use_native = True
