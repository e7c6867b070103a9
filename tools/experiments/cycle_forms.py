"""Experiment (round 6): does the eigensolver need the SYMMETRIC two-level cycle (pre-smoothing, corner-level correction,
post-smoothing: ~6 fine-level products) or does a cheaper form keep the iteration count - post-smoothing only, pre-smoothing only
(3 products each, not symmetric) or the additive form S R + P C P^T R (2 products, symmetric, the two levels independent)?
Python loop, fp32 cycle kernels, the C3 mesh, iteration counts only.    python tools/experiments/cycle_forms.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from diffsound_amd import meshgen
from diffsound_amd.diffelastic.mesh import TetMesh
from diffsound_amd.lobpcg import modal_solver as ms
from diffsound_amd.modal_ops import HipModalOps, TetSystem
from diffsound_amd.diffelastic.diff_model import _lame

dev = torch.device("cuda:0")
v, t = meshgen.kuhn_box(26)
mesh = TetMesh(torch.from_numpy(v).to(dev), torch.from_numpy(t).long().to(dev)).to_high_order(2)
sysd = TetSystem(mesh.vertices, mesh.tets, 2, bench.MAT[0])
for E, nu in ((5e10, 0.25), (7.1e10, 0.40), (2e11, 0.14)):
    lam, mu = (float(x) for x in _lame(E, nu))
    ops = HipModalOps(sysd, lam, mu)
    for form in ("symmetric (native bf16)", "symmetric", "post", "pre", "additive"):
        for sd in ((3,) if "native" in form else (3, 4, 2)):
            cfg = bench.solver_config()
            cfg.native = False
            cfg.smooth_degree = sd
            ms.TwoLevelChebyshev.use_native = "native" in form
            ms.TwoLevelChebyshev.cycle_form = form.split()[0]
            try:
                res = ms.ModalSolver(ops, cfg).solve(64)
                print(f"nu={nu:.2f} {form:24s} smoother degree {sd}: corner {res.coarse_iterations}, fine {res.iterations}, worst {float(res.rerr.max()):.1e}", flush=True)
            except Exception as e:
                print(f"nu={nu:.2f} {form:24s} smoother degree {sd}: {type(e).__name__}: {e}", flush=True)
ms.TwoLevelChebyshev.use_native = True
ms.TwoLevelChebyshev.cycle_form = "symmetric"
