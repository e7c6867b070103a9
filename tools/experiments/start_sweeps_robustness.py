"""Experiment (round 6): are the start sweeps robust to the number of modes / the block width?  Cold solves on the C3 mesh,
native iteration, iteration counts of the corner-node phase / the fine level per variant.
    python tools/experiments/start_sweeps_robustness.py [modes:block ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from diffsound_amd import meshgen
from diffsound_amd.diffelastic.mesh import TetMesh
from diffsound_amd.lobpcg.modal_solver import ModalSolver
from diffsound_amd.modal_ops import HipModalOps, TetSystem
from diffsound_amd.diffelastic.diff_model import _lame

dev = torch.device("cuda:0")
v, t = meshgen.kuhn_box(26)
mesh = TetMesh(torch.from_numpy(v).to(dev), torch.from_numpy(t).long().to(dev)).to_high_order(2)
sysd = TetSystem(mesh.vertices, mesh.tets, 2, bench.MAT[0])
VARIANTS = (("none", {"nested_ritz_tol": 0.0}),
            ("2 sweeps, backward error alone", {"start_sweeps": 2, "nested_ritz_tol": 0.0}),
            ("none, ritz 0.05", {"nested_ritz_tol": 0.05}),
            ("2 sweeps, ritz 0.05", {"start_sweeps": 2, "nested_ritz_tol": 0.05}),
            ("2 sweeps, ritz 0.1", {"start_sweeps": 2, "nested_ritz_tol": 0.1}),
            ("2 sweeps, ritz 0.2", {"start_sweeps": 2, "nested_ritz_tol": 0.2}),
            ("2 sweeps, ritz 0.4", {"start_sweeps": 2, "nested_ritz_tol": 0.4}),
            ("3 sweeps, ritz 0.05", {"start_sweeps": 3, "nested_ritz_tol": 0.05}))
SHAPES = ((16, 24), (32, 40), (64, 72), (64, 80), (128, 136)) if len(sys.argv) < 2 else tuple(tuple(int(x) for x in a.split(":")) for a in sys.argv[1:])
for modes, block in SHAPES:
    for E, nu in ((5e10, 0.25), (7.1e10, 0.40), (2e11, 0.14)):
        lam, mu = (float(x) for x in _lame(E, nu))
        ops = HipModalOps(sysd, lam, mu)
        for name, knobs in VARIANTS:
            cfg = bench.solver_config(block=block, start_sweeps=0)
            for k_, v_ in knobs.items():
                setattr(cfg, k_, v_)
            ModalSolver(ops, cfg).solve(modes)  # (warm-up: allocations)
            ts = []
            for _ in range(3):
                torch.cuda.synchronize()
                t0 = time.time()
                res = ModalSolver(ops, cfg).solve(modes)
                torch.cuda.synchronize()
                ts.append(time.time() - t0)
            print(f"modes {modes:3d} block {block:3d} nu={nu:.2f} {name:30s}: corner {res.coarse_iterations}, fine {res.iterations}, {1e3 * sorted(ts)[1]:6.1f} ms (median of 3), "
                  f"worst {float(res.rerr.max()):.1e}", flush=True)
