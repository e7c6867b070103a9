"""Experiment (round 6): iteration counts of a cold benchmark pass when Rayleigh-Ritz steps are traded for preconditioner sweeps -
start_sweeps (inverse-power steps on the random start block, corner-node level), precond_sweeps on the fine / corner level.  Python
loop (the native driver knows neither): counts only, not times.   python tools/experiments/sweeps_vs_ritz_steps.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from diffsound_amd import meshgen
from diffsound_amd.diffelastic.mesh import TetMesh
from diffsound_amd.lobpcg.modal_solver import ModalSolver
from diffsound_amd.modal_ops import HipModalOps, TetSystem
from diffsound_amd.diffelastic.diff_model import _lame

dev = torch.device("cuda:0")
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 26
v, t = meshgen.kuhn_box(cells)
mesh = TetMesh(torch.from_numpy(v).to(dev), torch.from_numpy(t).long().to(dev)).to_high_order(2)
sysd = TetSystem(mesh.vertices, mesh.tets, 2, bench.MAT[0])
for E, nu in ((5e10, 0.25), (2.9e10, 0.14), (7.1e10, 0.40)):
    lam, mu = (float(x) for x in _lame(E, nu))
    ops = HipModalOps(sysd, lam, mu)
    for knobs in ({}, {"start_sweeps": 1}, {"start_sweeps": 2}, {"precond_sweeps": 2}, {"start_sweeps": 1, "precond_sweeps": 2},
                  {"start_sweeps": 2, "nested_precond_sweeps": 2}, {"start_sweeps": 2, "nested_precond_sweeps": 2, "precond_sweeps": 2}):
        cfg = bench.solver_config(native=0)
        cfg.native = False
        for k_, v_ in knobs.items():
            setattr(cfg, k_, v_)
        s = ModalSolver(ops, cfg)
        res = s.solve(64)
        print(f"E={E:.2e} nu={nu:.2f} {knobs}: corner-level its {res.coarse_iterations}, fine its {res.iterations}, worst rel {float(res.rerr.max()):.2e}, "
              f"history {' '.join(f'{h[1]:.1e}' for h in res.history)}", flush=True)
