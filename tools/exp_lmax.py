"""Experiment: power-iteration estimates of lambda_max(T K) on the C3 mesh against the rigorous caps."""
import sys; sys.path.insert(0, ".")
import torch, bench
from diffsound_amd import meshgen
from diffsound_amd.diffelastic.mesh import TetMesh
from diffsound_amd.lobpcg.modal_solver import TwoLevelChebyshev
from diffsound_amd.modal_ops import HipModalOps, TetSystem
from oracle import fem
dev = torch.device("cuda:0")
v, t = meshgen.kuhn_box(26)
mesh = TetMesh(torch.from_numpy(v).to(dev), torch.from_numpy(t).long().to(dev)).to_high_order(2)
sysd = TetSystem(mesh.vertices, mesh.tets, 2, 2700.0)
for E, nu in ((5e10, .25), (9e10, .38), (2e10, .12)):
    lam, mu = fem.lame(E, nu)
    ops = HipModalOps(sysd, lam, mu)
    cfg = bench.solver_config(); cfg.lmax_cap = 0.0
    p = TwoLevelChebyshev(ops, cfg)
    print(E, nu, "fine lmax (x1.2 safety)", p.smooth.lmax, "coarse", p.coarse.lmax)
