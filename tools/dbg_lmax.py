"""Print the spectral bounds the preconditioner estimates on the benchmark mesh for a few materials."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffsound_amd import meshgen
from diffsound_amd.diffelastic.mesh import TetMesh
from diffsound_amd.modal_ops import TetSystem, HipModalOps
from diffsound_amd.lobpcg.modal_solver import ChebyshevBlockJacobi
dev = torch.device('cuda')
v, t = meshgen.kuhn_box(26)
mesh = TetMesh(torch.from_numpy(v).to(dev), torch.from_numpy(t).long().to(dev)).to_high_order(2)
sysd = TetSystem(mesh.vertices, mesh.tets, 2, 2700.0)
for E, nu in ((5e10, 0.25), (1e10, 0.1), (1e11, 0.4), (3e10, 0.33)):
    lam, mu = E * nu / ((1 + nu) * (1 - 2 * nu)), E / (2 * (1 + nu))
    ops = HipModalOps(sysd, lam, mu)
    f = ChebyshevBlockJacobi(ops, 3, 10.0, 30, 0, 1.0, 0.0)
    c = ChebyshevBlockJacobi(ops.coarse, 24, 400.0, 30, 0, 1.0, 0.0)
    print(f"E={E:g} nu={nu}: lambda_max(T K) fine {f.lmax:.4f} (cap 10), corner-node level {c.lmax:.4f} (cap 4)")
