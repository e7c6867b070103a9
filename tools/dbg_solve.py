"""Debug aid: eigensolve on the bowl (ord-1) with the batched SpMM enabled per product type."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from diffsound_amd.diffelastic.mesh import TetMesh
from diffsound_amd.modal_ops import TetSystem, HipModalOps, _HipBlockOps
from diffsound_amd.lobpcg.modal_solver import ModalSolver, SolverConfig
dev = torch.device('cuda')
m = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "g0_bowl_mesh.npz"))
v, t = m[m.files[0]], m[m.files[1]]
mesh = TetMesh(torch.from_numpy(v).to(dev), torch.from_numpy(t).long().to(dev)).to_high_order(1)
sysd = TetSystem(mesh.vertices, mesh.tets, 1, 2700.0)
for flags in ("", "K", "M", "C", "KMCR"):
    _HipBlockOps.batch_ops = flags
    ops = HipModalOps(sysd, 2e10, 2e10)
    res = ModalSolver(ops, SolverConfig(lmax_cap=4.0, maxit=40)).solve(32)
    print(f"batched ops '{flags}': iterations {res.iterations}, lam[0..2] {res.eigenvalues[:3].tolist()}, rerr max {float(res.rerr.max()):.2e}", flush=True)
