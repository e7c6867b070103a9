"""Experiment: nested iteration - solve the corner-node (P1) level first, prolongate its block as the start of the
fine solve.  Prints iteration counts and wall times against the cold random start (C3 mesh, bench settings)."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import bench
from diffsound_amd import meshgen
from diffsound_amd.diffelastic.mesh import TetMesh
from diffsound_amd.lobpcg.modal_solver import ModalSolver, SolverConfig
from diffsound_amd.modal_ops import HipModalOps, TetSystem
from oracle import fem

dev = torch.device("cuda:0")
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 26
v, t = meshgen.kuhn_box(cells)
mesh = TetMesh(torch.from_numpy(v).to(dev), torch.from_numpy(t).long().to(dev)).to_high_order(2)
sysd = TetSystem(mesh.vertices, mesh.tets, 2, 2700.0)
lam, mu = fem.lame(6.3e10, 0.31)
ops = HipModalOps(sysd, lam, mu)
cfg = bench.solver_config()
def timed(fn):
    torch.cuda.synchronize(); t0 = time.time(); r = fn(); torch.cuda.synchronize(); return r, time.time() - t0
res, dt = timed(lambda: ModalSolver(ops, cfg).solve(64))
res, dt = timed(lambda: ModalSolver(ops, cfg).solve(64))
print(f"cold: {res.iterations} iterations, {dt*1e3:.1f} ms")
csys = ops._xfer["sys"]
for deg, ratio, tolc in ((28, 550.0, 1e-3), (28, 550.0, 1e-4), (16, 200.0, 1e-3), (28, 550.0, 2e-6)):
    cops = HipModalOps(csys, lam, mu, two_level=False)
    ccfg = SolverConfig(block=80, cheb_degree=deg, cheb_ratio=ratio, lmax_cap=4.0, tol=tolc)
    for rep in range(2):
        rc, dtc = timed(lambda: ModalSolver(cops, ccfg).solve(64))
    Xc = rc.block_vectors.contiguous()
    X0 = torch.zeros((ops.n, Xc.shape[1]), device=dev)
    ops.prolong_add(Xc, X0)
    r2, dt2 = timed(lambda: ModalSolver(ops, cfg).solve(64, X0=X0))
    err = float((r2.eigenvalues / res.eigenvalues - 1).abs().max())
    ec = float((rc.eigenvalues / res.eigenvalues - 1).abs().max())
    print(f"coarse cheb({deg},{ratio:g}) tol {tolc:g}: {rc.iterations} it {dtc*1e3:.1f} ms (eig err vs fine {ec:.2e}); "
          f"fine from prolongated start: {r2.iterations} it {dt2*1e3:.1f} ms; total {1e3*(dtc+dt2):.1f} ms; eig diff {err:.1e}")
