"""cProfile of the host side of fwd+bwd passes (one lane) on the benchmark mesh."""
import cProfile, os, pstats, sys, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from diffsound_amd import meshgen
from diffsound_amd.diffelastic.mesh import TetMesh
from diffsound_amd.lobpcg.modal_solver import SolverConfig
from diffsound_amd.pipeline import ModalPipeline
MAT = (2700.0, 5e10, 0.25, 6.0, 1e-7)
dev = torch.device("cuda")
v, t = meshgen.kuhn_box(26)
mesh = TetMesh(torch.from_numpy(v).to(dev), torch.from_numpy(t).long().to(dev)).to_high_order(2)
cfg = SolverConfig(block=80, cheb_degree=48, cheb_ratio=800.0, lmax_cap=10.0)
pipe = ModalPipeline(mesh.vertices, mesh.tets, 2, 64, MAT, solver_config=cfg)
pipe.assemble(); _, _, a0 = pipe.run_pass(MAT[1], MAT[2], backward=False); pipe.set_target(a0)
pipe.run_pass(6e10, 0.3)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
import time; t0 = time.time()
for E, nu in ((4e10, 0.2), (7e10, 0.33), (3e10, 0.28)):
    pipe.assemble(); pipe.run_pass(E, nu)
torch.cuda.synchronize(); dt = time.time() - t0
pr.disable()
print(f"3 passes: {dt*1e3/3:.1f} ms per pass")
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(70); print(s.getvalue()[:14000])
