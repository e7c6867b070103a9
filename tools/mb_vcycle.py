"""One application of the two-level V-cycle (the eigensolver's preconditioner, ~45 launches issued by ONE native call) on the
benchmark mesh, 80 columns: wall time per application by HIP events, unprofiled - to hold against the sum of its kernels'
durations from a rocprofv3 --kernel-trace --stats run of this same script (python tools/mb_vcycle.py; launch gaps = the difference)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffsound_amd import meshgen
from diffsound_amd.diffelastic.mesh import TetMesh
from diffsound_amd.lobpcg.modal_solver import SolverConfig, TwoLevelChebyshev
from diffsound_amd.modal_ops import HipModalOps, TetSystem

dev = torch.device("cuda")
v, t = meshgen.kuhn_box(26)
mesh = TetMesh(torch.from_numpy(v).to(dev), torch.from_numpy(t).long().to(dev)).to_high_order(2)
sysd = TetSystem(mesh.vertices, mesh.tets, 2, 2700.0)
ops = HipModalOps(sysd, 2e10, 2e10)
cfg = SolverConfig(smooth_degree=3, smooth_ratio=10.0, coarse_degree=22, coarse_ratio=350.0)
pre = TwoLevelChebyshev(ops, cfg)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
for c in (80, 32):
    R, W = torch.randn(sysd.n, c, device=dev), torch.empty(sysd.n, c, device=dev)
    for _ in range(5):
        pre.apply(R, W)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        pre.apply(R, W)
    e1.record()
    torch.cuda.synchronize()
    print(f"V-cycle on {c} columns: {e0.elapsed_time(e1) / reps * 1e3:.1f} us per application ({reps} back to back)", flush=True)
