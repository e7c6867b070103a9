"""Time K X (80 columns) and the fused Chebyshev term on the benchmark mesh; env DS_SPMM_* pick the variant."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffsound_amd import meshgen
from diffsound_amd.diffelastic.mesh import TetMesh
from diffsound_amd.modal_ops import TetSystem, HipModalOps
dev = torch.device('cuda')
v, t = meshgen.kuhn_box(26)
mesh = TetMesh(torch.from_numpy(v).to(dev), torch.from_numpy(t).long().to(dev)).to_high_order(2)
sysd = TetSystem(mesh.vertices, mesh.tets, 2, 2700.0)
ops = HipModalOps(sysd, 2e10, 2e10, two_level=False)
X = torch.randn(sysd.n, 80, device=dev); Y = torch.empty_like(X); Wp = torch.randn_like(X); R0 = torch.randn_like(X)
def tm(fn, reps=30):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
tag = " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("DS_SPMM"))
print(f"[{tag}] K80 {tm(lambda: ops.apply_K(X, Y)):.3f} ms  M80 {tm(lambda: ops.apply_M(X, Y)):.3f} ms  "
      f"fused {tm(lambda: ops._cheb_spmm_launch(X, Wp, R0, 0.3, 0.7, False)):.3f} ms", flush=True)
