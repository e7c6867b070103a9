"""Time the fp64-value read-out products on the benchmark mesh (64 columns)."""
import os, sys
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
X = torch.randn(sysd.n, 64, device=dev); Y = torch.empty((sysd.n, 64), dtype=torch.float64, device=dev)
def tm(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
print(f"[{os.environ.get('DS_SPMM_F64_GENERIC','')}] K_lambda(f64) X: {tm(lambda: ops._spmm(2, sysd.klam, X, Y)):.3f} ms   M(f64) X: {tm(lambda: ops._spmm(3, sysd.ms, X, Y)):.3f} ms   polish_products: {tm(lambda: ops.polish_products(X)):.3f} ms")
