"""The fused residual (ds_union_residual) against the three launches it replaces, C3, 80 columns, on the solver's operand layout;
interleaved, warmed up."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffsound_amd import meshgen
from diffsound_amd.diffelastic.mesh import TetMesh
from diffsound_amd.modal_ops import HipModalOps, TetSystem
dev = torch.device("cuda")
v, t = meshgen.kuhn_box(26)
mesh = TetMesh(torch.from_numpy(v).to(dev), torch.from_numpy(t).long().to(dev)).to_high_order(2)
sysd = TetSystem(mesh.vertices, mesh.tets, 2, 2700.0)
ops = HipModalOps(sysd, 2e10, 2e10, two_level=False, mfma_groups=(0, 0))
n = sysd.n
S, KS = torch.randn(n, 256, device=dev), torch.empty(n, 256, device=dev)
X, KX = S[:, 8:88], KS[:, :80]
MX, R = torch.empty(n, 80, device=dev), torch.empty(n, 80, device=dev)
lam = torch.rand(80, device=dev, dtype=torch.float64) * 1e9
def three():
    ops._union(0, X, KX); ops._union(3, X, MX); ops.residual(R, MX, X, lam, src=KX)
def one():
    ops.residual_fused(X, lam, R)
t0 = time.time()
while time.time() - t0 < 2.0:
    three(); one()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for rnd in range(3):
    for name, fn in (("K X + M X + residual (three launches)", three), ("fused residual (one walk)", one)):
        fn(); e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        print(f"{name}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us", flush=True)
