"""The bf16 fused Chebyshev term on the matrix cores (fine level and corner-node level, 80 columns) alone on the device, warmed up."""
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
ops = HipModalOps(sysd, 2e10, 2e10)
cases = []
for o, name in ((ops, "fine level"), (ops.coarse, "corner-node level")):
    mk = lambda: torch.randn(o.n, 80, device=dev).bfloat16()
    cases.append((name, o, mk(), mk(), mk()))
t0 = time.time()
while time.time() - t0 < 2.0:
    for name, o, a, b, c in cases: o.cheb_spmm16(a, b, c, 0.3, 0.7, False)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for rnd in range(3):
    for name, o, a, b, c in cases:
        e0.record()
        for _ in range(30): o.cheb_spmm16(a, b, c, 0.3, 0.7, False)
        e1.record(); torch.cuda.synchronize()
        print(f"bf16 term, {name}: {e0.elapsed_time(e1) / 30 * 1e3:.1f} us", flush=True)
