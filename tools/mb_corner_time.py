"""Time K X / fused Chebyshev term (80 and 40 columns) on the corner-node level of the benchmark mesh (27^3 nodes, ord-1)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffsound_amd import meshgen
from diffsound_amd.diffelastic.mesh import TetMesh
from diffsound_amd.modal_ops import TetSystem, HipModalOps
dev = torch.device('cuda')
if len(sys.argv) > 1:
    HipModalOps.mfma_groups = tuple(int(x) for x in sys.argv[1].split(','))
v, t = meshgen.kuhn_box(26)
mesh = TetMesh(torch.from_numpy(v).to(dev), torch.from_numpy(t).long().to(dev)).to_high_order(2)
fine = TetSystem(mesh.vertices, mesh.tets, 2, 2700.0)
ops = HipModalOps(fine, 2e10, 2e10, two_level=True).coarse
sysd = ops.sys
def tm(fn, reps=200):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
tag = "mfma_groups=" + str(HipModalOps.mfma_groups)
out = [f"[{tag}] nv={sysd.nv} nnzb={sysd.nnzb}"]
for c in (80, 40):
    X = torch.randn(sysd.n, c, device=dev); Y = torch.empty_like(X); Wp = torch.randn_like(X); R0 = torch.randn_like(X)
    out.append(f"c={c}: K {tm(lambda: ops.apply_K(X, Y)):.1f} us  fused {tm(lambda: ops._cheb_spmm_launch(X, Wp, R0, 0.3, 0.7, False)):.1f} us  "
               f"bytes(fused) {ops.cheb_term_bytes(c)/1e6:.1f} MB")
Xb, Wb, Rb = (torch.randn(sysd.n, 80, device=dev).bfloat16() for _ in range(3))
out.append(f"bf16 fused term (80 columns): {tm(lambda: ops.cheb_spmm16(Xb, Wb, Rb, 0.3, 0.7, False)):.1f} us")
print("  ".join(out), flush=True)
