"""Debug aid: ONE fused Chebyshev-term launch, batched vs wave-per-node kernels, error pattern."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from diffsound_amd import meshgen
from diffsound_amd.diffelastic.mesh import TetMesh
from diffsound_amd.modal_ops import TetSystem, HipModalOps
dev = torch.device('cuda')
v, t = meshgen.kuhn_box(int(sys.argv[1])); order = int(sys.argv[2]); ncols = int(sys.argv[3])
mesh = TetMesh(torch.from_numpy(v).to(dev), torch.from_numpy(t).long().to(dev)).to_high_order(order)
sysd = TetSystem(mesh.vertices, mesh.tets, order, 2700.0)
ops = HipModalOps(sysd, 2e10, 2e10, two_level=False)
torch.manual_seed(0)
for scale_w, scale_r, first, c1, c2 in ((1.0, 1e10, False, 0.31, 0.77), (1e-6, 1e4, False, 0.2, 0.25), (1e-6, 1e4, True, 0.0, 0.25), (0.0, 1e4, False, 0.2, 0.25)):
    W = torch.randn(sysd.n, ncols, device=dev) * scale_w
    Wp = torch.randn(sysd.n, ncols, device=dev) * scale_w
    R0 = torch.randn(sysd.n, ncols, device=dev) * scale_r
    a = Wp.clone(); ops.cheb_spmm(W, a, R0, c1, c2, first)
    bt = ops.batches; ops.batches = None
    b = Wp.clone(); ops.cheb_spmm(W, b, R0, c1, c2, first)
    ops.batches = bt
    err = (a - b).abs() / b.abs().max()
    e3 = err.reshape(sysd.nv, 3, ncols)
    print(f"w~{scale_w:g} r~{scale_r:g} first={first}: max rel err {float(err.max()):.2e}; by row mod 3 {[f'{float(e3[:, r].max()):.1e}' for r in range(3)]}; "
          f"by col mod 4 {[f'{float(err[:, c::4].max()):.1e}' for c in range(4)]}; bad nodes {int((e3.amax((1, 2)) > 1e-5).sum())}/{sysd.nv}")
bt = ops.batches.cpu()
bad = (e3.amax((1, 2)) > 1e-5).cpu() if False else None
W = torch.randn(sysd.n, ncols, device=dev); Wp = torch.randn_like(W); R0 = torch.randn_like(W) * 1e10
a = Wp.clone(); ops.cheb_spmm(W, a, R0, 0.31, 0.77, False)
sv = ops.batches; ops.batches = None
b = Wp.clone(); ops.cheb_spmm(W, b, R0, 0.31, 0.77, False); ops.batches = sv
e3 = ((a - b).abs() / b.abs().max()).reshape(sysd.nv, 3, ncols)
badn = torch.nonzero((e3.amax((1, 2)) > 1e-5).cpu()).flatten()
bid = torch.searchsorted(bt[:, 1].contiguous(), badn, right=True)
pos = badn - bt[bid, 0]
nn = (bt[bid, 1] - bt[bid, 0])
rows = (sysd.rowptr[1:] - sysd.rowptr[:-1]).cpu()
import collections
print("position in batch of bad nodes:", collections.Counter(pos.tolist()))
print("nodes in batch (bad):", collections.Counter(nn.tolist()))
print("from end:", collections.Counter((nn - 1 - pos).tolist()))
print("row lengths of bad nodes:", collections.Counter(rows[badn].tolist()))
print("row lengths overall:", collections.Counter(rows.tolist()))
print("kb0 & 3 of bad batches:", collections.Counter((bt[bid, 2] & 3).tolist()), " (rowptr[node]-kb0)&7:", collections.Counter(((sysd.rowptr.cpu()[badn] - bt[bid, 2]) & 7).tolist()))
