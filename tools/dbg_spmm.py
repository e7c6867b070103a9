"""Debug aid: compare the f32 fast-path SpMM (whatever DS_SPMM_* selects) with the fp64-valued kernel path."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffsound_amd import meshgen
from diffsound_amd.diffelastic.mesh import TetMesh
from diffsound_amd.modal_ops import TetSystem, HipModalOps
order = int(sys.argv[2]) if len(sys.argv) > 2 else 2
ncols = int(sys.argv[3]) if len(sys.argv) > 3 else 80
dev = torch.device('cuda')
if sys.argv[1:2] == ["bowl"]:
    import numpy as np
    m = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "g0_bowl_mesh.npz"))
    v, t = m[m.files[0]], m[m.files[1]]
    print("bowl", {k: m[k].shape for k in m.files})
else:
    cells = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    v, t = meshgen.kuhn_box(cells)
mesh = TetMesh(torch.from_numpy(v).to(dev), torch.from_numpy(t).long().to(dev)).to_high_order(order)
sysd = TetSystem(mesh.vertices, mesh.tets, order, 2700.0)
ops = HipModalOps(sysd, 2e10, 2e10, two_level=False)
X = torch.randn(sysd.n, ncols, device=dev)
Y = torch.empty_like(X)
ops._spmm(0, ops.k32, X, Y)
Y64 = torch.empty(sysd.n, ncols, dtype=torch.float64, device=dev)
ops._spmm(2, ops.k32.double(), X, Y64)
err = (Y.double() - Y64).abs() / Y64.abs().max()
print("max rel err", float(err.max()))
print("by column mod 4:", [float(err[:, c::4].max()) for c in range(4)])
print("by row mod 3:", [float(err[r::3].max()) for r in range(3)])
print("by column:", [f"{float(err[:, c].max()):.1e}" for c in range(ncols)])
rows = sysd.rowptr[1:] - sysd.rowptr[:-1]
print("row lengths min/max", int(rows.min()), int(rows.max()))
bad = (err.reshape(sysd.nv, 3, ncols).amax((1, 2)) > 1e-5)
print("bad nodes", int(bad.sum()), "of", sysd.nv, "lengths of bad rows:", sorted(set(rows[bad].tolist()))[:40])
print("lengths of good rows:", sorted(set(rows[~bad].tolist()))[:40])
if ops.batches is not None and int(bad.sum()):
    bt = ops.batches.cpu()
    badn = torch.nonzero(bad.cpu()).flatten()
    bid = torch.searchsorted(bt[:, 1].contiguous(), badn, right=True)
    print("bad nodes (first 20):", badn[:20].tolist(), "their batches:", bid[:20].tolist())
    for b_ in sorted(set(bid[:6].tolist())):
        print("batch", b_, bt[b_].tolist(), "nodes", int(bt[b_, 1] - bt[b_, 0]), "blocks", int(bt[b_, 3] - bt[b_, 2]),
              "position in batch of bad nodes:", [int(x - bt[b_, 0]) for x in badn[bid == b_][:10]])
M = torch.empty_like(X)
ops._spmm(1, ops.ms32, X, M)
M64 = torch.empty_like(Y64)
ops._spmm(3, ops.ms32.double(), X, M64)
print("M max rel err", float(((M.double() - M64).abs() / M64.abs().max()).max()))
