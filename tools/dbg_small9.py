"""Diagnostic build (make EXPERIMENTAL=1, DS_SPMM_DBG=9): dump what the batched kernel's node loop sees on failing launches."""
import os, sys
os.environ["DS_SPMM_DBG"] = "9"; os.environ["DS_SPMM_BATCHED"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffsound_amd import meshgen, _hip
from diffsound_amd.diffelastic.mesh import TetMesh
from diffsound_amd.modal_ops import TetSystem, HipModalOps, _ld
dev = torch.device('cuda')
v, t = meshgen.kuhn_box(3)
mesh = TetMesh(torch.from_numpy(v).to(dev), torch.from_numpy(t).long().to(dev)).to_high_order(1)
sysd = TetSystem(mesh.vertices, mesh.tets, 1, 2700.0, reorder=False)
ops = HipModalOps(sysd, 2e10, 2e10)
bt = ops.batches; pp = _hip.ptr
k64 = ops.k32.double()
names = "ccnt crs id0 id1 x0 x1 x2 x3 c0 c1 c2 c3 tot acc0 acc1 acc2".split()
good = None
for rep in range(40):
    torch.manual_seed(0); X = torch.randn(sysd.n, 8, device=dev); out = torch.full((sysd.n, 8), 7.0, device=dev)
    ref = torch.empty(sysd.n, 8, dtype=torch.float64, device=dev); ops._spmm(2, k64, X, ref)
    dbg = torch.full((sysd.nv, 16), -777, dtype=torch.int32, device=dev)
    _hip.check(ops._L.ds_spmm_batched(0, 0, pp(bt), bt.shape[0], pp(ops.rowptr), pp(ops.colidx), pp(ops.k32t), ops.k32t.shape[0],
                                      ops.nv, pp(X), _ld(X), pp(out), _ld(out), None, 0, pp(dbg), 8, 0.0, 0.0, 0, _hip.stream_ptr()), "dbg")
    e = ((out.double() - ref).abs() / ref.abs().max()).reshape(sysd.nv, -1).amax(1)
    bad = torch.nonzero(e > 1e-5).flatten().tolist()
    d = dbg.cpu()
    if not bad and good is None:
        good = d.clone()
    if bad:
        print(f"rep {rep}: bad nodes {bad}")
        for n_ in bad[:4]:
            f = lambda row: {k: (int(x) if i < 4 else float(torch.tensor(int(x), dtype=torch.int32).view(torch.float32))) for i, (k, x) in enumerate(zip(names, row))}
            print("   bad ", f(d[n_]))
            if good is not None: print("   good", f(good[n_]))
            print("   X rows of id0/id1 (lane 0 = row 0, cols 0):", float(X[3 * int(d[n_][2]), 0]), float(X[3 * int(d[n_][3]), 0]))
        break
else:
    print("no failure in 40 reps")
