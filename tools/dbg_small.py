"""Debug aid: cube3 ord-1 (64 nodes, caller's numbering), K X and M X with few columns, against the fp64-valued kernels."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffsound_amd import meshgen
from diffsound_amd.diffelastic.mesh import TetMesh
from diffsound_amd.modal_ops import TetSystem, HipModalOps
dev = torch.device('cuda')
v, t = meshgen.kuhn_box(3)
mesh = TetMesh(torch.from_numpy(v).to(dev), torch.from_numpy(t).long().to(dev)).to_high_order(1)
sysd = TetSystem(mesh.vertices, mesh.tets, 1, 2700.0, reorder=False)
ops = HipModalOps(sysd, 2e10, 2e10)
print("batches:", ops.batches.cpu().tolist())
print("rowptr:", sysd.rowptr.cpu().tolist())
k64 = ops.k32.double(); m64 = ops.ms32.double()
for rep in range(int(os.environ.get("REPS", "3"))):
    for ncols in (8, 24, 4):
        X = torch.randn(sysd.n, ncols, device=dev)
        for name, kind, vals in (("K", 2, k64), ("M", 3, m64)):
            ref = torch.empty(sysd.n, ncols, dtype=torch.float64, device=dev); ops._spmm(kind, vals, X, ref)
            out = torch.full((sysd.n, ncols), 7.0, device=dev)
            (ops.apply_K if name == "K" else ops.apply_M)(X, out)
            e = ((out.double() - ref).abs() / ref.abs().max()).reshape(sysd.nv, -1).amax(1)
            bad = torch.nonzero(e > 1e-5).flatten().tolist()
            zero = torch.nonzero((out.reshape(sysd.nv, -1) == 0).all(1)).flatten().tolist()
            if zero: print(f"rep {rep} {name} ncols {ncols}: ALL-ZERO nodes {zero}", flush=True)
            print(f"rep {rep} {name} ncols {ncols}: max err {float(e.max()):.2e} bad nodes {bad[:24]}"
                  + (f" out[bad0]={out.reshape(sysd.nv, -1)[bad[0]][:6].tolist()}" if bad else ""), flush=True)
