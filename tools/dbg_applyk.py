"""Debug aid: K X through the batched kernel and through ds_spmm_bsr3's fast path, each against the fp64-valued kernel,
on the views the solver uses."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from diffsound_amd.diffelastic.mesh import TetMesh
from diffsound_amd.modal_ops import TetSystem, HipModalOps
dev = torch.device('cuda')
m = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "g0_bowl_mesh.npz"))
v, t = m[m.files[0]], m[m.files[1]]
mesh = TetMesh(torch.from_numpy(v).to(dev), torch.from_numpy(t).long().to(dev)).to_high_order(1)
sysd = TetSystem(mesh.vertices, mesh.tets, 1, 2700.0)
ops = HipModalOps(sysd, 2e10, 2e10)
n, b = sysd.n, 40
S = torch.randn(n, 8 + 3 * b, device=dev); KS = torch.zeros(n, 3 * b, device=dev)
k64 = ops.k32.double()
def check(X, out, label):
    ref = torch.empty(n, X.shape[1], dtype=torch.float64, device=dev)
    Xc = X.contiguous()
    ops._spmm(2, k64, Xc, ref)
    res = []
    for mode in ("batched", "bsr3"):
        bt = ops.batches
        if mode == "bsr3": ops.batches = None
        out.zero_(); ops.apply_K(X, out); a = out.double().clone(); ops.batches = bt
        res.append(float((a - ref).abs().max() / ref.abs().max()))
    print(f"{label:34s} batched err {res[0]:.2e}   bsr3 fast path err {res[1]:.2e}", flush=True)
for rep in range(2):
    check(S[:, 8:48], KS[:, :40], "X=S[:,8:48] -> KS[:,:40]")
    check(S[:, 88:128], KS[:, 80:120], "W=S[:,88:128] -> KS[:,80:120]")
    check(S[:, 88:124], KS[:, 76:112], "W=S[:,88:124] -> KS[:,76:112]")
    G0 = torch.randn(n, 8, device=dev); G1 = torch.empty_like(G0); check(G0, G1, "8 cols contiguous")
    X12 = torch.randn(n, 12, device=dev); Y12 = torch.empty_like(X12); check(X12, Y12, "12 cols")
    X4 = torch.randn(n, 4, device=dev); Y4 = torch.empty_like(X4); check(X4, Y4, "4 cols")
