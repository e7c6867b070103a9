"""Check (EXPERIMENTAL build, DS_SPMM_UNION=1) the neighbour-union SpMM against the production kernels."""
import os, sys
os.environ["DS_SPMM_UNION"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from diffsound_amd import meshgen
from diffsound_amd.diffelastic.mesh import TetMesh
from diffsound_amd.modal_ops import TetSystem, HipModalOps
dev = torch.device('cuda')
def rel(a, b): return float((a - b).abs().max() / b.abs().max())
for name, order, ncols in (("3", 1, 8), ("3", 2, 80), ("bowl", 1, 40), ("6", 2, 72), ("6", 2, 80)):
    if name == "bowl":
        m = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "g0_bowl_mesh.npz"))
        v, t = m[m.files[0]], m[m.files[1]]
    else:
        v, t = meshgen.kuhn_box(int(name))
    mesh = TetMesh(torch.from_numpy(v).to(dev), torch.from_numpy(t).long().to(dev)).to_high_order(order)
    sysd = TetSystem(mesh.vertices, mesh.tets, order, 2700.0)
    ops = HipModalOps(sysd, 2e10, 2e10, two_level=False)
    assert sysd.groups is not None and sysd.groups["union"] is not None, "union tables missing"
    u = sysd.groups["union"]
    big = torch.randn(sysd.n, ncols + 16, device=dev)
    X = big[:, 8:8 + ncols]
    Wp = torch.randn(sysd.n, ncols, device=dev); R0 = torch.randn(sysd.n, ncols, device=dev) * 1e10
    res = {}
    for mode in ("union", "prod"):
        if mode == "prod": sysd.groups["union"] = None
        Y = torch.zeros(sysd.n, ncols, device=dev); ops.apply_K(X, Y)
        a = Wp.clone(); ops.cheb_spmm(X, a, R0, 0.31, 0.77, False)
        b = Wp.clone(); ops.cheb_spmm(X, b, R0, 0.0, 0.5, True)
        c = torch.zeros(sysd.n, ncols, device=dev); ops.spmm_residual(X, R0, c)
        res[mode] = (Y, a, b, c)
    sysd.groups["union"] = u
    print(f"mesh {name} ord {order} ncols {ncols} capb {u['capb']}: K {rel(res['union'][0], res['prod'][0]):.1e}  cheb {rel(res['union'][1], res['prod'][1]):.1e}  "
          f"cheb-first {rel(res['union'][2], res['prod'][2]):.1e}  residual {rel(res['union'][3], res['prod'][3]):.1e}", flush=True)
# closer look at the residual and the non-first Chebyshev term on the last mesh
KX = torch.zeros(sysd.n, ncols, device=dev); ops.apply_K(X, KX)
c = torch.zeros(sysd.n, ncols, device=dev); ops.spmm_residual(X, R0, c)
print("residual vs R0 - KX:", rel(c, R0 - KX), " vs -KX:", rel(c, -KX), " vs R0:", rel(c, R0))
e = ((c - (R0 - KX)).abs() / (R0 - KX).abs().max()).reshape(sysd.nv, 3, ncols)
print("  err by node mod 4:", [f"{float(e[s::4].max()):.1e}" for s in range(4)], " by row:", [f"{float(e[:, r].max()):.1e}" for r in range(3)])
a = Wp.clone(); ops.cheb_spmm(X, a, R0, 0.31, 0.77, False)
ref = res["prod"][1]
e = ((a - ref).abs() / ref.abs().max()).reshape(sysd.nv, 3, ncols)
print("cheb err by node mod 4:", [f"{float(e[s::4].max()):.1e}" for s in range(4)], " by row:", [f"{float(e[:, r].max()):.1e}" for r in range(3)])
