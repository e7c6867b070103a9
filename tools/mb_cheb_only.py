"""Launch the fused Chebyshev-term SpMM on an 80-column block a few times (for rocprofv3 --pmc runs)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffsound_amd import meshgen
from diffsound_amd.diffelastic.mesh import TetMesh
from diffsound_amd.modal_ops import TetSystem, HipModalOps
dev = torch.device('cuda')
v, t = meshgen.kuhn_box(26)
mesh = TetMesh(torch.from_numpy(v).to(dev), torch.from_numpy(t).long().to(dev)).to_high_order(2)
sysd = TetSystem(mesh.vertices, mesh.tets, 2, 2700.0)
kind = sys.argv[2] if len(sys.argv) > 2 else "fp32"  # fp32 | bf16 (VALU kernel on bf16 blocks) | mfma (ds_spmm_union16m) | kx (Y = K X, fp32) | km | resid
ops = HipModalOps(sysd, 2e10, 2e10, two_level=False, mfma_groups=(8, 0) if kind == "mfma" else (0, 0))
X = torch.randn(sysd.n, 80, device=dev); W = torch.randn(sysd.n, 80, device=dev); R0 = torch.randn(sysd.n, 80, device=dev)
bf = kind in ("bf16", "mfma")
if bf:
    X, W, R0 = X.bfloat16(), W.bfloat16(), R0.bfloat16()
if kind in ("kx", "km", "resid"):  # the iteration's own operands: column ranges of 256-column buffers (rows 1 KiB apart, a 16-column
    Sb, KSb = torch.randn(sysd.n, 256, device=dev), torch.empty(sysd.n, 256, device=dev)  # rigid block in front), as bench.py times it
    X, Yk, Xr = Sb[:, 176:256], KSb[:, 160:240], Sb[:, 16:96]
    Rr, lam = torch.empty(sysd.n, 80, device=dev), torch.rand(80, device=dev, dtype=torch.float64) * 1e9
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    if kind == "kx":  # the eigensolver's own product Y = K X (fp32)
        ops._union(0, X, Yk)
    elif kind == "km":  # [K W | M W] in one walk (round 5)
        ops.apply_KM(X, KSb[:, :80], KSb[:, 80:160])
    elif kind == "resid":  # the fused residual
        ops.residual_fused(Xr, lam, Rr)
    else:
        (ops.cheb_spmm16 if bf else ops.cheb_spmm)(X, W, R0, 0.3, 0.7, False)
torch.cuda.synchronize()
print("done", flush=True)
