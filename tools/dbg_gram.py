"""Error of the folded-fp32 Gram against fp64, by row count."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from diffsound_amd.modal_ops import _HipBlockOps
dev = torch.device('cuda')
for nv in (75, 1000, 20000, 148877):
    n = 3 * nv
    ops = _HipBlockOps(); ops._init_common(None, None, nv, dev)
    for p, q, sym in ((8, 40, False), (240, 80, False), (240, 240, True), (33, 47, False)):
        g = torch.Generator().manual_seed(1)
        A = torch.randn((n, p), generator=g).to(dev); B = torch.randn((n, q), generator=g).to(dev)
        if sym: B = A * 1.5
        ref = (A.double().T @ B.double()).cpu().numpy()
        G = ops.gram(A, B, symmetric=sym).cpu().numpy()
        Ge = ops.gram(A, B, symmetric=sym, exact=True).cpu().numpy()
        sc = np.sqrt(np.outer((A.double() ** 2).sum(0).cpu().numpy(), (B.double() ** 2).sum(0).cpu().numpy()))
        print(f"n={n:7d} {p:3d}x{q:3d} sym={int(sym)}  fast err/scale {np.abs(G - ref).max() / sc.max():.2e}  exact {np.abs(Ge - ref).max() / sc.max():.2e}")
