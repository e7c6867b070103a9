"""Union SpMM on a mesh with vertices no tet references (empty rows, possibly a whole empty group)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from diffsound_amd import meshgen
from diffsound_amd.modal_ops import TetSystem, HipModalOps
dev = torch.device('cuda')
v, t = meshgen.kuhn_box(3)
far = np.array([[9.0, 9.0, 9.0 + 0.01 * i] for i in range(9)], dtype=v.dtype)  # 9 unused vertices, clustered: one whole group
v2 = np.concatenate([v, far], 0)
for reorder in (True, False):
    s = TetSystem(torch.from_numpy(v2).to(dev), torch.from_numpy(t).to(dev), 1, 2700.0, reorder=reorder)
    ops = HipModalOps(s, 2e10, 2e10, two_level=False)
    X = torch.randn(s.n, 16, device=dev); R0 = torch.randn_like(X); Wp = torch.randn_like(X)
    def run():
        Y = torch.zeros_like(X); ops.apply_K(X, Y)
        a = Wp.clone(); ops.cheb_spmm(X, a, R0, 0.3, 0.7, False)
        c = torch.zeros_like(X); ops.spmm_residual(X, R0, c)
        return Y, a, c
    assert s.groups["union"] is not None
    got = run(); u = s.groups["union"]; s.groups["union"] = None; ref = run(); s.groups["union"] = u
    print("reorder", reorder, "empty rows", int((s.rowptr[1:] == s.rowptr[:-1]).sum()), [float((g - r).abs().max() / r.abs().max()) for g, r in zip(got, ref)])
