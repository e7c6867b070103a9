"""ds_mix alone at the solver's main shape (debug-variant timing)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffsound_amd.modal_ops import _HipBlockOps
dev = torch.device('cuda')
nv = 148877; n = 3 * nv
ops = _HipBlockOps(); ops._init_common(None, None, nv, dev)
S = torch.randn((n, 248), device=dev); out = torch.empty((n, 160), device=dev)
for p, q in ((240, 80), (160, 80), (80, 80), (240, 160), (216, 144), (168, 112)):
    C = torch.randn((p, q), dtype=torch.float64, device=dev)
    A = S[:, :p]; O = out[:, :q]
    f = lambda: ops.mix(A, C, O)
    f(); torch.cuda.synchronize(); t = time.time()
    for _ in range(20): f()
    torch.cuda.synchronize(); ms = (time.time() - t) / 20 * 1e3
    print(f"mix {p}->{q}: {ms:.3f} ms  {2.0*n*p*q/ms/1e9:6.1f} TF/s")
