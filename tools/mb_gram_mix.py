"""Microbenchmark of the tall-skinny dense kernels (ds_gram, ds_mix) at the solver's shapes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffsound_amd.modal_ops import _HipBlockOps
dev = torch.device('cuda')
nv = int(os.environ.get('NV', 148877)); n = 3 * nv
ops = _HipBlockOps(); ops._init_common(None, None, nv, dev)
def tm(f, reps=20):
    f(); torch.cuda.synchronize(); t = time.time()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.time() - t) / reps * 1e3
S = torch.randn((n, 248), device=dev); KS = torch.randn((n, 240), device=dev)
only = os.environ.get('ONLY')
shapes = ((248, 80, False), (240, 80, False), (240, 240, True), (168, 80, False), (88, 80, False), (80, 80, True))
if only:
    a_, b_, c_ = only.split(','); shapes = ((int(a_), int(b_), bool(int(c_))),)
for p, q, sym in (() if os.environ.get('MIX_ONLY') else shapes):
    A = S[:, 8:8 + p] if sym else S[:, :p]
    B = KS[:, :q]
    ms = tm(lambda: ops.gram(A, B, symmetric=sym))
    fl = 2.0 * n * p * q * (0.5 if sym else 1.0)
    by = n * 4.0 * (p + q)
    print(f"gram {p:3d}x{q:3d} sym={int(sym)}: {ms:.3f} ms  {fl/ms/1e9:6.1f} TF/s(f64)  {by/ms/1e6:6.0f} GB/s min-traffic")
mix_only = os.environ.get('MIX_ONLY')
if only and not mix_only: sys.exit(0)
out = torch.empty((n, 160), device=dev)
for p, q in (((240, 80), (240, 160)) if mix_only else ((160, 72), (72, 72), (224, 72), (240, 80), (240, 160), (80, 80), (80, 64))):
    C = torch.randn((p, q), dtype=torch.float64, device=dev)
    A = S[:, :p]; O = out[:, :q]
    ms = tm(lambda: ops.mix(A, C, O))
    by = n * 4.0 * (p + q)
    print(f"mix  {p:3d}->{q:3d}: {ms:.3f} ms  {2.0*n*p*q/ms/1e9:6.1f} TF/s(f32)  {by/ms/1e6:6.0f} GB/s min-traffic")
if mix_only: sys.exit(0)
X = torch.randn((n, 72), device=dev)
print(f"copy n x72: {tm(lambda: out[:, :72].copy_(X)):.3f} ms")
