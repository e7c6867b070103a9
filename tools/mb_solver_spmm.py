"""The SpMM forms of one eigensolver iteration on the benchmark mesh (C3), on the solver's own operand layout, interleaved and warmed
up: [K W | M W] in one walk, the fused residual, K W, M W, the bf16 fused term on both levels.  One process per library build
(DS_EXP_LIB): tools/sweep runs print comparable tables.   python tools/mb_solver_spmm.py [tag]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffsound_amd import meshgen
from diffsound_amd.diffelastic.mesh import TetMesh
from diffsound_amd.modal_ops import HipModalOps, TetSystem
tag = sys.argv[1] if len(sys.argv) > 1 else os.path.basename(os.environ.get("DS_EXP_LIB", "libdiffsound_hip.so"))
dev = torch.device("cuda")
v, t = meshgen.kuhn_box(26)
mesh = TetMesh(torch.from_numpy(v).to(dev), torch.from_numpy(t).long().to(dev)).to_high_order(2)
sysd = TetSystem(mesh.vertices, mesh.tets, 2, 2700.0)
ops = HipModalOps(sysd, 2e10, 2e10)
n, b, ny = sysd.n, 80, 16
S, KS = torch.randn(n, 256, device=dev), torch.empty(n, 256, device=dev)
X, W = S[:, ny:ny + b], S[:, ny + 2 * b:ny + 3 * b]
R = torch.empty(n, b, device=dev)
lam = torch.rand(b, device=dev, dtype=torch.float64) * 1e9
mk = lambda o: torch.randn(o.n, 80, device=dev).bfloat16()
fa, fb, fc = mk(ops), mk(ops), mk(ops)
ca, cb, cc = mk(ops.coarse), mk(ops.coarse), mk(ops.coarse)
cases = [("[K W | M W] one walk", lambda: ops.apply_KM(W, KS[:, :b], KS[:, b:2 * b])),
         ("fused residual", lambda: ops.residual_fused(X, lam, R)),
         ("K W", lambda: ops._union(0, W, KS[:, 2 * b:3 * b])),
         ("M W", lambda: ops._union(3, W, R)),
         ("bf16 term, fine level", lambda: ops.cheb_spmm16(fa, fb, fc, 0.3, 0.7, False)),
         ("bf16 term, corner-node level", lambda: ops.coarse.cheb_spmm16(ca, cb, cc, 0.3, 0.7, False))]
t0 = time.time()
while time.time() - t0 < 2.0:
    for _, fn in cases: fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
res = {nm: [] for nm, _ in cases}
for rnd in range(3):
    for nm, fn in cases:
        fn(); e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        res[nm].append(e0.elapsed_time(e1) / 20 * 1e3)
for nm, _ in cases:
    print(f"[{tag}] {nm}: " + " / ".join(f"{x:.1f}" for x in res[nm]) + " us", flush=True)
