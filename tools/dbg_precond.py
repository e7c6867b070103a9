"""Debug aid: preconditioner application with the batched SpMM kernels vs the wave-per-node kernels (same process)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from diffsound_amd import meshgen
from diffsound_amd.diffelastic.mesh import TetMesh
from diffsound_amd.modal_ops import TetSystem, HipModalOps
from diffsound_amd.lobpcg.modal_solver import ChebyshevBlockJacobi, TwoLevelChebyshev, SolverConfig
dev = torch.device('cuda')
order = int(sys.argv[2]); ncols = int(sys.argv[3])
if sys.argv[1] == "bowl":
    m = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "g0_bowl_mesh.npz"))
    v, t = m[m.files[0]], m[m.files[1]]
else:
    v, t = meshgen.kuhn_box(int(sys.argv[1]))
mesh = TetMesh(torch.from_numpy(v).to(dev), torch.from_numpy(t).long().to(dev)).to_high_order(order)
sysd = TetSystem(mesh.vertices, mesh.tets, order, 2700.0)
ops = HipModalOps(sysd, 2e10, 2e10)
cfg = SolverConfig(cheb_degree=8, lmax_cap=float({1: 4, 2: 10}[order]))
def rel(a, b): return float((a - b).norm() / b.norm())
def both(fn):
    saved = ops.batches, (ops.coarse.batches if ops.coarse is not None else None)
    out1 = fn()
    ops.batches = None
    if ops.coarse is not None: ops.coarse.batches = None
    out2 = fn()
    ops.batches = saved[0]
    if ops.coarse is not None: ops.coarse.batches = saved[1]
    return out1, out2
big = torch.zeros(sysd.n, 8 + 3 * ncols, device=dev)
R = torch.randn(sysd.n, ncols, device=dev) * 1e9
pre = ChebyshevBlockJacobi(ops, 8, 100.0, cap=cfg.lmax_cap)
for first_cols in (ncols, ncols - 4):
    def run():
        W = big[:, 8 + ncols: 8 + ncols + first_cols]; W.zero_()
        pre.apply(R[:, :first_cols].clone(), W); return W.clone()
    a, b = both(run); print(f"cheb apply {first_cols} cols: rel diff {rel(a, b):.2e}  norm {float(b.norm()):.3e}")
    def run2():
        W = big[:, 8 + ncols: 8 + ncols + first_cols]; W.copy_(torch.randn_like(W))
        torch.manual_seed(1); W.copy_(torch.randn(W.shape, device=dev))
        pre.apply(R[:, :first_cols].clone(), W, from_guess=True); return W.clone()
    a, b = both(run2); print(f"cheb from_guess {first_cols} cols: rel diff {rel(a, b):.2e}")
    def run3():
        X = big[:, 8: 8 + first_cols]; torch.manual_seed(2); X.copy_(torch.randn(X.shape, device=dev))
        Y = torch.empty(sysd.n, first_cols, device=dev); ops.spmm_residual(X, R[:, :first_cols], Y); return Y
    a, b = both(run3); print(f"spmm_residual {first_cols} cols: rel diff {rel(a, b):.2e}")
    def run4():
        X = big[:, 8: 8 + first_cols]; torch.manual_seed(2); X.copy_(torch.randn(X.shape, device=dev))
        Y = torch.empty(sysd.n, first_cols, device=dev); ops.apply_M(X, Y); Z = torch.empty_like(Y); ops.apply_K(X, Z); return torch.cat([Y, Z * 1e-12], 1)
    a, b = both(run4); print(f"apply_M/K {first_cols} cols: rel diff {rel(a, b):.2e}")
if ops.coarse is not None:
    two = TwoLevelChebyshev(ops, cfg)
    def run5():
        W = big[:, 8 + ncols: 8 + 2 * ncols]; W.zero_(); two.apply(R.clone(), W); return W.clone()
    a, b = both(run5); print(f"two-level apply: rel diff {rel(a, b):.2e}")
