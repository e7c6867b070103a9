import os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, scipy.linalg as sla
dev = torch.device("cuda:0")
src = open("tests/test_api_gpu.py").read()
ns = {}
exec(src[src.index("def _random_spd_pencil"):src.index("@pytest.mark.parametrize(\"n,method\"")], {"np": np}, ns)
from src.lobpcg import lobpcg_func
for n, method in ((1000, "basic"), (998, "basic")):
    try:
        Ad, Bd = ns["_random_spd_pencil"](n, 0.01, n)
        w = sla.eigh(Ad, Bd, eigvals_only=True)
        A = torch.from_numpy(Ad).float().to(dev).to_sparse(); B = torch.from_numpy(Bd).float().to(dev).to_sparse()
        hist = []
        E, X, rerr = lobpcg_func(A, B, 10, n=16, largest=False, niter=400, method=method, return_rerr=True,
                                 tracker=lambda st: hist.append((st.ivars["istep"], st.ivars["converged_count"], float(st.tvars["rerr"].max())) if "rerr" in st.tvars else None))
        print(n, method, "E", E.cpu().numpy()[:5], "w", w[:5], "rerr", float(rerr.max()), "iters", len(hist), hist[-3:])
        Xd = X.double().cpu().numpy()
        print("  orth", np.abs(Xd.T @ Bd @ Xd - np.eye(10)).max(), "eig err", np.abs(E.double().cpu().numpy() - w[:10]).max() / w[9])
    except Exception:
        traceback.print_exc()
from oracle import fem
from diffsound_amd import meshgen
from diffsound_amd.lobpcg.modal_solver import ModalSolver, SolverConfig
from diffsound_amd.modal_ops import HipModalOps, TetSystem
MAT = (2700.0, 5e10, 0.25)
v, t = meshgen.kuhn_box(8)
v, t = fem.to_high_order(torch.from_numpy(v), torch.from_numpy(t).long(), 2)
sysd = TetSystem(v.to(dev), t.to(dev), 2, MAT[0])
lam, mu = fem.lame(MAT[1], MAT[2])
for native in (True, False):
    for name, kw in (("fresh", dict(kx_fresh=True, fused_residual=False)), ("fused", dict(kx_fresh=True, fused_residual=True))):
        ops = HipModalOps(sysd, lam, mu)
        cfg = SolverConfig(block=40, lmax_cap=10.0, tol=1e-5, nested_tol=1e-2, native=native, raw_rr=False, raw_start=False, **kw)
        res = ModalSolver(ops, cfg).solve(32)
        print(native, name, res.iterations, res.coarse_iterations, " ".join(f"{h[1]:.6e}" for h in res.history), "rerr", res.rerr.cpu().numpy()[[0, 5, 31]])
