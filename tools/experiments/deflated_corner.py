"""EXPERIMENT (iteration counts only, interpreter path): the V-cycle's corner-level solve with the corner-level eigenvectors of the
nested start DEFLATED - exact on their span, a Chebyshev polynomial of lower degree on the rest:
    E = X_c L^-1 X_c^T r  +  (I - X_c X_c^T M_c) p(K_c) (I - M_c X_c X_c^T) r .
Question: how many fine iterations does the benchmark's solve need with (degree, ratio) of p below the production 22 / 350?
    python tools/experiments/deflated_corner.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from diffsound_amd import meshgen
from diffsound_amd.diffelastic.mesh import TetMesh
from diffsound_amd.lobpcg import modal_solver as ms
from diffsound_amd.modal_ops import HipModalOps, TetSystem

dev = torch.device("cuda")
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 26
v, t = meshgen.kuhn_box(cells)
mesh = TetMesh(torch.from_numpy(v).to(dev), torch.from_numpy(t).long().to(dev)).to_high_order(2)
sysd = TetSystem(mesh.vertices, mesh.tets, 2, 2700.0)
E, nu = 7.0e10, 0.3
lam_, mu_ = E * nu / ((1 + nu) * (1 - 2 * nu)), E / (2 * (1 + nu))
ops = HipModalOps(sysd, lam_, mu_)
ms.TwoLevelChebyshev.use_native = False


class Deflated:
    def __init__(self, base, co, Xc, lamc):
        self.base, self.co = base, co
        self.X = Xc.double()
        MX = torch.empty_like(Xc)
        co.apply_M(Xc.contiguous(), MX)
        self.MX = MX.double()
        self.lam = lamc.double()
        self.degree, self.lmax, self.lmin = base.degree, base.lmax, base.lmin

    def _buffers(self, R):
        return self.base._buffers(R)

    def apply(self, Rc, Ec, from_guess=False):
        r = Rc.double()
        a = self.X.T @ r
        rp = (r - self.MX @ a).to(Rc.dtype).contiguous()
        self.base.apply(rp, Ec)
        e = Ec.double()
        e = e - self.X @ (self.MX.T @ e) + self.X @ (a / self.lam[:, None])
        Ec.copy_(e.to(Ec.dtype))


def run(tag, cdeg, cratio, deflate, ndefl=80):
    cfg = bench.solver_config(coarse_degree=cdeg, coarse_ratio=cratio)
    cfg.native = False
    cfg.precond_storage = "fp32"
    cfg.nested_cheb_degree, cfg.nested_cheb_ratio = 22, 350.0  # the nested start itself as in production
    s = ms.ModalSolver(ops, cfg)
    if deflate:
        orig = s._nested_start

        def nested(k, b):
            co = ops.coarse
            if co.rigid is None:
                co.rigid = co._rigid_basis()
            ccfg = ms.SolverConfig(block=b, guard=cfg.guard, tol=cfg.nested_tol, maxit=cfg.nested_maxit, seed=cfg.seed, cheb_degree=22,
                                   cheb_ratio=350.0, power_iters=cfg.power_iters, lmax_safety=cfg.lmax_safety, lmax_cap=4.0,
                                   precond="chebyshev", native=False)
            cs = ms.ModalSolver(co, ccfg)
            rc = cs.solve(k, polish=False)
            s.nested_iterations = rc.iterations
            Xc = rc.block_vectors[:, :ndefl].contiguous()
            # Ritz values of the block on the corner level
            KX = torch.empty_like(Xc)
            co.apply_K(Xc, KX)
            lamc = (Xc.double() * KX.double()).sum(0)
            s.precond.coarse = Deflated(s.precond.coarse, co, Xc, lamc)
            X0 = torch.zeros((ops.n, b), dtype=ops.dtype, device=ops.device)
            ops.prolong_add(rc.block_vectors, X0)
            return X0

        s._nested_start = nested
    res = s.solve(64)
    hist = getattr(res, "history", None)
    print(f"{tag}: corner polynomial {cdeg} / {cratio:g}, deflation {'on (%d vectors)' % ndefl if deflate else 'off'}: fine iterations {res.iterations}, "
          f"corner-level iterations {s.nested_iterations}, lowest eigenvalues {res.eigenvalues[:3].tolist()}", flush=True)


run("production", 22, 350.0, False)
for cdeg, cratio in ((22, 350.0), (22, 60.0), (12, 60.0), (10, 40.0), (8, 30.0), (8, 20.0), (6, 15.0), (12, 100.0)):
    run("deflated", cdeg, cratio, True)
run("no deflation", 12, 60.0, False)
