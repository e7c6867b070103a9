"""Wall-clock breakdown of one cold eigensolve on the benchmark mesh (synchronising around each stage)."""
import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffsound_amd import meshgen
from diffsound_amd.diffelastic.mesh import TetMesh
from diffsound_amd.modal_ops import TetSystem, HipModalOps
from diffsound_amd.lobpcg import modal_solver as ms

dev = torch.device('cuda')
v, t = meshgen.kuhn_box(int(os.environ.get('CELLS', 26)))
mesh = TetMesh(torch.from_numpy(v).to(dev), torch.from_numpy(t).long().to(dev)).to_high_order(2)
sysd = TetSystem(mesh.vertices, mesh.tets, 2, 2700.0)
E, nu = 5e10, 0.25
ops = HipModalOps(sysd, E * nu / ((1 + nu) * (1 - 2 * nu)), E / (2 * (1 + nu)))
acc = collections.defaultdict(float); cnt = collections.Counter()
def wrap(obj, name, label=None):
    f = getattr(obj, name)
    def g(*a, **k):
        torch.cuda.synchronize(); t0 = time.time(); r = f(*a, **k); torch.cuda.synchronize()
        acc[label or name] += time.time() - t0; cnt[label or name] += 1; return r
    setattr(obj, name, g)
for n in ('apply_K', 'apply_M', 'gram', 'mix', 'mix_inplace', 'residual', 'cheb_init', 'cheb_spmm', 'polish_products'):
    wrap(ops, n)
wrap(ms, '_small', 'host_dense')
E_ = os.environ.get
cfg = ms.SolverConfig(block=int(E_('BLOCK', 80)), cheb_degree=int(E_('DEG', 48)), cheb_ratio=float(E_('RATIO', 800)), lmax_cap=10.0,
                      precond=E_('PRECOND', 'auto'), ortho_passes=int(E_('OP', 2)), ortho_tol=float(E_('OT', 2e-6)), smooth_degree=int(E_('SD', 3)), smooth_ratio=float(E_('SR', 10)),
                      coarse_degree=int(E_('CD', 24)), coarse_ratio=float(E_('CR', 400)))
if ops.coarse is not None:
    for n in ('apply_K', 'cheb_init', 'cheb_spmm'):
        wrap(ops.coarse, n, 'coarse_' + n)
    for n in ('spmm_residual', 'restrict', 'prolong_add'):
        wrap(ops, n)
for rep in range(2):
    acc.clear(); cnt.clear()
    torch.cuda.synchronize(); t0 = time.time()
    solver = ms.ModalSolver(ops, cfg)
    torch.cuda.synchronize(); t_setup = time.time() - t0
    res = solver.solve(64)
    torch.cuda.synchronize(); tot = time.time() - t0
print("ortho amp", " ".join(f"{a:.1f}" for a in solver.ortho_log))
print(f"max rerr {float(res.rerr.max()):.2e}" if hasattr(res, "rerr") else "", end=" ")
print(f"total {tot*1e3:.1f} ms (solver setup incl. lmax power iteration {t_setup*1e3:.1f} ms), iterations {res.iterations}")
for k, v_ in sorted(acc.items(), key=lambda kv: -kv[1]):
    print(f"  {k:18s} {v_*1e3:8.1f} ms  calls {cnt[k]:5d}  avg {v_/cnt[k]*1e3:7.3f} ms")
print(f"  accounted {sum(acc.values())*1e3:.1f} ms")
