#!/usr/bin/env python3
"""Where a step of the fp64 refinement at configs[4] spends its time: the solver's calls wrapped in synchronising timers
(python tools/c5_refine_breakdown.py [cells] [modes] [block])."""
import collections
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from diffsound_amd import meshgen  # noqa: E402
from diffsound_amd.diffelastic.diff_model import _lame  # noqa: E402
from diffsound_amd.diffelastic.mesh import TetMesh  # noqa: E402
from diffsound_amd.lobpcg import modal_solver as ms  # noqa: E402
from diffsound_amd.modal_ops import HipModalOps, TetSystem  # noqa: E402

cells = int(sys.argv[1]) if len(sys.argv) > 1 else 55
modes = int(sys.argv[2]) if len(sys.argv) > 2 else 128
block = int(sys.argv[3]) if len(sys.argv) > 3 else 136
sweeps = int(sys.argv[4]) if len(sys.argv) > 4 else 2
dev = torch.device("cuda", 0)
v, t = meshgen.kuhn_box(cells)
mesh = TetMesh(torch.from_numpy(v).to(dev), torch.from_numpy(t).long().to(dev)).to_high_order(2)
sysd = TetSystem(mesh.vertices, mesh.tets, 2, bench.MAT[0])
lam, mu = (float(x) for x in _lame(bench.MAT[1], bench.MAT[2]))
ops = HipModalOps(sysd, lam, mu)
acc, cnt = collections.defaultdict(float), collections.defaultdict(int)
on = [False]


def wrap(obj, name, label=None):
    f = getattr(obj, name)
    label = label or name

    def g(*a, **k):
        if not on[0]:
            return f(*a, **k)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = f(*a, **k)
        torch.cuda.synchronize()
        acc[label] += time.perf_counter() - t0
        cnt[label] += 1
        return r
    setattr(obj, name, g)


for nm in ("mix64", "gram", "gram_blocks", "apply_K64", "apply_M64", "apply_K"):
    wrap(ops, nm)
wrap(ms, "_small", "host Rayleigh-Ritz (_small)")
nest = dict(nested_tol=3e-3, nested_maxit=8, nested_cheb_degree=22, nested_cheb_ratio=350.0)  # bench.py's defaults
for rep in range(2):
    solver = ms.ModalSolver(ops, ms.SolverConfig(block=block, lmax_cap=10.0, refine_tol=1e-10, refine_sweeps=sweeps, **nest))
    wrap(solver, "precond_apply", "preconditioner (fp32 V-cycle)")
    orig = solver.refine64

    def timed_refine(*aa, **kk):
        acc.clear(), cnt.clear()
        on[0] = True
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = orig(*aa, **kk)
        torch.cuda.synchronize()
        acc["refine64 total"] = time.perf_counter() - t0
        on[0] = False
        return r
    solver.refine64 = timed_refine
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r64 = solver.solve(modes)
    torch.cuda.synchronize()
    tot = time.perf_counter() - t0
print(f"cells {cells}: n = {sysd.n}, {modes} modes, block {block}, {sweeps} preconditioner sweeps per step; solve {tot:.3f} s, {r64.refine_iterations} fp64 steps, "
      f"worst backward error {float(r64.rerr.max()):.2e} (second solve; timers synchronise, so the parts add up to more than an "
      "unsynchronised run)")
total = acc.pop("refine64 total")
rest = total - sum(acc.values())
for k, s in sorted(acc.items(), key=lambda kv: -kv[1]):
    print(f"  {k:34s} {s * 1e3:9.1f} ms  {cnt[k]:4d} calls  {100 * s / total:5.1f} %")
print(f"  {'everything else (torch elementwise)':34s} {rest * 1e3:9.1f} ms              {100 * rest / total:5.1f} %")
print(f"  {'refine64 total':34s} {total * 1e3:9.1f} ms")
