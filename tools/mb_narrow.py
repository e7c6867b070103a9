"""8- and 16-column products on the benchmark mesh (C3): the narrow kernel (ds_spmm_union_narrow: lanes dealt over the union's
entries) against the production neighbour-union kernel on the same block; interleaved, warmed up."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffsound_amd import meshgen
from diffsound_amd.diffelastic.mesh import TetMesh
from diffsound_amd.modal_ops import HipModalOps, TetSystem
dev = torch.device("cuda")
v, t = meshgen.kuhn_box(26)
mesh = TetMesh(torch.from_numpy(v).to(dev), torch.from_numpy(t).long().to(dev)).to_high_order(2)
sysd = TetSystem(mesh.vertices, mesh.tets, 2, 2700.0)
ops = HipModalOps(sysd, 2e10, 2e10, two_level=True, mfma_groups=(0, 0))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for o, name in ((ops, "fine level"), (ops.coarse, "corner-node level")):
    for c in (8, 16):
        X, Y = torch.randn(o.n, c, device=dev), torch.empty(o.n, c, device=dev)
        cases = [("production K X", lambda: o._union(0, X, Y)), ("narrow K X", lambda: o._narrow(0, X, Y)),
                 ("production M X", lambda: o._union(3, X, Y)), ("narrow M X", lambda: o._narrow(3, X, Y))]
        t0 = time.time()
        while time.time() - t0 < 1.0:
            for _, fn in cases:
                fn()
        torch.cuda.synchronize()
        for rnd in range(2):
            for nm, fn in cases:
                fn(); e0.record()
                for _ in range(20): fn()
                e1.record(); torch.cuda.synchronize()
                print(f"{name}, {c} columns, {nm}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us", flush=True)
