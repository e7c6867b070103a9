"""Where does the host spend the time between two bench steps (device idle 7-13 ms in the kernel trace)?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from diffsound_amd import meshgen, pipeline
from diffsound_amd.diffelastic.mesh import TetMesh
from diffsound_amd.lobpcg.modal_solver import SolverConfig
from diffsound_amd.pipeline import ModalPipeline
MAT = (2700.0, 5e10, 0.25, 6.0, 1e-7)
dev = torch.device("cuda")
v, t = meshgen.kuhn_box(26)
mesh = TetMesh(torch.from_numpy(v).to(dev), torch.from_numpy(t).long().to(dev)).to_high_order(2)
cfg = SolverConfig(block=80, cheb_degree=48, cheb_ratio=800.0, lmax_cap=10.0)
pipe = ModalPipeline(mesh.vertices, mesh.tets, 2, 64, MAT, solver_config=cfg)
pipe.assemble(); _, _, a0 = pipe.run_pass(MAT[1], MAT[2], backward=False); pipe.set_target(a0)
hyps = [(4e10 + 1e10 * i, 0.2 + 0.02 * i) for i in range(6)]
marks = []
orig_assemble = type(pipe.system).assemble
def traced(self, *a, **k):
    marks.append(("assemble_enter", time.perf_counter()))
    return orig_assemble(self, *a, **k)
type(pipe.system).assemble = traced
import threading
orig_run_pass = ModalPipeline.run_pass
def traced_rp(self, *a, **k):
    marks.append(("run_pass_enter", time.perf_counter()))
    return orig_run_pass(self, *a, **k)
ModalPipeline.run_pass = traced_rp
from diffsound_amd.pipeline import DirectLinear
orig_lame = DirectLinear.lame
def traced_lame(self):
    marks.append(("lame_enter", time.perf_counter()))
    r = orig_lame(self)
    marks.append(("lame_exit", time.perf_counter()))
    return r
DirectLinear.lame = traced_lame
for s in range(4):
    t0 = time.perf_counter()
    outs = pipe.run_batch(hyps, lanes=3)
    t1 = time.perf_counter()
    loss = sum(o[0].loss for o in outs)
    tot = pipeline.all_reduce_loss(loss, dev)
    t2 = time.perf_counter()
    first = min(tm for nm, tm in marks if tm >= t0 and nm == "assemble_enter")
    for name in ("run_pass_enter", "lame_enter", "lame_exit"):
        print(f"      first {name}: {1e3*(min(tm for nm, tm in marks if tm >= t0 and nm == name)-t0):.2f} ms")
    print(f"step {s}: run_batch {1e3*(t1-t0):.1f} ms, first assemble call {1e3*(first-t0):.2f} ms after step start, reduce {1e3*(t2-t1):.2f} ms")
    del outs
    t3 = time.perf_counter()
    print(f"   del outs {1e3*(t3-t2):.2f} ms")
