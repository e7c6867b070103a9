"""Where does the HOST time of a pass go?  cProfile over a few single-lane passes at C3 (the native eigensolver loops show
as one ctypes call each).  python tools/host_profile.py [passes]"""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from diffsound_amd import meshgen
from diffsound_amd.diffelastic.mesh import TetMesh
from diffsound_amd.pipeline import ModalPipeline

dev = torch.device("cuda:0")
v, t = meshgen.kuhn_box(26)
mesh = TetMesh(torch.from_numpy(v).to(dev), torch.from_numpy(t).long().to(dev)).to_high_order(2)
pipe = ModalPipeline(mesh.vertices, mesh.tets, 2, 64, bench.MAT, solver_config=bench.solver_config())
pipe.assemble()
_, _, audio = pipe.run_pass(bench.MAT[1], bench.MAT[2], backward=False)
pipe.set_target(audio)
for E in (3e10, 4e10):
    pipe.run_pass(E, 0.3)
torch.cuda.synchronize()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
pr = cProfile.Profile()
pr.enable()
for i in range(n):
    pipe.run_pass(2e10 + 1e10 * i, 0.2 + 0.02 * i)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
st.sort_stats("cumulative").print_stats(35)
