"""cProfile of TetSystem construction at C3 (host-side view of the symbolic phase)."""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from diffsound_amd import meshgen
from diffsound_amd.diffelastic.mesh import TetMesh
from diffsound_amd.modal_ops import TetSystem

dev = torch.device("cuda:0")
v, t = meshgen.kuhn_box(26)
m = TetMesh(torch.from_numpy(v).to(dev), torch.from_numpy(t).long().to(dev)).to_high_order(2)
s = TetSystem(m.vertices, m.tets, 2, 2700.0)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
s2 = TetSystem(m.vertices, m.tets, 2, 2700.0)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
