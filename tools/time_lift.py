"""Times the mesh front end at C3 on the device: ord-2 lifting (edge table + radix unique) and the symbolic phase."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from diffsound_amd import meshgen
from diffsound_amd.diffelastic.mesh import TetMesh
from diffsound_amd.modal_ops import TetSystem

dev = torch.device("cuda:0")
v, t = meshgen.kuhn_box(26)
v, t = torch.from_numpy(v).to(dev), torch.from_numpy(t).long().to(dev)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.time()
    m = TetMesh(v, t).to_high_order(2)
    torch.cuda.synchronize(); t1 = time.time()
    s = TetSystem(m.vertices, m.tets, 2, 2700.0)
    torch.cuda.synchronize(); t2 = time.time()
    print(f"rep {rep}: lifting {1e3 * (t1 - t0):.2f} ms, TetSystem (ordering + symbolic + tables) {1e3 * (t2 - t1):.2f} ms", flush=True)
# the torch.unique route for comparison
vf = v[t]
allv = torch.cat([v] + [(vf[:, a] + vf[:, b]) / 2 for a, b in ((0, 1), (1, 2), (0, 2), (0, 3), (1, 3), (2, 3))])
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.time()
    u, inv = torch.unique(allv, dim=0, return_inverse=True)
    torch.cuda.synchronize()
    print(f"torch.unique(dim=0) on {allv.shape[0]} rows: {1e3 * (time.time() - t0):.2f} ms", flush=True)

# where the symbolic time goes
from diffsound_amd import _hip
from diffsound_amd.modal_ops import UNION_CAP, morton_order

for rep in range(2):
    torch.cuda.synchronize(); t0 = time.time()
    perm = morton_order(m.vertices)
    inv = torch.empty_like(perm); inv[perm] = torch.arange(perm.numel(), device=dev)
    tets = inv[m.tets.long()].to(torch.int32).contiguous()
    torch.cuda.synchronize(); t1 = time.time()
    pat = _hip.DevicePattern(tets, m.vertices.shape[0], UNION_CAP)
    torch.cuda.synchronize(); t2 = time.time()
    s = TetSystem(m.vertices, m.tets, 2, 2700.0)
    torch.cuda.synchronize(); t3 = time.time()
    s.assemble()
    torch.cuda.synchronize(); t4 = time.time()
    c = s.coarse_level()
    torch.cuda.synchronize(); t5 = time.time()
    print(f"morton+renumber {1e3*(t1-t0):.2f} ms, DevicePattern {1e3*(t2-t1):.2f} ms, whole TetSystem {1e3*(t3-t2):.2f} ms, "
          f"numeric assembly alone {1e3*(t4-t3):.2f} ms, coarse level (sub-mesh + its TetSystem + transfer) {1e3*(t5-t4):.2f} ms", flush=True)
