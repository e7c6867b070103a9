"""Time the numeric assembly (geometry + block kernels) and the material combine on the benchmark mesh."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffsound_amd import meshgen
from diffsound_amd.diffelastic.mesh import TetMesh
from diffsound_amd.modal_ops import TetSystem, HipModalOps
dev = torch.device('cuda')
v, t = meshgen.kuhn_box(26)
mesh = TetMesh(torch.from_numpy(v).to(dev), torch.from_numpy(t).long().to(dev)).to_high_order(2)
sysd = TetSystem(mesh.vertices, mesh.tets, 2, 2700.0)
ops = HipModalOps(sysd, 2e10, 2e10, two_level=True)
def tm(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
nb = sysd.nnzb * 19 * 8 + sysd.cptr[-1].item() * 4
print(f"assemble (both levels) {tm(sysd.assemble):.3f} ms ; fine level writes {nb/1e6:.0f} MB ; set_material {tm(lambda: ops.set_material(2.1e10, 1.9e10)):.3f} ms")
