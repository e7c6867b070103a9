"""Launch the fused Chebyshev-term SpMM a few times (for rocprofv3 --pmc runs)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffsound_amd import meshgen
from diffsound_amd.diffelastic.mesh import TetMesh
from diffsound_amd.modal_ops import TetSystem, HipModalOps
dev = torch.device('cuda')
v, t = meshgen.kuhn_box(26)
mesh = TetMesh(torch.from_numpy(v).to(dev), torch.from_numpy(t).long().to(dev)).to_high_order(2)
sysd = TetSystem(mesh.vertices, mesh.tets, 2, 2700.0)
ops = HipModalOps(sysd, 2e10, 2e10)
W = torch.randn(sysd.n, 80, device=dev); Wp = torch.randn_like(W); R0 = torch.randn_like(W); Y = torch.empty_like(W)
for _ in range(5):
    ops._cheb_spmm_launch(W, Wp, R0, 0.3, 0.7, False)
    ops.apply_M(W, Y)
torch.cuda.synchronize()
