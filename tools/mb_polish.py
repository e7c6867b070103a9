"""Read-out products at C3 (64 columns): three ds_spmm_bsr3 launches against ds_spmm_f64_polish (one walk)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from diffsound_amd import _hip, meshgen
from diffsound_amd.diffelastic.mesh import TetMesh
from diffsound_amd.modal_ops import HipModalOps, TetSystem

dev = torch.device("cuda")
v, t = meshgen.kuhn_box(26)
mesh = TetMesh(torch.from_numpy(v).to(dev), torch.from_numpy(t).long().to(dev)).to_high_order(2)
sysd = TetSystem(mesh.vertices, mesh.tets, 2, 2700.0)
ops = HipModalOps(sysd, 2e10, 2e10, two_level=False)
X = torch.randn(sysd.n, 64, device=dev)
Y = torch.empty((3, sysd.n, 64), dtype=torch.float64, device=dev)
L, p = _hip.lib(), _hip.ptr


def separate():
    for (kind, vals), out in zip(((2, sysd.klam), (2, sysd.kmu), (3, sysd.ms)), Y):
        ops._spmm(kind, vals, X, out)


def fused():
    _hip.check(L.ds_spmm_f64_polish(p(sysd.rowptr), p(sysd.colidx), p(sysd.klam), p(sysd.kmu), p(sysd.ms), sysd.nv, p(X), 64,
                                    p(Y[0]), p(Y[1]), p(Y[2]), 64, 64, _hip.stream_ptr()), "ds_spmm_f64_polish")


for name, fn in (("three launches", separate), ("one walk", fused)):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"{name}: {e0.elapsed_time(e1) / 10:.3f} ms", flush=True)
