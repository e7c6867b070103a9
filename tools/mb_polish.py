"""The read-out's fp64-value products (ds_spmm_f64_polish: K_lambda X, K_mu X, M_s X in one walk) at C3, 80 columns."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffsound_amd import _hip, meshgen
from diffsound_amd.diffelastic.mesh import TetMesh
from diffsound_amd.modal_ops import HipModalOps, TetSystem
dev = torch.device("cuda")
v, t = meshgen.kuhn_box(26)
mesh = TetMesh(torch.from_numpy(v).to(dev), torch.from_numpy(t).long().to(dev)).to_high_order(2)
sysd = TetSystem(mesh.vertices, mesh.tets, 2, 2700.0)
ops = HipModalOps(sysd, 2e10, 2e10, two_level=False, mfma_groups=(0, 0))
X = torch.randn(sysd.n, 80, device=dev)
Y3 = torch.empty(sysd.n, 240, dtype=torch.float64, device=dev)
p = _hip.ptr
def spmm():
    _hip.check(_hip.lib().ds_spmm_f64_polish(p(sysd.rowptr), p(sysd.colidx), p(sysd.klam), p(sysd.kmu), p(sysd.ms), sysd.nv, p(X), 80,
                                             p(Y3[:, :80]), p(Y3[:, 80:160]), p(Y3[:, 160:]), 240, 80, _hip.stream_ptr()), "polish")
t0 = time.time()
while time.time() - t0 < 1.5:
    spmm(); ops.polish_products(X)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for rnd in range(3):
    for name, fn in (("ds_spmm_f64_polish alone", spmm), ("polish products (SpMM + Gram 80 x 240)", lambda: ops.polish_products(X))):
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        print(f"{name}: {e0.elapsed_time(e1) / 10 * 1e3:.1f} us", flush=True)
