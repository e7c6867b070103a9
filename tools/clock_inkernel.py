"""Diagnostic build only (DS_SPMM_DBG=7): in-kernel clock of the batched K X product from s_memtime / s_memrealtime stamps."""
import os, sys, time
os.environ["DS_SPMM_DBG"] = "7"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffsound_amd import meshgen, _hip
from diffsound_amd.diffelastic.mesh import TetMesh
from diffsound_amd.modal_ops import TetSystem, HipModalOps, _ld
dev = torch.device('cuda')
v, t = meshgen.kuhn_box(26)
mesh = TetMesh(torch.from_numpy(v).to(dev), torch.from_numpy(t).long().to(dev)).to_high_order(2)
sysd = TetSystem(mesh.vertices, mesh.tets, 2, 2700.0)
ops = HipModalOps(sysd, 2e10, 2e10, two_level=False)
X = torch.randn(sysd.n, 80, device=dev); Y = torch.empty_like(X)
bt = ops.batches
dbg = torch.zeros((bt.shape[0], 2), dtype=torch.int64, device=dev)
pp = _hip.ptr
def launch():
    _hip.check(ops._L.ds_spmm_batched(0, 0, pp(bt), bt.shape[0], pp(ops.rowptr), pp(ops.colidx), pp(ops.k32t), ops.k32t.shape[0],
                                      ops.nv, pp(X), _ld(X), pp(Y), _ld(Y), None, 0, pp(dbg), 80, 0.0, 0.0, 0, _hip.stream_ptr()), "dbg")
t0 = time.time()
while time.time() - t0 < 2.5:
    for _ in range(200): launch()
    torch.cuda.synchronize()
d = dbg.double().cpu()
ratio = d[:, 0] / d[:, 1] * 100e6
print(f"in-kernel clock: median {ratio.median()/1e9:.3f} GHz  (10%..90%: {ratio.quantile(0.1)/1e9:.3f}..{ratio.quantile(0.9)/1e9:.3f}); "
      f"phase B per batch: median {d[:,1].median()*10:.0f} ns, batches {bt.shape[0]}")
