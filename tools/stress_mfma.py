"""Determinism stress of ds_spmm_union16m: the same launch repeated on one stream while a second stream keeps the device
busy with other launches; every result must equal the solo result bit for bit.  python tools/stress_mfma.py [cells] [G] [order]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from diffsound_amd import _hip, meshgen
from diffsound_amd.diffelastic.mesh import TetMesh
from diffsound_amd.modal_ops import HipModalOps, TetSystem

dev = torch.device("cuda")
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 26
G = int(sys.argv[2]) if len(sys.argv) > 2 else 8
order = int(sys.argv[3]) if len(sys.argv) > 3 else 2
ncols = 80
v, t = meshgen.kuhn_box(cells)
mesh = TetMesh(torch.from_numpy(v).to(dev), torch.from_numpy(t).long().to(dev)).to_high_order(order)
sysd = TetSystem(mesh.vertices, mesh.tets, order, 2700.0)
ops = HipModalOps(sysd, 2e10, 2e10, two_level=False, mfma_groups=(G, G))
mt = ops._mfma
L, p = _hip.lib(), _hip.ptr
mk = lambda: torch.randn(sysd.n, ncols, device=dev).bfloat16()
X, W0, R0 = mk(), mk(), mk()
X2, W2, R2 = mk(), mk(), mk()
Xf2, Wf2, Rf2 = X2.float(), W2.float(), R2.float()


INPLACE = len(sys.argv) > 4 and sys.argv[4] == "inplace"


def mfma(Xa, Wa, Ra, out):
    if INPLACE:
        out.copy_(Wa)
    _hip.check(L.ds_spmm_union16m(1, G, p(mt["gptr"]), p(mt["gcol"]), p(mt["gmeta"]), p(mt["gbase"]), p(ops.kc), sysd.nnzb,
                                  mt["ngroups"], mt["max_entries"], mt["max_batch_blocks"], sysd.nv, p(Xa), ncols, p(out), ncols, 0,
                                  p(Ra), ncols, p(ops.dinv), ncols, 0.3, 0.7, 0, None if INPLACE else p(Wa), 0 if INPLACE else ncols,
                                  _hip.stream_ptr()), "ds_spmm_union16m")


ref = torch.empty_like(W0)
mfma(X, W0, R0, ref)
torch.cuda.synchronize()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
outs = [torch.empty_like(W0) for _ in range(40)]
junk = torch.empty_like(W0)
big = torch.randn(64 * 1024 * 1024, device=dev)
for rnd in range(3):
    with torch.cuda.stream(s2):
        for i in range(60):
            if rnd == 0:
                mfma(X2, W2, R2, junk)
            elif rnd == 1:
                ops._cheb_spmm_launch(Xf2, Wf2, Rf2, 0.3, 0.7, False)  # the VALU kernel (static LDS, 4 waves per workgroup)
            else:
                torch.mm(big[:4096 * 4096].view(4096, 4096), big[:4096 * 4096].view(4096, 4096))
    with torch.cuda.stream(s1):
        for o in outs:
            mfma(X, W0, R0, o)
    torch.cuda.synchronize()
    bad = [i for i, o in enumerate(outs) if not torch.equal(o, ref)]
    worst = max((float((o.float() - ref.float()).abs().max()) for o in outs), default=0.0)
    print(f"round {rnd}: {len(bad)} of {len(outs)} results differ from the solo result (max abs diff {worst:.3g})", flush=True)
