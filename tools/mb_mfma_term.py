"""Fused bf16 Chebyshev term at C3: VALU kernel (ds_spmm_union16) against the MFMA form (ds_spmm_union16m), event-timed
alone on the device.  python tools/mb_mfma_term.py [cells] [ncols] [group_nodes]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from diffsound_amd import _hip, meshgen
from diffsound_amd.diffelastic.mesh import TetMesh
from diffsound_amd.modal_ops import HipModalOps, TetSystem

dev = torch.device("cuda")
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 26
ncols = int(sys.argv[2]) if len(sys.argv) > 2 else 80
v, t = meshgen.kuhn_box(cells)
mesh = TetMesh(torch.from_numpy(v).to(dev), torch.from_numpy(t).long().to(dev)).to_high_order(2)
sysd = TetSystem(mesh.vertices, mesh.tets, 2, 2700.0)
ops = HipModalOps(sysd, 2e10, 2e10, two_level=False)
mk = lambda: torch.randn(sysd.n, ncols, device=dev).bfloat16()
X, W, R0 = mk(), mk(), mk()
L, p, u, gr = _hip.lib(), _hip.ptr, sysd.groups["union"], sysd.groups
ut = None if u["single"] else p(u["utab"])
G = int(sys.argv[3]) if len(sys.argv) > 3 else 8
mt = sysd.mfma_tables(G)
kc = torch.empty((sysd.nnzb, 3, 4), dtype=torch.bfloat16, device=dev)
print(f"groups of {G}: {mt['ngroups']}, entries {mt['gcol'].numel()}, kc {kc.numel() * 2 / 1e6:.0f} MB, max entries {mt['max_entries']}, max blocks per batch {mt['max_batch_blocks']}")


def pack():
    _hip.check(L.ds_pack_kc(p(ops.k32), p(mt["kperm"]), sysd.nnzb, p(kc), _hip.stream_ptr()), "ds_pack_kc")


def valu():
    _hip.check(L.ds_spmm_union16(1, ut, p(u["ctab"]), u["ngroups"], u["capb"], p(gr["gent"]), p(ops.kgrp), ops.kgrp.shape[0],
                                 sysd.nv, p(X), ncols, p(W), ncols, 0, p(R0), ncols, p(ops.dinv), ncols, 0.3, 0.7, 0, None, 0,
                                 _hip.stream_ptr()), "ds_spmm_union16")


def mfma():
    _hip.check(L.ds_spmm_union16m(1, G, 0, p(mt["gptr"]), p(mt["gcol"]), p(mt["gmeta"]), p(mt["gbase"]), p(kc), sysd.nnzb, mt["ngroups"],
                                  mt["max_entries"], mt["max_batch_blocks"], sysd.nv, p(X), ncols, p(W), ncols, 0, p(R0), ncols, p(ops.dinv), ncols, 0.3, 0.7,
                                  0, None, 0, _hip.stream_ptr()), "ds_spmm_union16m")


def timeit(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for i in range(reps):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(reps))
    return ts[len(ts) // 2], ts[0]


for name, fn in (("pack_kc", pack), ("VALU term", valu), ("MFMA term", mfma)):
    med, best = timeit(fn)
    print(f"{name}: median {med * 1e3:.1f} us, best {best * 1e3:.1f} us", flush=True)
