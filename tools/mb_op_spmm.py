"""The eigensolver's fp32 products at C3: the 4-node VALU union kernel against the outer-product MFMA kernel on 8-node
unions (ds_spmm_union8): parity and time, alone on the device.  python tools/mb_op_spmm.py [cells] [order]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from diffsound_amd import _hip, meshgen
from diffsound_amd.diffelastic.mesh import TetMesh
from diffsound_amd.modal_ops import HipModalOps, TetSystem

dev = torch.device("cuda")
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 26
order = int(sys.argv[2]) if len(sys.argv) > 2 else 2
v, t = meshgen.kuhn_box(cells)
mesh = TetMesh(torch.from_numpy(v).to(dev), torch.from_numpy(t).long().to(dev)).to_high_order(order)
sysd = TetSystem(mesh.vertices, mesh.tets, order, 2700.0)
ops = HipModalOps(sysd, 2e10, 2e10, two_level=False)
L, p = _hip.lib(), _hip.ptr
mt = sysd.mfma_tables(8)
kop = torch.empty((sysd.nnzb * 9 + 1024,), device=dev)  # (+ 4 KiB: the LDS-DMA staging reads whole 1 KiB pieces)
mop = torch.empty((sysd.nnzb + 1024,), device=dev)
_hip.check(L.ds_pack_op(p(ops.k32), p(ops.ms32), p(mt["kperm"]), sysd.nnzb, p(kop), p(mop), _hip.stream_ptr()), "ds_pack_op")
print(f"nv {sysd.nv}, groups {mt['ngroups']}, entries {mt['gcol'].numel()}, max entries {mt['max_entries']}, max group blocks {mt['max_group_blocks']}")


def op(kind, X, Y):
    _hip.check(L.ds_spmm_union8(kind, p(mt["gptr"]), p(mt["gcol"]), p(mt["gmeta"]), p(mt["gbase"]), p(kop if kind == 0 else mop),
                                sysd.nnzb, mt["ngroups"], mt["max_entries"], mt["max_group_blocks"], sysd.nv, p(X), X.stride(0),
                                p(Y), Y.stride(0), X.shape[1], _hip.stream_ptr()), "ds_spmm_union8")


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
    return ts[len(ts) // 2] * 1e3


for c in (80, 84, 64, 40, 68, 4):
    Xw = torch.randn((sysd.n, c + 8), device=dev)
    X = Xw[:, 4:4 + c]  # a column range of a wider block (leading dimension != ncols)
    for kind, name, ref in ((0, "K X", ops.apply_K), (3, "M X", ops.apply_M)):
        Y0, Y1 = torch.empty((sysd.n, c), device=dev), torch.full((sysd.n, c), float("nan"), device=dev)
        ref(X, Y0)
        op(kind, X, Y1)
        err = float((Y1 - Y0).abs().max() / Y0.abs().max())
        t0, t1 = timeit(lambda: ref(X, Y0)), timeit(lambda: op(kind, X, Y1))
        print(f"{name} {c:3d} columns: max rel difference {err:.2e}; union4 {t0:.1f} us, outer-product union8 {t1:.1f} us", flush=True)
