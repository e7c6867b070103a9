#!/usr/bin/env python3
"""The numeric assembly of one hypothesis (ds_assemble_kml on the benchmark mesh: geometry + K_lambda, K_mu, M_s blocks) and the
level transfers of the V-cycle, alone on the device: python tools/mb_assemble.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from diffsound_amd import meshgen  # noqa: E402
from diffsound_amd.diffelastic.mesh import TetMesh  # noqa: E402
from diffsound_amd.pipeline import ModalPipeline  # noqa: E402

sys.argv = [sys.argv[0], "--no-cpu-baseline"]
a = bench.parse()
dev = torch.device("cuda", 0)
v, t = meshgen.kuhn_box(a.cells)
mesh = TetMesh(torch.from_numpy(v).to(dev), torch.from_numpy(t).long().to(dev)).to_high_order(a.order)
pipe = ModalPipeline(mesh.vertices, mesh.tets, a.order, a.modes, bench.MAT, solver_config=bench.solver_config(a))


def timed(f, n=40):
    for _ in range(20):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


print(f"numeric assembly (fine + corner-node level): {timed(pipe.assemble):.1f} us")
pipe.assemble()
pipe.run_pass(bench.MAT[1], bench.MAT[2], backward=False)  # (builds the lane's operators and the level transfers)
ops = pipe.ops
if getattr(ops, "coarse", None) is not None:
    from diffsound_amd import _hip  # noqa: E402

    L, t, c = _hip.lib(), ops._xfer, a.block
    Rf = torch.randn((ops.n, c), device=dev).bfloat16()
    Rc = torch.empty((ops.coarse.n, c), device=dev, dtype=torch.bfloat16)
    Wf = torch.randn((ops.n, c), device=dev).bfloat16()

    def xfer(ptr_, col, w, nrows, X, Y, beta):
        _hip.check(L.ds_scalar_csr_spmm16(ptr_.data_ptr(), col.data_ptr(), w.data_ptr(), nrows, X.data_ptr(), X.stride(0),
                                          Y.data_ptr(), Y.stride(0), c, beta, _hip.stream_ptr()), "ds_scalar_csr_spmm16")

    print(f"restriction ({c} bf16 columns): {timed(lambda: xfer(t['rptr'], t['rcol'], t['rw'], ops.coarse.nv, Rf, Rc, 0.0)):.1f} us")
    print(f"prolongation + add: {timed(lambda: xfer(t['pptr'], t['pcol'], t['pw'], ops.nv, Rc, Wf, 1.0)):.1f} us")
