"""Micro-benchmark of the BSR-3 SpMM on the benchmark mesh: node orderings x block widths."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffsound_amd import meshgen
from diffsound_amd.diffelastic.mesh import TetMesh
from diffsound_amd.modal_ops import TetSystem, HipModalOps
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 26
dev = torch.device('cuda')
v, t = meshgen.kuhn_box(cells)
mesh = TetMesh(torch.from_numpy(v).to(dev), torch.from_numpy(t).long().to(dev)).to_high_order(2)
for reorder in (True,):
    sysd = TetSystem(mesh.vertices, mesh.tets, 2, 2700.0, reorder=reorder)
    ops = HipModalOps(sysd, 2e10, 2e10)
    for ncols in (80, 240):
        X = torch.randn(sysd.n, ncols, device=dev); Y = torch.empty_like(X)
        for name, fn in (('K', ops.apply_K), ('M', ops.apply_M)):
            fn(X, Y); torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): fn(X, Y)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 20
            vb = 36 if name == 'K' else 4
            bytes_ = sysd.nnzb * (vb + 4) + (sysd.nv + 1) * 4 + 2 * sysd.n * ncols * 4
            print(f"reorder={reorder} {name} ncols={ncols}: {ms:.3f} ms  {bytes_/ms/1e6:.0f} GB/s algorithmic")

    # fused Chebyshev term, tiled vs untiled
    ncols = 80
    W = torch.randn(sysd.n, ncols, device=dev); Wp = torch.randn_like(W); R0 = torch.randn_like(W)
    for label, fn in (("tiled" if sysd.tiles else ("grouped" if sysd.groups else "untiled"), lambda: ops._cheb_spmm_launch(W, Wp, R0, 0.3, 0.7, False)),):
        fn(); torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        bytes_ = sysd.nnzb * 40 + (sysd.nv + 1) * 4 + sysd.nv * 36 + 4 * sysd.n * ncols * 4
        print(f"fused cheb term ({label}) ncols={ncols}: {ms:.3f} ms  {bytes_/ms/1e6:.0f} GB/s algorithmic; tiles={sysd.tiles['ntiles'] if sysd.tiles else 0}")
