"""Time the eigensolver's fp32 products alone on the device at C3 (80 columns): K X, M X and the fp32 fused Chebyshev term.
python tools/mb_kx.py [cells]   (DS_EXP_LIB selects an experimental build of the library)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from diffsound_amd import meshgen  # noqa: E402
from diffsound_amd.diffelastic.mesh import TetMesh  # noqa: E402
from diffsound_amd.modal_ops import HipModalOps, TetSystem  # noqa: E402

dev = torch.device("cuda")
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 26
v, t = meshgen.kuhn_box(cells)
mesh = TetMesh(torch.from_numpy(v).to(dev), torch.from_numpy(t).long().to(dev)).to_high_order(2)
sysd = TetSystem(mesh.vertices, mesh.tets, 2, 2700.0)
ops = HipModalOps(sysd, 2e10, 2e10, two_level=False, mfma_groups=(0, 0), mfma32=True)
X, Y = torch.randn(sysd.n, 80, device=dev), torch.empty(sysd.n, 80, device=dev)
W, R0 = torch.randn(sysd.n, 80, device=dev), torch.randn(sysd.n, 80, device=dev)


def timeit(fn, label, reps=30):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"{label}: {e0.elapsed_time(e1) / reps * 1e3:.1f} us", flush=True)


timeit(lambda: ops._union(0, X, Y), "K X  (80 columns, fp32, VALU union kernel)")
timeit(lambda: ops._union(3, X, Y), "M X  (80 columns, fp32, VALU union kernel)")
if ops._mfma32 is not None:
    timeit(lambda: ops._union32(0, X, Y), "K X  (80 columns, fp32, matrix cores)")
    timeit(lambda: ops._union32(3, X, Y), "M X  (80 columns, fp32, matrix cores)")
    m4 = ops._mfma32
    print("4-node unions: entries", m4["gcol"].numel(), "blocks", sysd.nnzb, "max entries / group", m4["max_entries"],
          "max blocks / batch", m4["max_batch_blocks"], flush=True)
timeit(lambda: ops._cheb_spmm_launch(X, W, R0, 0.3, 0.7, False), "fused Chebyshev term (fp32)")

# the same products on column ranges of a wider buffer, as the solver calls them: leading dimension 248 (its [Y | X | P | W]
# basis: rows 992 bytes apart) and 256 (rows 1 024 bytes apart: every panel row starts at the same offset inside a cache line)
for ld in (248, 256):
    big = torch.randn(sysd.n, ld, device=dev)
    for c0 in (8, 168):
        Xs = big[:, c0:c0 + 80]
        timeit(lambda: ops._union(0, Xs, Y), f"K X  (80 columns at column {c0} of a {ld}-column buffer, VALU)")
        timeit(lambda: ops._union(3, Xs, Y), f"M X  (80 columns at column {c0} of a {ld}-column buffer, VALU)")
