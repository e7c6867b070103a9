"""Where the cycles of a wave of the fp32 matrix-core product kernel go (diagnostic build -DDS_M32_DIAG of the library,
DS_EXP_LIB=.../libds_m32diag.so): s_memtime stamps summed over the waves of one K X launch at C3, and the clock the chip
holds inside the kernel (s_memtime ticks per 100 MHz s_memrealtime tick)."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from diffsound_amd import _hip, meshgen  # noqa: E402
from diffsound_amd.diffelastic.mesh import TetMesh  # noqa: E402
from diffsound_amd.modal_ops import HipModalOps, TetSystem  # noqa: E402

dev = torch.device("cuda")
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 26
v, t = meshgen.kuhn_box(cells)
mesh = TetMesh(torch.from_numpy(v).to(dev), torch.from_numpy(t).long().to(dev)).to_high_order(2)
sysd = TetSystem(mesh.vertices, mesh.tets, 2, 2700.0)
ops = HipModalOps(sysd, 2e10, 2e10, two_level=False, mfma_groups=(0, 0), mfma32=True)
X, Y = torch.randn(sysd.n, 80, device=dev), torch.empty(sysd.n, 80, device=dev)
L = _hip.lib()
L.ds_m32_diag.restype = ctypes.c_int
L.ds_m32_diag.argtypes = [ctypes.c_void_p, ctypes.c_int]
import numpy as np  # noqa: E402

for epi, name in ((0, "K X"), (3, "M X")):
    for _ in range(20):
        ops._union32(epi, X, Y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops._union32(epi, X, Y)
    e1.record()
    torch.cuda.synchronize()
    n = ops._mfma32["ngroups"]
    rec = np.zeros((n, 8), dtype=np.uint64)
    L.ds_m32_diag(rec.ctypes.data, n)
    r = rec.astype(np.float64)
    t0, t1, rt, wait, a, m, head, tail = r.T
    life = t1 - t0
    clock = life.sum() / rt.sum() * 100.0  # MHz
    us = e0.elapsed_time(e1) * 1e3
    conc = life.sum() / clock / us  # (s_memtime is per XCD: no chip-wide first-start / last-end; the event time stands in)
    print(f"{name}: launch {us:.1f} us by events; in-kernel clock {clock:.0f} MHz; {conc:.0f} waves resident on average "
          f"({conc / 1024:.2f} per SIMD)")
    print(f"   cycles per wave {life.mean():.0f} ({life.mean() / clock:.1f} us): head {head.mean():.0f}, per-batch wait + LDS writes + issue "
          f"{wait.mean():.0f}, fragments + MFMAs {m.mean():.0f}, epilogue {tail.mean():.0f}", flush=True)

# the bf16 fused Chebyshev term on the matrix cores (ds_spmm_union16m, groups of 8 nodes), the benchmark's dominant kernel
from diffsound_amd.modal_ops import MF_BATCH  # noqa: E402

ops8 = HipModalOps(sysd, 2e10, 2e10, two_level=False, mfma_groups=(8, 8))
mk = lambda: torch.randn(sysd.n, 80, device=dev).bfloat16()
Wk, Wp, R0 = mk(), mk(), mk()
L.ds_mf_diag.restype = ctypes.c_int
L.ds_mf_diag.argtypes = [ctypes.c_void_p, ctypes.c_int]
for _ in range(20):
    ops8.cheb_spmm16(Wk, Wp, R0, 0.3, 0.7, False)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
ops8.cheb_spmm16(Wk, Wp, R0, 0.3, 0.7, False)
e1.record()
torch.cuda.synchronize()
n = ops8._mfma["ngroups"]
rec = np.zeros((n, 8), dtype=np.uint64)
L.ds_mf_diag(rec.ctypes.data, n)
t0, t1, rt, wait, a, m, head, tail = rec.astype(np.float64).T
life = t1 - t0
clock = life.sum() / rt.sum() * 100.0
us = e0.elapsed_time(e1) * 1e3
conc = life.sum() / clock / us
print(f"bf16 term (8-node groups, batches of {MF_BATCH}): launch {us:.1f} us by events; in-kernel clock {clock:.0f} MHz; {conc:.0f} waves resident "
      f"on average ({conc / 1024:.2f} per SIMD)")
print(f"   cycles per wave {life.mean():.0f} ({life.mean() / clock:.1f} us): head {head.mean():.0f}, per-batch wait + LDS writes + issue "
      f"{wait.mean():.0f}, fragments + MFMAs {m.mean():.0f}, epilogue {tail.mean():.0f}; batches per group {ops8._mfma['gcol'].numel() / n / MF_BATCH:.1f}",
      flush=True)

# the same kernel on the corner-node level (2 461 groups at C3: every wave resident at once, a launch = one wave lifetime)
opsc = HipModalOps(sysd, 2e10, 2e10).coarse
mkc = lambda: torch.randn(opsc.n, 80, device=dev).bfloat16()
Wk, Wp, R0 = mkc(), mkc(), mkc()
for _ in range(20):
    opsc.cheb_spmm16(Wk, Wp, R0, 0.3, 0.7, False)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
opsc.cheb_spmm16(Wk, Wp, R0, 0.3, 0.7, False)
e1.record()
torch.cuda.synchronize()
n = opsc._mfma["ngroups"]
rec = np.zeros((n, 8), dtype=np.uint64)
L.ds_mf_diag(rec.ctypes.data, n)
t0, t1, rt, wait, a, m, head, tail = rec.astype(np.float64).T
life = t1 - t0
clock = life.sum() / rt.sum() * 100.0
us = e0.elapsed_time(e1) * 1e3
print(f"bf16 term, corner-node level ({n} groups): launch {us:.1f} us by events; in-kernel clock {clock:.0f} MHz; cycles per wave {life.mean():.0f} "
      f"({life.mean() / clock:.1f} us; longest {life.max() / clock:.1f} us): head {head.mean():.0f}, per-batch wait + LDS writes + issue {wait.mean():.0f}, "
      f"fragments + MFMAs {m.mean():.0f}, epilogue {tail.mean():.0f}; batches per group {opsc._mfma['gcol'].numel() / n / MF_BATCH:.1f}", flush=True)
# start of each wave relative to the earliest wave of its XCD is not comparable across XCDs (s_memtime is per XCD); the spread of the
# starts inside the launch = how long the dispatcher takes to place the grid
