"""K X / M X (VALU neighbour-union kernel, C3, 80 columns) against the LAYOUT of its operands: compact (n x 80) blocks or column
ranges of wider buffers (rows 992 B / 1 024 B apart).  Variants are timed interleaved, three rounds, after a two-second warm-up
(a cold device clocks differently: single measurements taken one after the other are not comparable)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from diffsound_amd import meshgen  # noqa: E402
from diffsound_amd.diffelastic.mesh import TetMesh  # noqa: E402
from diffsound_amd.modal_ops import HipModalOps, TetSystem  # noqa: E402

dev = torch.device("cuda")
v, t = meshgen.kuhn_box(26)
mesh = TetMesh(torch.from_numpy(v).to(dev), torch.from_numpy(t).long().to(dev)).to_high_order(2)
sysd = TetSystem(mesh.vertices, mesh.tets, 2, 2700.0)
ops = HipModalOps(sysd, 2e10, 2e10, two_level=False, mfma_groups=(0, 0))
n = sysd.n


def block(ld, c0):
    return torch.randn(n, ld, device=dev)[:, c0:c0 + 80] if ld > 80 else torch.randn(n, 80, device=dev)


cases = {}
for xl, xc in ((80, 0), (248, 168), (256, 168)):
    for yl, yc in ((80, 0), (240, 160), (256, 160)):
        cases[f"X ld {xl:3d} -> Y ld {yl:3d}"] = (block(xl, xc), block(yl, yc))
t0 = time.time()
while time.time() - t0 < 2.0:  # warm-up
    for X, Y in cases.values():
        ops._union(0, X, Y)
torch.cuda.synchronize()
res = {k: {0: [], 3: []} for k in cases}
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for rnd in range(3):
    for epi in (0, 3):
        for k, (X, Y) in cases.items():
            ops._union(epi, X, Y)
            e0.record()
            for _ in range(20):
                ops._union(epi, X, Y)
            e1.record()
            torch.cuda.synchronize()
            res[k][epi].append(e0.elapsed_time(e1) / 20 * 1e3)
for k in cases:
    print(f"{k}:  K X {' / '.join(f'{x:.1f}' for x in res[k][0])} us   M X {' / '.join(f'{x:.1f}' for x in res[k][3])} us", flush=True)
