"""Determinism stress of the bf16 V-cycle (ds_twolevel_apply) with two hypothesis lanes on one mesh: each lane applies its
own preconditioner repeatedly on its own stream / thread; every result must equal that lane's solo result bit for bit.
python tools/stress_vcycle.py [cells] [fine_G,coarse_G]"""
import os
import sys
import threading

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from diffsound_amd import meshgen
from diffsound_amd.diffelastic.mesh import TetMesh
from diffsound_amd.lobpcg.modal_solver import SolverConfig, TwoLevelChebyshev
from diffsound_amd.modal_ops import HipModalOps, TetSystem

dev = torch.device("cuda")
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 16
groups = tuple(int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "8,0").split(","))
v, t = meshgen.kuhn_box(cells)
mesh = TetMesh(torch.from_numpy(v).to(dev), torch.from_numpy(t).long().to(dev)).to_high_order(2)
base = TetSystem(mesh.vertices, mesh.tets, 2, 2700.0)
cfg = SolverConfig(block=80, lmax_cap=10.0, smooth_degree=3, coarse_degree=22, coarse_ratio=350.0)
lanes = []
SAME = len(sys.argv) > 4 and sys.argv[4] == "same"
for i, (lam, mu) in enumerate(((2e10, 2e10), (2e10, 2e10) if SAME else (3e10, 1.5e10))):
    sysd = base if i == 0 else base.with_own_values()
    ops = HipModalOps(sysd, lam, mu, mfma_groups=groups)
    pre = TwoLevelChebyshev(ops, cfg)
    R = torch.randn((ops.n, 80), generator=torch.Generator(device=dev).manual_seed(0 if SAME else i), device=dev)
    ref = torch.empty_like(R)
    pre.apply(R.clone(), ref)
    torch.cuda.synchronize()
    lanes.append(dict(ops=ops, pre=pre, R=R, ref=ref, stream=torch.cuda.Stream(), outs=[torch.empty_like(R) for _ in range(12)]))


MODE = sys.argv[3] if len(sys.argv) > 3 else "two"  # two: both lanes apply; noise: lane 1 runs unrelated kernels
big = torch.randn(32 * 1024 * 1024, device=dev)


def work(l):
    with torch.cuda.stream(l["stream"]):
        if MODE == "noise" and l is lanes[1]:
            for _ in range(400):
                big.mul_(1.0000001)
        else:
            for o in l["outs"]:
                l["pre"].apply(l["R"].clone(), o)
    l["stream"].synchronize()


for rnd in range(3):
    th = [threading.Thread(target=work, args=(l,)) for l in lanes]
    for x in th:
        x.start()
    for x in th:
        x.join()
    torch.cuda.synchronize()
    for i, l in enumerate(lanes):
        if MODE == "noise" and i == 1:
            continue
        bad = sum(not torch.equal(o, l["ref"]) for o in l["outs"])
        worst = max(float((o - l["ref"]).abs().max() / l["ref"].abs().max()) for o in l["outs"])
        print(f"round {rnd} lane {i}: {bad} of {len(l['outs'])} differ (max rel diff {worst:.3g})", flush=True)
