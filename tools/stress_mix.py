"""Two different launches side by side on two streams (e.g. the MFMA term with 8-node groups on an ord-2 mesh and with
4-node groups on an ord-1 mesh); every result must equal its solo result.  python tools/stress_mix.py GA GB"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from diffsound_amd import _hip, meshgen
from diffsound_amd.diffelastic.mesh import TetMesh
from diffsound_amd.modal_ops import HipModalOps, TetSystem

dev = torch.device("cuda")
GA, GB = int(sys.argv[1]), int(sys.argv[2])
ncols = 80
L, p = _hip.lib(), _hip.ptr


def make(cells, order, G, seed):
    v, t = meshgen.kuhn_box(cells)
    mesh = TetMesh(torch.from_numpy(v).to(dev), torch.from_numpy(t).long().to(dev)).to_high_order(order)
    sysd = TetSystem(mesh.vertices, mesh.tets, order, 2700.0)
    ops = HipModalOps(sysd, 2e10, 2e10, two_level=False, mfma_groups=(max(G, 0), max(G, 0)))
    g = torch.Generator(device=dev).manual_seed(seed)
    mk = lambda: torch.randn(sysd.n, ncols, generator=g, device=dev).bfloat16()
    c = dict(ops=ops, X=mk(), W=mk(), R=mk(), G=G)
    if G < 0:
        c["Xf"] = c["X"].float()
        c["W"] = c["W"].float()
    return c


def term(c, out):
    if c["G"] < 0:  # fp32 VALU kernels: K X (the eigensolver's product) into an fp32 block
        c["ops"].apply_K(c["Xf"], out)
        return
    out.copy_(c["W"])
    c["ops"].cheb_spmm16(c["X"], out, c["R"], 0.3, 0.7, False)  # MFMA when G > 0, else the VALU kernel


OB = int(sys.argv[3]) if len(sys.argv) > 3 else 1
OA = int(sys.argv[4]) if len(sys.argv) > 4 else 2
A, B = make(20, OA, GA, 1), make(20, OB, GB, 2)
for c in (A, B):
    c["ref"] = torch.empty_like(c["W"])
    term(c, c["ref"])
    c["outs"] = [torch.empty_like(c["W"]) for _ in range(40)]
    c["stream"] = torch.cuda.Stream()
torch.cuda.synchronize()
for rnd in range(int(sys.argv[5]) if len(sys.argv) > 5 else 4):
    for c in (A, B):
        with torch.cuda.stream(c["stream"]):
            for o in c["outs"]:
                term(c, o)
    torch.cuda.synchronize()
    for name, c in (("A", A), ("B", B)):
        bad = [o for o in c["outs"] if not torch.equal(o, c["ref"])]
        msg = ""
        if bad:
            d = (bad[0].float() - c["ref"].float()) != 0
            rows = torch.nonzero(d.any(1)).flatten()
            cols = torch.nonzero(d.any(0)).flatten()
            msg = f"; first bad: {int(d.sum())} elements in {rows.numel()} rows (nodes {sorted(set((rows // 3).tolist()))[:12]}...), cols {cols.tolist()[:8]}... of {cols.numel()}"
        print(f"round {rnd} {name} (G={c['G']}): {len(bad)} of {len(c['outs'])} differ{msg}", flush=True)
