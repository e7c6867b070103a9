"""Gather probe on the real C3 union tables (tools/gather_probe.hip): time to pull every panel of every group into
registers, for the production layouts and for 16-byte-per-lane / two-panels-per-instruction bf16 loads, and for
workgroup-wide 16-node unions.  python tools/mb_gather_probe.py [cells]"""
import ctypes
import os
import subprocess
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
from diffsound_amd import _hip, meshgen  # noqa: E402
from diffsound_amd.diffelastic.mesh import TetMesh  # noqa: E402
from diffsound_amd.modal_ops import UNION_CAP, TetSystem  # noqa: E402

so = os.path.join(HERE, "gather_probe.so")
if not os.path.exists(so):
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", os.path.join(HERE, "gather_probe.hip"), "-o", so])
L = ctypes.CDLL(so)
L.gather_probe.restype = ctypes.c_float
L.gather_probe.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint, ctypes.c_void_p, ctypes.c_uint,
                           ctypes.c_int, ctypes.c_void_p, ctypes.c_int]

cells = int(sys.argv[1]) if len(sys.argv) > 1 else 26
dev = torch.device("cuda:0")
v, t = meshgen.kuhn_box(cells)
m = TetMesh(torch.from_numpy(v).to(dev), torch.from_numpy(t).long().to(dev)).to_high_order(2)
s = TetSystem(m.vertices, m.tets, 2, 2700.0)
pat = _hip.DevicePattern(s.tets, s.nv, UNION_CAP)
nv, ld = s.nv, 80
X16 = torch.randn((3 * nv, ld), device=dev).to(torch.bfloat16).contiguous()
X32 = torch.randn((3 * nv, ld), device=dev).contiguous()
out = torch.zeros(pat.ngroups + 8, dtype=torch.int32, device=dev)
ne = pat.ne
print(f"nv {nv}, groups of 4: {pat.ngroups}, union entries {ne} ({ne / pat.nnzb:.3f} of the blocks)")


L.gather_probe7.restype = ctypes.c_float
L.gather_probe7.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint, ctypes.c_void_p,
                            ctypes.c_uint, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]


def run7(depth, remap, colmask, gptr, gent, ng, X, panels, label):
    ms = L.gather_probe7(depth, remap, colmask, gptr.data_ptr(), gent.data_ptr(), ng, X.data_ptr(), X.numel() * X.element_size(), ld,
                         out.data_ptr(), 5)
    gb = panels * 960 / 1e9
    print(f"mode 7 depth {depth:2d} remap {remap} table {'all' if not colmask else colmask + 1:>6} panels, {label}: {ms * 1e3:.1f} us, "
          f"{gb / ms:.2f} TB/s gathered", flush=True)


def run(mode, gptr, gent, ng, X, panels, pbytes, label):
    ms = L.gather_probe(mode, gptr.data_ptr(), gent.data_ptr(), ng, X.data_ptr(), X.numel() * X.element_size(), ld, out.data_ptr(), 5)
    gb = panels * pbytes / 1e9
    print(f"mode {mode} {label}: {ms * 1e3:.1f} us, {gb / ms:.2f} TB/s gathered ({gb * 1e3:.0f} MB)", flush=True)


run(0, pat.gptr, pat.gent, pat.ngroups, X16, ne, 480, "bf16 panels, 8 B/lane, 1 panel/instr  ")
run(1, pat.gptr, pat.gent, pat.ngroups, X16, ne, 480, "bf16 panels, 16 B/lane, 2 panels/instr")
run(2, pat.gptr, pat.gent, pat.ngroups, X32, ne, 960, "fp32 panels, 16 B/lane, 1 panel/instr ")
# 16-node groups: union over 16 consecutive rows
rows = torch.repeat_interleave(torch.arange(nv, device=dev), (pat.rowptr[1:] - pat.rowptr[:-1]).long())
for G in (8, 16, 32):
    key = torch.unique((rows // G) * nv + pat.colidx.long())
    ng = (nv + G - 1) // G
    gptr = torch.searchsorted(key // nv, torch.arange(ng + 1, device=dev)).to(torch.int32).contiguous()
    gent = (key % nv).to(torch.int32).contiguous()
    run(3, gptr, gent, ng, X16, key.numel(), 480, f"bf16, 16 B/lane, workgroup-wide union of {G} nodes ({key.numel() / pat.nnzb:.3f})")
# fp32 panels, one wave per group of G nodes: the production load (16 B per lane, one instruction per panel) against the
# outer-product layout (one column per lane: 3 + 1 dword loads per panel)
run(5, pat.gptr, pat.gent, pat.ngroups, X32, ne, 960, "fp32 panels, 4-node unions, 4 dword loads per panel")
for G in (8, 16):
    key = torch.unique((rows // G) * nv + pat.colidx.long())
    ng = (nv + G - 1) // G
    gptr = torch.searchsorted(key // nv, torch.arange(ng + 1, device=dev)).to(torch.int32).contiguous()
    gent = (key % nv).to(torch.int32).contiguous()
    run(2, gptr, gent, ng, X32, key.numel(), 960, f"fp32 panels, one wave per {G}-node union, 16 B/lane")
    run(5, gptr, gent, ng, X32, key.numel(), 960, f"fp32 panels, one wave per {G}-node union, 4 dword loads per panel")
    run(6, gptr, gent, ng, X32, key.numel(), 960, f"fp32 panels, two waves (40 columns each) per {G}-node union, 3 dword loads per half panel")

# the knobs of the fp32 load: loads in flight, XCD-contiguous group ranges, and the ceiling of the load shape (small tables)
if os.environ.get("PROBE_KNOBS", "1") == "1":
    for G in (4, 8, 16):
        if G == 4:
            gptr, gent, ng, n_ent = pat.gptr, pat.gent, pat.ngroups, ne
        else:
            key = torch.unique((rows // G) * nv + pat.colidx.long())
            ng = (nv + G - 1) // G
            gptr = torch.searchsorted(key // nv, torch.arange(ng + 1, device=dev)).to(torch.int32).contiguous()
            gent = (key % nv).to(torch.int32).contiguous()
            n_ent = key.numel()
        for depth in (4, 8, 16):
            for remap in (0, 1):
                run7(depth, remap, 0, gptr, gent, ng, X32, n_ent, f"{G}-node unions")
        for colmask in (2047, 32767):
            run7(8, 1, colmask, gptr, gent, ng, X32, n_ent, f"{G}-node unions")
