#!/usr/bin/env python3
"""ds_mix64 (the fp64 refinement's dense updates) against the torch.mm / addmm chain it replaced, at configs[4]'s shapes:
python tools/mb_mix64.py [n] [b] [na]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffsound_amd import _hip  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4136393 // 3 * 3
b = int(sys.argv[2]) if len(sys.argv) > 2 else 136
na = int(sys.argv[3]) if len(sys.argv) > 3 else 64
dev = torch.device("cuda:0")
L = _hip.lib()
g = torch.Generator(device=dev).manual_seed(1)
f64 = dict(dtype=torch.float64, device=dev)
Y = torch.randn((n, 6), generator=g, **f64)
X = torch.randn((n, b), generator=g, **f64)
P = torch.randn((n, na), generator=g, **f64)
W = torch.randn((n, na), generator=g, **f64)
m = 6 + b + 2 * na
Z = torch.randn((m, b), generator=g, **f64)
Tp = torch.randn((m, na), generator=g, **f64)
offs = [0, 6, 6 + b, 6 + b + na]


def mix64(items, C, out):
    arr = (_hip.Block64 * len(items))()
    for d, (blk, r) in zip(arr, items):
        d.a, d.lda, d.p, d.offset = blk.data_ptr(), blk.stride(0), blk.shape[1], r
    _hip.check(L.ds_mix64(len(items), ctypes.addressof(arr), C.data_ptr(), C.stride(0), C.shape[1], out.data_ptr(), out.stride(0),
                          n, 1.0, 0.0, _hip.stream_ptr()), "ds_mix64")


def timed(f, reps=5):
    f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


Xn, Pn = torch.empty((n, b), **f64), torch.empty((n, na), **f64)
blocks = [Y, X, P, W]


def new_way():
    mix64([(blk, offs[i]) for i, blk in enumerate(blocks)], Z, Xn)
    mix64([(blk, offs[i]) for i, blk in enumerate(blocks) if i != 1], Tp, Pn)


def old_way():
    acc = torch.mm(Y, Z[0:6])
    acc.addmm_(P, Z[offs[2]:offs[3]])
    acc.addmm_(W, Z[offs[3]:])
    x = torch.addmm(acc, X, Z[6:6 + b])
    return x, acc[:, :na] * 1.0


t_new, t_old = timed(new_way), timed(old_way)
fl = 2.0 * n * (m * b + (m - b) * na)
by = 8.0 * n * ((6 + b + 2 * na) + b + (6 + 2 * na) + na)
print(f"n = {n}, b = {b}, active = {na}: one set of updates (X' and P' of one of the three families)")
print(f"  ds_mix64 (two launches): {t_new:8.3f} ms   {fl / t_new / 1e9:6.1f} TF/s fp64   {by / t_new / 1e6:7.0f} GB/s of minimal traffic")
print(f"  torch mm/addmm chain   : {t_old:8.3f} ms")
x_ref, _ = old_way()
new_way()
print(f"  max |difference| / max |X'| = {float((Xn - x_ref).abs().max() / x_ref.abs().max()):.2e}")
