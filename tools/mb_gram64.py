#!/usr/bin/env python3
"""ds_gram64_blocks ([Y X P W]^T [K W | M W] of one refinement step in one pass) against the eight ds_gram calls it replaces,
at configs[4]'s shapes: python tools/mb_gram64.py [n] [b] [na]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffsound_amd import _hip  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4102893
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
KW = torch.randn((n, na), generator=g, **f64)
MW = torch.randn((n, na), generator=g, **f64)
S = [Y, X, P, W]
m = 6 + b + 2 * na
ws = torch.empty((L.ds_gram_workspace_bytes(n, m, max(2 * na, m)),), dtype=torch.uint8, device=dev)


def table(blocks):
    arr, off = (_hip.Block64 * len(blocks))(), 0
    for d, blk in zip(arr, blocks):
        d.a, d.lda, d.p, d.offset = blk.data_ptr(), blk.stride(0), blk.shape[1], off
        off += blk.shape[1]
    return arr, off


def blocks_call(A, B, sym=0):
    (ta, p), (tb, q) = table(A), table(B)
    G = torch.empty((p, q), **f64)
    _hip.check(L.ds_gram64_blocks(len(A), ctypes.addressof(ta), len(B), ctypes.addressof(tb), n, sym, G.data_ptr(), ws.data_ptr(),
                                  ws.numel(), _hip.stream_ptr()), "ds_gram64_blocks")
    return G


def pair_call(A, B):
    G = torch.empty((A.shape[1], B.shape[1]), **f64)
    _hip.check(L.ds_gram(A.data_ptr(), 1, A.stride(0), A.shape[1], B.data_ptr(), 1, B.stride(0), B.shape[1], n, 0, G.data_ptr(),
                         ws.data_ptr(), ws.numel(), _hip.stream_ptr()), "ds_gram")
    return G


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


t_new = timed(lambda: blocks_call(S, [KW, MW]))
t_old = timed(lambda: [pair_call(a_, w_) for a_ in S for w_ in (KW, MW)])
fl = 2.0 * n * m * 2 * na
by = 8.0 * n * (m + 2 * na)
print(f"n = {n}, S = [6 | {b} | {na} | {na}] against [K W | M W] ({2 * na} columns)")
print(f"  ds_gram64_blocks, one launch : {t_new:8.3f} ms   {fl / t_new / 1e9:6.1f} TF/s fp64   {by / t_new / 1e6:7.0f} GB/s of minimal traffic")
print(f"  eight ds_gram calls          : {t_old:8.3f} ms")
KS = [torch.randn((n, x.shape[1]), generator=g, **f64) for x in S]
t_full = timed(lambda: blocks_call(S, KS, 1), 3)
t_full_old = timed(lambda: [pair_call(S[i], KS[j]) for i in range(4) for j in range(i, 4)], 3)
print(f"  all pairs, symmetric ({m} x {m}): one launch {t_full:8.3f} ms, ten ds_gram calls {t_full_old:8.3f} ms")
Gn = blocks_call(S, [KW, MW])
Go = torch.cat([torch.cat([pair_call(a_, w_) for w_ in (KW, MW)], 1) for a_ in S], 0)
print(f"  max |difference| / max |G| = {float((Gn - Go).abs().max() / Go.abs().max()):.2e}")
