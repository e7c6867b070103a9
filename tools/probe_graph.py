"""Feasibility probe: capture ctypes-launched kernels (ds_cheb_init + ds_spmm_union terms) into a HIP graph from two
host threads at once (thread-local capture mode), replay, compare with eager, time both.

Result on this image (torch 2.10 / ROCm 7): FAILS as written - `torch.cuda.graph.__enter__` and `__exit__` call the
device-wide `torch.cuda.synchronize()`, which is "not permitted when stream is capturing" as soon as ANOTHER thread
is inside its own capture, and the failed capture poisons the stream ("previous error during capture").  A lane-
parallel pipeline has to drive `CUDAGraph.capture_begin / capture_end` itself (no device-wide calls in between, other
lanes' pageable copies and allocations kept out of the capture window) - next round."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffsound_amd import meshgen
from diffsound_amd.diffelastic.mesh import TetMesh
from diffsound_amd.modal_ops import TetSystem, HipModalOps
dev = torch.device('cuda')
v, t = meshgen.kuhn_box(12)
mesh = TetMesh(torch.from_numpy(v).to(dev), torch.from_numpy(t).long().to(dev)).to_high_order(2)
res = {}
def work(i):
    torch.cuda.set_device(0)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        sysd = TetSystem(mesh.vertices, mesh.tets, 2, 2700.0)
        ops = HipModalOps(sysd, 2e10 * (1 + i), 2e10, two_level=False)
        R = torch.randn(sysd.n, 80, device=dev); W = torch.zeros_like(R); D = torch.zeros_like(R); AD = torch.zeros_like(R)
        def seq():
            ops.cheb_init(R, AD, W, 0.3)
            cur, oth = W, D
            for k in range(6):
                ops.cheb_spmm(cur, oth, R, 0.2, 0.1, first=(k == 0))
                cur, oth = oth, cur
        seq(); s.synchronize(); ref = W.clone(); refD = D.clone()
        W.zero_(); D.zero_()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
            seq()
        g.replay(); s.synchronize()
        ok = torch.equal(W, ref) and torch.equal(D, refD)
        t0 = time.time()
        for _ in range(50): seq()
        s.synchronize(); te = (time.time() - t0) / 50
        t0 = time.time()
        for _ in range(50): g.replay()
        s.synchronize(); tg = (time.time() - t0) / 50
        res[i] = (ok, te * 1e3, tg * 1e3)
ths = [threading.Thread(target=work, args=(i,)) for i in range(2)]
[t.start() for t in ths]; [t.join() for t in ths]
print(res)
