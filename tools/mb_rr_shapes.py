"""ONE Rayleigh-Ritz kernel shape of the eigensolver's iteration, launched a few times on the benchmark mesh's row count and the
solver's operand layout (for the rocprofv3 --pmc passes behind profiles/gram_mix_mfma_util.json, and for timing):
    python tools/mb_rr_shapes.py gram_256x160 [reps]     shapes: gram_PxQ = [first P columns of S]^T [first Q columns of KS], mix_PxQ = S[:, :P] C -> S2"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffsound_amd.modal_ops import _HipBlockOps
shape = sys.argv[1] if len(sys.argv) > 1 else "gram_256x160"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
kind, pq = shape.split("_")
p, q = (int(x) for x in pq.split("x"))
dev = torch.device("cuda")
nv = 148877
n = 3 * nv
ops = _HipBlockOps(); ops._init_common(None, None, nv, dev)
S, KS, S2 = torch.randn(n, 256, device=dev), torch.randn(n, 256, device=dev), torch.empty(n, 256, device=dev)
C = torch.randn(p, q, device=dev) / p
fn = (lambda: ops.gram(S[:, :p], KS[:, :q])) if kind == "gram" else (lambda: ops.mix(S[:, :p], C, S2[:, 16:16 + q]))
for _ in range(3): fn()
torch.cuda.synchronize(); t0 = time.time()
for _ in range(reps): fn()
torch.cuda.synchronize()
print(f"{shape}: {(time.time() - t0) / reps * 1e3:.3f} ms per call ({reps} calls)", flush=True)
