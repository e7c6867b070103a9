"""Victim-side matrix for the gfx950 finding of DESIGN.md 5b: which AGGRESSOR makes which VICTIM
return results that differ from its solo results, when the two run side by side on two streams?

  aggressors  register-only spin kernels of tests/probes/mfma_probe.hip (no memory, no LDS inside the loop):
              x32 = v_mfma_f32_16x16x32_bf16, x16 = v_mfma_f32_16x16x16_bf16, f32 = v_mfma_f32_16x16x4_f32, fma = v_fma_f32
  victims     (1) register-only chains of v_pk_fma_f32 / of v_fma_f32 (tests/probes/mfma_probe.hip);
              (2) the production kernels that accumulate with inline-asm v_pk_fma_f32: the eigensolver's fp32 K X and the
                  bf16 Chebyshev term of the VALU neighbour-union kernel (libdiffsound_hip.so);
              (3) the production MFMA term itself.
Every victim launch is compared bit for bit with the victim's solo result.   python tools/mfma_interference.py [rounds]"""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from diffsound_amd import _hip, meshgen
from diffsound_amd.diffelastic.mesh import TetMesh
from diffsound_amd.modal_ops import HipModalOps, TetSystem

dev = torch.device("cuda")
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
P = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "probes", "libmfma_probe.so"))
P.probe_mfma_spin.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
P.probe_fma_chain.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
ncu = torch.cuda.get_device_properties(0).multi_processor_count
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
FORMS = (("x32", 32), ("x16", 16), ("f32", 4), ("fma", 0))
spin_out = torch.empty(ncu * 8 * 64, device=dev)


def spin(form, iters, waves_per_cu, stream):
    rc = P.probe_mfma_spin(form, iters, ncu * waves_per_cu, spin_out.data_ptr(), stream.cuda_stream)
    assert rc == 0, rc


def calibrate(form, waves_per_cu, target_ms=12.0):
    iters = 20000
    for _ in range(2):
        torch.cuda.synchronize()
        t0 = time.time()
        spin(form, iters, waves_per_cu, sB)
        torch.cuda.synchronize()
        ms = (time.time() - t0) * 1e3
        iters = max(1000, int(iters * target_ms / max(ms, 1e-3)))
    return iters


def side_by_side(name, victim_launch, outs, ref, waves_per_cu=4):
    """victim_launch(out) on stream A, the aggressor spinning on stream B; returns a row of the matrix."""
    row = {}
    for fname, form in FORMS:
        iters = calibrate(form, waves_per_cu)
        bad = total = 0
        for _ in range(rounds):
            spin(form, iters, waves_per_cu, sB)
            with torch.cuda.stream(sA):
                for o in outs:
                    victim_launch(o)
            torch.cuda.synchronize()
            bad += sum(0 if torch.equal(o, ref) else 1 for o in outs)
            total += len(outs)
        row[fname] = (bad, total)
        print(f"  victim {name:34s} beside {fname}: {bad} of {total} launches differ from the solo result", flush=True)
    return row


print(f"device: {torch.cuda.get_device_name(0)}, {ncu} CUs; {rounds} rounds per cell")
# ---------------------------------------------------------------- (1) register-only victims
seed = torch.rand(4096, device=dev) - 0.5
nblk, citers = ncu * 4, 3000
for packed, name in ((1, "register chain v_pk_fma_f32"), (0, "register chain v_fma_f32")):
    ref = torch.empty(nblk * 256 * 12, device=dev)

    def launch(o, packed=packed):
        rc = P.probe_fma_chain(packed, citers, nblk, seed.data_ptr(), o.data_ptr(), torch.cuda.current_stream().cuda_stream)
        assert rc == 0, rc

    launch(ref)
    torch.cuda.synchronize()
    assert torch.isfinite(ref).all()
    outs = [torch.empty_like(ref) for _ in range(40)]
    for wpc in (4, 8):
        side_by_side(f"{name} ({wpc} spin waves/CU)", launch, outs, ref, wpc)


# ---------------------------------------------------------------- (2), (3) production victims
def make(cells, order, G, seed_):
    v, t = meshgen.kuhn_box(cells)
    mesh = TetMesh(torch.from_numpy(v).to(dev), torch.from_numpy(t).long().to(dev)).to_high_order(order)
    sysd = TetSystem(mesh.vertices, mesh.tets, order, 2700.0)
    ops = HipModalOps(sysd, 2e10, 2e10, two_level=False, mfma_groups=(max(G, 0), max(G, 0)))
    g = torch.Generator(device=dev).manual_seed(seed_)
    mk = lambda: torch.randn(sysd.n, 80, generator=g, device=dev).bfloat16()
    c = dict(ops=ops, X=mk(), W=mk(), R=mk(), G=G)
    if G < 0:
        c["Xf"], c["W"] = c["X"].float(), c["W"].float()
    return c


def term(c, out):
    if c["G"] < 0:
        c["ops"].apply_K(c["Xf"], out)
        return
    out.copy_(c["W"])
    c["ops"].cheb_spmm16(c["X"], out, c["R"], 0.3, 0.7, False)


for cells, order, G, name in ((20, 2, -1, "production fp32 K X (VALU union)"), (20, 1, 0, "production bf16 term (VALU union)"),
                              (20, 2, 8, "production bf16 term (MFMA x16)")):
    c = make(cells, order, G, 7)
    ref = torch.empty_like(c["W"])
    term(c, ref)
    torch.cuda.synchronize()
    outs = [torch.empty_like(ref) for _ in range(40)]
    side_by_side(name, lambda o, c=c: term(c, o), outs, ref, 4)
print("done")
