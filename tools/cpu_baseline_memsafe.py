#!/usr/bin/env python3
"""the MEMORY-SAFE restatement of the reference's loop body, measured at a stated mesh size on host
cores ("optimised CPU baseline": element matrices pre-summed over the Gauss points, oracle.fem.assemble_stiffness; the
read-out's matrix-free K(theta) U evaluated eight modes at a time under activation checkpointing - same arithmetic as the faithful
restatement bench.py times at 8^3, which needs ~25 GB of COO triplets and ~30 GB of read-out activations at the benchmark mesh).
    python tools/cpu_baseline_memsafe.py CELLS [THREADS] > gpurun_out/rXX_cpu_memsafe_CELLS.json
TEST / MEASUREMENT INFRASTRUCTURE: uses oracle/, never part of the product path."""
import json
import os
import resource
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.utils.checkpoint as ckpt  # noqa: E402

from bench import MAT  # noqa: E402
from diffsound_amd import meshgen  # noqa: E402
from oracle import fem, modal  # noqa: E402
from oracle import oscillator as oosc  # noqa: E402

cells = int(sys.argv[1]) if len(sys.argv) > 1 else 8
nthreads = int(sys.argv[2]) if len(sys.argv) > 2 else min(os.cpu_count() or 1, 16)
order, modes, chunk = 2, 64, 8
torch.set_num_threads(nthreads)
stages, clock = {}, [time.time()]


def _heartbeat():
    # (the shift-invert stage is minutes of silence: a GPU-box call with nothing new on stderr for 7 minutes is taken to be hung)
    import threading

    def beat():
        while True:
            time.sleep(60)
            print(f"... {time.time() - clock[0]:.0f} s into the current stage", file=sys.stderr, flush=True)

    threading.Thread(target=beat, daemon=True).start()


_heartbeat()


def lap(name):
    now = time.time()
    stages[name] = round(now - clock[0], 2)
    clock[0] = now
    print(f"{name}: {stages[name]} s  (peak RSS {resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6:.1f} GB)", file=sys.stderr, flush=True)


t0 = clock[0]
v, t = meshgen.kuhn_box(cells)
v, t = fem.to_high_order(torch.from_numpy(v), torch.from_numpy(t).long(), order)
d = fem.OracleDeform(v, t, order)
lap("mesh_lifting_and_shape_function_derivatives")
E = torch.tensor(MAT[1], requires_grad=True)
nu = torch.tensor(MAT[2], requires_grad=True)
lam, mu = fem.lame(MAT[1], MAT[2])
M3, _ = fem.assemble_mass(v, t, order, MAT[0])
lap("mass_assembly")
K = fem.assemble_stiffness(d, lam, mu)
lap("stiffness_assembly_presummed")
ev, U, _, _ = modal.eigsh_shift_invert(K, M3, modes)
lap("arpack_shift_invert")
del K
# read-out (diff_model.py:371-388), eight modes at a time, activations recomputed in the backward
Uf = torch.as_tensor(U).float()
vals = torch.as_tensor(ev).float()
MU = torch.from_numpy(M3.astype(np.float32) @ Uf.numpy())


def piece(Uc, MUc, vc, E_, nu_):
    lam_, mu_ = fem.lame(E_, nu_)
    KU = fem.stiff_func(d, lam_, mu_, Uc)
    return (Uc * KU).sum(0) - vc * (Uc * MUc).sum(0)


adds = [ckpt.checkpoint(piece, Uf[:, s:s + chunk], MU[:, s:s + chunk], vals[s:s + chunk], E, nu, use_reentrant=False)
        for s in range(0, modes, chunk)]
pred = torch.zeros(modes) + torch.as_tensor(ev)
pred = pred + torch.cat(adds)
f = (torch.sqrt(pred) / 2 / np.pi).unsqueeze(1)
lap("get_undamped_freqs_chunked")
force = torch.zeros((1, 150))
force[0, 0] = 1
sig, _ = oosc.bank(f.float(), force, 8000, 32000, MAT[3], MAT[4])
loss = (sig ** 2).mean()
lap("oscillator_and_loss")
loss.backward()
lap("backward")
dt = time.time() - t0
cpu_model = "unknown"
try:
    for line in open("/proc/cpuinfo"):
        if line.startswith("model name"):
            cpu_model = line.split(":", 1)[1].strip()
            break
except OSError:
    pass
print(json.dumps({
    "kind": "port, memory-safe (BASELINE.md section 3 (i): optimised CPU baseline)", "cells": cells, "tets": int(t.shape[0]),
    "n": int(3 * v.shape[0]), "modes": modes, "seconds": round(dt, 1), "passes_per_s": 1.0 / dt, "stage_seconds": stages,
    "threads": nthreads, "cpu_model": cpu_model, "host_hardware_threads": os.cpu_count(),
    "peak_rss_gb": round(resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6, 1),
    "loss": float(loss.detach()), "dloss_dE": float(E.grad), "dloss_dnu": float(nu.grad),
    "lowest_eigenvalues": [float(x) for x in ev[:4]]}, indent=1))
