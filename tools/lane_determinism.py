#!/usr/bin/env python3
"""Run-to-run reproducibility of concurrent hypothesis lanes at the benchmark's shape (the check of
tests/test_fullsize_gpu.py::test_c3_eight_lanes_are_bit_identical_from_run_to_run, with the quantities of a pass broken
out): 8 hypotheses on 8 lanes, 2 steps, run REPS times from identical fresh state; reports which quantity of which pass
differs first - the fp32 block before the polish (bit checksum), the polish's fp64 Gram matrices, eigenvalues, the quadratic
forms, loss, gradients.   python tools/lane_determinism.py [raw_rr 0|1] [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from diffsound_amd import meshgen  # noqa: E402
from diffsound_amd.diffelastic.mesh import TetMesh  # noqa: E402
from diffsound_amd.lobpcg import modal_solver  # noqa: E402
from diffsound_amd.pipeline import ModalPipeline  # noqa: E402

raw = int(sys.argv[1]) if len(sys.argv) > 1 else 1
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device("cuda:0")
MAT = bench.MAT
v, t = meshgen.kuhn_box(26)
mesh = TetMesh(torch.from_numpy(v).to(dev), torch.from_numpy(t).long().to(dev)).to_high_order(2)
rng = np.random.default_rng(2024)
hyps = [(float(E), float(nu)) for E, nu in zip(rng.uniform(1e10, 1e11, 8), rng.uniform(0.1, 0.4, 8))]

orig_polish = modal_solver.ModalSolver._polish


def polish(self, X, k, it, rerr, history):
    chk = int(X.contiguous().view(torch.int32).long().sum())
    GK, coef, GM = self.ops.polish_products(X)
    gchk = [g.cpu().numpy().tobytes() for g in list(GK) + [GM]]
    res = orig_polish(self, X, k, it, rerr, history)
    res.history = list(res.history) + [("x_bits", chk), ("gram_bytes", gchk)]
    return res


modal_solver.ModalSolver._polish = polish
runs = []
for rep in range(reps):
    pipe = ModalPipeline(mesh.vertices, mesh.tets, 2, 64, MAT, solver_config=bench.solver_config(raw_rr=raw))
    pipe.assemble()
    _, _, target = pipe.run_pass(MAT[1], MAT[2], backward=False)
    pipe.set_target(target)
    out = pipe.run_steps(hyps, 2, lanes=8)
    torch.cuda.synchronize()
    rec = []
    for step in out:
        for (r, res, _) in step:
            h = dict(x for x in res.history if isinstance(x[0], str))
            rec.append(dict(x_bits=h["x_bits"], gram=h["gram_bytes"], ev=res.eigenvalues.cpu().numpy().tobytes(),
                            a=res.a_lambda.cpu().numpy().tobytes(), b=res.b_mu.cpu().numpy().tobytes(),
                            m=res.m_diag.cpu().numpy().tobytes(), loss=r.loss, gE=r.grad_E, gnu=r.grad_nu,
                            it=(r.iterations, r.coarse_iterations)))
    runs.append(rec)
    del pipe, out
    torch.cuda.empty_cache()
ndiff = 0
for rep in range(1, reps):
    for i, (p, q) in enumerate(zip(runs[0], runs[rep])):
        for key in ("it", "x_bits", "gram", "ev", "a", "b", "m", "loss", "gE", "gnu"):
            if p[key] != q[key]:
                ndiff += 1
                print(f"raw_rr={raw} run {rep} pass {i} (step {i // 8}, hypothesis {i % 8}): FIRST difference in '{key}'",
                      (p[key], q[key]) if key in ("it", "x_bits", "loss", "gE", "gnu") else "", flush=True)
                break
print(f"raw_rr={raw}: {reps} runs x 16 passes, {ndiff} passes differ from run 0")
