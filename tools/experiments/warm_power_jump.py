"""EXPERIMENT: how low can the ONE-step warm estimate of lambda_max(T K) fall when the Poisson ratio jumps between two hypotheses on
the same mesh?  (The 1.2 safety factor must cover it: an under-estimated interval end makes the Chebyshev polynomial diverge.)
For pairs (nu_old -> nu_new): the estimate after 1 / 3 steps from the old material's converged block, against 200 cold steps.
    python tools/experiments/warm_power_jump.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from diffsound_amd import meshgen
from diffsound_amd.diffelastic.mesh import TetMesh
from diffsound_amd.lobpcg.modal_solver import ChebyshevBlockJacobi as C
from diffsound_amd.modal_ops import HipModalOps, TetSystem

dev = torch.device("cuda")
v, t = meshgen.kuhn_box(int(sys.argv[1]) if len(sys.argv) > 1 else 26)
mesh = TetMesh(torch.from_numpy(v).to(dev), torch.from_numpy(t).long().to(dev)).to_high_order(2)
sysd = TetSystem(mesh.vertices, mesh.tets, 2, 2700.0)
lame = lambda E, nu: (E * nu / ((1 + nu) * (1 - 2 * nu)), E / (2 * (1 + nu)))
ops = HipModalOps(sysd, *lame(7e10, 0.3))
for level, o in (("fine level", ops), ("corner-node level", ops.coarse)):
    for nu0, nu1 in ((0.12, 0.45), (0.45, 0.12), (0.30, 0.49), (0.05, 0.40), (0.25, 0.35)):
        ops.set_material(*lame(7e10, nu0))
        o._power_block = None
        C(o, 3, 10.0, power_iters=200, safety=1.0)          # converge the block on the old material
        ops.set_material(*lame(2e11, nu1))
        blk = o._power_block.clone()
        est = {}
        for steps in (1, 3):
            o._power_block = blk.clone()
            saved = C.warm_power_iters, C.warm_spread
            C.warm_power_iters, C.warm_spread = steps, 0.0
            est[steps] = C(o, 3, 10.0, power_iters=200, safety=1.0).lmax
            C.warm_power_iters, C.warm_spread = saved
        o._power_block = None
        true = C(o, 3, 10.0, power_iters=200, safety=1.0).lmax
        print(f"{level}: nu {nu0:.2f} -> {nu1:.2f}: one step {est[1]:.4f} ({est[1] / true:.4f} of the 200-step value {true:.4f}), "
              f"three steps {est[3]:.4f} ({est[3] / true:.4f})", flush=True)
