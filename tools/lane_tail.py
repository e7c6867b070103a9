"""How uneven do the 8 lanes of a bench step finish?  Per step: the host time at which each lane's pass returned, relative to the
step's start (the step ends when the last lane does: the device is shared by fewer and fewer lanes towards its end)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from diffsound_amd import meshgen, pipeline  # noqa: E402
from diffsound_amd.diffelastic.mesh import TetMesh  # noqa: E402

dev = torch.device("cuda", 0)
v, t = meshgen.kuhn_box(26)
mesh = TetMesh(torch.from_numpy(v).to(dev), torch.from_numpy(t).long().to(dev)).to_high_order(2)
pipe = pipeline.ModalPipeline(mesh.vertices, mesh.tets, 2, 64, bench.MAT, solver_config=bench.solver_config())
pipe.assemble()
_, _, audio0 = pipe.run_pass(bench.MAT[1], bench.MAT[2], backward=False)
pipe.set_target(audio0)
rng = np.random.default_rng(2024)
Es, nus = rng.uniform(1e10, 1e11, size=64), rng.uniform(0.1, 0.4, size=64)
hyps = [(float(Es[h]), float(nus[h])) for h in range(8)]
stamps = {}
orig = pipeline.ModalPipeline.run_pass


def timed(self, E, nu, **kw):
    r = orig(self, E, nu, **kw)
    stamps[(E, nu)] = time.time()
    return r


pipeline.ModalPipeline.run_pass = timed
for _ in range(3):
    pipe.run_batch(hyps, lanes=8)
rows = []
for s in range(8):
    torch.cuda.synchronize()
    t0 = time.time()
    outs = pipe.run_batch(hyps, lanes=8)
    t1 = time.time()
    ends = sorted(stamps[h] - t0 for h in hyps)
    its = [o[0].iterations for o in outs]
    rows.append((t1 - t0, ends, its))
    print(f"step {s}: {1e3 * (t1 - t0):6.1f} ms; lanes done at " + " ".join(f"{1e3 * e:6.1f}" for e in ends) + f"  iterations {its}", flush=True)
mean_end = np.mean([np.mean(r[1]) for r in rows])
last = np.mean([r[0] for r in rows])
print(f"mean lane finish {1e3 * mean_end:.1f} ms of a {1e3 * last:.1f} ms step: {100 * (1 - mean_end / last):.1f} % of the lane-time is tail")

# the same without the join per step (ModalPipeline.run_steps): when does each hypothesis' LAST step end?
K = 10
stamps.clear()
last = {}


def timed2(self, E, nu, **kw):
    r = orig(self, E, nu, **kw)
    last[(E, nu)] = time.time()
    return r


pipeline.ModalPipeline.run_pass = timed2
torch.cuda.synchronize()
t0 = time.time()
pipe.run_steps(hyps, K, lanes=8)
torch.cuda.synchronize()
t1 = time.time()
ends = sorted(last[h] - t0 for h in hyps)
print(f"{K} steps without a join: {1e3 * (t1 - t0):.1f} ms = {1e3 * (t1 - t0) / K:.1f} ms per step; the lanes' last passes end at "
      + " ".join(f"{1e3 * e:.0f}" for e in ends) + f" ms: {100 * (1 - np.mean(ends) / (t1 - t0)):.1f} % of the lane-time is tail")
