#!/usr/bin/env python3
"""What one hypothesis lane of the benchmark holds in HBM: every tensor reachable from a lane's objects (by attribute path),
split into storage shared with the other lanes (the topology's tables) and storage of its own (values, basis, scratch).
python tools/lane_memory.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from diffsound_amd import meshgen  # noqa: E402
from diffsound_amd.diffelastic.mesh import TetMesh  # noqa: E402
from diffsound_amd.pipeline import ModalPipeline  # noqa: E402

sys.argv = [sys.argv[0], "--no-cpu-baseline"]
a = bench.parse()
dev = torch.device("cuda", 0)
v, t = meshgen.kuhn_box(a.cells)
mesh = TetMesh(torch.from_numpy(v).to(dev), torch.from_numpy(t).long().to(dev)).to_high_order(a.order)
pipe = ModalPipeline(mesh.vertices, mesh.tets, a.order, a.modes, bench.MAT, solver_config=bench.solver_config(a))
pipe.assemble()
tgt, res0, audio0 = pipe.run_pass(bench.MAT[1], bench.MAT[2], backward=False)
pipe.set_target(audio0)
rng = np.random.default_rng(2024)
hyps = [(float(e), float(n)) for e, n in zip(rng.uniform(1e10, 1e11, 4), rng.uniform(0.1, 0.4, 4))]
torch.cuda.reset_peak_memory_stats()
pipe.run_steps(hyps, 2, lanes=2)
torch.cuda.synchronize()


def walk(obj, path, out, seen, depth=0):
    if depth > 6 or id(obj) in seen:
        return
    seen.add(id(obj))
    if torch.is_tensor(obj):
        if obj.is_cuda:
            st = obj.untyped_storage()
            out.setdefault(st.data_ptr(), (st.nbytes(), path))
        return
    if isinstance(obj, dict):
        for k, x in obj.items():
            walk(x, f"{path}[{k!r}]", out, seen, depth + 1)
    elif isinstance(obj, (list, tuple)):
        for i, x in enumerate(obj):
            walk(x, f"{path}[{i}]", out, seen, depth + 1)
    elif hasattr(obj, "__dict__") and not isinstance(obj, type) and type(obj).__module__.startswith("diffsound_amd"):
        for k, x in vars(obj).items():
            walk(x, f"{path}.{k}", out, seen, depth + 1)


lane0, lane1 = {}, {}
walk(pipe._lanes[0], "lane", lane0, set())
walk(pipe._lanes[1], "lane", lane1, set())
shared = {p: x for p, x in lane1.items() if p in lane0}
own = {p: x for p, x in lane1.items() if p not in lane0}
gib = 2.0 ** 30
print(f"benchmark mesh, 2 lanes after 2 steps: torch allocated {torch.cuda.memory_allocated() / gib:.2f} GiB, peak "
      f"{torch.cuda.max_memory_allocated() / gib:.2f} GiB, reserved {torch.cuda.memory_reserved() / gib:.2f} GiB")
print(f"reachable from lane 1: own {sum(x[0] for x in own.values()) / gib:.3f} GiB in {len(own)} storages, shared with lane 0 "
      f"{sum(x[0] for x in shared.values()) / gib:.3f} GiB in {len(shared)} storages")
for title, d in (("own", own), ("shared", shared)):
    print(f"-- {title}: the 40 largest")
    for nbytes, path in sorted(d.values(), reverse=True)[:40]:
        print(f"   {nbytes / 2 ** 20:9.1f} MiB  {path}")
