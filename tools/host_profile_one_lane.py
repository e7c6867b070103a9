"""cProfile of ONE hypothesis at a time at the benchmark's shape: where the host thread's time goes outside the native calls
(python tools/host_profile_one_lane.py [passes]) -> cumulative table on stdout."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from diffsound_amd import meshgen
from diffsound_amd.diffelastic.mesh import TetMesh
from diffsound_amd.pipeline import ModalPipeline
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda:0")
v, t = meshgen.kuhn_box(26)
mesh = TetMesh(torch.from_numpy(v).to(dev), torch.from_numpy(t).long().to(dev)).to_high_order(2)
pipe = ModalPipeline(mesh.vertices, mesh.tets, 2, 64, bench.MAT, solver_config=bench.solver_config())
pipe.assemble()
_, _, a0 = pipe.run_pass(bench.MAT[1], bench.MAT[2], backward=False)
pipe.set_target(a0)
for i in range(3):
    pipe.run_pass(5e10 * (1 + 0.01 * i), 0.25)
torch.cuda.synchronize()
pr = cProfile.Profile()
t0 = time.time()
pr.enable()
for i in range(n):
    pipe.run_pass(4e10 * (1 + 0.02 * i), 0.22 + 0.005 * i)
pr.disable()
torch.cuda.synchronize()
print(f"{n} passes, {(time.time() - t0) / n * 1e3:.2f} ms per pass (under the profiler)")
st = pstats.Stats(pr, stream=sys.stdout)
st.sort_stats("cumulative").print_stats(45)
st.sort_stats("tottime").print_stats(30)
