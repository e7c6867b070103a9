"""cProfile of the drop-in training loop at the benchmark's shape (experiments/material_sync_train.py:137-168 through DiffSoundObj /
TraditionalDampedOscillator / MSSLoss / Adam): where an epoch's host time goes.   python tools/host_profile_api_loop.py [epochs] [cycle]"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.optim import Adam
import bench
from diffsound_amd import meshgen
from src.ddsp.mss_loss import MSSLoss
from src.ddsp.oscillator import TraditionalDampedOscillator
from src.diffelastic.diff_model import Material, build_model

epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 10
cycle = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = torch.device("cuda:0")
v, t = meshgen.kuhn_box(26)
v, t = torch.from_numpy(v).to(dev), torch.from_numpy(t).long().to(dev)
mat = [2700.0, 4.6e10, 0.31, 6.0, 1e-7]
forces = torch.zeros((1, 150), device=dev)
forces[0, 0] = 1
model = build_model(None, mode_num=64, order=2, mat=mat, task="material", vertices=v, tets=t)
model.solver_config = bench.solver_config()
osc = TraditionalDampedOscillator(forces, 1, 64, 8000, 32000, Material(mat)).cuda()
late = MSSLoss([1024, 512, 256, 128, 64], 32000, type="l1_loss").cuda()
gt = torch.randn((1, 8000), device=dev) * 1e-3
opt = Adam(model.parameters(), lr=5e-3)


def epoch(i):
    if i % cycle == 0:
        model.eigen_decomposition()
    f = model.get_undamped_freqs().float()
    pred = osc(f)
    loss = late(pred, gt, osc.damped_freq, 1)
    opt.zero_grad()
    loss.backward()
    opt.step()
    return float(loss.detach())


for i in range(3):
    epoch(i)
torch.cuda.synchronize()
pr = cProfile.Profile()
t0 = time.time()
pr.enable()
for i in range(epochs):
    epoch(i)
pr.disable()
torch.cuda.synchronize()
print(f"{epochs} epochs, cycle {cycle}: {(time.time() - t0) / epochs * 1e3:.2f} ms per epoch (under the profiler)")
st = pstats.Stats(pr, stream=sys.stdout)
st.sort_stats("cumulative").print_stats(40)
st.sort_stats("tottime").print_stats(25)
