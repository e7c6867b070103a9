"""Run K X (80 columns) back to back for a few seconds while sampling the shader clock with rocm-smi."""
import os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffsound_amd import meshgen
from diffsound_amd.diffelastic.mesh import TetMesh
from diffsound_amd.modal_ops import TetSystem, HipModalOps
dev = torch.device('cuda')
v, t = meshgen.kuhn_box(26)
mesh = TetMesh(torch.from_numpy(v).to(dev), torch.from_numpy(t).long().to(dev)).to_high_order(2)
sysd = TetSystem(mesh.vertices, mesh.tets, 2, 2700.0)
ops = HipModalOps(sysd, 2e10, 2e10, two_level=False)
X = torch.randn(sysd.n, 80, device=dev); Y = torch.empty_like(X)
stop = False
def sample():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=10).stdout
            for ln in out.splitlines():
                if "sclk" in ln or "mclk" in ln or "Power" in ln or "fclk" in ln:
                    print(ln.strip(), flush=True)
            print("--", flush=True)
        except Exception as e:
            print("rocm-smi failed:", e, flush=True)
            return
        time.sleep(0.7)
th = threading.Thread(target=sample); th.start()
t0 = time.time()
while time.time() - t0 < 4.0:
    for _ in range(200):
        ops.apply_K(X, Y)
    torch.cuda.synchronize()
stop = True; th.join()
