"""CPU oracle sample pass of bench.py at several torch thread counts."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
for nt in (1, 4, 16, 64, 256):
    torch.set_num_threads(nt)
    real = os.cpu_count
    os.cpu_count = lambda nt=nt: nt   # bench.cpu_baseline caps its thread count at 16: this sweep is for <= 16 only
    try:
        t0 = time.time(); r = bench.cpu_baseline(5, 2, 64, 105456); dt = time.time() - t0
    finally:
        os.cpu_count = real
    print(nt, "threads:", round(r["sample_seconds"], 2), "s", flush=True)
