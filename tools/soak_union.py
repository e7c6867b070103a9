"""Soak: the union-SpMM parity tests repeated in ONE process (the bugs met while building the kernel were intermittent)."""
import sys, pytest
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
bad = 0
for i in range(n):
    rc = pytest.main(["-q", "-x", "-m", "gpu", "tests/test_hip_kernels.py", "tests/test_fullsize_gpu.py", "-k", "union", "-p", "no:cacheprovider"])
    bad += int(rc != 0)
    print(f"round {i}: rc={rc}", flush=True)
print("failures:", bad)
sys.exit(1 if bad else 0)
