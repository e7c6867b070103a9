#!/usr/bin/env python3
"""CPU baseline at several sizes: ONE measured full fwd+bwd pass of the CPU oracle per size with the
per-stage table, on the GPU box's host cores.   python tools/cpu_baseline_sizes.py 8 12 > gpurun_out/rXX_cpu_baseline_sizes.json"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

out = []
for cells in [int(a) for a in sys.argv[1:]] or [8]:
    r = bench.cpu_baseline(cells, 2, 64, 105456)
    out.append({"cells": cells, **r})
    print(json.dumps(out[-1]), file=sys.stderr, flush=True)
print(json.dumps(out, indent=1))
