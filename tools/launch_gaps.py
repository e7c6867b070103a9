#!/usr/bin/env python3
"""Idle time between consecutive kernels of a ONE-LANE run (rocprofv3 --kernel-trace CSV): the gap from a kernel's end to the next
kernel's start, overall and between consecutive corner-node level terms (the launch-bound chain of the V-cycle) - what a captured
graph of the cycle could remove.   python tools/launch_gaps.py <dir>"""
import csv, glob, os, sys
import numpy as np

f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))), key=lambda t: t[0])
rows = rows[len(rows) // 3:]  # the timed passes (set-up and warm-up are the first third of a short run)
gap = np.array([rows[i + 1][0] - rows[i][1] for i in range(len(rows) - 1)], dtype=np.float64) / 1e3
dur = np.array([e - s for s, e, _ in rows], dtype=np.float64) / 1e3
corner = lambda n: "spmm_union_mfma_kernel" in n and ", 16, 1," in n
cc = np.array([gap[i] for i in range(len(rows) - 1) if corner(rows[i][2]) and corner(rows[i + 1][2])])
small = gap[(gap >= 0) & (gap < 50)]
print(f"{len(rows)} launches, kernel time {dur.sum() / 1e3:.1f} ms, span {(rows[-1][1] - rows[0][0]) / 1e6:.1f} ms")
print(f"gaps under 50 us (back-to-back launches of one enqueue burst): {len(small)}, median {np.median(small):.2f} us, mean {small.mean():.2f} us, "
      f"total {small.sum() / 1e3:.2f} ms")
print(f"corner-level term -> corner-level term: {len(cc)} gaps, median {np.median(cc):.2f} us, mean {cc.mean():.2f} us, 90th percentile "
      f"{np.percentile(cc, 90):.2f} us; their kernels last {np.median([d for d, r in zip(dur, rows) if corner(r[2])]):.1f} us")
print(f"gaps of 50 us and more (the host is computing or waiting): {(gap >= 50).sum()}, total {gap[gap >= 50].sum() / 1e3:.1f} ms")
