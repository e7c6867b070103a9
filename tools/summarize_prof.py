#!/usr/bin/env python3
"""Condense a rocprofv3 --kernel-trace --stats output directory into a small CSV for profiles/.

    python tools/summarize_prof.py gpurun_out/prof_xxx profiles/r01_bench_kernel_stats.csv [--top 40]
"""
import csv
import glob
import os
import sys


def main():
    src, dst = sys.argv[1], sys.argv[2]
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 40
    files = glob.glob(os.path.join(src, "**", "*_kernel_stats.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no *_kernel_stats.csv under {src}")
    rows = list(csv.DictReader(open(files[0])))
    total = sum(float(r["TotalDurationNs"]) for r in rows)
    with open(dst, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_ms", "avg_us", "min_us", "max_us", "percent"])
        for r in rows[:top]:
            w.writerow([r["Name"], r["Calls"], f"{float(r['TotalDurationNs']) / 1e6:.3f}",
                        f"{float(r['AverageNs']) / 1e3:.2f}", f"{float(r['MinNs']) / 1e3:.2f}",
                        f"{float(r['MaxNs']) / 1e3:.2f}", f"{100 * float(r['TotalDurationNs']) / total:.2f}"])
        w.writerow(["TOTAL (all kernels)", sum(int(r["Calls"]) for r in rows), f"{total / 1e6:.3f}", "", "", "", "100"])
    print(f"wrote {dst}: {len(rows)} kernels, {total / 1e6:.1f} ms")


if __name__ == "__main__":
    main()
