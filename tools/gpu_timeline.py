#!/usr/bin/env python3
"""Concurrency profile of a rocprofv3 --kernel-trace run over the last `frac` of the run: time with 0, 1, 2, ... kernels
in flight, the idle gaps by size class, and which kernels border the SHORT gaps.  python tools/gpu_timeline.py <dir> [frac]"""
import collections, csv, glob, os, sys
src = sys.argv[1]; frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.4
f = glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("(anonymous namespace)::", "")[:48],
             r.get("Stream_Id", r.get("Queue_Id", "?"))) for r in rows)
t0, t1 = ev[0][0], max(e[1] for e in ev)
lo = t1 - int((t1 - t0) * frac)
ev = [e for e in ev if e[1] > lo]
pts = []
for s, e, _, _ in ev:
    pts.append((max(s, lo), 1)); pts.append((e, -1))
pts.sort()
level, last, hist = 0, lo, collections.Counter()
for t, d in pts:
    hist[level] += t - last; last = t; level += d
tot = sum(hist.values())
print("kernels in flight -> share of the window:", {k: round(100 * v / tot, 1) for k, v in sorted(hist.items())})
streams = collections.Counter()
for s, e, _, st in ev: streams[st] += e - max(s, lo)
print("busy share per stream:", {k: round(100 * v / tot, 1) for k, v in streams.most_common(12)})
cur_end, lastname = lo, None
classes = collections.Counter(); before = collections.Counter(); after = collections.Counter()
for s, e, name, st in ev:
    if s > cur_end:
        g = (s - cur_end) / 1e3
        c = "<5us" if g < 5 else "5-20us" if g < 20 else "20-100us" if g < 100 else "100-1000us" if g < 1000 else ">1ms"
        classes[c] += s - cur_end
        if g < 100 and lastname:
            before[lastname] += s - cur_end; after[name] += s - cur_end
    if e > cur_end: cur_end, lastname = e, name
print("idle time by gap size (ms):", {k: round(v / 1e6, 1) for k, v in classes.items()}, "window", round(tot / 1e6, 1), "ms")
print("short gaps: kernel that ended before (ms):", [(k, round(v / 1e6, 1)) for k, v in before.most_common(8)])
print("short gaps: kernel that started after (ms):", [(k, round(v / 1e6, 1)) for k, v in after.most_common(8)])
