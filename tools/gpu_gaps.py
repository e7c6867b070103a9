#!/usr/bin/env python3
"""Which kernels border the long device-idle gaps of a rocprofv3 --kernel-trace run (last `frac` of the run)?
   python tools/gpu_gaps.py <dir> [frac] [min_gap_us]"""
import csv, glob, os, sys, collections
src = sys.argv[1]; frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.6; ming = float(sys.argv[3]) * 1e3 if len(sys.argv) > 3 else 200e3
f = glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(f)))
print("columns:", list(rows[0].keys()))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-60:], r.get("Queue_Id", "?"), r.get("Thread_Id", "?")) for r in rows)
t0, t1 = ev[0][0], max(e[1] for e in ev)
lo = t1 - int((t1 - t0) * frac)
before, after = collections.Counter(), collections.Counter()
cur_end, last = 0, None
tot = 0
for s, e, name, q, th in ev:
    if e <= lo:
        if e > cur_end: cur_end, last = e, (name, q)
        continue
    if cur_end and s > cur_end and s - cur_end >= ming:
        before[last[0]] += s - cur_end; after[name] += s - cur_end; tot += s - cur_end
    if e > cur_end: cur_end, last = e, (name, q)
print(f"long gaps total {tot/1e6:.1f} ms")
print("kernel that ENDED before the gap (ms of gap):")
for k, v in before.most_common(8): print(f"  {v/1e6:7.1f}  {k}")
print("kernel that STARTED after the gap:")
for k, v in after.most_common(8): print(f"  {v/1e6:7.1f}  {k}")

# timeline around the three longest gaps: (time since gap start in us, duration, stream, kernel)
rows2 = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Stream_Id", "?"), r["Kernel_Name"].split("(")[0][-40:]) for r in rows))
gaps = []
cur_end = 0
for s, e, st, nm in rows2:
    if e <= lo:
        cur_end = max(cur_end, e); continue
    if cur_end and s > cur_end: gaps.append((s - cur_end, cur_end, s))
    cur_end = max(cur_end, e)
import os as _os
sel = [int(x) for x in _os.environ.get('GAPS', '10,40,90').split(',')]
sg = sorted(gaps, reverse=True)
print('gap size ranks (us):', [round(g[0]/1e3) for g in sg[:120:5]])
for g, a, b in [sg[i] for i in sel if i < len(sg)]:
    print(f"--- gap {g/1e3:.0f} us")
    for s, e, st, nm in rows2:
        if e > a - 400_000 and s < b + 400_000:
            print(f"  {(s - a)/1e3:9.1f} +{(e - s)/1e3:7.1f} us  stream {st}  {nm}")
