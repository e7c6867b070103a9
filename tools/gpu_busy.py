#!/usr/bin/env python3
"""Device busy fraction from a rocprofv3 --kernel-trace CSV: union of the kernel intervals over the last `frac` of the
run minus its last `tail` (the timed steps without the drain of the last one; the profiled command must end with the timed
region: --no-solo --amortised-cycle 1), plus how many kernels overlap on average.   python tools/gpu_busy.py <dir> [frac] [tail]"""
import csv, glob, os, sys
src = sys.argv[1]; frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.6
tail = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
f = glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)[0]
iv = []
for r in csv.DictReader(open(f)):
    iv.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
iv.sort()
t0, t1 = iv[0][0], max(e for _, e in iv)
lo = t1 - int((t1 - t0) * frac)
t1 -= int((t1 - t0) * tail)
iv = [(max(s, lo), min(e, t1)) for s, e in iv if e > lo and s < t1]
busy = 0; cur_s, cur_e = iv[0]; total = 0
gaps = []
for s, e in iv:
    total += e - s
    if s > cur_e:
        busy += cur_e - cur_s; gaps.append(s - cur_e); cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
wall = t1 - lo
gaps.sort()
big = [g for g in gaps if g > 200_000]
print(f"window {wall/1e6:.1f} ms: busy {busy/1e6:.1f} ms = {100*busy/wall:.1f} %, sum of kernel durations {total/1e6:.1f} ms (mean overlap {total/busy:.2f}); "
      f"{len(gaps)} idle gaps, {len(big)} longer than 0.2 ms totalling {sum(big)/1e6:.1f} ms, gaps <= 0.2 ms total {sum(g for g in gaps if g <= 200_000)/1e6:.1f} ms")
