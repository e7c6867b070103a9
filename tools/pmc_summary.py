#!/usr/bin/env python3
"""Average rocprofv3 --pmc counters per dispatch for kernels whose name contains a pattern.

    python tools/pmc_summary.py <dir> <pattern> [<pattern> ...]   -> JSON on stdout
"""
import csv, glob, json, os, sys
from collections import defaultdict

src, pats = sys.argv[1], sys.argv[2:]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r.get("Kernel_Name", "")
        for p in pats:
            if p in name:
                acc[p][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {p: {c: {"mean": sum(v) / len(v), "n": len(v)} for c, v in sorted(cs.items())} for p, cs in acc.items()}
print(json.dumps(out, indent=1))
