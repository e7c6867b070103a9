#!/usr/bin/env python3
"""profiles/spmm_pmc_bytes_per_launch.json from the per-kind PMC summaries of tools/collect_profiles.sh:

    python tools/pmc_bytes.py <tag>     reads gpurun_out/<tag>_spmm_pmc_{fp32,bf16,mfma,kx}.json (tools/pmc_summary.py)

HBM-side bytes per launch of the fused term (fp32 / bf16 / mfma) and of the eigensolver's Y = K X (kx) = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 - separate rocprofv3 --pmc passes; the
gfx950 correction of guides/MI355X_MICROARCH.md: FETCH_SIZE counts 64 of every 128 bytes of a wide streaming read.
Every record carries the hash of the SpMM kernel sources it was measured on; bench.py reports a record with another
hash as stale (traffic: null)."""
import datetime
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

tag = sys.argv[1]
out = {"_comment": __doc__.split("\n\n")[1].replace("\n", " ")}
for kind in ("fp32", "bf16", "mfma", "kx", "km", "resid"):
    path = os.path.join(ROOT, "gpurun_out", f"{tag}_spmm_pmc_{kind}.json")
    if not os.path.exists(path):
        continue
    doc = json.load(open(path))
    (name, c), = [(k_, v_) for k_, v_ in doc.items() if "FETCH_SIZE" in v_][:1]
    nbytes = int(round((2 * c["FETCH_SIZE"]["mean"] + c["WRITE_SIZE"]["mean"]) * 1024))
    out[f"cells26_cols80_{kind}"] = {"bytes": nbytes, "kernel": name, "launches": c["FETCH_SIZE"]["n"],
                                     "spmm_source_sha16": bench.spmm_source_hash(),
                                     "measured": f"{tag} ({datetime.date.today().isoformat()})"}
json.dump(out, open(os.path.join(ROOT, "profiles", "spmm_pmc_bytes_per_launch.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
