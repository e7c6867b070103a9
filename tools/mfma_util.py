#!/usr/bin/env python3
"""profiles/gram_mix_mfma_util.json from the per-shape counter summaries of tools/collect_profiles.sh (part 2):
    python tools/mfma_util.py <tag>    reads gpurun_out/<tag>_rr_pmc_<shape>.json (tools/pmc_summary.py)
Matrix-pipe utilisation of a launch = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs): the busy counter sums the
cycles every SIMD's matrix pipe was busy (guides/MI355X_MICROARCH.md), GRBM_GUI_ACTIVE sums the 8 XCDs' active cycles.  Each
record carries the hash of the dense kernels' sources; bench.py reports a record with another hash as stale."""
import datetime, glob, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
tag = sys.argv[1]
kernels = {}
for path in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", f"{tag}_rr_pmc_*.json"))):
    shape = re.search(r"_rr_pmc_(\w+)\.json$", path).group(1)
    doc = json.load(open(path))
    main = [(k, v) for k, v in doc.items() if "SQ_VALU_MFMA_BUSY_CYCLES" in v and "reduce" not in k]
    if not main:
        continue
    name, c = main[0]
    cyc = c["GRBM_GUI_ACTIVE"]["mean"] / 8.0
    busy = c["SQ_VALU_MFMA_BUSY_CYCLES"]["mean"]
    kernels[shape] = {"kernel": name, "launches": c["SQ_VALU_MFMA_BUSY_CYCLES"]["n"], "mfma_busy_cycles_all_simds": busy,
                      "active_cycles_per_xcd": cyc, "mfma_pipe_utilisation": busy / (cyc * 1024.0),
                      "mfma_f32_ops": c.get("SQ_INSTS_VALU_MFMA_MOPS_F32", {}).get("mean")}
out = {"_comment": __doc__.split("\n")[2] + " " + __doc__.split("\n")[3], "dense_source_sha16": bench.dense_source_hash(),
       "measured": f"{tag} ({datetime.date.today().isoformat()})", "kernels": kernels}
json.dump(out, open(os.path.join(ROOT, "profiles", "gram_mix_mfma_util.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
