#!/usr/bin/env python3
"""Regenerate profiles/README.md from the committed bench JSON and rocprofv3 kernel-stat summaries."""
import csv, json, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = lambda f: os.path.join(ROOT, "profiles", f)

def table(f, n=14):
    rows = list(csv.reader(open(P(f))))
    out = ["| kernel | calls | total ms | avg us | % |", "|---|---|---|---|---|"]
    for r in rows[1:n + 1]:
        name = r[0].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
        if name.startswith("Cijk"):
            name = "rocBLAS `Cijk_…` (fp64 6×n·n×6 in the rigid-basis set-up, outside the timed region)"
        elif name.startswith("void at::native::elementwise_kernel_manual_unroll"):
            name = "torch copy kernel (`Tensor.copy_`)"
        else:
            name = "`" + name.split("(")[0].replace("void ", "") + "`"
        out.append(f"| {name} | {r[1]} | {r[2]} | {r[3]} | {r[6]} |")
    t = rows[-1]
    out.append(f"| all kernels | {t[1]} | {t[2]} | | 100 |")
    return "\n".join(out), rows

d = json.load(open(P("r01_bench_n1.json")))
r = d["roofline"]; lv = r["levels"]; solo = r["solo"]
t1, rows1 = table("r01_bench_lanes1_kernel_stats.csv")
t3, rows3 = table("r01_bench_kernel_stats.csv")
fused1 = next(x for x in rows1 if "spmm_union_kernel<20, 1" in x[0])
fused3 = next(x for x in rows3 if "spmm_union_kernel<20, 1" in x[0])
pmc = json.load(open(P("spmm_pmc_bytes_per_launch.json")))["cells26_cols80"]
tot1 = float(rows1[-1][2]); npass1 = 11  # target pass + (1 warm-up + 3 timed) steps x 2 hypotheses on one lane + ... see bench.py
readme = f'''# profiles/ — round 1 evidence (MI355X, gfx950, ROCm 7.2)

All runs: `bench.py` defaults = workload C3 (Kuhn box 26³ = 105 456 tets, ord-2, n = 446 631, nnz = 37.2 M, 64 modes,
block 80, two-level Chebyshev preconditioner), 8 hypotheses per step, 4 in flight per GPU, cold-start eigensolve and
numeric assembly in every pass.  (regenerate this file with `python tools/make_profiles_readme.py`)

| file | what |
|---|---|
| `r01_bench_n1.json` | the JSON line of `python bench.py` (N = 1, 3 steps, 1 warm-up, CPU baseline included): {d["value"]:.1f} passes/s |
| `r01_bench_kernel_stats.csv` | `rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --no-cpu-baseline`, top 45 kernels (target pass + warm-up step + 3 timed steps; 4 hypothesis lanes overlap, so durations are stretched by sharing) |
| `r01_bench_lanes1_kernel_stats.csv` | the same with `--lanes 1 --hyp-per-gpu 2`: one hypothesis at a time, every kernel alone on the device - the table to read kernel durations from |
| `spmm_pmc_bytes_per_launch.json` | HBM-side traffic of the dominant kernel (fine level) from separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes (`tools/pmc_bytes.sh`), gfx950 correction (2·FETCH + WRITE)·1024 as prescribed by `guides/MI355X_MICROARCH.md` |

(summaries made on the GPU box by `tools/summarize_prof.py`; the raw traces exceed what travels back)

## One hypothesis at a time (`r01_bench_lanes1_kernel_stats.csv`, {npass1} passes)

{t1}

{(tot1 - 47.7) / npass1:.0f} ms of kernels per pass (without the one-off rocBLAS set-up product). `spmm_union_kernel<20,1,116>` is the fused
Chebyshev-term SpMM on a full 80-column block (one wavefront per 4 nodes walking the union of their neighbours); the
name covers the fine level (≈ 0.29 ms per launch = 743.1 MB algorithmic) and the roughly ten times more numerous launches of
the corner-node level (≈ 0.05 ms each), {float(fused1[3]):.0f} µs on average; `<0,1,…>` is the same kernel on the narrower blocks left after
locking, `<·,0,…>` K·W, `<·,2,…>` the residual handed to the corner-node level, `<·,3,…>` the mass product (node-scalar
values), `mix_lds_kernel<10>` the fused Ritz updates [X' P'] = [X P W][Z1 Zp], `gram32_partial_kernel` the folded-fp32 MFMA Gram blocks
[V W]ᵀ(MW) and [X P W]ᵀ(KW), `mix_lds_kernel` the Ritz / ortho updates, `spmm_f64_node_kernel` and
`gram_partial_kernel<double>` the fp64 read-out.

## Default run, 4 lanes (`r01_bench_kernel_stats.csv`)

{t3}

The dominant kernel by total time is the fused Chebyshev-term SpMM. Its rocprofv3 average over this run
({float(fused3[3]) / 1e3:.3f} ms over {fused3[1]} launches; includes the single-lane target pass and the warm-up) and the HIP-event average over the
timed region inside `bench.py` ({r["avg_launch_ms"]:.3f} ms over {r["launches_timed"]} launches: fine level {lv["fine"]["avg_launch_ms"]:.2f} ms, corner-node level {lv["corner_node"]["avg_launch_ms"]:.3f} ms;
a different run, without the profiler, and the launches of the first lane only - the lane that issues them one by one
from the solver loop) are both stretched by the four lanes sharing the device: `roofline.achieved` =
{r["achieved"]:.0f} GB/s. Alone on the device the fine-level launch takes {solo["avg_launch_ms"]:.3f} ms (`roofline.solo` in the JSON; the
single-lane profile above agrees; 0.260 ms on contiguous operands in `tools/mb_kx_time.py` - the solver's blocks are
column ranges of a 248-column buffer), i.e. {solo["achieved"] / 1e3:.2f} TB/s algorithmic = {100 * solo["frac"]:.1f} % of the 8 TB/s HBM peak, with {pmc / 1e6:.1f} MB of
PMC traffic per launch ({pmc / 743098484:.2f} × algorithmic). It is bound by the CU gather path (≈ 27 B/clk/CU for the 960-byte neighbour
panels) plus the per-entry LDS / FMA work, not by HBM: see DESIGN.md §5.

## History this round (C3, one MI355X)

| step | passes/s |
|---|---|
| first end-to-end version | 0.62 |
| fused Chebyshev-term SpMM, Chebyshev(48), locking, Cholesky-QR, host LAPACK for the small dense steps | 3.45 |
| two-level preconditioner (corner-node P1 level) | 6.9 |
| symmetric/skip-tile Gram, conditional second ortho sweep | 8.5 |
| 3 hypotheses in flight per GPU (stream + host thread each) | 13.9 |
| Rayleigh–Ritz by recurrence, single-sweep projected Cholesky-QR, assembly in every pass | 14.5 |
| persistent LDS-staged `mix` kernel | 15.8 |
| Gram on the fp32 MFMA folded into fp64 every 48 rows (240×80: 0.466 → 0.230 ms) | 16.5 |
| neighbour-union SpMM promoted to the default on both levels (fused term 0.332 → 0.275 ms), `mix` with `ds_read_b128` operands | 18.0 |
| assembly stores through LDS (2.0 → 0.45 ms), wave-per-node fp64 read-out products, in-place residual / ortho update | 18.4 |
| mass product on the union kernel, fused [X' P'] Ritz updates, leaner launch path | 19.2 |
| persistent lane threads (no 3-13 ms bubble per step), 4 lanes x 2 hypotheses per step | 20.3 |
| bench preconditioner Chebyshev(2) smoother / Chebyshev(28, ratio 550) corner-node level | 20.7 |
| union SpMM at five waves per SIMD with a window of four loads (fused term 0.275 -> 0.260 ms, corner-node term 30 -> 25 us) | 21.5 |
| V-cycle on compact iterates (out-of-place last term) | {d["value"]:.1f} |

SpMM kernel history (80 columns, K·X, micro-benchmark `tools/mb_spmm.py`): node groups per wave 0.441 ms → Morton order
0.426 → wave per node with scalar metadata 0.340 → cooperative row metadata (readlane ids, LDS coefficients) 0.286 →
buffer loads with scalar panel offsets 0.271. Variants measured and parked because they are not faster (the product is
gather-bound; `make EXPERIMENTAL=1`): LDS-tiled 0.69–0.94 ms, register-blocked 4 nodes per wave 0.273, software-pipelined
window 0.301, batched (one wave per run of nodes) 0.282.  The neighbour union (4 nodes share their panel loads) went from
0.289 / fused 0.311 ms (compiler-scheduled FMAs and coefficient reads) to 0.268 / 0.275 ms with inline-asm packed FMAs on
accumulator halves that never change registers and without the epilogue-row touch during staging, then to 0.245 / 0.260 ms
at five waves per SIMD with a window of four loads, against 0.272 / 0.322 ms for one wavefront per node: it is now the
production kernel.
'''
open(P("README.md"), "w").write(readme)
print("profiles/README.md written")
