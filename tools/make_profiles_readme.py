#!/usr/bin/env python3
"""profiles/README.md for round 4 from the files under profiles/ (python tools/make_profiles_readme.py)."""
import csv
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = lambda *a: os.path.join(ROOT, "profiles", *a)


def table(name, top):
    rows = list(csv.DictReader(open(P(name))))[:top]
    out = ["| kernel | calls | total ms | avg us | min | max | % |", "|---|---|---|---|---|---|---|"]
    for r in rows:
        k = r["kernel"].replace("(anonymous namespace)::", "").replace("void ", "")
        k = k.split("(")[0]
        out.append(f"| `{k}` | {r['calls']} | {float(r['total_ms']):.1f} | {float(r['avg_us']):.1f} | {float(r['min_us']):.1f} | "
                   f"{float(r['max_us']):.1f} | {float(r['percent']):.1f} |")
    return "\n".join(out)


def kernel_row(name, key):
    for r in csv.DictReader(open(P(name))):
        if key in r["kernel"]:
            return r
    return None


d = json.load(open(P("r04_bench_n1.json")))
r = d["roofline"]
kw = r["lobpcg_spmm"]
cb = d["cpu_baseline"]
pmc = json.load(open(P("spmm_pmc_bytes_per_launch.json")))
l1 = "r04_bench_lanes1_kernel_stats.csv"
stream_l1 = d["roofline"]["stream_triad"]  # (the one-lane table holds the passes' own launches only: no triad in it)
solo = "r04_bench_solo_kernel_stats.csv"
fine = kernel_row(l1, "spmm_union_mfma_kernel<8, 5, 1, false, 16, 0>")
fine_solo = kernel_row(solo, "spmm_union_mfma_kernel<8, 5, 1, false, 16, 0>")
kx_solo = kernel_row(solo, "spmm_union_kernel<20, 0, 140, false, false, true, 0>")
kx = kernel_row(l1, "spmm_union_kernel<20, 0, 140, false, false, true, 0>")
res = kernel_row(l1, "spmm_union_kernel<20, 4, 140, false, false, true, 0>")
fb, kb = r["algorithmic_bytes_per_launch"], kw["algorithmic_bytes_per_launch"]


def frac(nbytes, row):
    g = nbytes / (float(row["avg_us"]) * 1e-6) / 1e9
    return f"{row['calls']} launches, avg {float(row['avg_us']):.1f} us (min {float(row['min_us']):.1f}, max {float(row['max_us']):.1f}) -> {g:.0f} GB/s = {100 * g / 8000:.1f} % of 8 TB/s, {100 * g / stream_l1:.1f} % of that run's STREAM triad ({stream_l1:.0f} GB/s)"


lines = f'''# profiles/ — round 4 evidence (MI355X, gfx950, ROCm 7.2); round 3 as `r03_*`, round 2 as `r02_*`, round 1 under `r01/`

All runs: `bench.py` defaults = workload C3 (Kuhn box 26³ = 105 456 tets, ord-2, n = 446 631, nnz = 37.2 M, 64 modes, block 80,
two-level preconditioner on bf16 blocks with both levels' terms on the matrix cores, nested start to 3e-3, tolerance 1e-5),
8 hypotheses per step, 8 in flight per GPU, cold-start eigensolve and numeric assembly in every pass.
Collected by `tools/collect_profiles.sh r04 <part>` on the GPU box (this file: `python tools/make_profiles_readme.py`).

| file | what |
|---|---|
| `r04_bench_n1.json` | the JSON line of `python bench.py` (N = 1, 10 steps, 2 warm-up, CPU baseline at two sizes included): **{d["value"]:.1f} passes/s**; amortised variant {d["amortised"]["value"]:.0f} passes/s |
| `r04_bench_kernel_stats.csv`, `r04_gpu_busy.txt` | `rocprofv3 --kernel-trace --stats` of `python bench.py --no-cpu-baseline --no-solo --amortised-cycle 1 --steps 8` (the passes' own launches only; 8 lanes overlap: durations stretched by sharing) and the device-busy fraction inside its timed steps (55 % .. 95 % of the run) |
| `r04_bench_lanes1_kernel_stats.csv` | the same with `--lanes 1 --hyp-per-gpu 2 --steps 4`: one hypothesis at a time, every kernel alone on the device but between the other kernels of a pass — **the table to read in-pass kernel durations from; fine- and corner-node-level launches carry different symbols (the last template argument: 0 fine, 1 corner-node level)** |
| `r04_bench_solo_kernel_stats.csv`, `r04_solo_timing_transient.txt` | the bench line's kernel-ALONE figures under the profiler (`python bench.py --no-cpu-baseline --lanes 1 --hyp-per-gpu 1 --steps 1 --warmup 1 --amortised-cycle 1`: nearly all launches of the fused term, K W and the triad in this table are the 330 back-to-back ones of `roofline.avg_launch_ms`), and why those are steady-state figures since this round: successive 30-launch averages after the timed region fall from 0.201 to 0.181 ms over ~100 launches |
| `spmm_pmc_bytes_per_launch.json`, `r04_spmm_pmc_{{fp32,bf16,mfma,kx}}.json` | HBM-side traffic per launch from separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes; bytes = (2·FETCH + WRITE)·1024 (gfx950 correction of `guides/MI355X_MICROARCH.md`); every record carries the hash of the SpMM sources and of `modal_ops.py` (the ordering and the union tables) it was measured on — `bench.py` reports a record with another hash as stale (`traffic: null`) |
| `r04_node_order_ab.txt` | A/B on one box of the node ORDERING: Morton curve over raw coordinates (round 3) against bricks aligned to the mesh's node planes — 45.5 → 48.0 passes/s, K·W 210 → 198 µs, bf16 term 2 500 → 3 100 GB/s |
| `r04_union_cap_ab.txt` | chunk cap of the neighbour-union tables 116 → 140 blocks (every group one chunk): K·X 215 → 204 µs |
| `r04_kx_fresh_ab.txt` | K·X′ by one fresh product instead of the update of K·[X P W]: 48.3 → 49.6 passes/s |
| `r04_basis_row_pitch_ab.txt`, `r04_mb_kx_strided.txt`, `r04_mb_kx_layout.txt` | the solver's basis buffers with rows 1 KiB apart (256-column pitch instead of 248): K·X / M·X against operand layouts, interleaved and warmed up — compact blocks 177 / 137 µs, ranges of a 248-column buffer 193 / 165, of a 256-column buffer 182 / 152; +1–2 % passes/s |
| `r04_fused_residual_ab.txt`, `r04_mb_fused_residual.txt` | the residual of every iteration in ONE walk of the unions (`ds_union_residual`): 459 µs (K·X′ + M·X′ + residual) → 280 µs (239 after the norm reduction was parallelised); 49.3 → 51.8 passes/s |
| `r04_mfma32_knockout.txt`, `r04_m32_diag.txt`, `r04_mb_kx.txt`, `r04_mb_kx_raw_morton.txt`, `r04_mb_kx_plane_order.txt` | **the fp32 matrix-core form of the eigensolver's own products (`ds_spmm_union32m`): built, parity-green, slower** — K·X 263–277 µs against the VALU kernel's 205–235 on the same boxes; knock-out builds (no MFMAs 189, no gathers 249, no A-fragment chain 222, none 110); per-wave `s_memtime` stamps of a diagnostic build (`tools/m32_diag.py`): where a wave's cycles go in that kernel and in the bf16 term, the in-kernel clock (2.0–2.1 GHz), wave slots occupied |
| `r04_corner_batch_ab.txt`, `r04_mb_corner.txt` | the corner-node level's bf16 term with batches of 16 and of 32 entries at its real size (2 461 groups): 18.2 against 23.5 µs — both levels run 16 |
| `r04_union_waves_per_workgroup_ab.txt` | the VALU union kernel with 4 / 2 / 1 waves per workgroup: 208 / 213 / 215 µs (4 stays) |
| `r04_hw_queues.txt`, `r04_block_sweep.txt`, `r04_lane_memory.txt` | hardware queues 2 / 4 (default) / 8 / 16: no gain; eigensolver block 72 / 76 / 80: 48.8 / 49.2 / 53.2 passes/s (80 stays); what one hypothesis lane holds in HBM, by tensor |
| `r04_lanes_sweep.txt` | 6 / 8 / 12 / 16 hypothesis lanes: 47.9 / 49.2 / 49.7 / 49.0 passes/s (the device, not the host, is the bound) |
| `r04_gram_mix.txt`, `r04_gram_mix_pmc.json` | Gram / `mix` timings at the solver's shapes and their MFMA counters |
| `r04_symbolic_phase_timing.txt` | ord-2 lifting and the symbolic phase per topology at C3 |
| `r04_c5_bench.json`, `r04_c5_kernel_stats.csv`, `r04_c5_bench_noprof.json` | **configs[4]** (`bench.py --workload c5` under `rocprofv3 --kernel-trace --stats`, and plain): 998 250 tets, n = 4.1 M, 128 modes - solve with fp64 refinement 2.63 s / 2.93 s (2.6-2.9 s over the round's runs; round 3: 3.6-3.9); the SpMM forms alone in steady state |
| `r04_mb_mix64.txt`, `r04_mb_gram64.txt`, `r04_c5_refine_breakdown.txt`, `r04_c5_refine_sweeps.txt` | configs[4]'s fp64 refinement: `ds_mix64` (fp64 MFMA over the list of blocks of the basis) 7.9 ms against 18.0 ms for the `torch.mm`/`addmm` chain; `ds_gram64_blocks` 8.8 ms against 17.1 (11.6 after the fp64 Gram kernels got exact wait counts) for eight `ds_gram` calls; where the 1.65 s of 16 steps go; 1 / 2 / 3 preconditioner sweeps per step |
| `r04_gather_loops_ab.txt` | gather loops that issued one load per trip and waited for it (found by an ISA scan of all kernels): restriction 32.8 -> 22.4 us with four gathers in flight; the numeric assembly does not gain from it (write-bound) |
| `r04_mb_polish.txt` | the read-out's fp64 products at C3: `ds_spmm_f64_polish` 0.65 ms (0.86 before its gather loop kept eight panels in flight), with the 80 x 240 exact Gram 1.10 ms |

## A note on the profiled runs

About one in ten `rocprofv3 --kernel-trace --stats -- python3 bench.py ...` runs with 8 lanes ends in a host SIGSEGV inside the HIP
runtime's launch path under the profiler's hooks (first seconds of the run; it happened once in round 3 and once in round 4, each
time an immediate rerun was clean; no unprofiled run or test has ever shown it). `collect_profiles.sh` gives that step a second try.

## Roofline figures of `r04_bench_n1.json` (HIP events inside `bench.py`) and of the kernel table

* dominant kernel `{r["kernel"]}`: {fb / 1e6:.1f} MB algorithmic per launch (SURVEY.md §8(d), BSR-3 count), {r["avg_launch_ms"]:.4f} ms alone on the device in steady state (100 back-to-back launches after 230 untimed ones, right after the timed region; the first 30 of them: {r["avg_launch_ms_first_30"]:.4f} ms)
  → **{r["achieved"]:.0f} GB/s = {100 * r["frac"]:.1f} % of 8 TB/s, {100 * r["frac_of_stream"]:.1f} % of the STREAM triad measured in the same run ({r["stream_triad"]:.0f} GB/s)**; the same under rocprofv3 (`{solo}`): {frac(fb, fine_solo)}; PMC traffic {pmc["cells26_cols80_mfma"]["bytes"] / 1e6:.1f} MB = {pmc["cells26_cols80_mfma"]["bytes"] / fb:.2f} × algorithmic;
  inside a pass, from `{l1}` alone: {frac(fb, fine)};
  in situ (8 lanes sharing the chip): fine {r["in_situ"]["levels"]["fine"]["avg_launch_ms"]:.3f} ms, corner-node {r["in_situ"]["levels"]["corner_node"]["avg_launch_ms"]:.3f} ms per launch;
* the eigensolver's K·W (`{kw["kernel"][:60]}…`): {kb / 1e6:.1f} MB in {kw["avg_launch_ms"]:.4f} ms on the iteration's own operands alone in steady state ({kw["avg_launch_ms_on_compact_blocks"]:.4f} ms on compact blocks; first 30 launches {kw["avg_launch_ms_first_30"]:.4f} ms) = {kw["achieved"]:.0f} GB/s = **{100 * kw["frac_of_stream"]:.1f} % of STREAM**; under rocprofv3 (`{solo}`, strided and compact operands together): {frac(kb, kx_solo)}; PMC traffic {pmc["cells26_cols80_kx"]["bytes"] / 1e6:.1f} MB = {pmc["cells26_cols80_kx"]["bytes"] / kb:.2f} × algorithmic;
  inside a pass, from `{l1}` alone: {frac(kb, kx)};
  in a fresh process, interleaved and warmed up (`r04_mb_kx_layout.txt`): 177 µs on compact blocks (42 %), 182–186 µs on the solver's operands (40–41 %);
* the fused residual (`spmm_union_kernel<20,4,…,0>`, K·X′ and M·X′ of one block in one walk): {frac(kb + 4136393 * 4.0, res)} (bytes: K·X's plus the mass scalars; the block R replaces Y);
* CPU baseline: {cb["sample"]} on {cb["cores"]} threads of {cb["cpu_model"]} ({cb["host_hardware_threads"]} hardware threads on the host); measured exponent {cb["measured_exponent"]["value"]:.2f}; extrapolated with it to the benchmark mesh {cb["extrapolated_to_benchmark_mesh_measured_exponent"]["seconds_per_pass"]:.0f} s per pass.

## One hypothesis at a time (`{l1}`; 15 passes incl. the target render, one warm-up step and the 4 full passes of the amortised variant; the passes' own launches only)

{table(l1, 26)}

`spmm_union_mfma_kernel<8, NT, EPI, OUT32, 16, LVL>`: the bf16 terms on the matrix cores (8 nodes per wavefront, NT 16-column
tiles: 5 = the full 80-column block, fewer after locking; EPI 1 fused Chebyshev term, 2 residual handed to the corner-node level;
OUT32 = the term that leaves the V-cycle; LVL 0 fine level, 1 corner-node level).
`spmm_union_kernel<20|0, EPI, 140, BIG, BF, OUT32, LVL>`: the VALU kernel — LPN 20 = full 80-column blocks, 0 = the narrower blocks
after locking; EPI 0 K·W, 3 mass product (M·W of the orthonormalisation), 4 the fused residual.

## Default run, 8 lanes (`r04_bench_kernel_stats.csv`)

{table("r04_bench_kernel_stats.csv", 16)}
'''
open(P("README.md"), "w").write(lines)
print("wrote profiles/README.md")
