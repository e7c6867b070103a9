#!/usr/bin/env python3
"""profiles/README.md for round 5 from the files under profiles/ (python tools/make_profiles_readme.py)."""
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



TAG = "r05"
d = json.load(open(P(f"{TAG}_bench_n1.json")))
r = d["roofline"]
kw = r["lobpcg_spmm"]
cb = d["cpu_baseline"]
pmc = json.load(open(P("spmm_pmc_bytes_per_launch.json")))
util = json.load(open(P("gram_mix_mfma_util.json")))
l1 = f"{TAG}_bench_lanes1_kernel_stats.csv"
stream_l1 = d["roofline"]["stream_triad"]  # (the one-lane table holds the passes' own launches only: no triad in it)
solo = f"{TAG}_bench_solo_kernel_stats.csv"
fine = kernel_row(l1, "spmm_union_mfma_kernel<8, 5, 1, false, 16, 0,")
fine_solo = kernel_row(solo, "spmm_union_mfma_kernel<8, 5, 1, false, 16, 0,")
km = kernel_row(l1, "spmm_union_kernel<20, 5, 140, false, false, true, 0>")
res = kernel_row(l1, "spmm_union_kernel<20, 4, 140, false, false, true, 0>")
kx = kernel_row(l1, "spmm_union_kernel<20, 0, 140, false, false, true, 0>")
fb, kb = r["algorithmic_bytes_per_launch"], kw["algorithmic_bytes_per_launch"]
ip = r["in_pass"]["kernels"]
rr = r["rayleigh_ritz"]
ms = cb["memory_safe_restatement"]


def frac(nbytes, row):
    g = nbytes / (float(row["avg_us"]) * 1e-6) / 1e9
    return f"{row['calls']} launches, avg {float(row['avg_us']):.1f} us (min {float(row['min_us']):.1f}, max {float(row['max_us']):.1f}) -> {g:.0f} GB/s = {100 * g / 8000:.1f} % of 8 TB/s, {100 * g / stream_l1:.1f} % of that run's STREAM triad ({stream_l1:.0f} GB/s)"


def ipl(key):
    if key not in ip:  # (since the start block went into coefficients a pass launches K W / M W only inside [K W | M W])
        return "not launched in a pass any more (the start block's K X0 and M X0 come out of the one-walk form too)"
    v = ip[key]
    t = f"{v['launches']} launches, avg {v['avg_launch_ms'] * 1e3:.1f} us (min {v['min_launch_ms'] * 1e3:.1f}) -> {v['achieved']:.0f} GB/s = {100 * v['frac']:.1f} % of 8 TB/s, {100 * v['frac_of_stream']:.1f} % of STREAM"
    if v.get("traffic"):
        t += f"; PMC traffic {v['traffic'] / 1e6:.1f} MB = {v['traffic_over_algorithmic']:.2f} x algorithmic"
    return t


def rrl(dct, key):
    v = dct[key]
    return f"{v.get('avg_ms'):.3f} ms = {v['tflops']:.1f} TF/s = {100 * v['frac_of_mfma_f32_peak']:.0f} % of 157.3 TF/s"


HIST = "| `r04_node_order_ab.txt` | A/B on one box of the node ORDERING: Morton curve over raw coordinates (round 3) against bricks aligned to the mesh's node planes — 45.5 → 48.0 passes/s, K·W 210 → 198 µs, bf16 term 2 500 → 3 100 GB/s |\n| `r04_union_cap_ab.txt` | chunk cap of the neighbour-union tables 116 → 140 blocks (every group one chunk): K·X 215 → 204 µs |\n| `r04_kx_fresh_ab.txt` | K·X′ by one fresh product instead of the update of K·[X P W]: 48.3 → 49.6 passes/s |\n| `r04_basis_row_pitch_ab.txt`, `r04_mb_kx_strided.txt`, `r04_mb_kx_layout.txt` | the solver's basis buffers with rows 1 KiB apart (256-column pitch instead of 248): K·X / M·X against operand layouts, interleaved and warmed up — compact blocks 177 / 137 µs, ranges of a 248-column buffer 193 / 165, of a 256-column buffer 182 / 152; +1–2 % passes/s |\n| `r04_fused_residual_ab.txt`, `r04_mb_fused_residual.txt` | the residual of every iteration in ONE walk of the unions (`ds_union_residual`): 459 µs (K·X′ + M·X′ + residual) → 280 µs (239 after the norm reduction was parallelised); 49.3 → 51.8 passes/s |\n| `r04_mfma32_knockout.txt`, `r04_m32_diag.txt`, `r04_mb_kx.txt`, `r04_mb_kx_raw_morton.txt`, `r04_mb_kx_plane_order.txt` | **the fp32 matrix-core form of the eigensolver's own products (`ds_spmm_union32m`): built, parity-green, slower** — K·X 263–277 µs against the VALU kernel's 205–235 on the same boxes; knock-out builds (no MFMAs 189, no gathers 249, no A-fragment chain 222, none 110); per-wave `s_memtime` stamps of a diagnostic build (`tools/m32_diag.py`): where a wave's cycles go in that kernel and in the bf16 term, the in-kernel clock (2.0–2.1 GHz), wave slots occupied |\n| `r04_corner_batch_ab.txt`, `r04_mb_corner.txt` | the corner-node level's bf16 term with batches of 16 and of 32 entries at its real size (2 461 groups): 18.2 against 23.5 µs — both levels run 16 |\n| `r04_union_waves_per_workgroup_ab.txt` | the VALU union kernel with 4 / 2 / 1 waves per workgroup: 208 / 213 / 215 µs (4 stays) |\n| `r04_hw_queues.txt`, `r04_block_sweep.txt`, `r04_lane_memory.txt` | hardware queues 2 / 4 (default) / 8 / 16: no gain; eigensolver block 72 / 76 / 80: 48.8 / 49.2 / 53.2 passes/s (80 stays); what one hypothesis lane holds in HBM, by tensor |\n| `r04_lanes_sweep.txt` | 6 / 8 / 12 / 16 hypothesis lanes: 47.9 / 49.2 / 49.7 / 49.0 passes/s (the device, not the host, is the bound) |\n| `r04_gram_mix.txt`, `r04_gram_mix_pmc.json` | Gram / `mix` timings at the solver's shapes and their MFMA counters |\n| `r04_symbolic_phase_timing.txt` | ord-2 lifting and the symbolic phase per topology at C3 |\n| `r04_c5_bench.json`, `r04_c5_kernel_stats.csv`, `r04_c5_bench_noprof.json` | **configs[4]** (`bench.py --workload c5` under `rocprofv3 --kernel-trace --stats`, and plain): 998 250 tets, n = 4.1 M, 128 modes - solve with fp64 refinement 2.63 s / 2.93 s (2.6-2.9 s over the round's runs; round 3: 3.6-3.9); the SpMM forms alone in steady state |\n| `r04_mb_mix64.txt`, `r04_mb_gram64.txt`, `r04_c5_refine_breakdown.txt`, `r04_c5_refine_sweeps.txt` | configs[4]'s fp64 refinement: `ds_mix64` (fp64 MFMA over the list of blocks of the basis) 7.9 ms against 18.0 ms for the `torch.mm`/`addmm` chain; `ds_gram64_blocks` 8.8 ms against 17.1 (11.6 after the fp64 Gram kernels got exact wait counts) for eight `ds_gram` calls; where the 1.65 s of 16 steps go; 1 / 2 / 3 preconditioner sweeps per step |\n| `r04_gather_loops_ab.txt` | gather loops that issued one load per trip and waited for it (found by an ISA scan of all kernels): restriction 32.8 -> 22.4 us with four gathers in flight; the numeric assembly does not gain from it (write-bound) |\n| `r04_mb_polish.txt` | the read-out's fp64 products at C3: `ds_spmm_f64_polish` 0.65 ms (0.86 before its gather loop kept eight panels in flight), with the 80 x 240 exact Gram 1.10 ms |"
lines = f"""# profiles/ — round 5 evidence (MI355X, gfx950, ROCm 7.2); round 4 as `r04_*`, what is still cited of round 3 as `r03_*`, round 1 under `r01/`

All runs: `bench.py` defaults = workload C3 (Kuhn box 26³ = 105 456 tets, ord-2, n = 446 631, nnz = 37.2 M, 64 modes, block 80,
two-level preconditioner on bf16 blocks with both levels' terms on the matrix cores, nested start to 3e-3, tolerance 1e-5,
Rayleigh-Ritz on the raw basis), 8 hypotheses per step, 8 in flight per GPU, cold-start eigensolve and numeric assembly in every pass.
Collected by `tools/collect_profiles.sh r05 <part>` on the GPU box (this file: `python tools/make_profiles_readme.py`).

| file | what |
|---|---|
| `r05_bench_n1.json` | the JSON line of `python bench.py` (N = 1, 10 steps, 2 warm-up, CPU baseline at two sizes included): **{d["value"]:.1f} passes/s**; amortised variant {d["amortised"]["value"]:.0f} passes/s; carries `roofline.in_pass`, `roofline.rayleigh_ritz`, `cpu_baseline.memory_safe_restatement`, `host_threads` |
| `r05_bench_kernel_stats.csv`, `r05_gpu_busy.txt` | `rocprofv3 --kernel-trace --stats` of the 8-lane run (the passes' own launches only; durations stretched by sharing) and the device-busy fraction inside its timed steps |
| `r05_bench_lanes1_kernel_stats.csv` | the same with `--lanes 1 --hyp-per-gpu 2 --steps 4`: one hypothesis at a time, every kernel alone on the device but between the other kernels of a pass — **the table to read in-pass kernel durations from** (last template argument of the SpMM symbols: 0 fine level, 1 corner-node level) |
| `r05_bench_solo_kernel_stats.csv` | the bench line's kernel-ALONE figures under the profiler |
| `r05_raw_rr_ab_off.json`, `r05_raw_rr_ab_on.json` | **A/B on one box of the round's main change — the Ritz step on the raw basis (`--raw-rr 0 / 1`): 54.4 → 60.3 passes/s**, same iteration counts; the lines carry the in-pass tables of both routes (explicit: K W 192 µs + M W 162 µs + two Gram + two updates per iteration; raw: one `[K W \\| M W]` walk 234 µs + one Gram + one update) |
| `r05_m32_diag.txt`, `r05_precond_sweep.txt` | per-wave `s_memtime` stamps of the bf16 term on both levels (diagnostic build: slower than production, the SHARES are what it is for — fine level: head 15 %, batches 35 %, fragments + MFMAs 22 %, epilogue 23 %; corner-node level 18 / 32 / 18 / 28 %); passes/s against the preconditioner's degrees on the new kernels, two interleaved repeats (the defaults 22 / 350 / 3 stay: 62.4 / 61.5) |
| `r05_launch_gaps.txt`, `r05_vcycle_kernel_stats.csv` | do launch gaps matter for one hypothesis at a time? Under rocprofv3 every 15.9 µs corner-level term is followed by ~10 µs of idle device; unprofiled the V-cycle's 32 launches take 1 104 µs against 1 067 µs of kernel time (1.2 µs per launch) — the gaps are the profiler's; a captured graph of the cycle would buy 3 % and was not built |
| `r05_deflated_corner_experiment.txt` | an algorithmic try at the corner-node level's 21 launches per cycle: its solve exact on the span of the nested start's 80 corner-level eigenvectors and a polynomial of lower degree on the rest — degree 12 / ratio 60 needs 7 fine iterations with the deflation and 7 without (production 22 / 350: 5); not pursued |
| `r05_raw_start_ab.txt` | the start block's projection / orthonormalisation / first Ritz step in coefficients (`SolverConfig.raw_start`: one `[K X0 \| M X0]` walk, one Gram launch, one update) against the explicit sequence: 59.6 / 60.7 / 63.3 → 62.6 / 63.6 / 65.2 passes/s in three interleaved pairs; taken in every start block of a run |
| `r05_warm_power_ab.txt`, `r05_warm_power_jump.txt`, `r05_norm_probe_ab.txt` | what a pass keeps from the previous hypothesis on the same geometry: the warm power iteration of the Chebyshev intervals stops after ONE step when its block is still converged (61.3 / 62.5 → 63.4 / 63.0 passes/s, interleaved); how low that one-step estimate falls under jumps of the Poisson ratio (worst 0.878 of the 200-step value, three steps 0.881: the 1.2 safety factor covers both); the operator-norm probe kept per geometry generation |
| `r05_host_wait_mode.txt` | two boxes of the pool ran the default benchmark at 50–54 passes/s with every one-stream figure normal: eight spinning lanes on a host with few free cores (a fast box restricted to four CPUs: 52–53). Sleeping waits (`hipDeviceScheduleBlockingSync` + `ds_host_wait_mode 1`): **63.0 on four CPUs**, 63.0 against 63.8 with cores to spare |
| `r05_order_ab.txt`, `r05_nested_knobs.txt`, `r05_lanes_sync_sweep.txt`, `r05_concurrency_profile.txt`, `r05_device_slots_experiment.txt`, `r05_host_profile_one_lane.txt` | experiments of the round that changed nothing in the product: bricks along a Hilbert curve (same kernel times, 2.5 % more traffic); the nested start's own polynomial (flat optimum); 12 / 16 hypotheses in flight and the host's wait modes (slower than 8); how many kernels overlap in the 8-lane run and where the device idles; a gate that lets only 8 solves on the device of 16 hypotheses in flight (+3 % for twice the HBM); cProfile of one hypothesis at a time (`ds_lobpcg_iterate` 83 % of a pass's host time) |
| `r05_mf_tail.txt` | the bf16 term kernel's round trips: a wave's head in ONE (fixed-stride records of a group's first 64 entries: 110.9–112.6 → 108.0–109.2 µs), what the fifth batch of a 65-entry group costs (a timing-only build without it: 106.3 → 94.3 µs), the tail form (the last batch takes two entries more, bit-identical): **97.7–98.0 µs = 0.489 of 8 TB/s** |
| `r05_nt_hint_ab.txt`, `r05_nt_epi_ab.txt` | four library builds on one box: non-temporal hints on the value / table loads (nt1), on the result stores (nt2), both (nt3) — K W 182 → 177 / 169 µs, M W 149 → 130 / 129, bf16 term 117 → 112 (loads) / 111 (both), corner-node level +4 % with the load hint (kept off there); the same hint on the epilogue's once-read operands costs 6 % (not adopted) |
| `spmm_pmc_bytes_per_launch.json`, `r05_spmm_pmc_{{fp32,bf16,mfma,kx,km,resid}}.json` | HBM-side traffic per launch from separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes; bytes = (2·FETCH + WRITE)·1024 (gfx950 correction of `guides/MI355X_MICROARCH.md`); keyed by the hash of the SpMM sources + `modal_ops.py` — `bench.py` reports a record with another hash as stale |
| `gram_mix_mfma_util.json`, `r05_gram_mix.txt` | matrix-pipe utilisation of the Rayleigh-Ritz kernels per shape (one shape per profiled process: `SQ_VALU_MFMA_BUSY_CYCLES` ÷ (`GRBM_GUI_ACTIVE`/8 × 1 024 SIMDs)), keyed by the hash of `gram.hip` + `blockops.hip`; timings at the solver's shapes |
| `r05_mb_solver_spmm.txt`, `r05_mb_narrow.txt`, `r05_mb_corner.txt`, `r05_mb_kx.txt` | the SpMM forms of one iteration alone on the solver's operand layout; the narrow-block kernel against the production kernel (8 / 16 columns: K X 131 → 115 µs on the fine level, slower for M X and on the corner-node level: used for the fine level's K X only); the corner-node level; K X / M X on compact blocks |
| `r05_host_time_one_lane.txt`, `r05_host_eigh_probe.txt` | where the host time of ONE hypothesis at a time goes (`DS_EXP_TIMING=1`: per solve, waiting for the stream / `dsyevd` / `dgemm` / rest; per pass, by stage with a synchronisation per stage; the first pass of the file is the set-up pass) and host LAPACK timings of the Ritz problem on the EPYC host (dsyevd 240: 2.35 ms standalone, ssyevd 1.52, dsyevr lowest third 4.6, torch/MKL 2.9; dgemm 119 GF/s on one thread) |
| `r05_cpu_memsafe_8.json`, `_12`, `_16` | **the memory-safe CPU restatement on the GPU box's host** (EPYC 9575F, 16 threads): 18.5 s, 67.4 s, 496.5 s per pass at 3 072 / 10 368 / 24 576 tets; ARPACK's shift-invert 11.5 / 47.3 / 449.5 s; exponent {ms["measured_exponent_between_last_two"]:.2f} between the last two → {ms["extrapolated_to_benchmark_mesh"]["seconds_per_pass"]:.0f} s at the benchmark mesh |
| `r05_barrier_probe.txt` | `tools/probes/barrier_probe.hip`: a device-wide barrier for 2 461 single-wave workgroups costs 41 µs (616 four-wave workgroups: 15 µs) with per-wave agent-scope fences — against 16.4 µs for the launch it would replace: why the corner-node level's polynomial is not one persistent launch |
| `r05_rocprof_lanes_stress_30_sleeping_waits.txt` | the same stress on the round's final library (lanes sleep while they wait for the device): 0 of 30 profiled 8-lane runs failed |
| `r05_rocprof_lanes_stress_14.txt`, `r05_rocprof_lanes_stress_40.txt` | 54 profiled 8-lane runs with python's faulthandler on: 2 SIGSEGV, both inside the profiler's interception of a HIP call made from `ds_lobpcg_iterate` on a lane thread (backtraces in the file); no unprofiled run has ever shown it |
| `r05_c5_bench.json`, `r05_c5_kernel_stats.csv`, `r05_c5_bench_noprof.json` | **configs[4]** (`bench.py --workload c5`, under rocprofv3 and plain): fp32 phase 0.75 s, with the fp64 refinement 2.2-2.5 s over the round's runs (run-to-run spread, identical step counts); the 136-column block on the union kernels and the native driver (two 68-column slices), the refinement's fp64 K W / M W on the union tables (37-38 % of STREAM; wave-per-node kernel 27-29 %) |
| `r05_symbolic_phase_timing.txt` | ord-2 lifting and the symbolic phase per topology at C3 |
{HIST}

## A note on the profiled runs

`rocprofv3 --kernel-trace --stats -- python3 bench.py ...` with 8 lanes (8 host threads launching concurrently) occasionally ends in a
host SIGSEGV: about 1 run in 10 in round 4, **2 of 54 in round 5** (`r05_rocprof_lanes_stress_*.txt`, python's faulthandler on). Both
backtraces end inside the profiler's interception layer under a HIP call (`hipMemcpyAsync` / `hipStreamSynchronize`) issued directly
from `ds_lobpcg_iterate` on a lane thread; the fault address is not in this library's memory. No unprofiled run or test has ever
shown it (this round: ~40 unprofiled 8-lane bench runs, three full GPU suites). `collect_profiles.sh` gives that step a second try.

## Roofline figures of `r05_bench_n1.json` (HIP events inside `bench.py`) and of the kernel table

* dominant kernel `{r["kernel"]}`: {fb / 1e6:.1f} MB algorithmic per launch (SURVEY.md §8(d), BSR-3 count), {r["avg_launch_ms"]:.4f} ms alone on the device in steady state (100 back-to-back launches after 230 untimed ones; the first 30 of them: {r["avg_launch_ms_first_30"]:.4f} ms)
  → **{r["achieved"]:.0f} GB/s = {100 * r["frac"]:.1f} % of 8 TB/s, {100 * r["frac_of_stream"]:.1f} % of the STREAM triad measured in the same run ({r["stream_triad"]:.0f} GB/s)**; the same under rocprofv3 (`{solo}`): {frac(fb, fine_solo)}; PMC traffic {pmc["cells26_cols80_mfma"]["bytes"] / 1e6:.1f} MB = {pmc["cells26_cols80_mfma"]["bytes"] / fb:.2f} × algorithmic;
  **in a pass** (`roofline.in_pass`, HIP events around every launch of one-at-a-time passes): {ipl("fused_term_bf16")}; from `{l1}` (rocprofv3): {frac(fb, fine)};
* `[K W | M W]` in one walk (`spmm_union_kernel<20,5,…>`, round 5): in a pass {ipl("lobpcg_kw_mw")}; rocprofv3: {km["calls"]} launches, avg {float(km["avg_us"]):.1f} µs;
* the fused residual (`<20,4,…>`): in a pass {ipl("fused_residual")}; rocprofv3: {res["calls"]} launches, avg {float(res["avg_us"]):.1f} µs;
* K·W alone (`<20,0,…>`; now only on explicit-route iterations and at the start of a solve): {kb / 1e6:.1f} MB in {kw["avg_launch_ms"]:.4f} ms alone in steady state on the iteration's own operands ({kw["avg_launch_ms_on_compact_blocks"]:.4f} ms on compact blocks) = **{100 * kw["frac_of_stream"]:.1f} % of STREAM**; in a pass {ipl("lobpcg_kw")}; M·W in a pass {ipl("lobpcg_mw")};
* Rayleigh-Ritz kernels (fp32 MFMA, peak 157.3 TF/s): Gram 256×160 alone {rrl(rr["alone"], "gram_256x160")}, in a pass {rrl(rr["in_pass"], "gram_256x160")}, matrix pipe busy {100 * util["kernels"]["gram_256x160"]["mfma_pipe_utilisation"]:.0f} %; update 256→160 alone {rrl(rr["alone"], "mix_256x160")}, in a pass {rrl(rr["in_pass"], "mix_256x160")}, pipe busy {100 * util["kernels"]["mix_256x160"]["mfma_pipe_utilisation"]:.0f} %; explicit-route shapes alone: Gram 240×80 {rrl(rr["alone"], "gram_240x80")} (pipe {100 * util["kernels"]["gram_240x80"]["mfma_pipe_utilisation"]:.0f} %), update 240→160 {rrl(rr["alone"], "mix_240x160")} (pipe {100 * util["kernels"]["mix_240x160"]["mfma_pipe_utilisation"]:.0f} %);
* one hypothesis at a time: {r["in_pass"]["seconds_per_pass_one_at_a_time"] * 1e3:.1f} ms per pass;
* CPU baseline: {cb["sample"]} on {cb["cores"]} threads of {cb["cpu_model"]} ({cb["host_hardware_threads"]} hardware threads on the host); measured exponent {cb["measured_exponent"]["value"]:.2f}; memory-safe restatement on the same CPU model: {", ".join(f"{q['seconds_per_pass']:.1f} s at {q['tets']} tets" for q in ms["points"])}.

## One hypothesis at a time (`{l1}`; 15 passes incl. the target render and one warm-up step; the passes' own launches only)

{table(l1, 28)}

`spmm_union_mfma_kernel<8, NT, EPI, OUT32, 16, LVL>`: the bf16 terms on the matrix cores (8 nodes per wavefront, NT 16-column
tiles: 5 = the full 80-column block, fewer after locking; EPI 1 fused Chebyshev term, 2 residual handed to the corner-node level;
OUT32 = the term that leaves the V-cycle; LVL 0 fine level, 1 corner-node level).
`spmm_union_kernel<20|0, EPI, 140, BIG, BF, OUT32, LVL>`: the VALU kernel — LPN 20 = full 80-column blocks, 0 = other widths;
EPI 0 K·X, 3 mass product, 4 the fused residual, 5 `[K W | M W]` in one walk.  `spmm_union_narrow_kernel`: K·X on ≤ 16 columns.

## Default run, 8 lanes (`{TAG}_bench_kernel_stats.csv`)

{table(f"{TAG}_bench_kernel_stats.csv", 16)}
"""
open(P("README.md"), "w").write(lines)
print("wrote profiles/README.md")
