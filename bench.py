#!/usr/bin/env python3
"""Headline benchmark: forward+backward modal-analysis passes/sec on the 100k-tet ord-2 mesh, 64 modes.

    python bench.py --gpus N --steps K --warmup W
    (N > 1 without WORLD_SIZE in the environment: the script re-launches itself as
     python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...,
     one rank per GPU over RCCL, and exits with that job's status; fewer than N visible devices is an error)

One *step* = every rank runs ``--hyp-per-gpu`` complete passes (numeric assembly of K_lambda / K_mu / M, cold-start
block eigensolve for 64 elastic modes, differentiable frequency read-out, oscillator render, MSE loss,
backward to (E, nu)) for its own material hypotheses on the shared synthetic mesh, then the scalar
losses are all-reduced (RCCL) - weak scaling, no other collective.  Rank 0 prints ONE JSON line.
Inputs are synthetic (Kuhn box mesh) and resident in HBM before the timed region.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

MAT = (2700.0, 5e10, 0.25, 6.0, 1e-7)
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (guides/MI355X_MICROARCH.md); the STREAM triad below is MEASURED in the run
MFMA_F32_PEAK_TFS = 157.3  # v_mfma_f32_16x16x4_f32: 256 flop / cycle / CU x 256 CUs x 2.4 GHz (the same guide); fp32 Gram / update kernels
PROF_KINDS = {0: "term", 1: "kx", 2: "mx", 3: "resid", 4: "gram", 5: "mix", 6: "km"}  # DS_PROF_* of include/diffsound_hip.h


# (the node ordering and the union tables decide the traffic as much as the kernels do: modal_ops.py is part of the key)
SPMM_SOURCES = ("spmm.hip", "spmm_union.inc", "spmm_mfma.inc", "spmm_mfma32.inc", "ds_common.h", "ds_diag.h", "../modal_ops.py")


_LOAD_AT_START = None
try:
    _LOAD_AT_START = os.getloadavg()
except OSError:
    pass


def _cgroup_cpu_stat():
    """nr_periods / nr_throttled / throttled time of this process's CPU cgroup (v2: cpu.stat with throttled_usec; v1: throttled_time
    in ns) and its quota - a GPU box gives an ordinary user a share of the host's CPUs, and a process that asks for more is paused."""
    out = {}
    for path in ("/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu/cpu.stat", "/sys/fs/cgroup/cpu,cpuacct/cpu.stat"):
        try:
            with open(path) as f:
                for line in f:
                    k, _, v = line.partition(" ")
                    if k in ("nr_periods", "nr_throttled", "throttled_usec", "throttled_time", "usage_usec"):
                        out[k] = int(v)
            out["source"] = path
            break
        except OSError:
            continue
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                out["quota"] = f.read().strip()
            break
        except OSError:
            continue
    return out


_CGROUP_AT_START = _cgroup_cpu_stat()
_CGROUP_TIMED = None
_CGROUP_MARKS = [("process start", _CGROUP_AT_START)]


def _cg_mark(label):
    """A named point of the run for the per-leg view of the CPU cgroup's counters (host_load.cgroup_cpu.legs)."""
    _CGROUP_MARKS.append((label, _cgroup_cpu_stat()))


def host_load_record():
    """What else runs on the GPU box's host (a box is one GPU of a shared 8-GPU host): the 1 / 5 / 15-minute load averages when this
    process started and now (this run itself adds about its lane threads), and the time of one 240 x 240 dsyevd on one thread - the
    Ritz step's size - as a probe of the host's speed.  The eight-lane rate and the API legs move with these."""
    import time as _t

    rec = {"loadavg_at_start": None if _LOAD_AT_START is None else [round(x, 2) for x in _LOAD_AT_START]}
    now = _cgroup_cpu_stat()
    rec["cgroup_cpu"] = {"quota": now.get("quota"), "source": now.get("source"),
                         "during_this_process": {k: now[k] - _CGROUP_AT_START.get(k, 0) for k in now if isinstance(now[k], int)},
                         "during_the_timed_region": _CGROUP_TIMED, "torch_threads": __import__("torch").get_num_threads(),
                         "legs": [{"until": lb, **{k: st[k] - prev.get(k, 0) for k in ("usage_usec", "nr_throttled", "throttled_usec", "throttled_time")
                                                   if k in st}}
                                  for (_, prev), (lb, st) in zip(_CGROUP_MARKS[:-1], _CGROUP_MARKS[1:])]}
    try:
        rec["loadavg_now"] = [round(x, 2) for x in os.getloadavg()]
    except OSError:
        rec["loadavg_now"] = None
    try:
        import numpy as _np
        from scipy.linalg import lapack as _la

        a_ = _np.random.default_rng(0).standard_normal((240, 240))
        a_ = a_ + a_.T
        ts = []
        try:
            from threadpoolctl import threadpool_limits as _lim
        except Exception:
            import contextlib as _cl

            _lim = lambda n: _cl.nullcontext()
        with _lim(1):
            for _ in range(7):
                t0 = _t.perf_counter()
                _la.dsyevd(a_)
                ts.append(_t.perf_counter() - t0)
        rec["dsyevd_240_ms_median_of_7"] = round(1e3 * sorted(ts)[3], 3)
    except Exception as e:  # (a diagnostic: never the reason a run fails)
        rec["dsyevd_240_ms_median_of_7"] = None
        rec["probe_error"] = str(e)[:80]
    return rec


def spmm_source_hash():
    """sha256 (16 hex digits) over the sources of the SpMM kernels: the key that ties a PMC traffic figure under
    profiles/ to the kernel it was measured on (tools/pmc_bytes.py records it; a figure with another key is stale)."""
    import hashlib

    h = hashlib.sha256()
    for name in SPMM_SOURCES:
        with open(os.path.join(ROOT, "diffsound_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


DENSE_SOURCES = ("gram.hip", "blockops.hip", "ds_common.h")


def dense_source_hash():
    """The same key for the Gram / update kernels' counter record (profiles/gram_mix_mfma_util.json)."""
    import hashlib

    h = hashlib.sha256()
    for name in DENSE_SOURCES:
        with open(os.path.join(ROOT, "diffsound_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--cells", type=int, default=26, help="Kuhn box cells per edge (26 -> 105 456 tets)")
    ap.add_argument("--order", type=int, default=2)
    ap.add_argument("--modes", type=int, default=64)
    ap.add_argument("--hyp-per-gpu", type=int, default=8, help="material hypotheses per GPU per step")
    ap.add_argument("--lanes", type=int, default=8,
                    help="hypotheses in flight at once per GPU (own HIP stream + host thread each), so one lane's "
                         "host-side Rayleigh-Ritz step overlaps the other lane's kernels")
    ap.add_argument("--host-wait", default="sleep", choices=["sleep", "spin"],
                    help="how a lane's host thread waits for its stream: sleep (poll 20 us, then a blocking event; default with "
                         "several lanes) or spin (the HIP runtime's stream synchronisation)")
    ap.add_argument("--step-barrier", action="store_true",
                    help="join all hypothesis lanes after every step (the schedule of rounds 1-2). Default: a lane runs its "
                         "hypotheses' passes of consecutive steps back to back - a hypothesis' next step depends on its own "
                         "previous one only - and the per-step loss all-reduce is issued as soon as every lane has finished "
                         "that step")
    ap.add_argument("--cheb-degree", type=int, default=48)
    ap.add_argument("--cheb-ratio", type=float, default=800.0)
    ap.add_argument("--block", type=int, default=80)
    ap.add_argument("--precond", default="auto", choices=["auto", "chebyshev", "twolevel"],
                    help="auto = two-level V-cycle on ord-2 meshes, one-level Chebyshev polynomial otherwise")
    ap.add_argument("--smooth-degree", type=int, default=3)
    ap.add_argument("--smooth-ratio", type=float, default=10.0)
    ap.add_argument("--coarse-degree", type=int, default=22)
    ap.add_argument("--coarse-ratio", type=float, default=350.0)
    ap.add_argument("--power-iters", type=int, default=30,
                    help="power iterations for lambda_max(T K) of the Chebyshev intervals (a quarter of them on every pass "
                         "after the first); 0 = none, the rigorous bound (nodes per element: 10 / 4) is the interval's end")
    ap.add_argument("--warm-power-iters", type=int, default=-1,
                    help="power iterations on every pass after the first (the dominant block of the previous material's "
                         "estimate is the start); -1 = library default")
    ap.add_argument("--rr-refresh", type=int, default=-1,
                    help="recompute K [X P W] and the whole Gram matrix every this many iterations (-1 = solver default)")
    ap.add_argument("--kx-fresh", type=int, default=-1,
                    help="1: K X' of every new Ritz block by one product (solver default), 0: by the update of K [X P W]")
    ap.add_argument("--raw-rr", type=int, default=-1,
                    help="1: Rayleigh-Ritz on the raw basis (solver default: [K W | M W] in one walk, one Gram, one update per "
                         "iteration), 0: the explicit sequence M W, Gram, update of W, K W, Gram, update")
    ap.add_argument("--fused-residual", type=int, default=-1,
                    help="1: the residual of every iteration in one walk of the unions (solver default), 0: K X', M X', residual")
    ap.add_argument("--tol", type=float, default=1e-5,
                    help="backward-error tolerance of the eigensolve, ||K u - lambda M u|| / (||u|| (||K|| + lambda ||M||)) "
                         "per wanted pair: 1e-5 is the tolerance the metric states for fp32 iterates (SURVEY.md 8(d)); "
                         "0 = the library default 2e-6")
    ap.add_argument("--nested-tol", type=float, default=3e-3,
                    help="> 0: nested iteration - the random start is first iterated on the corner-node level to this "
                         "backward error and prolongated (SolverConfig.nested_tol); 0 = start the fine level from the "
                         "random block directly")
    ap.add_argument("--nested-maxit", type=int, default=8)
    ap.add_argument("--start-sweeps", type=int, default=2,
                    help="applications of the preconditioner to the random start block before its first Ritz step "
                         "(SolverConfig.start_sweeps; 0 = rounds 1-5)")
    ap.add_argument("--nested-ritz-tol", type=float, default=0.2,
                    help="a pair of the corner-node phase counts as converged only when its Ritz value also moved by less than this "
                         "(relative) in the last step (SolverConfig.nested_ritz_tol; 0 = the backward error alone)")
    ap.add_argument("--warm-start", action="store_true", help="amortised variant: reuse the previous block")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (the product); gloo = rehearsal of the N > 1 path without RCCL")
    ap.add_argument("--share-devices", action="store_true",
                    help="rehearsal on a box with fewer GPUs than ranks: rank r uses device r %% device_count "
                         "(needs --dist-backend gloo: RCCL refuses two ranks on one device)")
    ap.add_argument("--ortho-tol", type=float, default=-1.0,
                    help="a second orthogonalisation sweep runs only when eps x amplification of the first exceeds this "
                         "(< 0: the library default 2e-6; on the benchmark the first sweep suffices either way - the launch "
                         "counts with 2e-6 and 1e-5 are identical)")
    ap.add_argument("--ortho-passes", type=int, default=-1)
    ap.add_argument("--mfma-groups", default="8,8",
                    help="nodes per wavefront of the MFMA form of the preconditioner's bf16 terms on the fine and on the "
                         "corner-node level (8, or 0 = the VALU kernel on that level)")
    ap.add_argument("--group-jacobi", type=int, default=8, choices=[0, 8],
                    help="the corner-node level's polynomial on the group-block Jacobi (inverse of the 24 x 24 diagonal block of every "
                         "8-node group of the matrix-core tables; degree / ratio: --group-degree / --group-ratio) or, 0, on the "
                         "3 x 3 node blocks (--coarse-degree / --coarse-ratio; rounds 2-5)")
    ap.add_argument("--one-level-group", type=int, default=-1, choices=[-1, 0, 8],
                    help="--workload geom: the one-level polynomial of the ord-1 meshes on the group-block Jacobi (8) or on the node blocks "
                         "(0); -1 = the library's default; its degree / ratio: --cheb-group-degree / --cheb-group-ratio")
    ap.add_argument("--cheb-group-degree", type=int, default=0)
    ap.add_argument("--cheb-group-ratio", type=float, default=0.0)
    ap.add_argument("--group-degree", type=int, default=14)
    ap.add_argument("--group-ratio", type=float, default=150.0)
    ap.add_argument("--loss", default="mse", choices=["mse", "mss"],
                    help="scalar loss head: mse = the headline metric's; mss = the reference experiments' multi-scale "
                         "spectral loss (MSSLoss [1024..64], 'l1_loss', material_sync_train.py:124) on the STFT kernels")
    ap.add_argument("--geom-iters", type=int, default=200, help="--workload geom: iterations of the shape loop (HBM growth is read over them)")
    ap.add_argument("--workload", default="c3", choices=["c3", "c5", "geom"],
                    help="c3 = the headline benchmark (BASELINE.json configs[2]); c5 = configs[4], the 1M-tet / 128-mode / fp64 "
                         "stress (one rank): SpMM bandwidth of every product form at that size and the fp32 + fp64-refined "
                         "solve - its own JSON line, not the headline metric; geom = the shape loop of SURVEY.md 8(f2, f3) "
                         "(src/dmtet/geometry/dmtet_thickness.py:237-299): every iteration new vertices AND a new topology, a fresh "
                         "DiffSoundObj, eigen_decomposition, get_vals, relative MSE, backward to the vertices - its own JSON line")
    ap.add_argument("--no-solo", action="store_true",
                    help="skip the roofline object's kernel-alone measurements (hundreds of launches of the fused term, K W and "
                         "the STREAM triad after the timed region): for a rocprofv3 kernel table that should hold nothing but "
                         "the passes' own launches (tools/collect_profiles.sh, the one-lane table)")
    ap.add_argument("--amortised-cycle", type=int, default=15,
                    help="also time the amortised variant the reference trains with (eigendecomposition every this many "
                         "passes, material_sync_train.py:135-141: EIGEN_DECOMPOSE_CYCLE = 15); 0 = skip")
    ap.add_argument("--no-affinity", action="store_true",
                    help="do not bind the rank to the CPUs of its device's NUMA node (diffsound_amd.hostcpu; default: bind, from sysfs, "
                         "before the first GPU call)")
    ap.add_argument("--no-api-path", action="store_true",
                    help="skip the legs behind the timed region that time the path a reference user runs: one hypothesis at a time, and "
                         "the literal loop body of experiments/material_sync_train.py:137-168 through DiffSoundObj / "
                         "TraditionalDampedOscillator / MSSLoss / Adam")
    ap.add_argument("--api-epochs", type=int, default=30, help="epochs of the EIGEN_DECOMPOSE_CYCLE = 15 leg of the API loop")
    ap.add_argument("--same-material", action="store_true",
                    help="every step repeats each hypothesis' (E, nu) exactly (rounds 1-5); default: the material moves every step, "
                         "E (1 + 1e-3 s), nu (1 - 5e-4 s), as an optimiser's would - no pass sees the material its lane saw before")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--geom-cpu-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-reps", type=int, default=3, help="full passes of the CPU oracle (a fresh process each)")
    ap.add_argument("--cpu-sample-cells", type=int, default=8,
                    help="the CPU oracle runs ONE full pass on a Kuhn box of this many cells per edge (8 -> 3072 tets, the "
                         "smallest ord-2 size of BASELINE.md section 3; 12 takes several minutes)")
    ap.add_argument("--cpu-second-cells", type=int, default=12,
                    help="a SECOND measured size of the CPU oracle (one pass; 12 -> 10 368 tets, about a minute on the GPU box's "
                         "host): the line then carries a measured scaling exponent between the two sizes instead of only the "
                         "linear extrapolation; 0 = skip")
    return ap.parse_args()


def solver_config(a=None, **over):
    """The eigensolver settings of the benchmark (``a``: parsed arguments; None = the defaults) - also what the
    parity tests of tests/test_parity_gpu.py run, so the numbers timed here are the numbers checked there."""
    from diffsound_amd.lobpcg.modal_solver import SolverConfig

    if a is None:
        saved, sys.argv = sys.argv, sys.argv[:1]
        try:
            a = parse()
        finally:
            sys.argv = saved
    for k, v in over.items():
        setattr(a, k, v)
    cfg = SolverConfig(block=a.block, cheb_degree=a.cheb_degree, cheb_ratio=a.cheb_ratio,
                       lmax_cap=float({1: 4, 2: 10}[a.order]), precond=a.precond, smooth_degree=a.smooth_degree,
                       smooth_ratio=a.smooth_ratio, coarse_degree=a.coarse_degree, coarse_ratio=a.coarse_ratio)
    if a.rr_refresh >= 0:
        cfg.rr_refresh = a.rr_refresh
    if getattr(a, "kx_fresh", -1) >= 0:
        cfg.kx_fresh = bool(a.kx_fresh)
    if getattr(a, "fused_residual", -1) >= 0:
        cfg.fused_residual = bool(a.fused_residual)
    if getattr(a, "raw_rr", -1) >= 0:
        cfg.raw_rr = bool(a.raw_rr)
    cfg.tol = a.tol
    cfg.power_iters = a.power_iters
    if a.warm_power_iters > 0:  # (a field of THIS configuration - round 5 wrote a class attribute of the library)
        cfg.warm_power_iters = a.warm_power_iters
        cfg.warm_power_spread = 0.0
    if getattr(a, "ortho_tol", -1.0) >= 0:
        cfg.ortho_tol = a.ortho_tol
    if getattr(a, "ortho_passes", -1) > 0:
        cfg.ortho_passes = a.ortho_passes
    cfg.start_sweeps = getattr(a, "start_sweeps", 0)
    cfg.nested_ritz_tol = getattr(a, "nested_ritz_tol", 0.2)
    cfg.group_degree, cfg.group_ratio = getattr(a, "group_degree", 14), getattr(a, "group_ratio", 150.0)
    cfg.nested_tol, cfg.nested_maxit = a.nested_tol, a.nested_maxit
    cfg.nested_cheb_degree, cfg.nested_cheb_ratio = a.coarse_degree, a.coarse_ratio
    return cfg


def relaunch_as_ranks(a):
    """``--gpus N`` with N > 1 and no WORLD_SIZE: start N ranks (one per GPU) under torch.distributed.run as a CHILD
    process and exit with its status.  Runs before anything touches the GPU (``device_count`` does not initialise it).
    Never falls back to fewer ranks."""
    if "WORLD_SIZE" in os.environ:
        world = int(os.environ["WORLD_SIZE"])
        if world != a.gpus:
            raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}; launch with "
                             f"`python bench.py --gpus {a.gpus}` or torch.distributed.run --nproc-per-node {a.gpus}")
        return
    if a.gpus <= 1:
        return
    ndev = torch.cuda.device_count()
    if a.share_devices and a.dist_backend != "gloo":
        raise SystemExit("bench.py: --share-devices needs --dist-backend gloo")
    if ndev < a.gpus and not (a.share_devices and ndev >= 1):
        raise SystemExit(f"bench.py: --gpus {a.gpus} but only {ndev} HIP device(s) are visible")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // a.gpus)))
    sys.exit(subprocess.call(cmd, env=env))


def cpu_baseline(sample_cells, order, modes, full_tets, reps=3, second_cells=0):
    """The CPU oracle timed in a FRESH CHILD PROCESS (its own BLAS / OpenMP state: nothing the GPU run did to the
    process - thread limits, pinned memory, lane threads - can skew it), ``reps`` full passes; ``value`` = 1 / median."""
    nthreads = min(os.cpu_count() or 1, 16)
    env = dict(os.environ)
    env["OMP_NUM_THREADS"] = env["OPENBLAS_NUM_THREADS"] = env["MKL_NUM_THREADS"] = str(nthreads)
    env["HIP_VISIBLE_DEVICES"] = env["CUDA_VISIBLE_DEVICES"] = ""  # the child never touches the GPU
    def child(cells):
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-child", "--cpu-sample-cells",
                              str(cells), "--order", str(order), "--modes", str(modes), "--cells",
                              str(round((full_tets / 6) ** (1 / 3)))], capture_output=True, text=True, env=env, cwd=ROOT)
        if out.returncode != 0:
            raise SystemExit("bench.py: CPU baseline child failed:\n" + out.stderr[-2000:])
        return json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])

    runs = [child(sample_cells) for _ in range(max(1, reps))]
    secs = sorted(r["sample_seconds"] for r in runs)
    med = runs[[r["sample_seconds"] for r in runs].index(secs[len(secs) // 2])]
    med = dict(med)
    med["repetitions"] = len(runs)
    med["sample_seconds_all"] = [round(r["sample_seconds"], 3) for r in runs]
    med["sample"] = med["sample"] + f" (median of {len(runs)} passes, each in a fresh process: {med['sample_seconds_all']} s)"
    if second_cells and second_cells > sample_cells:
        # the FAITHFUL restatement's measured scaling between two sizes on this host, and the
        # benchmark-mesh time extrapolated with the measured exponent (still optimistic: ARPACK's LU grows as n^2.5 between
        # 10k and 25k tets, profiles/r03_cpu_memsafe_*_container.json); (i) - a measured pass at the benchmark mesh - is about
        # 7 hours of shift-invert alone and is not attempted
        big = child(second_cells)
        n0 = int(med["unit"].split(" at ")[1].split(" tets")[0])
        n1 = int(big["unit"].split(" at ")[1].split(" tets")[0])
        p = float(np.log(big["sample_seconds"] / med["sample_seconds"]) / np.log(n1 / n0))
        t_full = big["sample_seconds"] * (full_tets / n1) ** max(p, 1.0)
        med["second_size"] = {"tets": n1, "sample_seconds": big["sample_seconds"], "stage_seconds": big["stage_seconds"],
                              "value": big["value"], "unit": big["unit"]}
        med["measured_exponent"] = {"value": p, "between_tets": [n0, n1],
                                    "what": "d log(seconds per pass) / d log(tets) of the faithful CPU restatement on this host"}
        med["extrapolated_to_benchmark_mesh_measured_exponent"] = {
            "value": 1.0 / t_full, "unit": "passes/s", "seconds_per_pass": t_full,
            "how": (f"{big['sample_seconds']:.1f} s at {n1} tets x ({full_tets} / {n1}) ^ {max(p, 1.0):.2f} - BASELINE.md section 3 (ii), "
                    "the faithful restatement's curve; (i), a measured pass at the benchmark mesh, is infeasible (hours of "
                    "shift-invert LU, > 58 GB)")}
        med["sample"] += (f"; second size {second_cells}^3 cells ({n1} tets): {big['sample_seconds']:.1f} s -> measured exponent "
                          f"{p:.2f} in the tet count")
    return med


def memsafe_curve(full_tets, cpu_model):
    """the MEMORY-SAFE restatement (tools/cpu_baseline_memsafe.py: element matrices pre-summed over
    the Gauss points, read-out eight modes at a time under checkpointing - the same arithmetic) MEASURED on the GPU box's host in
    round 5 at 3 072, 10 368 and 24 576 tets (profiles/r05_cpu_memsafe_*.json; the largest size is 8 minutes, the benchmark mesh
    would be hours: ARPACK's shift-invert LU grows as tets^2.3 between the last two sizes).  A record, not a measurement of THIS run:
    it says which CPU it was taken on."""
    pts = []
    for c in (8, 12, 16):
        try:
            d = json.load(open(os.path.join(ROOT, "profiles", f"r05_cpu_memsafe_{c}.json")))
            pts.append({"tets": d["tets"], "seconds_per_pass": d["seconds"], "arpack_shift_invert_seconds": d["stage_seconds"]["arpack_shift_invert"],
                        "threads": d["threads"], "cpu_model": d["cpu_model"]})
        except (OSError, ValueError, KeyError):
            pass
    if len(pts) < 2:
        return None
    p = float(np.log(pts[-1]["seconds_per_pass"] / pts[-2]["seconds_per_pass"]) / np.log(pts[-1]["tets"] / pts[-2]["tets"]))
    t_full = pts[-1]["seconds_per_pass"] * (full_tets / pts[-1]["tets"]) ** p
    return {"points": pts, "same_cpu_model_as_this_run": bool(cpu_model) and all(q["cpu_model"] == cpu_model for q in pts),
            "measured_exponent_between_last_two": p,
            "extrapolated_to_benchmark_mesh": {"seconds_per_pass": t_full, "value": 1.0 / t_full, "unit": "passes/s",
                                               "how": f"{pts[-1]['seconds_per_pass']:.0f} s at {pts[-1]['tets']} tets x ({full_tets} / {pts[-1]['tets']}) ^ {p:.2f}"},
            "source": "profiles/r05_cpu_memsafe_{8,12,16}.json (tools/cpu_baseline_memsafe.py on the GPU box's host, 16 threads)"}


def cpu_baseline_pass(sample_cells, order, modes, full_tets):
    """The CPU oracle (faithful restatement of the reference loop body) MEASURED on one full
    fwd+bwd pass at a stated size, with per-stage times.  ``value`` is passes/s AT THAT SIZE; the linear extrapolation
    to the benchmark mesh is reported separately (optimistic for the CPU: the reference's assembly and ARPACK's LU are
    super-linear)."""
    from diffsound_amd import meshgen
    from oracle import fem, modal
    from oracle import oscillator as oosc

    # 16 threads = the host share of one GPU on this pool; measured on the MI355X host: 1 thread 3.9 s, 4-64 threads
    # 2.9 s, all 256 hardware threads 18.2 s on the 750-tet sample (oversubscribed: every small op forks a big team)
    nthreads = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(nthreads)
    blas_threads = None
    try:
        from threadpoolctl import threadpool_info

        blas_threads = {i.get("internal_api", i.get("user_api")): i.get("num_threads") for i in threadpool_info()}
    except Exception:
        pass
    v, t = meshgen.kuhn_box(sample_cells)
    stages = {}
    clock = [time.time()]

    def lap(name):
        now = time.time()
        stages[name] = round(now - clock[0], 3)
        clock[0] = now

    t0 = clock[0]
    v, t = fem.to_high_order(torch.from_numpy(v), torch.from_numpy(t).long(), order)
    d = fem.OracleDeform(v, t, order)
    lap("mesh_lifting_and_shape_function_derivatives")
    E = torch.tensor(MAT[1], requires_grad=True)
    nu = torch.tensor(MAT[2], requires_grad=True)
    lam, mu = fem.lame(MAT[1], MAT[2])
    M3, _ = fem.assemble_mass(v, t, order, MAT[0])
    lap("mass_assembly")
    K = fem.assemble_stiffness_faithful(d, lam, mu)
    lap("stiffness_assembly")
    ev, U, _, _ = modal.eigsh_shift_invert(K, M3, modes)
    lap("arpack_shift_invert")
    f = modal.undamped_freqs_material(d, M3, ev, U, E, nu)
    lap("get_undamped_freqs")
    force = torch.zeros((1, 150))
    force[0, 0] = 1
    sig, _ = oosc.bank(f.float(), force, 8000, 32000, MAT[3], MAT[4])
    loss = (sig ** 2).mean()
    lap("oscillator_and_loss")
    loss.backward()
    lap("backward")
    dt = time.time() - t0
    ntets = t.shape[0]
    cpu_model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu_model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {
        "value": 1.0 / dt,
        "unit": f"passes/s at {ntets} tets (NOT the benchmark mesh)",
        "cores": nthreads,
        "thread_pools": blas_threads,
        "cpu_model": cpu_model,
        "host_hardware_threads": os.cpu_count(),
        "kind": "port",
        "sample": (f"ONE full fwd+bwd pass of the CPU oracle on a {sample_cells}^3-cell Kuhn box ({ntets} tets, "
                   f"ord-{order}, n={3 * v.shape[0]}, {modes} modes), measured: {dt:.1f} s"),
        "sample_seconds": dt,
        "stage_seconds": stages,
        "extrapolated_to_benchmark_mesh": {
            "value": 1.0 / (dt * full_tets / ntets), "unit": "passes/s",
            "how": f"linear in tets to {full_tets} tets (optimistic for the CPU: assembly and the LU are super-linear)"},
    }


def main_c5(a):
    """BASELINE.json configs[4]: 55^3 Kuhn cells = 998 250 tets, ord-2 (n = 4.1 M, nnz = 0.34 G), 128 modes, fp64
    eigenvalues - "HBM-roofline SpMM stress".  Times every product form of the solve alone on the device against its
    algorithmic bytes and the measured STREAM triad, then the solve itself: fp32 iteration + fp64
    Rayleigh-Ritz polish, and the fp64 refinement to a backward error < 1e-10 of all 128 pairs."""
    from diffsound_amd import _hip, meshgen
    from diffsound_amd.diffelastic.mesh import TetMesh
    from diffsound_amd.lobpcg.modal_solver import ModalSolver, SolverConfig
    from diffsound_amd.modal_ops import HipModalOps, TetSystem
    from diffsound_amd.diffelastic.diff_model import _lame

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    cells, modes, block = (a.cells if a.cells != 26 else 55), (a.modes if a.modes != 64 else 128), (a.block if a.block != 80 else 136)
    v, t = meshgen.kuhn_box(cells)
    t0 = time.time()
    mesh = TetMesh(torch.from_numpy(v).to(dev), torch.from_numpy(t).long().to(dev)).to_high_order(2)
    sysd = TetSystem(mesh.vertices, mesh.tets, 2, MAT[0])
    lam, mu = (float(x) for x in _lame(MAT[1], MAT[2]))
    ops = HipModalOps(sysd, lam, mu, coarse_group_jacobi=a.group_jacobi)
    torch.cuda.synchronize()
    t_setup = time.time() - t0
    n, nv, nnzb = sysd.n, sysd.nv, sysd.nnzb
    L = _hip.lib()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def timed(fn, reps):
        """Seconds per launch in steady state: the launches that follow a synchronisation run 5-10 % longer for ~20 ms
        (profiles/r04_solo_timing_transient.txt), so ~40 ms of untimed launches go first and at least ~60 ms are timed."""
        fn()
        torch.cuda.synchronize()
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        one = max(e0.elapsed_time(e1), 1e-3)
        for _ in range(int(40.0 / one) + 1):
            fn()
        reps = max(reps, int(60.0 / one) + 1)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e-3

    ne = 1 << 28
    ta, tb, tc = (torch.empty(ne, device=dev) for _ in range(3))
    tb.fill_(1.0), tc.fill_(2.0)
    stream_gbs = 3.0 * ne * 4 / timed(lambda: _hip.check(L.ds_stream_triad(ta.data_ptr(), tb.data_ptr(), tc.data_ptr(), ne, 0.5,
                                                                           _hip.stream_ptr()), "ds_stream_triad"), 10) / 1e9
    del ta, tb, tc
    products = []

    def record(name, nbytes, secs, what):
        products.append({"product": name, "what": what, "algorithmic_bytes": int(nbytes), "ms": secs * 1e3,
                         "achieved_gbs": nbytes / secs / 1e9, "frac_of_peak": nbytes / secs / 1e9 / HBM_PEAK_GBS,
                         "frac_of_stream": nbytes / secs / 1e9 / stream_gbs})

    for c in (84, 68):  # the widths the neighbour-union kernels serve; a 136-column block is two launches of 68
        X, Y = torch.randn((n, c), device=dev), torch.empty((n, c), device=dev)
        record(f"K X fp32, {c} columns", nnzb * 40 + (nv + 1) * 4 + 2 * n * c * 4, timed(lambda: ops.apply_K(X, Y), 5),
               "spmm_union_kernel<.,0>: fp32 3x3 blocks, the eigensolver's stiffness product")
        record(f"M X fp32, {c} columns", nnzb * 8 + (nv + 1) * 4 + 2 * n * c * 4, timed(lambda: ops.apply_M(X, Y), 5),
               "spmm_union_kernel<.,3>: node-scalar mass values")
        del X, Y
    c = 136
    X, Y = torch.randn((n, c), device=dev), torch.empty((n, c), device=dev)
    record("K X fp32, 136 columns", nnzb * 40 + (nv + 1) * 4 + 2 * n * c * 4, timed(lambda: ops.apply_K(X, Y), 3),
           "the form the solve runs on its 136-column block")
    del X, Y
    Xb, Wb, Rb = (torch.randn((n, 84), device=dev).bfloat16() for _ in range(3))
    record("fused Chebyshev term bf16, 84 columns", ops.cheb_term_bytes(84, elem_bytes=2),
           timed(lambda: ops.cheb_spmm16(Xb, Wb, Rb, 0.3, 0.7, False), 5),
           "spmm_union_mfma_kernel: the preconditioner's term on the matrix cores")
    del Xb, Wb, Rb
    X64, Y64 = torch.randn((n, 80), device=dev, dtype=torch.float64), torch.empty((n, 80), device=dev, dtype=torch.float64)
    record("K X fp64 values and vectors, 80 columns, term by term", 2 * nnzb * 76 + 2 * (nv + 1) * 4 + 4 * n * 80 * 8,
           timed(lambda: ops.apply_K64(X64, Y64), 2), "two spmm_f64_node launches (K_lambda, K_mu) + combination")
    ops.combined_k64(True)
    try:
        ops.apply_K64(X64, Y64)  # (forms the combined array)
        record("K X fp64 values and vectors, 80 columns", nnzb * 76 + (nv + 1) * 4 + 2 * n * 80 * 8,
               timed(lambda: ops.apply_K64(X64, Y64), 2),
               "spmm_f64_union_kernel<0> on the combined fp64 K array in group order: the refinement's K W (round 5; the wave-per-node kernel before)")
        record("M X fp64 values and vectors, 80 columns", nnzb * 12 + (nv + 1) * 4 + 2 * n * 80 * 8,
               timed(lambda: ops.apply_M64(X64, Y64), 2), "spmm_f64_union_kernel<1>, node-scalar values: the refinement's M W")
    finally:
        ops.combined_k64(False)
    record("M X fp64 values and vectors, 80 columns, wave-per-node kernel", nnzb * 12 + (nv + 1) * 4 + 2 * n * 80 * 8,
           timed(lambda: ops.apply_M64(X64, Y64), 2), "spmm_f64_node, node-scalar values (outside the refinement)")
    del X64, Y64
    torch.cuda.empty_cache()
    # ---- the solve
    torch.cuda.synchronize()
    t0 = time.time()
    nest = dict(nested_tol=a.nested_tol, nested_maxit=a.nested_maxit, nested_cheb_degree=a.coarse_degree,
                nested_cheb_ratio=a.coarse_ratio, nested_ritz_tol=a.nested_ritz_tol,
                start_sweeps=a.start_sweeps, group_degree=a.group_degree, group_ratio=a.group_ratio)  # the nested start of the headline benchmark
    # every solve is run twice and the second is reported: the first one allocates its multi-GB blocks (hipMalloc of
    # 7 GB pieces costs hundreds of ms and varies from run to run); a user's second eigendecomposition pays none of it
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.time()
        res = ModalSolver(ops, SolverConfig(block=block, lmax_cap=10.0, **nest)).solve(modes)
        torch.cuda.synchronize()
        t32 = time.time() - t0
    first64 = None
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.time()
        r64 = ModalSolver(ops, SolverConfig(block=block, lmax_cap=10.0, refine_tol=1e-10, **nest)).solve(modes)
        torch.cuda.synchronize()
        t64 = time.time() - t0
        first64 = t64 if first64 is None else first64
    if not (float(res.rerr.max()) < 2e-6 and float(r64.rerr.max()) < 1e-10):
        raise SystemExit(f"bench.py --workload c5: not converged (fp32 {float(res.rerr.max()):.3g}, fp64 {float(r64.rerr.max()):.3g})")
    free_b, total_b = torch.cuda.mem_get_info(dev)
    print(json.dumps({
        "metric": "seconds per eigensolve, 1M-tet ord-2 mesh, 128 modes, fp64 eigenvalues (BASELINE.json configs[4])",
        "value": t64, "unit": "s", "n_gpus": 1, "higher_is_better": False, "dtype": "f64", "data": "synthetic",
        "config": {"workload": (f"Kuhn box {cells}^3 cells = {sysd.T} tets, ord-2 ({nv} nodes, n={n}, nnz={nnzb * 9}), {modes} modes, "
                                f"block {block}; one eigensolve, no gradient (configs[4] is the SpMM stress, not a pass)"),
                   "setup_seconds": t_setup},
        "solve": {"fp32_iteration_plus_fp64_polish": {"seconds": t32, "iterations": res.iterations,
                                                      "coarse_iterations": res.coarse_iterations,
                                                      "worst_backward_error": float(res.rerr.max())},
                  "with_fp64_refinement": {"seconds": t64, "fp32_iterations": r64.iterations,
                                           "fp64_steps": r64.refine_iterations,
                                           "worst_backward_error": float(r64.rerr.max()), "tolerance": 1e-10,
                                           "seconds_first_call_with_allocations": first64}},
        "stream_triad_gbs": stream_gbs, "hbm_peak_gbs": HBM_PEAK_GBS, "spmm": products,
        "hbm_in_use_gib": (total_b - free_b) / 2 ** 30}))


def main_geom(a):
    """The geometry loop (SURVEY.md 8 rows f2 + f3; reference src/dmtet/geometry/dmtet_thickness.py:237-299 ``getMesh`` + ``tick``,
    experiments/thickness_train.py:104-106: ord-1, 32 modes): per iteration the marching-tets stage hands over NEW vertices and a NEW
    topology; a fresh ``DiffSoundObj`` is built on them (symbolic phase: pattern, contribution lists, union / MFMA tables; first numeric
    assembly), ``eigen_decomposition()`` (assembly + cold eigensolve), ``get_vals()``, the relative MSE against target values,
    ``backward()`` to the vertices and on to the one shape parameter (the thickness), Adam.  Synthetic stand-in of the marching-tets
    output (DMTet itself is out of scope): a 32 x 32 x nz-cell shell whose layer count nz cycles 7, 8, 9, 8 - a
    different tet count and node count every iteration, the size of a resolution-32 grid's output - scaled in z by the thickness.
    Reports iterations/s, the split of an iteration, HBM at iteration 10 and at the end (handles that leak would show), and the
    CPU oracle's forward part on one of the meshes."""
    from diffsound_amd import meshgen
    from src.diffelastic.diff_model import DiffSoundObj, MatSet

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    cells = a.cells if a.cells != 26 else 32
    modes = a.modes if a.modes != 64 else 32
    order = 1 if a.order == 2 and "--order" not in sys.argv else a.order
    layers = (7, 8, 9, 8)
    meshes = []
    for nz in layers:
        v, t = meshgen.kuhn_box(cells, cells, nz, box=(0.10, 0.10, 0.10 * nz / cells))
        meshes.append((torch.from_numpy(v).to(dev), torch.from_numpy(t).long().to(dev)))
    mat = MatSet.Ceramic
    geom_cfg = None  # (None: DiffSoundObj's own default, lobpcg.modal_solver.tuned_config(order))
    if a.one_level_group >= 0:  # (a process-wide default of the library, set here for the A/B: DiffSoundObj builds the operator objects)
        from diffsound_amd.modal_ops import HipModalOps

        HipModalOps.one_level_group_jacobi = a.one_level_group
    if any(f in sys.argv for f in ("--cheb-degree", "--cheb-ratio", "--block", "--start-sweeps", "--nested-ritz-tol", "--cheb-group-degree", "--cheb-group-ratio")):
        from diffsound_amd.lobpcg.modal_solver import tuned_config

        geom_cfg = tuned_config(order)
        if "--cheb-degree" in sys.argv or "--cheb-ratio" in sys.argv:
            geom_cfg.cheb_degree, geom_cfg.cheb_ratio = a.cheb_degree, a.cheb_ratio
        if "--block" in sys.argv:
            geom_cfg.block = a.block
        if "--start-sweeps" in sys.argv:
            geom_cfg.start_sweeps = a.start_sweeps
        if a.cheb_group_degree > 0:
            geom_cfg.cheb_group_degree = a.cheb_group_degree
        if a.cheb_group_ratio > 0:
            geom_cfg.cheb_group_ratio = a.cheb_group_ratio
    theta = torch.nn.Parameter(torch.tensor(1.0, device=dev))
    opt = torch.optim.Adam([theta], lr=2e-3)
    zscale = lambda: torch.stack([torch.ones((), device=dev), torch.ones((), device=dev), theta])

    def iteration(i, target, sync=None):
        v0, t0 = meshes[i % len(meshes)]
        verts = v0 * zscale()[None, :]
        if sync:
            sync("vertices")
        obj = DiffSoundObj(verts, t0, mode_num=modes, order=order, mat=mat, solver_config=geom_cfg)
        _ = obj.system  # (the symbolic phase + first numeric assembly: built on first use)
        if sync:
            sync("symbolic + tables")
        obj.eigen_decomposition()
        if sync:
            sync("assembly + eigensolve")
        vals = obj.get_vals()
        loss = ((vals - target) ** 2 / target ** 2).mean()
        opt.zero_grad()
        loss.backward()
        opt.step()
        if sync:
            sync("get_vals + loss + backward + Adam")
        return float(loss.detach()), obj.last_result.iterations, obj.system.T, obj.system.nv

    # target values: the shell at thickness 1.15 (outside the timed region)
    with torch.no_grad():
        v0, t0 = meshes[1]
        gt = DiffSoundObj(v0 * torch.tensor([1.0, 1.0, 1.15], device=dev), t0, mode_num=modes, order=order, mat=mat)
        gt.eigen_decomposition()
        target = gt.get_vals().detach().clone()
        del gt
    for i in range(6):  # untimed: first calls allocate, kernels load
        iteration(i, target)
    torch.cuda.synchronize()
    # the split of an iteration, with a device synchronisation after every stage (its own short loop: the timed loop has none)
    stages = {}
    for i in range(8):
        clock = [time.time()]

        def sync(name):
            torch.cuda.synchronize()
            now = time.time()
            stages[name] = stages.get(name, 0.0) + (now - clock[0]) / 8
            clock[0] = now

        torch.cuda.synchronize()
        clock[0] = time.time()
        iteration(i, target, sync)
    import gc

    gc.collect()
    torch.cuda.synchronize()
    free10 = alloc10 = None
    losses, its, sizes = [], [], []
    t0c = time.time()
    for i in range(a.geom_iters):
        if i == 10:
            gc.collect()
            torch.cuda.synchronize()
            alloc10, free10 = torch.cuda.memory_allocated(dev), torch.cuda.mem_get_info(dev)[0]
            t10 = time.time()
        lo, it_, T_, nv_ = iteration(i, target)
        losses.append(lo), its.append(it_), sizes.append((T_, nv_))
    torch.cuda.synchronize()
    dt = time.time() - t0c
    gc.collect()
    torch.cuda.synchronize()
    alloc_end, free_end = torch.cuda.memory_allocated(dev), torch.cuda.mem_get_info(dev)[0]
    if not np.isfinite(losses).all():
        raise SystemExit("bench.py --workload geom: non-finite loss")
    out = {"metric": "shape-loop iterations/s: fresh DiffSoundObj on new vertices + new topology, eigen_decomposition, get_vals, backward to vertices",
           "value": a.geom_iters / dt, "unit": "iterations/s", "n_gpus": 1, "higher_is_better": True, "dtype": "f32", "data": "synthetic",
           "config": {"workload": (f"{cells} x {cells} x nz-cell Kuhn shell, nz cycling {layers} ({min(s[0] for s in sizes)}-{max(s[0] for s in sizes)} tets, "
                                   f"{min(s[1] for s in sizes)}-{max(s[1] for s in sizes)} nodes), ord-{order}, {modes} modes, a new topology every iteration"),
                      "reference_loop": "src/dmtet/geometry/dmtet_thickness.py:237-299 (getMesh + tick), experiments/thickness_train.py:104-106"},
           "iterations": a.geom_iters, "ms_per_iteration": 1e3 * dt / a.geom_iters, "mean_solver_iterations": float(np.mean(its)),
           "split_ms_with_a_sync_per_stage": {k_: 1e3 * v_ for k_, v_ in stages.items()},
           "loss_first_last": [losses[0], losses[-1]], "thickness_after": float(theta.detach()),
           "hbm": {"torch_allocated_mib_at_iteration_10": alloc10 / 2 ** 20, "torch_allocated_mib_at_end": alloc_end / 2 ** 20,
                   "device_free_mib_at_iteration_10": free10 / 2 ** 20, "device_free_mib_at_end": free_end / 2 ** 20,
                   "growth_mib_torch": (alloc_end - alloc10) / 2 ** 20, "growth_mib_device": (free10 - free_end) / 2 ** 20,
                   "what": f"{a.geom_iters - 10} fresh objects between the two readings: pattern handles, tables or blocks that leaked would show here"}}
    if not a.no_cpu_baseline:
        env = dict(os.environ)
        nthreads = min(os.cpu_count() or 1, 16)
        env["OMP_NUM_THREADS"] = env["OPENBLAS_NUM_THREADS"] = env["MKL_NUM_THREADS"] = str(nthreads)
        env["HIP_VISIBLE_DEVICES"] = env["CUDA_VISIBLE_DEVICES"] = ""
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--geom-cpu-child", "--cells", str(cells), "--modes", str(modes),
                            "--order", str(order)], capture_output=True, text=True, env=env, cwd=ROOT)
        if r.returncode != 0:
            raise SystemExit("bench.py: geometry CPU baseline child failed:\n" + r.stderr[-2000:])
        out["cpu_baseline"] = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    print(json.dumps(out))


def geom_cpu_child(a):
    """The CPU oracle on ONE iteration of the geometry loop at the same size (fresh process, no GPU): shape-function derivatives,
    faithful K assembly, M assembly, ARPACK shift-invert, get_vals.  FORWARD ONLY: the reference differentiates the whole assembly by
    autograd (diff_model.py:390-399), which the oracle does not restate - the figure is optimistic for the CPU and says so."""
    from diffsound_amd import meshgen
    from oracle import fem, modal

    nthreads = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(nthreads)
    cells, nz = a.cells, 8
    v, t = meshgen.kuhn_box(cells, cells, nz, box=(0.10, 0.10, 0.10 * nz / cells))
    v, t = torch.from_numpy(v), torch.from_numpy(t).long()
    mat = (2700.0, 7.2e10, 0.19)
    stages, clock = {}, [time.time()]

    def lap(name):
        now = time.time()
        stages[name] = round(now - clock[0], 3)
        clock[0] = now

    t0 = clock[0]
    if a.order > 1:
        v, t = fem.to_high_order(v, t, a.order)
    d = fem.OracleDeform(v, t, a.order)
    lap("shape_function_derivatives")
    lam, mu = fem.lame(mat[1], mat[2])
    M3, _ = fem.assemble_mass(v, t, a.order, mat[0])
    lap("mass_assembly")
    K = fem.assemble_stiffness_faithful(d, lam, mu)
    lap("stiffness_assembly")
    ev, U, _, _ = modal.eigsh_shift_invert(K, M3, a.modes)
    lap("arpack_shift_invert")
    _ = modal.get_vals(K, M3, ev, U)
    lap("get_vals")
    dt = time.time() - t0
    print(json.dumps({"value": 1.0 / dt, "unit": f"iterations/s at {t.shape[0]} tets, FORWARD ONLY", "cores": nthreads, "kind": "port",
                      "sample": (f"ONE forward iteration of the CPU oracle on the {cells} x {cells} x {nz}-cell shell ({t.shape[0]} tets, ord-{a.order}, "
                                 f"{a.modes} modes): {dt:.1f} s; no backward to the vertices (the oracle does not restate autograd through "
                                 "the assembly) - optimistic for the CPU"),
                      "sample_seconds": dt, "stage_seconds": stages}))


def in_pass_profile(pipe, hyps, dev, block, passes=2):
    """``roofline.in_pass``: complete passes run ONE AT A TIME on one stream with every SpMM form, Gram and update launch of
    the pass bracketed by HIP events inside the library (ds_profile_stream / ds_profile_kinds) - each kernel alone on the device,
    but BETWEEN the other kernels of a pass, which is where the caches are in the state a real pass leaves them in (a back-to-back
    series of one kernel on the same operands keeps part of them in the 256 MB Infinity Cache: that is ``avg_launch_ms``).
    Returns the records as arrays (kind, ms, a, b, c, flags)."""
    import ctypes

    from diffsound_amd import _hip

    L = _hip.lib()
    if pipe._lanes:
        lane, stream = pipe._lanes[0], pipe._lanes[0].stream
    else:
        lane, stream = None, torch.cuda.Stream(device=dev)
    cap = 6000 * passes
    torch.cuda.synchronize()
    from diffsound_amd.pipeline import set_wait_mode

    holder = lane if lane is not None else pipe
    if holder.ops is not None:
        set_wait_mode(holder.ops, 0)  # (one at a time: the solve's waits spin, as in the one-hypothesis leg; run_batch sets the lanes' mode again)
    with torch.cuda.stream(stream):
        pipe.run_pass(*hyps[0], _lane=lane)  # (untimed: the stream's first pass after the lanes' concurrent run)
        stream.synchronize()
        _hip.check(L.ds_profile_kinds(0x7F), "ds_profile_kinds")
        _hip.check(L.ds_profile_stream(stream.cuda_stream, cap), "ds_profile_stream")
        t0 = time.time()
        for i in range(passes):
            pipe.run_pass(*hyps[(i + 1) % len(hyps)], _lane=lane)
        stream.synchronize()
        secs = (time.time() - t0) / passes
    ms = (ctypes.c_float * cap)()
    a_, b_ = (ctypes.c_int64 * cap)(), (ctypes.c_int64 * cap)()
    c_, f_ = (ctypes.c_int32 * cap)(), (ctypes.c_int32 * cap)()
    n = int(L.ds_profile_collect(ms, a_, b_, c_, f_, cap))
    _hip.check(L.ds_profile_kinds(0), "ds_profile_kinds")
    fl = np.array(f_[:n], dtype=np.int64)
    return {"kind": fl >> 16, "ms": np.array(ms[:n], dtype=np.float64), "a": np.array(a_[:n], dtype=np.float64),
            "b": np.array(b_[:n], dtype=np.float64), "c": np.array(c_[:n], dtype=np.int64), "flags": fl & 0xFFFF,
            "passes": passes, "seconds_per_pass": secs}


def summarize_in_pass(rec, sysd, block, stream_gbs, mf_levels):
    """Per kernel family of a pass: launches at the full block width on the fine level, average duration, algorithmic bytes
     or flops, and the fractions of the HBM roofline / the in-run STREAM triad / the fp32 matrix-core peak."""
    nv, nnzb, n = sysd.nv, sysd.nnzb, sysd.n
    out = {}
    kind, ms, a, b, c, fl = (rec[k] for k in ("kind", "ms", "a", "b", "c", "flags"))

    def hbm(name, sel, nbytes, what):
        if sel.any():
            t = float(ms[sel].mean())
            gbs = nbytes / (t * 1e-3) / 1e9
            out[name] = {"what": what, "launches": int(sel.sum()), "avg_launch_ms": t, "min_launch_ms": float(ms[sel].min()),
                         "algorithmic_bytes_per_launch": float(nbytes), "achieved": gbs, "unit": "GB/s",
                         "frac": gbs / HBM_PEAK_GBS, "frac_of_stream": gbs / stream_gbs if stream_gbs else None}

    fine = a == nv
    full = c == block
    eb = (fl >> 8).astype(np.float64)
    sel = (kind == 0) & fine & full & ((fl & 1) == 0) & (eb == 2)
    vb = 2.0 if nv in mf_levels else 4.0
    hbm("fused_term_bf16", sel, nnzb * (9 * vb + 4) + (nv + 1) * 4 + nv * 36 + 4 * n * block * 2,
        "the dominant kernel (fused bf16 Chebyshev term, fine level), 4 vector streams")
    hbm("lobpcg_kw", (kind == 1) & fine & full, nnzb * 40 + (nv + 1) * 4 + 2 * n * block * 4, "Y = K W of the iteration (fp32)")
    hbm("lobpcg_mw", (kind == 2) & fine & full, nnzb * 8 + (nv + 1) * 4 + 2 * n * block * 4, "Y = M_s W of the orthonormalisation (fp32)")
    hbm("fused_residual", (kind == 3) & fine & full, nnzb * 44 + (nv + 1) * 4 + 2 * n * block * 4,
        "R = K X - (M X) diag(lam) in one walk: K's and M's values, X gathered, R written")
    hbm("lobpcg_kw_mw", (kind == 6) & fine & full, nnzb * 44 + (nv + 1) * 4 + 3 * n * block * 4,
        "[K W | M W] of the raw preconditioned residuals in one walk (round 5): K's and M's values, W gathered, two blocks written")
    rr = {}
    for kd, nm in ((4, "gram"), (5, "mix")):
        m = (kind == kd) & (b == n)
        for p_, q_ in sorted({(int(x), int(y)) for x, y in zip(a[m], c[m])}, key=lambda t: -t[0] * t[1])[:4]:
            sel = m & (a == p_) & (c == q_)
            if sel.sum() < 2:
                continue
            t = float(ms[sel].mean())
            flops, nbytes = 2.0 * n * p_ * q_, n * (p_ + q_) * 4.0
            rr[f"{nm}_{p_}x{q_}"] = {"launches": int(sel.sum()), "avg_ms": t, "min_ms": float(ms[sel].min()),
                                      "tflops": flops / (t * 1e-3) / 1e12, "frac_of_mfma_f32_peak": flops / (t * 1e-3) / 1e12 / MFMA_F32_PEAK_TFS,
                                      "gbs": nbytes / (t * 1e-3) / 1e9, "frac_of_hbm_peak": nbytes / (t * 1e-3) / 1e9 / HBM_PEAK_GBS}
    by_kind = {PROF_KINDS[k]: {"launches": int((kind == k).sum()), "total_ms_per_pass": float(ms[kind == k].sum() / rec["passes"])}
               for k in sorted(PROF_KINDS) if (kind == k).any()}
    return out, rr, by_kind


def pipeline_ops(pipe):
    """Every operator object of a pipeline (one per hypothesis lane, plus their corner-node levels)."""
    out = [ln.ops for ln in pipe._lanes if ln.ops is not None] or ([pipe.ops] if pipe.ops is not None else [])
    return out + [o.coarse for o in out if getattr(o, "coarse", None) is not None]


def summed_stats(pipe, name):
    tot = [0, 0]
    for o in pipeline_ops(pipe):
        st = getattr(o, name, None) or [0, 0]
        tot[0] += st[0]
        tot[1] += st[1]
    return tot


def one_hypothesis_leg(pipe, hyps, dev, passes=8, steps_done=0):
    """ONE hypothesis at a time: complete cold passes back to back on one stream and one host thread - the
    latency a user sees who runs a single material (the headline keeps 8 in flight).  The native solve's waits spin here (one
    thread has a core to itself).  Every pass a material this lane has not seen."""
    from diffsound_amd.pipeline import set_wait_mode

    lane = pipe._lanes[0] if pipe._lanes else None
    holder = lane if lane is not None else pipe
    stream = lane.stream if lane is not None else torch.cuda.current_stream(dev)
    saved = getattr(holder.ops, "host_wait_mode", 0)
    set_wait_mode(holder.ops, 0)
    keep_saved, pipe.keep_blocks = getattr(pipe, "keep_blocks", True), False  # (cold passes: nobody takes the rotated block, as in the timed region)
    its, cits, each = [], [], []
    try:
        with torch.cuda.stream(stream):
            E, nu = hyps[0]
            for w in range(3):  # (untimed: the stream's first passes after the lanes' concurrent run)
                pipe.run_pass(E * (1.37 + 0.01 * w), nu * 0.93, _lane=lane)
            stream.synchronize()
            t0 = time.time()
            for i in range(passes):
                # ONE hypothesis, as a user who fits one material runs it: its material moves from pass to pass as an optimiser's
                # would (the same move as in the timed region), so every pass is a material this lane has not seen
                s = steps_done + i + 7
                t1 = time.time()
                r, _, _ = pipe.run_pass(E * (1 + 1e-3 * s), nu * (1 - 5e-4 * s), _lane=lane)  # (returns after the pass's last wait)
                each.append(1e3 * (time.time() - t1))
                its.append(r.iterations)
                cits.append(r.coarse_iterations)
            stream.synchronize()
            dt = (time.time() - t0) / passes
    finally:
        set_wait_mode(holder.ops, saved)
        pipe.keep_blocks = keep_saved
    return {"passes_per_s": 1.0 / dt, "ms_per_pass": 1e3 * dt, "ms_per_pass_median": float(np.median(each)), "ms_per_pass_each": [round(x, 2) for x in each],
            "passes_timed": passes, "mean_fine_iterations": float(np.mean(its)),
            "mean_corner_level_iterations": float(np.mean(cits)),
            "what": "complete cold-start passes (assembly + eigensolve + read-out + render + loss + backward) of ONE hypothesis one after the "
                    "other on ONE stream and ONE host thread, its material moved every pass as in the timed region (E (1 + 1e-3 s), "
                    "nu (1 - 5e-4 s)); the host thread's waits spin"}


def kernel_stats_leg(pipe, hyps, dev, passes=2):
    """Kernel time and launch count of one pass, one hypothesis at a time, from torch's own kernel tracer (kineto over roctracer):
    every device kernel of the passes, the library's and torch's.  None when the tracer is not available (e.g. the process already
    runs under rocprofv3) - the rocprofv3 one-lane table under profiles/ is the reference figure."""
    if any(k.startswith("ROCPROF") or k.startswith("ROCP_") for k in os.environ):
        return None
    try:
        from torch.profiler import ProfilerActivity, profile

        lane = pipe._lanes[0] if pipe._lanes else None
        stream = lane.stream if lane is not None else torch.cuda.current_stream(dev)
        with torch.cuda.stream(stream):
            pipe.run_pass(*hyps[0], _lane=lane)
            stream.synchronize()
            with profile(activities=[ProfilerActivity.CUDA]) as prof:
                for i in range(passes):
                    pipe.run_pass(*hyps[(i + 1) % len(hyps)], _lane=lane)
                stream.synchronize()
        ev = [e for e in prof.events() if getattr(e, "device_type", None) is not None and "cuda" in str(e.device_type).lower()]
        kern = [e for e in ev if not any(w in e.name.lower() for w in ("memcpy", "memset", "copybuffer"))] if ev else []
        if not ev:
            return None
        tot_us = sum(float(getattr(e, "device_time", 0.0) or getattr(e, "cuda_time", 0.0) or 0.0) for e in ev)
        own = sum(1 for e in kern if ("ds_" in e.name or "kernel" in e.name) and "at::" not in e.name)
        return {"kernel_ms_per_pass": tot_us / 1e3 / passes, "launches_per_pass": len(ev) / passes,
                "copies_per_pass": (len(ev) - len(kern)) / passes, "passes": passes,
                "source": "torch.profiler (kineto / roctracer), device activities of one-at-a-time passes"}
    except Exception as ex:  # the tracer is optional evidence, never a reason to lose the line
        return {"error": f"{type(ex).__name__}: {ex}"[:300]}


def api_path_leg(a, verts, tets, dev, cfg):
    """The path a user of the reference runs, literally: the loop body of experiments/material_sync_train.py:137-168 through the
    reference's own names - ``src.diffelastic.diff_model.build_model`` / ``DiffSoundObj.eigen_decomposition`` /
    ``get_undamped_freqs``, ``src.ddsp.oscillator.TraditionalDampedOscillator``, ``src.ddsp.mss_loss.MSSLoss([1024 ... 64],
    'l1_loss')``, Adam + StepLR - on the benchmark mesh, with EIGEN_DECOMPOSE_CYCLE = 1 and = 15 (the script's value).  The early
    epochs' Sinkhorn loss (geomloss, a third-party package the image lacks) is not part of it: the late loss from epoch 0.
    The model runs the benchmark's eigensolver settings (``cfg``)."""
    from torch.optim import Adam, lr_scheduler

    from src.ddsp.mss_loss import MSSLoss
    from src.ddsp.oscillator import TraditionalDampedOscillator
    from src.diffelastic.diff_model import Material, build_model

    gt_coeff = [MAT[0], 6.2e10, 0.27, MAT[3], MAT[4]]
    init_coeff = [MAT[0], 4.6e10, 0.31, MAT[3], MAT[4]]
    forces = torch.zeros((1, 150), device=dev)
    forces[0, 0] = 1
    t_build = time.time()
    gt_osc = TraditionalDampedOscillator(forces, 1, a.modes, 8000, 32000, Material(gt_coeff)).cuda()
    gt = build_model(None, mode_num=a.modes, order=a.order, mat=gt_coeff, task="gt", vertices=verts, tets=tets)
    gt.solver_config = cfg
    gt.eigen_decomposition()
    with torch.no_grad():
        gt_audios = gt_osc(gt.get_undamped_freqs().float())
    del gt
    torch.manual_seed(0)
    model = build_model(None, mode_num=a.modes, order=a.order, mat=init_coeff, task="material", vertices=verts, tets=tets)
    model.solver_config = cfg
    osc = TraditionalDampedOscillator(forces, len(gt_audios), a.modes, 8000, 32000, Material(init_coeff)).cuda()
    late = MSSLoss([1024, 512, 256, 128, 64], 32000, type="l1_loss").cuda()
    torch.cuda.synchronize()
    t_build = time.time() - t_build
    out = {"what": ("experiments/material_sync_train.py:137-168, literally: every CYCLE-th epoch model.eigen_decomposition() (numeric "
                    "assembly + eigensolve; DiffSoundObj starts it from the previous block, the reference's ARPACK call starts cold), "
                    "every epoch get_undamped_freqs -> TraditionalDampedOscillator -> MSSLoss([1024..64], 'l1_loss')(pred, gt, "
                    "damped_freq, 1) -> zero_grad / backward / Adam.step / StepLR.step; task='material' (E and nu trainable)"),
           "build_seconds_not_timed": t_build}
    for cycle, epochs, cold in ((1, 8, False), (1, 8, True), (15, max(15, a.api_epochs), False)):
        opt = Adam(model.parameters(), lr=5e-3)
        sched = lr_scheduler.StepLR(opt, step_size=100, gamma=0.9)
        losses, t_eig, n_eig, its = [], 0.0, 0, []
        for epoch in range(-2, epochs):  # (epochs -2, -1: untimed - first calls allocate)
            if epoch == 0:
                torch.cuda.synchronize()
                t0 = time.time()
                t_eig, n_eig, its = 0.0, 0, []
            if epoch % cycle == 0 or epoch < 0:
                torch.cuda.synchronize()
                te = time.time()
                if cold:
                    model._warm = None  # (the reference's ARPACK call starts from nothing: the same here)
                model.eigen_decomposition()
                torch.cuda.synchronize()
                t_eig += time.time() - te
                n_eig += 1
                its.append(model.last_result.iterations)
            f = model.get_undamped_freqs().float()
            pred = osc(f)
            loss = late(pred, gt_audios, osc.damped_freq, 1)
            opt.zero_grad()
            loss.backward()
            opt.step()
            sched.step()
            losses.append(float(loss.detach()))
        torch.cuda.synchronize()
        dt = time.time() - t0
        if not np.isfinite(losses).all():
            raise SystemExit("bench.py: non-finite loss in the API loop")
        out[f"cycle_{cycle}" + ("_cold_start" if cold else "")] = {
                                 "eigen_decompose_cycle": cycle, "epochs": epochs, "ms_per_epoch": 1e3 * dt / epochs,
                                 "start_of_each_eigensolve": ("cold: random block + nested start, as a pass of the headline" if cold else
                                                              "the previous eigensolve's block (DiffSoundObj's default)"),
                                 "epochs_per_s": epochs / dt, "eigen_decompositions": n_eig,
                                 "ms_per_eigen_decomposition": 1e3 * t_eig / max(1, n_eig), "mean_iterations": float(np.mean(its)),
                                 "ms_per_epoch_outside_eigen_decomposition": 1e3 * (dt - t_eig) / epochs,
                                 "loss_first_last": [losses[2], losses[-1]]}
    out["youngs_poisson_after"] = [float(model.material_model.youngs()), float(model.material_model.poisson())]
    return out


def cap_torch_threads_at_the_cpu_quota():
    """torch sizes its CPU thread team by the CPUs it may run on (128 on a GPU box once the rank is bound to its NUMA node); the box's
    cgroup gives 16 CPUs' worth of time per 100 ms, and a team larger than that is paused the moment it all runs (cpu.stat:
    nr_throttled).  The product's hot path is one BLAS thread per lane either way; this keeps torch's own CPU ops inside the quota."""
    q = str(_CGROUP_AT_START.get("quota", "")).split()
    try:
        cpus = int(q[0]) // int(q[1]) if len(q) == 2 and q[0] != "max" else 0
    except (ValueError, ZeroDivisionError):
        cpus = 0
    if cpus > 0:
        import torch

        if torch.get_num_threads() > cpus:
            torch.set_num_threads(max(1, cpus))


def main():
    a = parse()
    if not a.cpu_baseline_child and not a.geom_cpu_child:  # (the CPU oracle's children size their own pools)
        cap_torch_threads_at_the_cpu_quota()
    if a.geom_cpu_child:
        return geom_cpu_child(a)
    if a.workload == "c5":
        return main_c5(a)
    if a.workload == "geom":
        return main_geom(a)
    if a.cpu_baseline_child:  # one pass of the CPU oracle, nothing else (no GPU, no torch.distributed)
        print(json.dumps(cpu_baseline_pass(a.cpu_sample_cells, a.order, a.modes, 6 * a.cells ** 3)))
        return
    relaunch_as_ranks(a)  # N > 1: becomes N ranks under torch.distributed.run (never a silent single rank)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == a.gpus
    # CPU placement of this rank, from sysfs alone and BEFORE the process touches the GPU (threads the HIP runtime and RCCL start
    # later inherit the mask): the CPUs of the NUMA node its device hangs off, its share of them when ranks share the node
    from diffsound_amd import hostcpu

    local_world_ = int(os.environ.get("LOCAL_WORLD_SIZE", world))
    affinity = ({"bound": False, "why": "--no-affinity"} if a.no_affinity else
                hostcpu.bind_rank_to_device_numa(local_rank % max(1, torch.cuda.device_count()) if a.share_devices else local_rank,
                                                 1 if a.share_devices else local_world_, min_cpus=max(4, min(a.lanes, a.hyp_per_gpu) + 2)))
    # (before anything creates the device's context: with several lanes every wait of a host thread for the device sleeps)
    sleeping_waits = False
    if a.host_wait == "sleep" and a.lanes > 1:
        from diffsound_amd.pipeline import prefer_sleeping_waits

        ndev_ = torch.cuda.device_count()  # (does not initialise the device)
        sleeping_waits = prefer_sleeping_waits(local_rank % max(1, ndev_) if a.share_devices else local_rank)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU path)")
    if a.share_devices:
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(a.dist_backend, rank=rank, world_size=world)
    # host-thread budget of the node: every rank runs `lanes` hypothesis lanes, each ONE host thread that issues launches and solves
    # the <= 3b x 3b dense problems on ONE LAPACK thread (the lanes pin themselves, pipeline._lane_pool) - ranks x lanes threads
    # in all; they should fit half the node's hardware threads (the other half: the runtime's own threads, RCCL's proxies)
    hw_threads = os.cpu_count() or 1
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", world))
    if a.lanes * local_world > hw_threads:  # (outright oversubscription; between a half and all of them the line says so)
        raise SystemExit(f"bench.py: {local_world} ranks x {a.lanes} lanes = {a.lanes * local_world} host threads exceed the "
                         f"node's {hw_threads} hardware threads; lower --lanes")

    def barrier():
        if world > 1:
            if a.dist_backend == "nccl":
                dist.barrier(device_ids=[local_rank])  # RCCL: on this rank's own device, not a guessed one
            else:
                dist.barrier()

    from diffsound_amd import meshgen
    from diffsound_amd.diffelastic.mesh import TetMesh
    from diffsound_amd.lobpcg.modal_solver import SolverConfig
    from diffsound_amd.pipeline import ModalPipeline, all_reduce_loss, gather_rank_stats, shard_hypotheses

    # ---- synthetic inputs, resident in HBM before timing -------------------------------------
    v, t = meshgen.kuhn_box(a.cells)
    mesh = TetMesh(torch.from_numpy(v).to(dev), torch.from_numpy(t).long().to(dev)).to_high_order(a.order)
    cfg = solver_config(a)
    t_sym = time.time()
    loss_fn = None
    if a.loss == "mss":
        from diffsound_amd.ddsp.mss_loss import MSSLoss

        loss_fn = MSSLoss([1024, 512, 256, 128, 64], 32000, type="l1_loss")
    pipe = ModalPipeline(mesh.vertices, mesh.tets, a.order, a.modes, MAT, solver_config=cfg, loss_fn=loss_fn,
                         mfma_groups=tuple(int(x) for x in a.mfma_groups.split(",")), host_wait=a.host_wait,
                         coarse_group_jacobi=a.group_jacobi)
    torch.cuda.synchronize()
    t_sym = time.time() - t_sym
    nhyp = a.hyp_per_gpu * world
    rng = np.random.default_rng(2024)  # SURVEY.md 8(d) C4 hypothesis ranges
    Es = rng.uniform(1e10, 1e11, size=max(nhyp, 64))
    nus = rng.uniform(0.1, 0.4, size=max(nhyp, 64))
    mine = shard_hypotheses(nhyp, rank, world)
    # fixed target audio: the material-table hypothesis rendered once (outside the timed region)
    pipe.assemble()
    tgt, res0, audio0 = pipe.run_pass(MAT[1], MAT[2], backward=False)
    pipe.set_target(audio0)

    tol = cfg.tol or 2e-6
    worst = [0.0]
    cits = []

    hyps = [(float(Es[h]), float(nus[h])) for h in mine]
    its_all = []

    def on_step(s, outs):
        """Called once per step, as soon as every lane has finished it (the lanes run on): the convergence gate of each
        pass and the step's ONE collective, the all-reduce of the scalar loss."""
        loss_sum = 0.0
        for h, (r, res, _) in zip(mine, outs):
            # convergence gate: an unconverged pass is not a pass
            if not (r.iterations < cfg.maxit and r.max_rerr < tol and np.isfinite(r.loss)
                    and np.isfinite(r.grad_E) and np.isfinite(r.grad_nu)):
                raise SystemExit(f"bench.py: hypothesis {h} (E={Es[h]:.4g}, nu={nus[h]:.4g}) did not converge: "
                                 f"{r.iterations} iterations, backward error {r.max_rerr:.3g} (tol {tol:.1g}), "
                                 f"loss {r.loss}, grad ({r.grad_E}, {r.grad_nu})")
            worst[0] = max(worst[0], r.max_rerr)
            loss_sum += r.loss
            its_all.append(r.iterations)
            cits.append(r.coarse_iterations)
        on_step.total = all_reduce_loss(loss_sum, dev)
        on_step.stamps.append(time.time())

    on_step.stamps = []

    steps_done = [0]  # steps run so far (warm-up included): the material of a step depends on its global number

    def moved(s, E, nu):
        """(E, nu) of a hypothesis at global step s: an optimiser's small move per step (the amortised leg below moves the material
        the same way) - every pass of the run assembles, solves and differentiates a material nobody has seen before, so no warm
        estimate (power block, norm probe) finds its own previous operator again.  --same-material: the repetition of rounds 1-5."""
        return (E, nu) if a.same_material else (E * (1 + 1e-3 * (s + 1)), nu * (1 - 5e-4 * (s + 1)))

    def run_steps(nsteps, warm):
        """``nsteps`` steps (one pass of every hypothesis of this rank per step; every pass runs its own numeric assembly).
        Default: the hypothesis lanes run their passes of consecutive steps back to back (ModalPipeline.run_steps - a
        hypothesis' next step depends on its own previous one only); --step-barrier: all lanes join after every step."""
        if nsteps <= 0:
            return warm
        s0 = steps_done[0]
        steps_done[0] += nsteps
        if a.step_barrier:
            for s in range(nsteps):
                outs = pipe.run_batch([moved(s0 + s, E, nu) for E, nu in hyps], lanes=a.lanes, warm=warm if a.warm_start else None)
                if a.warm_start:
                    warm = [res.block_vectors for _, res, _ in outs]
                on_step(s, outs)
            return warm
        outs = pipe.run_steps(hyps, nsteps, lanes=a.lanes, on_step=on_step, warm_start=a.warm_start,
                              warm_init=warm if a.warm_start else None, material_at=lambda s, i, E, nu: moved(s0 + s, E, nu))
        return [res.block_vectors for _, res, _ in outs[-1]] if a.warm_start else None

    _cg_mark("setup (mesh, tables, pipeline)")
    warm = run_steps(max(a.warmup, 0), None)
    _cg_mark("warm-up steps")
    if a.warmup <= 0 and a.lanes > 1 and len(mine) > 1:
        # --warmup 0: the lanes' streams, operators and value arrays are set-up, not part of a step
        pipe.run_batch([(MAT[1], MAT[2])] * min(a.lanes, len(mine)), lanes=a.lanes, backward=False)

    # ---- instrument the dominant kernel: every launch of the fused Chebyshev-term SpMM (fine and corner-node level
    #      alike: they are ONE kernel) on the FIRST hypothesis lane's stream is bracketed by HIP events inside the
    #      library (ds_profile_stream) - the launches come from the native drivers, not from Python
    from diffsound_amd import _hip
    lane_ops = [ln.ops for ln in pipe._lanes if ln.ops is not None] or [pipe.ops]
    prof_stream = pipe._lanes[0].stream.cuda_stream if pipe._lanes else torch.cuda.current_stream(dev).cuda_stream
    PROF_CAP = 20000
    if rank == 0:
        _hip.check(_hip.lib().ds_profile_stream(prof_stream, PROF_CAP), "ds_profile_stream")
    barrier()
    torch.cuda.synchronize()
    cg0 = _cgroup_cpu_stat()
    t0 = time.time()
    del its_all[:]
    run_steps(a.steps, warm)
    iters, total = list(its_all), on_step.total
    torch.cuda.synchronize()
    own_dt = time.time() - t0  # this rank's own work, before it waits for the slowest one
    cg1 = _cgroup_cpu_stat()
    global _CGROUP_TIMED
    _CGROUP_TIMED = {k: cg1[k] - cg0.get(k, 0) for k in cg1 if isinstance(cg1[k], int)}  # (CPU time used / periods throttled inside the timed region)
    barrier()
    dt = time.time() - t0
    cdev = dev if (world > 1 and a.dist_backend == "nccl") else torch.device("cpu")
    tmax = torch.tensor([dt], dtype=torch.float64, device=cdev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax[0])
    # per-rank view of the step: the only expected scaling loss is the imbalance of iteration counts between the
    # ranks' hypotheses; one small all-gather AFTER the timed region makes it visible
    free_b, total_b = torch.cuda.mem_get_info(dev)
    gathered = gather_rank_stats([len(mine), np.sum(iters), np.max(iters), own_dt,
                                  torch.cuda.max_memory_allocated(dev) / 2 ** 30, (total_b - free_b) / 2 ** 30, local_rank], dev)
    aff_all = [affinity]
    if world > 1:
        aff_all = [None] * world
        dist.all_gather_object(aff_all, affinity)
    per_rank = [{"rank": r, "device": int(g[6]), "cpu_affinity": aff_all[r], "hypotheses_per_step": int(g[0]), "fine_iterations": int(g[1]),
                 "max_iterations_of_a_pass": int(g[2]), "busy_seconds": round(float(g[3]), 4),
                 "idle_fraction": round(max(0.0, 1 - float(g[3]) / dt), 4),
                 "hbm_peak_allocated_gib": round(float(g[4]), 3), "hbm_in_use_on_device_gib": round(float(g[5]), 3)}
                for r, g in enumerate(gathered)]

    import ctypes
    pms = (ctypes.c_float * PROF_CAP)()
    pnv, pnz = (ctypes.c_int64 * PROF_CAP)(), (ctypes.c_int64 * PROF_CAP)()
    pnc, pfi = (ctypes.c_int32 * PROF_CAP)(), (ctypes.c_int32 * PROF_CAP)()
    nrec = int(_hip.lib().ds_profile_collect(pms, pnv, pnz, pnc, pfi, PROF_CAP)) if rank == 0 else 0
    # ---- the amortised variant: one full pass, then cycle - 1 passes that only read out, render and
    #      differentiate on the kept eigenvectors while the material moves (an optimiser's small steps); reported BESIDE
    #      the headline, never instead of it
    amortised = None
    if world == 1 and a.amortised_cycle > 1:
        cyc, ncyc = a.amortised_cycle, 2
        torch.cuda.synchronize()
        ta = time.time()
        for c in range(ncyc):
            outs = pipe.run_batch([(float(Es[h]), float(nus[h])) for h in mine], lanes=a.lanes)
            for k in range(1, cyc):
                for h, (_, res, _) in zip(mine, outs):
                    r, _, _ = pipe.run_cached_pass(res, float(Es[h]) * (1 + 1e-3 * k), float(nus[h]) * (1 - 5e-4 * k))
                    if not (np.isfinite(r.loss) and np.isfinite(r.grad_E) and np.isfinite(r.grad_nu)):
                        raise SystemExit("bench.py: non-finite loss or gradient in a cached pass")
        torch.cuda.synchronize()
        ta = time.time() - ta
        amortised = {"value": ncyc * cyc * len(mine) / ta, "unit": "passes/s", "eigen_decompose_cycle": cyc,
                     "cycles_timed": ncyc, "seconds": ta,
                     "what": (f"per hypothesis one complete pass (assembly + cold-start eigensolve + read-out + render + loss + "
                              f"backward) followed by {cyc - 1} passes of read-out + render + loss + backward on the kept "
                              f"eigenvectors with (E, nu) moved by 0.1 % / 0.05 % per pass - the reference's training loop "
                              f"with EIGEN_DECOMPOSE_CYCLE = {cyc} (experiments/material_sync_train.py:135-167)")}

    # ---- the path a user of the reference runs: one hypothesis at a time, its kernel time and launch
    #      count, and the literal loop of experiments/material_sync_train.py through the drop-in API - rank 0 of a 1-rank job
    one_hyp = kstats = api = None
    _cg_mark("timed steps + amortised variant")
    if world == 1 and not a.no_api_path:
        one_hyp = one_hypothesis_leg(pipe, hyps, dev, passes=12, steps_done=steps_done[0])
        _cg_mark("one hypothesis at a time")
        kstats = kernel_stats_leg(pipe, hyps, dev)
        _cg_mark("kernel statistics (torch profiler)")
        v0, t0_ = meshgen.kuhn_box(a.cells)
        api = api_path_leg(a, torch.from_numpy(v0).to(dev), torch.from_numpy(t0_).long().to(dev), dev, cfg)
        _cg_mark("API loop")
        del v0, t0_
        torch.cuda.empty_cache()

    sysd = pipe.system
    roof = None
    if nrec and not a.no_solo:
        ms_all = np.array(pms[:nrec], dtype=np.float64)
        nv_a, nz_a = np.array(pnv[:nrec], dtype=np.float64), np.array(pnz[:nrec], dtype=np.float64)
        nc_a = np.array(pnc[:nrec], dtype=np.float64)
        fl_a = np.array(pfi[:nrec], dtype=np.int64)
        fi_a, eb_a = (fl_a & 1).astype(np.float64), (fl_a >> 8).astype(np.float64)  # `first` flag, bytes per vector element
        # algorithmic bytes of a fused term (HipModalOps.cheb_term_bytes): values + ids, row pointers, block-Jacobi blocks,
        # W_k gathered, R0 and W_{k-1} read (not when `first`), W_{k+1} written
        # (9 values + one int32 id per block; the values count 2 bytes each where the level's bf16 term
        # runs on the matrix cores - the same rule as cheb_term_bytes)
        mf_levels = {int(o.nv) for o in (lane_ops[0], getattr(lane_ops[0], "coarse", None))
                     if o is not None and o._mfma is not None and o.kc is not None}
        vb_a = np.where((eb_a == 2) & np.isin(nv_a.astype(np.int64), sorted(mf_levels)), 2.0, 4.0)
        by_all = nz_a * (9 * vb_a + 4) + (nv_a + 1) * 4 + nv_a * 36 + (4 - fi_a) * 3 * nv_a * nc_a * eb_a
        kd_a = fl_a >> 16  # kind of the record (DS_PROF_*): the timed region records the fused term only
        assert (kd_a == 0).all()
        fl_a = fl_a & 0xFFFF
        fi_a, eb_a = (fl_a & 1).astype(np.float64), (fl_a >> 8).astype(np.float64)
        full = nc_a == a.block  # full-width blocks (after locking the narrower ones run another instantiation)
        ms, nbytes = ms_all[full], by_all[full]
        achieved = float(nbytes.sum() / (ms.sum() * 1e-3) / 1e9)
        nlanes = min(a.lanes, a.hyp_per_gpu)
        fine = nv_a[full] == sysd.nv
        levels = {name: {"launches": int(m.sum()), "avg_launch_ms": float(ms[m].mean()),
                         "algorithmic_bytes_per_launch": float(nbytes[m].mean()),
                         "achieved": float(nbytes[m].sum() / (ms[m].sum() * 1e-3) / 1e9)}
                  for name, m in (("fine", fine), ("corner_node", ~fine)) if m.any()}
        all_widths = {"launches": int(nrec), "achieved": float(by_all.sum() / (ms_all.sum() * 1e-3) / 1e9)}
        # ---- complete passes one at a time with every SpMM / Gram / update launch timed in place (roofline.in_pass)
        inpass_rec = in_pass_profile(pipe, hyps, dev, a.block, passes=2)
        # the same kernel alone on the device (fine level, 80 columns): what one launch achieves when it does not
        # share the chip with the other hypothesis lanes' kernels
        ops0 = lane_ops[0]
        bf = cfg.precond_storage == "bf16"
        vdt = torch.bfloat16 if bf else torch.float32
        Wk = torch.randn((sysd.n, a.block), device=dev).to(vdt)
        Wp, R0 = torch.randn((sysd.n, a.block), device=dev).to(vdt), torch.randn((sysd.n, a.block), device=dev).to(vdt)
        term = (lambda: ops0.cheb_spmm16(Wk, Wp, R0, 0.3, 0.7, False)) if bf else (lambda: ops0._cheb_spmm_launch(Wk, Wp, R0, 0.3, 0.7, False))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

        def alone(fn):
            """Average launch duration of ``fn`` alone on the device by HIP events: (the first 30 back-to-back launches after
            3 untimed ones, 100 launches after 200 more).  The first figure carries a transient - right after the synchronising
            end of the timed region successive 30-launch averages of K W fall from 0.201 to 0.181 ms over the first ~100
            launches and stay there (DS_BENCH_KW_SERIES=40 prints the series) - the second is the steady state the roofline
            fractions are quoted on; both are in the line."""
            out = []
            for warm, reps in ((3, 30), (200, 100)):
                for _ in range(warm):
                    fn()
                e0.record()
                for _ in range(reps):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                out.append(e0.elapsed_time(e1) / reps)
            return out

        solo_ms_first, solo_ms = alone(term)
        fine_bytes = ops0.cheb_term_bytes(a.block, elem_bytes=2 if bf else 4)
        solo = fine_bytes / (solo_ms * 1e-3) / 1e9
        del Wk, Wp, R0
        # the eigensolver's own stiffness product K W (fp32 blocks, the "LOBPCG SpMM" of the north star), alone as well - ON THE
        # OPERANDS THE ITERATION GIVES IT: W = the last b columns of the solver's [Y | X | P | W] basis buffer, K W = the last b
        # columns of its K [X P W] buffer (both with rows 1 KiB apart, lobpcg/modal_solver.py), and beside it on compact blocks
        ny_ = 16 if a.block % 16 == 0 else 8  # (the solver pads the rigid block to 16 columns: modal_solver.py)
        pitch = -(-(ny_ + 3 * a.block) // 256) * 256
        Sb, KSb = torch.randn((sysd.n, pitch), device=dev), torch.empty((sysd.n, pitch), device=dev)
        Xk, Yk = Sb[:, ny_ + 2 * a.block:ny_ + 3 * a.block], KSb[:, 2 * a.block:3 * a.block]

        kw_ms_first, kw_ms = alone(lambda: ops0.apply_K(Xk, Yk))
        if int(os.environ.get("DS_BENCH_KW_SERIES", "0")) > 0:
            series = [alone(lambda: ops0.apply_K(Xk, Yk))[0] for _ in range(int(os.environ["DS_BENCH_KW_SERIES"]))]
            print("K W, successive 30-launch averages (ms):", " ".join(f"{x:.4f}" for x in [kw_ms_first] + series), file=sys.stderr)
        Xc, Yc = torch.randn((sysd.n, a.block), device=dev), torch.empty((sysd.n, a.block), device=dev)
        kw_ms_compact = alone(lambda: ops0.apply_K(Xc, Yc))[1]
        del Sb, KSb, Xc, Yc
        kw_bytes = sysd.nnzb * 40 + (sysd.nv + 1) * 4 + 2 * sysd.n * a.block * 4
        del Xk, Yk
        # STREAM triad a = b + s c on the same device, right here: 3 arrays of 1 GiB (4x the Infinity Cache)
        ne = 1 << 28
        ta, tb, tc = (torch.empty(ne, device=dev) for _ in range(3))
        tb.fill_(1.0), tc.fill_(2.0)
        L = _hip.lib()
        triad_first, triad_ms = alone(lambda: _hip.check(L.ds_stream_triad(ta.data_ptr(), tb.data_ptr(), tc.data_ptr(), ne, 0.5,
                                                                           _hip.stream_ptr()), "ds_stream_triad"))
        stream_gbs = 3.0 * ne * 4 / (triad_ms * 1e-3) / 1e9
        del ta, tb, tc
        # ---- the Rayleigh-Ritz kernels alone on the device, at the iteration's shapes and on its operand layout (north star:
        #      "MFMA utilisation on the Rayleigh-Ritz reported against gfx950 peaks"): [Y X P W]^T (M W) of the orthonormalisation,
        #      [X P W]^T (K W) of the Ritz step, the in-place update W <- [Y X P W] C and [X' P'] = [X P W] [Z1 Zp]
        b_ = a.block
        Sb = torch.randn((sysd.n, pitch), device=dev)
        S2b = torch.empty((sysd.n, pitch), device=dev)
        KSb = torch.randn((sysd.n, pitch), device=dev)
        MWb = torch.randn((sysd.n, b_), device=dev)
        rr_solo = {}

        def rr_record(name, p_, q_, fn, what):
            first_, t = alone(fn)
            flops, nbytes = 2.0 * sysd.n * p_ * q_, sysd.n * (p_ + q_) * 4.0
            rr_solo[name] = {"what": what, "avg_ms": t, "avg_ms_first_30": first_, "tflops": flops / (t * 1e-3) / 1e12,
                             "frac_of_mfma_f32_peak": flops / (t * 1e-3) / 1e12 / MFMA_F32_PEAK_TFS,
                             "gbs": nbytes / (t * 1e-3) / 1e9, "frac_of_hbm_peak": nbytes / (t * 1e-3) / 1e9 / HBM_PEAK_GBS}

        pw = ny_ + 3 * b_
        rr_record(f"gram_{pw}x{b_}", pw, b_, lambda: ops0.gram(Sb[:, :pw], MWb), "[Y X P W]^T (M W): gram32_partial_kernel + gram_reduce_kernel")
        rr_record(f"gram_{3 * b_}x{b_}", 3 * b_, b_, lambda: ops0.gram(Sb[:, ny_:pw], KSb[:, 2 * b_:3 * b_]),
                  "[X P W]^T (K W): the Ritz step's Gram block")
        Cw = torch.randn((pw, b_), device=dev) / pw
        rr_record(f"mix_{pw}x{b_}", pw, b_, lambda: ops0.mix(Sb[:, :pw], Cw, Sb[:, pw - b_:pw]), "W <- [Y X P W] C in place: mix_lds_kernel<5>")
        Cz = torch.randn((3 * b_, 2 * b_), device=dev) / pw
        rr_record(f"mix_{3 * b_}x{2 * b_}", 3 * b_, 2 * b_, lambda: ops0.mix(Sb[:, ny_:pw], Cz, S2b[:, ny_:ny_ + 2 * b_]),
                  "[X' P'] = [X P W] [Z1 Zp]: mix_lds_kernel<10>")
        rr_record(f"gram_{pw}x{2 * b_}", pw, 2 * b_, lambda: ops0.gram(Sb[:, :pw], KSb[:, :2 * b_]),
                  "[Y X P W]^T [K W | M W]: the ONE Gram launch of an iteration on the raw basis (round 5)")
        Cr = torch.randn((pw, 2 * b_), device=dev) / pw
        rr_record(f"mix_{pw}x{2 * b_}", pw, 2 * b_, lambda: ops0.mix(Sb[:, :pw], Cr, S2b[:, ny_:ny_ + 2 * b_]),
                  "[X' P'] = [Y X P W] Z_raw: the ONE update of an iteration on the raw basis (round 5)")
        del Sb, S2b, KSb, MWb, Cw, Cz, Cr
        # MFMA pipe utilisation of those kernels from the rocprofv3 --pmc passes of tools/collect_profiles.sh (a counter pass cannot
        # run inside this process): SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 4 SIMDs x CUs), valid for the kernel sources it names
        rr_pmc, rr_pmc_note = None, "no counter record under profiles/ (tools/collect_profiles.sh part 2 writes profiles/gram_mix_mfma_util.json)"
        try:
            rec = json.load(open(os.path.join(ROOT, "profiles", "gram_mix_mfma_util.json")))
            if rec.get("dense_source_sha16") == dense_source_hash():
                rr_pmc, rr_pmc_note = rec["kernels"], f"counter passes of {rec.get('measured', '?')} on these kernel sources"
            else:
                rr_pmc_note = f"stale: measured on kernel sources {rec.get('dense_source_sha16')}, this run's are {dense_source_hash()}"
        except (OSError, ValueError, KeyError):
            pass
        mf_lv = {int(o.nv) for o in (lane_ops[0], getattr(lane_ops[0], "coarse", None))
                 if o is not None and o._mfma is not None and o.kc is not None}
        inpass, rr_inpass, inpass_kinds = summarize_in_pass(inpass_rec, sysd, a.block, stream_gbs, mf_lv)
        pmc = os.path.join(ROOT, "profiles", "spmm_pmc_bytes_per_launch.json")
        try:  # PMC bytes per launch of the same forms (records of that file, valid for these kernel sources only)
            doc = json.load(open(pmc)) if os.path.exists(pmc) else {}
            for name, key in (("fused_term_bf16", "mfma" if mf_lv else "bf16"), ("lobpcg_kw", "kx"), ("lobpcg_kw_mw", "km"),
                              ("fused_residual", "resid")):
                rec = doc.get(f"cells{a.cells}_cols{a.block}_{key}")
                if name in inpass and isinstance(rec, dict) and rec.get("spmm_source_sha16") == spmm_source_hash():
                    inpass[name]["traffic"] = rec["bytes"]
                    inpass[name]["traffic_over_algorithmic"] = rec["bytes"] / inpass[name]["algorithmic_bytes_per_launch"]
        except (OSError, ValueError, KeyError):
            pass
        # PMC bytes of one such launch: a figure measured in its own rocprofv3 --pmc passes (tools/collect_profiles.sh)
        # and valid ONLY for the kernel sources it was measured on - another source hash means stale, reported as null
        traffic, traffic_note = None, "no PMC figure under profiles/ for this shape"
        pmc = os.path.join(ROOT, "profiles", "spmm_pmc_bytes_per_launch.json")
        if os.path.exists(pmc):
            try:
                mf = bf and getattr(ops0, "_mfma", None) is not None and ops0.kc is not None
                doc = json.load(open(pmc))
                rec = doc.get(f"cells{a.cells}_cols{a.block}_{'mfma' if mf else ('bf16' if bf else 'fp32')}")
                if isinstance(rec, dict):
                    if rec.get("spmm_source_sha16") == spmm_source_hash():
                        traffic, traffic_note = rec["bytes"], f"PMC passes of {rec.get('measured', '?')} on these kernel sources"
                    else:
                        traffic_note = (f"stale: measured on kernel sources {rec.get('spmm_source_sha16')}, this run's are "
                                        f"{spmm_source_hash()}")
            except Exception as ex:
                traffic_note = f"unreadable PMC record: {ex}"
        kw_traffic = None  # the same for the eigensolver's Y = K X (record "kx")
        try:
            rec = json.load(open(pmc)).get(f"cells{a.cells}_cols{a.block}_kx") if os.path.exists(pmc) else None
            if isinstance(rec, dict) and rec.get("spmm_source_sha16") == spmm_source_hash():
                kw_traffic = rec["bytes"]
        except Exception:
            pass
        # The figure a pass sees: ``achieved`` / ``frac`` / ``avg_launch_ms`` are the dominant kernel's launches INSIDE
        # complete passes (one at a time, every launch bracketed by HIP events on its own stream); the kernel's steady state in a
        # back-to-back series - the best case, what rounds 1-5 quoted - is kept beside it as ``*_alone``.
        ip = inpass.get("fused_term_bf16") if isinstance(inpass, dict) else None
        in_ms = ip["avg_launch_ms"] if ip else solo_ms
        in_gbs = fine_bytes / (in_ms * 1e-3) / 1e9
        roof = {"bound": "hbm", "achieved": in_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": in_gbs / HBM_PEAK_GBS, "traffic": traffic, "traffic_note": traffic_note,
                "stream_triad": stream_gbs, "frac_of_stream": in_gbs / stream_gbs,
                "achieved_alone": solo, "frac_alone": solo / HBM_PEAK_GBS, "frac_of_stream_alone": solo / stream_gbs,
                "launches_timed_in_pass": (ip or {}).get("launches"),
                "kernel": ((f"spmm_union_mfma_kernel<8,{(a.block + 15) // 16},1>: W' = W + c1(W - W_prev) + c2 T(R0 - K W) on a "
                            f"{a.block}-column block, fine level (bf16 K blocks and iterates, block products on the matrix "
                            "cores with fp32 accumulation, fp32 epilogue)")
                           if bf and getattr(ops0, "_mfma", None) is not None and ops0.kc is not None else
                           (f"spmm_union_kernel<{a.block // 4},1,...,{'bf16' if bf else 'fp32'} blocks>: W' = W + c1(W - W_prev) + "
                            f"c2 T(R0 - K W) on a {a.block}-column block, fine level (fp32 K blocks, "
                            f"{'bf16' if bf else 'fp32'} iterates, fp32 arithmetic)")),
                "algorithmic_bytes_per_launch": fine_bytes, "avg_launch_ms": in_ms, "avg_launch_ms_alone": solo_ms,
                "avg_launch_ms_first_30": solo_ms_first,
                "stream_triad_first_30": 3.0 * ne * 4 / (triad_first * 1e-3) / 1e9,
                "how": ("'achieved' = algorithmic bytes of ONE fine-level fused-term launch / the mean HIP-event time of its launches "
                        "inside complete passes run one at a time ('in_pass', below: each launch alone on the device but between "
                        "the other kernels of a pass); 'achieved_alone' / 'avg_launch_ms_alone' = the same kernel "
                        "alone on the device, on the compact blocks the V-cycle runs it on, right after the timed "
                        "region: 100 back-to-back launches after 230 untimed ones - the steady state; the first 30 launches "
                        "after the synchronising end of the timed region run 5-10 % longer (a transient of ~100 launches, "
                        "'avg_launch_ms_first_30'; rounds 1-3 quoted that figure); 'stream_triad' = ds_stream_triad on 3 x 1 GiB "
                        "arrays, same device, same run, timed the same way; 'traffic' = PMC bytes of one such launch "
                        "(profiles/)"),
                "lobpcg_spmm": {"kernel": (f"spmm_union_kernel<{a.block // 4},0,140,...,0>: Y = K X on a {a.block}-column fp32 block (K W of the "
                                           f"iteration), on the iteration's own operands: W = columns {ny_ + 2 * a.block}..{ny_ + 3 * a.block} of the "
                                           f"[Y | X | P | W] basis buffer, K W = columns {2 * a.block}..{3 * a.block} of the K [X P W] buffer, rows "
                                           f"{pitch * 4} bytes apart; alone on the device, 100 back-to-back launches after 230 untimed "
                                           "ones, HIP events"),
                                "algorithmic_bytes_per_launch": kw_bytes, "avg_launch_ms": kw_ms, "avg_launch_ms_first_30": kw_ms_first,
                                "avg_launch_ms_on_compact_blocks": kw_ms_compact,
                                "achieved": kw_bytes / (kw_ms * 1e-3) / 1e9, "frac": kw_bytes / (kw_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                "frac_of_stream": kw_bytes / (kw_ms * 1e-3) / 1e9 / stream_gbs, "traffic": kw_traffic},
                "in_pass": {"how": ("complete passes one at a time on one stream (one hypothesis lane), every launch of these kernels "
                                    "bracketed by HIP events inside the library: alone on the device but BETWEEN the other kernels of a "
                                    "pass - the figure to hold against the roofline of a real pass; 'avg_launch_ms' above is the kernel's "
                                    "own steady state in a back-to-back series"),
                            "passes": inpass_rec["passes"], "seconds_per_pass_one_at_a_time": inpass_rec["seconds_per_pass"],
                            "kernels": inpass, "event_time_per_pass_ms_by_kind": inpass_kinds},
                "rayleigh_ritz": {"peak": MFMA_F32_PEAK_TFS, "unit": "TFLOP/s", "peak_what": "v_mfma_f32_16x16x4_f32, dense, 256 CUs x 2.4 GHz",
                                  "alone": rr_solo, "in_pass": rr_inpass, "mfma_busy": rr_pmc, "mfma_busy_note": rr_pmc_note,
                                  "how": ("Gram G = A^T B (fp32 MFMA folded into fp64) and update Out = A C at the iteration's shapes and "
                                          "operand layout; 'alone' = 100 back-to-back launches after 230 untimed ones, 'in_pass' = the "
                                          "launches of the one-at-a-time passes; flops = 2 n p q, bytes = n (p + q) 4")},
                "in_situ": {"achieved": achieved, "frac": achieved / HBM_PEAK_GBS,
                            "algorithmic_bytes_per_launch": float(nbytes.mean()), "avg_launch_ms": float(ms.mean()),
                            "launches_timed": int(len(ms)), "levels": levels, "all_block_widths": all_widths,
                            "note": (f"HIP events (ds_profile_stream) around every fused-term launch on the FIRST hypothesis "
                                     f"lane's stream inside the timed region, {a.block}-column blocks, fine and corner-node "
                                     f"level; {nlanes} lanes launch concurrently on separate streams, so a launch shares "
                                     "the device with the other lanes' kernels and its duration is stretched accordingly")}}

    if getattr(pipe.ops, "coarse", None) is not None and a.precond != "chebyshev":
        precond_desc = (f"two-level V-cycle: Chebyshev({a.smooth_degree}, ratio {a.smooth_ratio:g}) block-Jacobi smoother + "
                        f"corner-node P1 level Chebyshev({a.coarse_degree}, ratio {a.coarse_ratio:g})")
    else:
        precond_desc = f"Chebyshev({a.cheb_degree}) block-Jacobi"
    if rank == 0:
        passes = a.steps * nhyp
        out = {
            "metric": "fwd+bwd modal-analysis passes/sec, 100k-tet ord-2 mesh, 64 modes",
            "value": passes / dt,
            "unit": "passes/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": 1e3 * dt / a.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": (f"Kuhn box {a.cells}^3 cells = {sysd.T} tets, ord-{a.order} ({sysd.nv} nodes, n={sysd.n}, "
                             f"nnz={sysd.nnzb * 9}), {a.modes} modes, fwd+bwd w.r.t. (E, nu), S=8000 @ 32 kHz"),
                "hypotheses_per_gpu_per_step": a.hyp_per_gpu,
                "loss": "MSE against a fixed target clip" if a.loss == "mse" else "MSSLoss [1024..64] l1_loss (STFT kernels)",
                "parallelism": (f"dp{world} over material hypotheses, scalar loss all-reduce; {min(a.lanes, a.hyp_per_gpu)} "
                                "hypotheses in flight per GPU (one HIP stream + host thread each)"),
                "step_schedule": ("all lanes join after every step" if a.step_barrier else
                                  "no join between steps: a lane runs its hypotheses' passes of consecutive steps back to back (a "
                                  "hypothesis' next step depends on its own previous one only); the step's loss all-reduce is issued "
                                  "as soon as every lane of the rank has finished that step"),
                "precision": ("fp32 block vectors and SpMM, fp64 Gram accumulation / Rayleigh-Ritz / read-out; the preconditioner's "
                              f"internal blocks in {cfg.precond_storage}"),
                "eigensolver": (f"LOBPCG(ortho) block {a.block}, {precond_desc}, "
                                f"cold start{' (warm)' if a.warm_start else ''}, mean iterations {np.mean(iters):.1f}"
                                + (f" after a nested start (mean {np.mean(cits):.1f} corner-node level iterations to "
                                   f"{a.nested_tol:g}" + (f", its random block through {cfg.start_sweeps} preconditioner sweeps first" if cfg.start_sweeps else "")
                                   + ")" if a.nested_tol > 0 else "")
                                + f", backward-error tolerance {tol:g}"),
                "host_wait": (f"{a.host_wait}: " + ("a lane's host thread sleeps while it waits for its stream - the native solve polls "
                                                   "20 us, then waits on a blocking event (ds_host_wait_mode 1); every other wait through "
                                                   f"hipDeviceScheduleBlockingSync ({'set' if sleeping_waits else 'NOT set: the context existed'}) - "
                                                   "waiting lanes leave their cores to the computing ones"
                                                   if a.host_wait == "sleep" and a.lanes > 1 else "hipStreamSynchronize (the runtime spins)")),
                "raw_start": (lambda st: {"taken": st[0], "explicit_route": st[1],
                                          "what": "start blocks (corner-node level: random; fine level: the prolonged corner-level vectors) "
                                                  "orthonormalised and rotated in coefficients from one [K X0 | M X0] walk and one Gram launch; "
                                                  "a block too ill-conditioned for one sweep takes the explicit route"}
                              )(summed_stats(pipe, "raw_start_stats")),
                "warm_power_iteration": (lambda st: {"estimates": st[0], "mean_steps": round(st[1] / max(1, st[0]), 2),
                                                     "what": "lambda_max(T K) of the two Chebyshev intervals per pass, from the previous "
                                                             "material's block: steps until two successive estimates agree to 0.3 % (at least 2)"}
                                         )(summed_stats(pipe, "warm_stats")),
                "symbolic_seconds_not_timed": t_sym,
                "symbolic_phase": ("pattern + contribution lists + neighbour-union tables on the device (ds_dpattern_build), "
                                   "plus the corner-node level, operators and first assembly; once per topology"),
                "convergence_gate": (f"every timed pass converged: backward error of all {a.modes} pairs < {tol:.1g} "
                                     f"(worst {worst[0]:.3g}), iterations < {cfg.maxit}, finite loss and gradients"),
            },
            "roofline": roof,
            # one hypothesis at a time (the headline keeps `lanes` in flight), its kernel time / launches, and the reference's own
            # training loop through the drop-in API: VERDICT r05 items 1 and 5
            "one_hypothesis_passes_per_s": None if one_hyp is None else one_hyp["passes_per_s"],
            "one_hypothesis": one_hyp,
            "kernel_ms_per_pass": (kstats or {}).get("kernel_ms_per_pass"),
            "launches_per_pass": (kstats or {}).get("launches_per_pass"),
            "kernel_stats": kstats,
            "api_path": api,
            "materials": ("every step each hypothesis' material moves: E (1 + 1e-3 s), nu (1 - 5e-4 s) at global step s (warm-up "
                          "included) - no pass repeats a material" if not a.same_material else "the same (E, nu) every step"),
            "loss_sum_last_step": total,
            "ranks": per_rank,
            # wall-clock seconds between the completions of consecutive steps on rank 0 (warm-up steps first): shows whether the
            # timed steps ran in a steady state
            "step_completion_intervals_s": [round(b - a_, 4) for a_, b in zip(on_step.stamps[:-1], on_step.stamps[1:])],
            "amortised": amortised,
            "collective": (f"{a.dist_backend} all-reduce of the scalar loss over {world} ranks" if world > 1 else "none (1 rank)"),
            "process_group": ({"backend": dist.get_backend(), "world_size": dist.get_world_size(),
                               "is_rccl": dist.get_backend() == "nccl"} if world > 1 else None),
            "host_load": host_load_record(),
            "host_threads": {"hardware_threads": hw_threads, "ranks_on_node": local_world, "lanes_per_rank": min(a.lanes, a.hyp_per_gpu),
                             "lane_threads_on_node": local_world * min(a.lanes, a.hyp_per_gpu), "lapack_threads_per_lane": 1,
                             "within_half_of_hardware_threads": local_world * min(a.lanes, a.hyp_per_gpu) <= hw_threads // 2},
        }
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.cpu_sample_cells, a.order, a.modes, sysd.T, a.cpu_reps, a.cpu_second_cells)
            out["cpu_baseline"]["memory_safe_restatement"] = memsafe_curve(sysd.T, out["cpu_baseline"].get("cpu_model"))
        print(json.dumps(out))
    if world > 1:
        barrier()  # rank 0's solo kernel timings above ran while the others wait here (a shared device stays quiet)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
