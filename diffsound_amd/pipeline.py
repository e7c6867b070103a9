"""One forward+backward modal-analysis pass and its data-parallel batching over hypotheses.

A *pass* is the body of the reference training loop with the eigendecomposition every epoch
(reference experiments/material_sync_train.py:137-167 with EIGEN_DECOMPOSE_CYCLE = 1):
  1. numeric assembly of K_lambda, K_mu, M_s (symbolic pattern reused),
  2. K = lam K_lambda + mu K_mu, block eigensolve for ``modes`` elastic modes (cold start),
  3. f = get_undamped_freqs()         (differentiable read-out through a_i, b_i, m_i),
  4. oscillator render (A = 1, S samples, impulse force),
  5. scalar loss (MSE against a fixed target signal),
  6. backward to (E, nu).
Hypotheses (E_i, nu_i) on one mesh are independent: rank r of a ``torch.distributed`` job takes
hypotheses r, r + world, ... ; the ONLY collective is the all-reduce of the scalar loss (RCCL).
"""
from dataclasses import dataclass

import numpy as np
import torch
import torch.nn as nn

from .ddsp.oscillator import TraditionalDampedOscillator
from .diffelastic.material_model import Material
from .lobpcg.modal_solver import ModalSolver, SolverConfig
from . import _hip
from .modal_ops import HipModalOps, TetSystem


class DirectLinear(nn.Module):
    """(E, nu) as direct leaves - one material hypothesis of the inverse-rendering batch."""

    def __init__(self, youngs, poisson, mat):
        super().__init__()
        self.E = nn.Parameter(torch.tensor(float(youngs), dtype=torch.float64))
        self.nu = nn.Parameter(torch.tensor(float(poisson), dtype=torch.float64))
        self.mat = mat

    def youngs(self):
        return self.E

    def poisson(self):
        return self.nu

    def lame(self):
        E, nu = self.E, self.nu
        return E * nu / ((1 + nu) * (1 - 2 * nu)), E / (2 * (1 + nu))


import os as _os

_PASS_TIMING = bool(_os.environ.get("DS_EXP_TIMING"))


def shard_hypotheses(num, rank, world):
    """Indices of the hypotheses owned by ``rank`` (round-robin; embarrassingly parallel)."""
    return list(range(rank, num, world))


@dataclass
class PassResult:
    loss: float
    grad_E: float
    grad_nu: float
    freqs: torch.Tensor
    iterations: int
    eigenvalues: torch.Tensor
    coarse_iterations: int = 0
    max_rerr: float = float("nan")  # largest backward error ||K u - lambda M u|| / (||u|| (||K|| + lambda ||M||)) of the wanted pairs


class ModalPipeline:
    """Mesh-bound state shared by all hypotheses on this GPU: pattern, assembled K_lambda/K_mu/M_s,
    oscillator, target audio."""

    def __init__(self, vertices, tets, order, modes, mat, sample_num=8000, sr=32000, force_frames=150,
                 solver_config=None, target_freqs=None, loss_fn=None, mfma_groups=None, host_wait="sleep",
                 coarse_group_jacobi=None):
        """``mfma_groups``: nodes per wavefront of the MFMA form of the preconditioner's bf16 terms on (fine, corner-node) level,
        handed to every operator object THIS pipeline builds (None: the library default).  ``host_wait``: how this pipeline's
        lanes wait for their streams when several are in flight ("sleep" | "spin").  Both are per pipeline (round 6): two
        pipelines in one process - or ranks run as threads - do not share a knob."""
        self.device = vertices.device
        self.mfma_groups = None if mfma_groups is None else tuple(mfma_groups)
        # (group-block Jacobi of the corner-node level's polynomial: 8 / 0, None = HipModalOps' default; per pipeline)
        self.coarse_group_jacobi = coarse_group_jacobi
        self.host_wait = host_wait
        # scalar loss head (audio, target) -> 0-dim tensor; None = the MSE of the headline metric.  The reference's
        # experiments put MSSLoss here (experiments/material_sync_train.py:124,159): pass that module.
        self.loss_fn = loss_fn
        self.modes = modes
        self.mat = Material(mat)
        self.cfg = solver_config or SolverConfig()
        self.system = TetSystem(vertices, tets, order, self.mat.density)
        self.ops = None
        self._lanes = []
        force = torch.zeros((1, force_frames), device=self.device)
        force[0, 0] = 1  # impulse (reference material_sync_train.py:103-104)
        self.osc = TraditionalDampedOscillator(force, 1, modes, sample_num, sr, self.mat)
        self.target = None
        self.target_freqs = target_freqs
        self.vertices = vertices

    def set_target(self, audio):
        self.target = audio.detach()

    def assemble(self):
        self.system.assemble()

    def run_pass(self, youngs, poisson, warm=None, backward=True, _lane=None, assemble=True):
        """One complete pass for one hypothesis: numeric assembly (step 1; ``assemble=False`` skips it when the
        caller has just run ``assemble()``), then steps 2-6."""
        model = DirectLinear(youngs, poisson, self.mat)
        lam, mu = model.lame()
        holder = self if _lane is None else _lane
        timing = _PASS_TIMING  # (DS_EXP_TIMING=1: where the host time of one pass goes, with a device synchronisation per stage)
        if timing:
            import sys
            import time

            stamps = []

            def lap(name):
                torch.cuda.current_stream(self.device).synchronize()
                stamps.append((name, time.time()))

            lap("start")
        if assemble:
            holder.system.assemble()
        lam_f, mu_f = float(lam.detach()), float(mu.detach())
        if holder.ops is None:
            holder.ops = HipModalOps(holder.system, lam_f, mu_f, mfma_groups=self.mfma_groups,
                                     coarse_group_jacobi=self.coarse_group_jacobi)
            set_wait_mode(holder.ops, getattr(holder, "host_wait_mode", 0))  # (the native solve's waits: this lane's own setting)
        else:
            holder.ops.set_material(lam_f, mu_f)
        if timing:
            lap("assemble + set_material")
        solver = ModalSolver(holder.ops, self.cfg)
        solver.keep_block = bool(getattr(self, "keep_blocks", True))
        if timing:
            lap("preconditioner set-up (power iterations)")
        res = solver.solve(self.modes, X0=warm)
        if timing:
            lap("solve (nested start + iteration + polish)")
        out = self._readout(holder, model, res, backward)
        if timing:
            lap("read-out + render + loss + backward")
            print("pass: " + ", ".join(f"{n} {1e3 * (t - stamps[i][1]):.2f} ms" for i, (n, t) in enumerate(stamps[1:])) +
                  f", total {1e3 * (stamps[-1][1] - stamps[0][1]):.2f} ms", file=sys.stderr, flush=True)
        return out

    def run_cached_pass(self, res, youngs, poisson, backward=True, _lane=None):
        """A pass BETWEEN eigendecompositions (reference experiments/material_sync_train.py:135-141 with
        EIGEN_DECOMPOSE_CYCLE = 15: ``eigen_decomposition()`` only every 15th epoch, ``get_undamped_freqs()`` - the
        first-order read-out on the kept eigenvectors - every epoch): steps 3-6 of a pass on the quadratic forms
        a_i, b_i, m_i of an earlier solve ``res`` with the current (E, nu).  No assembly, no eigensolve."""
        model = DirectLinear(youngs, poisson, self.mat)
        return self._readout(self if _lane is None else _lane, model, res, backward)

    def _readout(self, holder, model, res, backward):
        if self.loss_fn is None and type(holder.osc) is TraditionalDampedOscillator and holder.osc.audio_num == 1:
            return self._readout_native(holder, model, res, backward)
        return self._readout_torch(holder, model, res, backward)

    def _call_loss(self, audio, damped_freq):
        import inspect

        takes = getattr(self, "_loss_takes_freq", None)
        if takes is None:
            try:
                params = inspect.signature(self.loss_fn.forward if isinstance(self.loss_fn, nn.Module) else self.loss_fn).parameters
                takes = "freq" in params
            except (TypeError, ValueError):
                takes = False
            self._loss_takes_freq = takes
        if takes and damped_freq is not None:
            return self.loss_fn(audio, self.target, freq=damped_freq)
        return self.loss_fn(audio, self.target)

    def _readout_torch(self, holder, model, res, backward):
        """Steps 3-6 as torch operations with autograd (any loss head, e.g. MSSLoss; also the cross-check of the native path)."""
        lam, mu = model.lame()
        ev = res.eigenvalues
        dev = ev.device
        pred = ev + (lam.to(dev) * res.a_lambda + mu.to(dev) * res.b_mu) - ev * res.m_diag
        freqs = (torch.sqrt(pred) / 2 / np.pi).float().unsqueeze(1)
        audio = holder.osc(freqs)
        if self.target is None:
            loss = (audio ** 2).mean()
        elif self.loss_fn is not None:
            # (the Sinkhorn head of the reference's MSSLoss wants the damped frequencies beside the audio:
            # src/ddsp/mss_loss.py:104-117; the l1 / rmse heads ignore them)
            loss = self._call_loss(audio, getattr(holder.osc, "damped_freq", None))
        else:
            loss = ((audio - self.target) ** 2).mean()
        gE = gnu = float("nan")
        if backward:
            loss.backward()
            gE, gnu = float(model.E.grad), float(model.nu.grad)
        rerr = float(res.rerr.max()) if res.rerr is not None else float("nan")
        return (PassResult(float(loss.detach()), gE, gnu, freqs.detach(), res.iterations, ev, res.coarse_iterations, rerr),
                res, audio.detach())


    def _readout_native(self, holder, model, res, backward):
        """Steps 3-6 with the headline's MSE loss as ONE native call (``ds_readout_pass``: three small kernels around the two
        oscillator launches) and one 24-byte copy back - the torch formulation below it is ~45 element-wise launches on
        64-element vectors plus the autograd engine, 0.83 ms of host time per pass.  Same arithmetic (reference
        diff_model.py:371-388, oscillator.py:282-310); gradients by the chain rule in closed form."""
        from . import _hip

        E, nu = float(model.E.detach()), float(model.nu.detach())
        lam, mu = E * nu / ((1 + nu) * (1 - 2 * nu)), E / (2 * (1 + nu))
        dlam_dE, dlam_dnu = nu / ((1 + nu) * (1 - 2 * nu)), E * (1 + 2 * nu * nu) / ((1 + nu) ** 2 * (1 - 2 * nu) ** 2)
        dmu_dE, dmu_dnu = 1 / (2 * (1 + nu)), -E / (2 * (1 + nu) ** 2)
        osc = holder.osc
        ev = res.eigenvalues
        dev = ev.device
        m, S = int(ev.shape[0]), int(osc.sample_num)
        if m != int(osc.mode_num):  # (the torch path fails the same way, in the bank's reshape)
            raise ValueError(f"the solve returned {m} modes, the oscillator bank was built for {osc.mode_num}")
        buf = getattr(holder, "_readout_buf", None)
        if buf is None or buf[0].shape[0] != 6 * m or buf[1].shape[0] != 2 * S or buf[0].device != dev:
            buf = (torch.empty(6 * m, dtype=torch.float64, device=dev), torch.empty(2 * S, dtype=torch.float32, device=dev),
                   torch.empty(3, dtype=torch.float64, device=dev))
            holder._readout_buf = buf
        work, fwork, out = buf
        if osc._force.device != dev:
            osc._force = osc._force.to(dev)
        audio = torch.empty((1, S), dtype=torch.float32, device=dev)
        freqs = torch.empty(m, dtype=torch.float32, device=dev)
        tgt = None if self.target is None else self.target.reshape(-1).float().contiguous()
        if tgt is not None and tgt.shape[0] != S:
            raise ValueError("target clip length differs from the oscillator's sample_num")
        a, b, md = res.a_lambda.double().contiguous(), res.b_mu.double().contiguous(), res.m_diag.double().contiguous()
        p = _hip.ptr
        _hip.check(_hip.lib().ds_readout_pass(p(ev.double().contiguous()), p(a), p(b), p(md), m, lam, mu, dlam_dE, dlam_dnu,
                                              dmu_dE, dmu_dnu, float(osc.alpha), float(osc.beta), p(osc._force),
                                              int(osc._force.shape[1]), S, float(osc.sr), p(tgt), 1 if backward else 0,
                                              p(audio), p(freqs), p(work), p(fwork), p(out), _hip.stream_ptr()),
                   "ds_readout_pass")
        host = out.tolist()  # the pass's one synchronisation
        # what code written against the reference reads off the bank after a forward (oscillator.py:303-305; e.g.
        # experiments/material_sync_train.py:167): the undamped and damped frequencies of this pass
        osc.undamped_freq = freqs.reshape(1, m, 1)
        osc.damped_freq = (work[m:2 * m] / (2 * np.pi)).float().reshape(1, m, 1)  # (work[m:2m]: damped angular frequencies)
        gE, gnu = (host[1], host[2]) if backward else (float("nan"), float("nan"))
        rerr = getattr(res, "_rerr_max", None)
        if rerr is None:
            rerr = float(res.rerr.max()) if res.rerr is not None else float("nan")
            try:
                res._rerr_max = rerr
            except AttributeError:
                pass
        return (PassResult(host[0], gE, gnu, freqs.unsqueeze(1), res.iterations, ev, res.coarse_iterations, rerr), res, audio)


class _Lane:
    """Per-stream state of one of the concurrent hypothesis lanes (own material operators and scratch)."""

    def __init__(self, pipe, ops=None):
        self.ops = ops
        # the first lane adopts the pipeline's system (and operators); the others assemble into their own arrays
        self.system = pipe.system if ops is not None or not pipe._lanes else pipe.system.with_own_values()
        self.stream = torch.cuda.Stream(device=pipe.device)
        self.host_wait_mode = 0  # 0: the runtime's stream synchronisation, 1: poll, then sleep on a blocking event (set with the pool)
        self.osc = TraditionalDampedOscillator(pipe.osc._force.clone(), 1, pipe.modes, pipe.osc.sample_num, pipe.osc.sr,
                                               pipe.mat)


def set_wait_mode(ops, mode):
    """How the native solves on this operator object (and on its corner-node level) wait for their stream: 0 = the runtime's
    stream synchronisation (a spin), 1 = poll, then sleep on a blocking event.  Per operator object, i.e. per lane."""
    ops.host_wait_mode = int(mode)
    if getattr(ops, "coarse", None) is not None:
        ops.coarse.host_wait_mode = int(mode)


def prefer_sleeping_waits(device=None):
    """Ask the HIP runtime to SLEEP, not spin, whenever a host thread of this process waits for the device
    (hipSetDeviceFlags(hipDeviceScheduleBlockingSync)) - for processes that run several hypothesis lanes: every wait of a lane,
    the interpreter's included, then leaves its core to the lanes that are computing.  Must run BEFORE the process first touches
    the device (the flag belongs to the device's primary context); returns False when it is too late or the runtime refuses -
    the native solve's own waits sleep regardless (ds_host_wait_mode, set when the lane pool is built).
    Measured on the benchmark (8 lanes): 61.8 passes/s on FOUR host cores against 52 - 53 with spinning waits, 62.4 against 63.1
    with cores to spare (profiles/r05_host_wait_mode.txt)."""
    import ctypes

    try:
        hip = ctypes.CDLL("libamdhip64.so")
        if device is not None and int(hip.hipSetDevice(ctypes.c_int(int(device)))) != 0:  # (the flag goes to the CURRENT device)
            return False
        return int(hip.hipSetDeviceFlags(ctypes.c_uint(0x4))) == 0
    except (OSError, AttributeError):
        return False


def _lane_pool(pipe, lanes):
    """The pipeline's persistent lane threads; a pool that is too small is shut down (its threads and their thread-local
    pinned staging rings released) before a larger one replaces it."""
    pool = getattr(pipe, "_lane_pool", None)
    if pool is None or pool._max_workers < lanes:
        from concurrent.futures import ThreadPoolExecutor

        if pool is not None:
            pool.shutdown(wait=True)
        from .lobpcg.modal_solver import one_blas_thread_for_this_thread

        pool = pipe._lane_pool = ThreadPoolExecutor(max_workers=lanes, thread_name_prefix="ds-lane",
                                                    initializer=one_blas_thread_for_this_thread)
    # Several lanes: a lane's host thread SLEEPS while it waits for its stream (round 5, ds_host_wait_mode) - the lanes wait four
    # fifths of the time, and spinning they would hold one core each, which a host under CPU quota or beside busy neighbours does
    # not have to give (on four cores the benchmark runs 60 passes/s with sleeping waits against 52 with spinning ones; with
    # cores to spare the sleep's wake-up costs 2 %).  ``host_wait`` = "spin" keeps the runtime's own synchronisation.
    # Per LANE since round 6 (ds_lobpcg_t.wait_mode; ds_host_wait_mode only sets the default of descriptors that say -1): another
    # pipeline of the process, or one hypothesis at a time on this one, keeps its own setting.
    mode = 0 if getattr(pipe, "host_wait", "sleep") == "spin" or lanes <= 1 else 1
    for ln in pipe._lanes:
        ln.host_wait_mode = mode
        if ln.ops is not None:
            set_wait_mode(ln.ops, mode)
    return pool


def _run_batch(pipe, hyps, lanes=2, warm=None, backward=True):
    """Hypotheses are independent, so ``lanes`` of them are in flight at once, each on its own HIP stream driven
    by its own host thread: while one lane's small dense Rayleigh-Ritz step runs on the host (the GPU would idle
    for ~2.5 ms per iteration), the other lane's kernels keep the device busy.  Returns the per-hypothesis
    ``run_pass`` results in order.  ``warm``: optional list of start blocks."""
    import threading

    n = len(hyps)
    if lanes <= 1 or n <= 1:
        return [pipe.run_pass(E, nu, warm=None if warm is None else warm[i], backward=backward)
                for i, (E, nu) in enumerate(hyps)]
    while len(pipe._lanes) < lanes:
        pipe._lanes.append(_Lane(pipe, ops=pipe.ops if not pipe._lanes else None))
    main = torch.cuda.current_stream(pipe.device)
    ready = torch.cuda.Event()
    ready.record(main)
    out, errs, done = [None] * n, [], []

    def work(li):
        lane = pipe._lanes[li]
        try:
            torch.cuda.set_device(pipe.device)
            with torch.cuda.stream(lane.stream):
                lane.stream.wait_event(ready)
                for i in range(li, n, lanes):
                    E, nu = hyps[i]
                    out[i] = pipe.run_pass(E, nu, warm=None if warm is None else warm[i], backward=backward, _lane=lane)
                e = torch.cuda.Event()
                e.record(lane.stream)
                done.append(e)
        except BaseException as ex:  # surfaced in the caller's thread
            errs.append(ex)

    # persistent worker threads: starting three threads per batch cost 2-9 ms of interpreter-lock hand-offs before the
    # first lane reached its first launch (device idle at every step boundary in the kernel trace)
    pool = _lane_pool(pipe, lanes)
    futures = [pool.submit(work, li) for li in range(min(lanes, n))]
    for f in futures:
        f.result()
    if errs:
        raise errs[0]
    for e in done:
        main.wait_event(e)
    if pipe.ops is None:
        pipe.ops = pipe._lanes[0].ops
    return out


ModalPipeline.run_batch = _run_batch


def _run_steps(pipe, hyps, steps, lanes=2, on_step=None, warm_start=False, backward=True, warm_init=None, material_at=None):
    """``steps`` consecutive passes of every hypothesis WITHOUT a join between the steps: lane ``li`` owns the hypotheses
    ``li, li + lanes, ...`` and runs their passes of step 0, then of step 1, ... back to back on its own stream and thread.
    A hypothesis' step s + 1 depends on nothing but its own step s (the inverse-rendering batch of the reference has one
    optimiser state per material hypothesis, experiments/material_sync_train.py:137-167), so the lanes need not wait for
    each other: joined per step (``run_batch`` in a loop) the 8 lanes of the benchmark finish between 90 and 195 ms of a
    195 ms step - 14 % of the lane time is the tail of the step (tools/lane_tail.py).
    ``on_step(s, results_of_step_s)`` is called on the CALLER's thread as soon as every lane has finished step s (in step
    order; the lanes keep running meanwhile): that is where the per-step scalar all-reduce of the loss goes.
    ``warm_start``: a hypothesis' step s + 1 starts from the block its step s ended with (``warm_init``: optional list of
    start blocks for step 0).  ``material_at(s, i, E, nu) -> (E, nu)``: the material hypothesis i runs step s with (an optimiser's
    move between steps; default: the same (E, nu) every step).  Returns ``out[s][i]``."""
    import threading

    n = len(hyps)
    if steps <= 0 or n == 0:
        return []
    lanes = max(1, min(lanes, n))
    pipe.keep_blocks = bool(warm_start)  # (cold passes do not form the block a warm start would take)
    while len(pipe._lanes) < lanes:
        pipe._lanes.append(_Lane(pipe, ops=pipe.ops if not pipe._lanes else None))
    main = torch.cuda.current_stream(pipe.device)
    ready = torch.cuda.Event()
    ready.record(main)
    out = [[None] * n for _ in range(steps)]
    finished = [0] * steps  # lanes that have completed step s
    cond = threading.Condition()
    errs, done = [], []

    def work(li):
        lane = pipe._lanes[li]
        try:
            torch.cuda.set_device(pipe.device)
            with torch.cuda.stream(lane.stream):
                lane.stream.wait_event(ready)
                warm = {} if warm_init is None else {i: warm_init[i] for i in range(li, n, lanes)}
                for s in range(steps):
                    for i in range(li, n, lanes):
                        if errs:  # another lane failed: stop here instead of running every remaining step first
                            return
                        E, nu = hyps[i] if material_at is None else material_at(s, i, *hyps[i])
                        out[s][i] = pipe.run_pass(E, nu, warm=warm.get(i), backward=backward, _lane=lane)
                        if warm_start:
                            warm[i] = out[s][i][1].block_vectors
                    with cond:
                        finished[s] += 1
                        cond.notify_all()
                e = torch.cuda.Event()
                e.record(lane.stream)
                done.append(e)
        except BaseException as ex:  # surfaced in the caller's thread
            with cond:
                errs.append(ex)
                cond.notify_all()

    pool = _lane_pool(pipe, lanes)
    futures = [pool.submit(work, li) for li in range(lanes)]
    try:
        for s in range(steps):
            with cond:
                while finished[s] < lanes and not errs:
                    cond.wait()
            if errs:
                break
            if on_step is not None:
                on_step(s, out[s])
    finally:
        for f in futures:
            f.result()
        pipe.keep_blocks = True  # (what single passes and batches outside this loop get, as before)
    if errs:
        raise errs[0]
    for e in done:
        main.wait_event(e)
    if pipe.ops is None:
        pipe.ops = pipe._lanes[0].ops
    return out


ModalPipeline.run_steps = _run_steps


def all_reduce_loss(loss_sum, device):
    """Sum of per-rank loss sums.  The data path has no other collective."""
    import torch.distributed as dist

    on = dist.is_available() and dist.is_initialized()
    if on and dist.get_backend() != "nccl":
        device = "cpu"  # gloo rehearsal of the N > 1 path
    t = torch.tensor([loss_sum], dtype=torch.float64, device=device)
    if on:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t[0])


def gather_rank_stats(values, device):
    """Every rank's small vector of step statistics (hypotheses, iteration counts, busy seconds, memory) on every
    rank - one all-gather of a few doubles AFTER the timed region, so that load imbalance between the ranks'
    hypotheses (the only expected scaling loss) shows in the benchmark line.  Returns a list of
    lists, one per rank (a single entry without torch.distributed)."""
    import torch.distributed as dist

    on = dist.is_available() and dist.is_initialized()
    if on and dist.get_backend() != "nccl":
        device = "cpu"
    t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=device)
    if not on:
        return [t.tolist()]
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [o.tolist() for o in out]
