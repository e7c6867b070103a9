"""Device-resident block eigensolver for  K u = lambda M u  (lowest elastic modes of a free body).

This is the engine behind ``lobpcg_func`` / ``DiffSoundObj.eigen_decomposition``.  It is the
"ortho" LOBPCG of Duersch et al. 2018 that the reference's ``src/lobpcg/_lobpcg.py:433-477``
implements, re-designed for MI355X:

  * the search basis  S = [X | P | W]  lives in ONE row-major (n x 3b) fp32 buffer so that the
    stiffness product  K S  is a single BSR-3 block-SpMM launch (the HBM-roofline kernel) and
    the Rayleigh-Ritz matrix  S^T (K S)  is a single tall-skinny MFMA Gram launch with fp64
    accumulation;
  * the six rigid-body modes are deflated analytically (the reference instead asks for k+6
    pairs and drops six, src/utils/utils.py:80-90, and does not converge - SURVEY.md 0.4);
  * the preconditioner is a Chebyshev polynomial in (block-Jacobi)^-1 K, i.e. only more SpMMs;
  * all large operations go through an ``ops`` object (HIP kernels in the product); the small
    (<= 3b x 3b) dense algebra is fp64 ``torch.linalg`` on the same device;
  * a final fp64 Rayleigh-Ritz "polish" on the converged block returns eigenvalues accurate to
    second order in the fp32 iteration error together with the quadratic forms
    u^T K_lambda u, u^T K_mu u needed by the differentiable read-out.

The ``ops`` protocol (see ``diffsound_amd/modal_ops.py`` for the HIP implementation):
  n, device, dtype, rigid (n x 6, M-orthonormal), apply_K, apply_M, gram, mix, residual,
  precond, polish_products.
"""
import os
import threading
from dataclasses import dataclass, field
from typing import Callable, Optional

import torch


@dataclass
class SolverConfig:
    block: int = 0  # search block width b (0 -> k rounded up to a multiple of 8, plus guards)
    guard: int = 8
    # backward-stable criterion of the reference (_lobpcg.py:307-333):
    #   ||K x - lambda M x|| / (||x|| (||K|| + lambda ||M||)) < tol   per wanted pair.
    # fp32 iterates stored in HBM carry rounding noise that K amplifies to ~3 eps32 = 3.5e-7 on this
    # scale, so 2e-6 is ~6x above the floor; the fp64 polish then yields eigenvalues good to ~1e-8.
    tol: float = 0.0  # 0 -> 2e-6 for fp32 iterates, 1e-10 for fp64
    maxit: int = 400
    ortho_passes: int = 2  # at most; a pass is skipped when the previous one left eps * amplification < ortho_tol
    ortho_tol: float = 2e-6
    check_every: int = 1
    lock: bool = True  # hard-lock converged leading columns (reference S_ = S[:, nc:ns])
    seed: int = 0
    cheb_degree: int = 8  # terms of the Chebyshev polynomial preconditioner (1 = plain block-Jacobi)
    cheb_ratio: float = 100.0  # the polynomial targets the interval [lmax/ratio, lmax] of T K
    power_iters: int = 30
    # estimates from the previous material's dominant block (same geometry): stop when two successive estimates agree to
    # ``warm_power_spread`` (at least two steps); spread 0: exactly ``warm_power_iters`` steps
    warm_power_iters: int = 3
    warm_power_spread: float = 0.003
    lmax_safety: float = 1.2
    lmax_cap: float = 0.0  # rigorous bound lambda_max(T K) <= nodes per element (4 / 10); 0 = none
    # Two-level preconditioner (ops with a ``coarse`` level, i.e. ord-2 meshes): symmetric V-cycle with a
    # Chebyshev block-Jacobi smoother on the fine level and a Chebyshev polynomial solve on the corner-node level.
    precond: str = "auto"  # "auto" (two-level when the ops offer a coarse level) | "chebyshev" | "twolevel"
    smooth_degree: int = 3  # terms of the fine smoother (pre: degree-1 SpMMs from a zero guess, post: degree)
    smooth_ratio: float = 10.0  # the smoother damps [lmax/ratio, lmax] of T K
    coarse_degree: int = 24
    coarse_ratio: float = 400.0
    # Rayleigh-Ritz by recurrence: K X and K P of the new basis are the same linear combinations of
    # K [X P W] as X and P themselves, and X^T K X, X^T K P, P^T K P follow from the small Ritz algebra, so
    # an iteration multiplies only the b new columns W by K and forms only the [X P W]^T (K W) block of
    # the Gram matrix (a third of the SpMM columns, half of the Gram flops).  Every ``rr_refresh``-th
    # iteration recomputes K [X P W] and the whole Gram matrix from the vectors (0 = every iteration).
    rr_refresh: int = 8
    # K X' of the new Ritz block by ONE product K X' (b columns, 0.19 ms at the benchmark size) instead of the update
    # [K X' | K P'] = K [X P W] [Z1 Zp] (a 3b -> 2b column mix, 0.37 ms): K P is then never formed - the Gram blocks among X
    # and P come from the small Ritz algebra and only the residual needs K X - and K X' carries no recurrence error
    kx_fresh: bool = True
    # ... and then K X' and M X' feed nothing but the residual: ops that offer ``residual_fused`` form R = K X' - (M X') diag(lam)
    # and its column norms in ONE walk of the neighbour unions - neither product is written, X' is gathered once instead of
    # twice, and the separate residual pass over three blocks is gone (needs kx_fresh)
    fused_residual: bool = True
    # Round 5 - Rayleigh-Ritz on the RAW basis [Y X P W] (needs fused_residual): W stays as the preconditioner left it; K W and
    # M W come out of ONE walk of the unions (ops.apply_KM), [Y X P W]^T [K W | M W] out of ONE Gram launch; [Y X P] is
    # M-orthonormal, so the projected Cholesky-QR transform of W is known in coefficients only, every block of the Ritz matrix
    # follows from those Gram rows and the recurrence's [X P]^T K [X P], and ONE update [X' P'] = [Y X P W] Z_raw writes the new
    # basis.  Per iteration: [K W | M W], Gram, update - instead of M W, Gram, update of W, K W, Gram, update.  An iteration whose
    # W is too ill-conditioned for a single sweep (eps x amplification >= ortho_tol) takes the explicit route.
    raw_rr: bool = True
    # ... and the same for the START block (round 5): its projection against the rigid block, its M-orthonormalisation and its first
    # Ritz step from ONE [K X0 | M X0] walk, ONE Gram launch and ONE update (needs raw_rr's operators; a start block too
    # ill-conditioned for one sweep takes the explicit route)
    raw_start: bool = True
    # storage of the preconditioner's internal blocks (V-cycle iterates, residuals, corner-level vectors): "bf16" halves
    # the bytes of every fused term - the cycle is bound by them - and leaves the outer iteration counts unchanged
    # (fp32 arithmetic in registers; the cycle's input R and output W stay fp32); "fp32" keeps everything in fp32
    precond_storage: str = "bf16"
    native: bool = True  # run the iteration through ds_lobpcg_iterate when possible (False: the Python loop below)
    # Nested iteration (ops with a ``coarse`` level, cold starts only): the random start block is first iterated on the
    # corner-node (P1) level - 14x fewer non-zeros, the same block width - to ``nested_tol``, and its prolongation
    # P X_c starts the fine solve.  The P1 spectrum is ~6 % off the P2 one, so a loose coarse tolerance is enough;
    # the fine solve then needs ~4 iterations fewer.  0 = off.
    # fp64 refinement (BASELINE.json configs[4], "fp64 eigenvalues"): after the fp32 iteration has converged, the block
    # is iterated further with fp64 vectors and fp64 block values - Rayleigh-Ritz on [Y | X | W], W the (fp32) two-level
    # preconditioner applied to the fp64 residual - until the backward error of every wanted pair is below this
    # (1e-10).  0 = off (the fp64 Rayleigh-Ritz polish of the fp32 block is the result).
    refine_tol: float = 0.0
    refine_maxit: int = 40
    refine_refresh: int = 8  # every this many fp64 steps all Gram blocks are recomputed from the vectors (else by recurrence)
    refine_sweeps: int = 2   # preconditioner sweeps per fp64 step (2: W = B R + B (R - K B R); C5: 21 -> 17 steps, 4.3 -> 3.8 s)
    # Round 6: Rayleigh-Ritz steps (host-bound for one hypothesis alone) traded for preconditioner sweeps (device work).
    # ``start_sweeps`` applications of the preconditioner to the RANDOM start block of a nested start's corner-node phase before its
    # first Ritz step (inverse-power steps: the block arrives dominated by the low end of the spectrum; solves without a nested
    # start ignore it - nothing would project their swept block again).
    # ``precond_sweeps`` / ``nested_precond_sweeps`` - W = B R + B (R - K B R) per iteration on the fine / corner-node level: measured
    # and NOT adopted (one iteration less for twice the cycle: profiles/r06_start_sweeps.txt); Python loop only.
    start_sweeps: int = 0
    # ``ritz_tol`` > 0: a pair counts as converged (and is locked) only when, besides its backward error < tol, its Ritz value moved by
    # less than this (relative) in the last step; ``nested_ritz_tol`` is the corner-node phase's.  The backward error is relative to
    # ||K|| + lambda ||M||, ~1e3 x the wanted eigenvalues: a SMOOTH vector passes the corner phase's loose 3e-3 whatever its Rayleigh
    # quotient is.  A random start block never met that case (its error is high-frequency until the wanted pairs have settled); a swept
    # one did - with 32 modes in a block of 40 the first 16 columns were locked at the first test with Ritz values 2 x off (1.1e10
    # for 5.6e9), the corner phase ran to its iteration cap and the fine level took 10 iterations instead of 5
    # (profiles/r06_start_sweeps.txt).  With the settled test the sweeps help at every block width measured there; any value forbids a
    # lock at the FIRST test, which is what went wrong - 0.05 .. 0.4 measure alike, 0.2 keeps the benchmark's block of 80 at the
    # time it had without the test (0.05 locks one step later there: +1 ms on one hypothesis).
    ritz_tol: float = 0.0
    nested_ritz_tol: float = 0.2
    start_sweeps_fp32: bool = False  # (experiment: the sweeps through the fp32 preconditioner kernels instead of the bf16 driver)
    start_sweeps_qr: bool = False    # (experiment: M-orthonormalise the block after every sweep)
    precond_sweeps: int = 1
    nested_precond_sweeps: int = 1
    # Corner-node levels whose operator object runs the GROUP-block Jacobi (HipModalOps.group_jacobi = 8: T = the inverse of the
    # 24 x 24 diagonal block of every 8-node group of the matrix-core tables): degree and interval ratio of that level's polynomial -
    # in the V-cycle and in the nested start's corner phase - in the place of coarse_degree / coarse_ratio and nested_cheb_*.
    # Chebyshev(14, 150) in T_g K follows Chebyshev(22, 350) in the node blocks' T K iteration for iteration
    # (profiles/r06_group_block_jacobi_gpu.txt).
    group_degree: int = 14
    group_ratio: float = 150.0
    # ... and of the ONE-level polynomial of an operator object that runs the group blocks itself (HipModalOps.one_level_group_jacobi,
    # ord-1 meshes), in the place of cheb_degree / cheb_ratio
    cheb_group_degree: int = 16
    cheb_group_ratio: float = 300.0
    nested_tol: float = 0.0
    nested_maxit: int = 8
    nested_cheb_degree: int = 28
    nested_cheb_ratio: float = 550.0


def tuned_config(order, **over):
    """The eigensolver settings the benchmarks measure (bench.py) as the library's suggestion for a tet mesh of this order:
    the rigorous bound lambda_max(T K) <= nodes per element caps the Chebyshev intervals; on ord-2 meshes the two-level
    V-cycle with Chebyshev(22, ratio 350) on the corner-node level, a nested start to 3e-3 whose random block takes two
    preconditioner sweeps before its first Ritz step; on ord-1 meshes the one-level polynomial Chebyshev(24, ratio 600) (round 6,
    the shape loop of bench.py --workload geom at 50k tets / 32 modes: 10 iterations and 13.0 ms per eigendecomposition against 19
    and 17.6 with the library's plain default Chebyshev(8, 100); start sweeps apply to the corner-node phase of a nested start only: ModalSolver.solve).  ``DiffSoundObj`` uses it when the caller gives no
    ``solver_config`` - a script written against the reference (build_model(...); model.eigen_decomposition()) then runs the
    configuration whose numbers DESIGN.md quotes; ``tol`` stays the library default (2e-6) unless overridden."""
    o2 = int(order) == 2
    cfg = SolverConfig(lmax_cap=float({1: 4, 2: 10}.get(int(order), 0)), coarse_degree=22, coarse_ratio=350.0,
                       nested_tol=3e-3 if o2 else 0.0, nested_maxit=8, nested_cheb_degree=22, nested_cheb_ratio=350.0,
                       start_sweeps=2 if o2 else 0, cheb_degree=8 if o2 else 24, cheb_ratio=100.0 if o2 else 600.0)
    for k_, v_ in over.items():
        if not hasattr(cfg, k_):
            raise TypeError(f"tuned_config: SolverConfig has no field {k_!r}")
        setattr(cfg, k_, v_)
    return cfg


@dataclass
class ModalResult:
    eigenvalues: torch.Tensor  # (k,) fp64, ascending
    vectors: torch.Tensor  # (n, k) M-orthonormal
    a_lambda: torch.Tensor  # (k,) u^T K_lambda u   fp64
    b_mu: torch.Tensor  # (k,) u^T K_mu u       fp64
    m_diag: torch.Tensor  # (k,) u^T M u          fp64 (== 1 up to rounding)
    iterations: int = 0
    rerr: Optional[torch.Tensor] = None  # (k,) last relative residuals
    history: list = field(default_factory=list)
    block_vectors: Optional[torch.Tensor] = None  # (n, b) whole converged block (warm start)
    coarse_iterations: int = 0  # iterations of the corner-node phase of a nested start
    refine_iterations: int = 0  # fp64 refinement steps (SolverConfig.refine_tol)
    refine_history: list = field(default_factory=list)


def _sym(G):
    return 0.5 * (G + G.transpose(0, 1))


def _svqb_transform(G, tau=1e-12):
    """T such that (W T)^T M (W T) = I given G = W^T M W  (reference _get_svqb, _lobpcg.py:527-585,
    non-dropping branch: tiny eigenvalues are clamped, not removed)."""
    d = torch.rsqrt(torch.clamp(G.diagonal(), min=1e-300))
    E, Z = torch.linalg.eigh(_sym(G) * d[:, None] * d[None, :])
    E = torch.clamp(E, min=tau * E.abs().max())
    return (d[:, None] * Z) * torch.rsqrt(E)[None, :]


def _orthonormalizer(G):
    """T with (W T)^T M (W T) = I from G = W^T M W: Cholesky-QR on the diagonally scaled Gram matrix
    (one potrf + one small triangular solve), falling back to the clamped-eigenvalue transform of the
    reference's svqb when the factorisation breaks down (rank-deficient block)."""
    return _orthonormalizer_q(G)[0]


def _orthonormalizer_q(G):
    """(T, amp): T as in ``_orthonormalizer``; ``amp`` (python float) estimates by how much the storage rounding
    of W is amplified in the orthogonality of the result, amp = max(1 / min diag(chol) (a lower bound of the
    scaled block's condition), sqrt(removed / kept) per column).  An optional extra last row of G carries the
    squared M-norms of what the preceding projection removed from each column."""
    rem = None
    if G.shape[0] == G.shape[1] + 1:
        G, rem = G[:-1], G[-1]
    G = _sym(G)
    diag = torch.clamp(G.diagonal(), min=1e-300)
    d = torch.rsqrt(diag)
    L, info = torch.linalg.cholesky_ex(G * d[:, None] * d[None, :])
    if int(info) != 0 or not bool(torch.isfinite(L).all()):
        return _svqb_transform(G), float("inf")
    amp = 1.0 / max(float(L.diagonal().min()), 1e-300)
    if rem is not None:
        amp = max(amp, float(torch.sqrt(rem / diag).max()))
    Li = torch.linalg.solve_triangular(L, torch.eye(L.shape[0], dtype=L.dtype, device=L.device), upper=False)
    return d[:, None] * Li.transpose(0, 1), amp


def _orthonormal_columns(Tm):
    """Orthonormal basis (Euclidean, coefficient space) of the columns of the small dense Tm (r x c):
    scaled Cholesky-QR applied twice (one pass leaves an orthogonality error eps*cond^2, which would
    put a 1e-8 floor under the fp64 residuals), with a Householder-QR fallback."""
    Q = Tm
    for _ in range(2):
        G = Q.transpose(0, 1) @ Q
        d = torch.rsqrt(torch.clamp(G.diagonal(), min=1e-300))
        L, info = torch.linalg.cholesky_ex(_sym(G) * d[:, None] * d[None, :])
        if int(info) != 0 or not bool(torch.isfinite(L).all()):
            return torch.linalg.qr(Tm).Q
        Li = torch.linalg.solve_triangular(L, torch.eye(L.shape[0], dtype=L.dtype, device=L.device), upper=False)
        Q = Q @ (d[:, None] * Li.transpose(0, 1))
    return Q


def _thread_local_setters():
    """(mkl_set_num_threads_local, omp_set_num_threads, omp_get_max_threads) of the libraries torch's CPU kernels run on, or None.
    All three act on the CALLING THREAD only - unlike ``torch.set_num_threads``, which also stores a process-wide default and
    re-creates torch's pthreadpool with the new thread count on every call."""
    global _TLS_SETTERS
    if _TLS_SETTERS is False:
        _TLS_SETTERS = None
        try:
            import ctypes

            lib = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libtorch_cpu.so"))
            # (the C interface: the lower-case mkl_set_num_threads_local is the Fortran one and takes a pointer)
            mkl, oset, oget = lib.MKL_Set_Num_Threads_Local, lib.omp_set_num_threads, lib.omp_get_max_threads
            mkl.restype, mkl.argtypes = ctypes.c_int, [ctypes.c_int]
            oset.restype, oset.argtypes = None, [ctypes.c_int]
            oget.restype, oget.argtypes = ctypes.c_int, []
            _TLS_SETTERS = (mkl, oset, oget)
        except (OSError, AttributeError):
            pass
    return _TLS_SETTERS


_TLS_SETTERS = False


class _one_thread:
    """LAPACK on <= 3b x 3b matrices is fastest single-threaded (measured on the MI355X host: 240 x 240
    fp64 eigh 2.5 ms with 1 thread, no faster with 2-8, 150 ms with the default 128 threads; rocSOLVER's
    launch-bound syevd takes 5.7 ms).  Every thread that enters lowers ITS OWN MKL / OpenMP thread count
    (``mkl_set_num_threads_local``, ``omp_set_num_threads``: per-thread settings) and restores its own previous values when it
    leaves; re-entrant; nothing process-wide is touched.
    History: until round 4 only the first thread to enter lowered a count - the reason two identical 8-lane runs could differ in
    the last bits of a gradient (tests/test_fullsize_gpu.py::test_c3_eight_lanes_are_bit_identical_from_run_to_run).  Round 4
    saved ONE value for all threads and restored it from the last thread to leave, which could leave 1 as the process-wide
    default - and, as a side effect, kept the hypothesis lanes' threads at one thread for good; the lanes now
    pin themselves (``pin_thread_to_one_core`` from the lane pool's initializer), this guard only brackets the dense steps."""

    _tls = threading.local()

    def __enter__(self):
        st = _one_thread._tls
        depth = getattr(st, "depth", 0)
        if depth == 0:
            fns = _thread_local_setters()
            if fns is not None:
                mkl, oset, oget = fns
                torch.get_num_threads()  # (a thread's first torch call sizes its OpenMP team from MKL's count: before we lower it)
                st.saved = (mkl(1), oget())  # (MKL_Set_Num_Threads_Local returns the previous local value; 0 = unset)
                oset(1)
            else:  # another BLAS behind torch: its own (heavier) switch, and only when there is something to lower
                st.saved = torch.get_num_threads()
                if st.saved != 1:
                    torch.set_num_threads(1)
        st.depth = depth + 1

    def __exit__(self, *a):
        st = _one_thread._tls
        st.depth -= 1
        if st.depth == 0:
            fns = _thread_local_setters()
            if fns is not None:
                fns[0](st.saved[0])
                fns[1](st.saved[1])
            elif st.saved != 1:
                torch.set_num_threads(st.saved)


def one_blas_thread_for_this_thread():
    """For a private worker thread (a hypothesis lane): one MKL / OpenMP thread for its whole life, thread-local settings only.
    (It limits the thread's BLAS / OpenMP team, not the cores it may run on - which CPUs a rank's threads use is decided once per
    process by diffsound_amd.hostcpu.bind_rank_to_device_numa.  Named ``pin_thread_to_one_core`` until round 6, which it never did.)
    A lane issues launches and solves <= 3b x 3b dense problems; left at the host's full thread count (256 hardware threads on
    the GPU box) any OpenMP region it enters would spin up a team of that size next to the other lanes' teams."""
    fns = _thread_local_setters()
    if fns is not None:
        torch.get_num_threads()  # (sizes the thread's team first, see _one_thread)
        fns[0](1)
        fns[1](1)
    else:
        torch.set_num_threads(1)


pin_thread_to_one_core = one_blas_thread_for_this_thread  # (the old name: kept for callers written against round 5)


def _small(fn, dev, *mats):
    """Run the small dense step ``fn`` on host copies of ``mats`` (fp64) and return its results on ``dev``."""
    if dev.type != "cuda":
        return fn(*mats)
    host = [m.cpu() for m in mats]
    with _one_thread():
        out = fn(*host)
    if out is None:
        return None
    if isinstance(out, tuple):
        return tuple(o.to(dev, non_blocking=True) if torch.is_tensor(o) else o for o in out)
    return out.to(dev, non_blocking=True)


def _raw_basis_transform(GG, Gxp, lam_locked, ny, ncl, nxp, na, ortho_tol, eps):
    """Round 5, Rayleigh-Ritz on the raw basis (host, fp64; the same steps as csrc/lobpcg.cpp).  GG = [Y X P W]^T [K W | M W]
    ((w0 + na) x 2 na, w0 = ny + ncl + nxp columns of the M-orthonormal V = [Y | X_locked | X_active P]).  Returns (G, Q):
    G = S_a^T K S_a for S_a = [X_a P W_o] (W_o = the M-orthonormalised projection of W) and Q = S_a in coordinates of the raw
    basis - or None when a single sweep would not be enough for this W."""
    w0 = ny + ncl + nxp
    pr = w0 + na
    GK, GM = GG[:, :na], GG[:, na:]
    C = GM[:w0]
    G0 = _sym(GM[w0:])
    CtC = C.transpose(0, 1) @ C
    Gp = G0 - CtC
    if bool((Gp.diagonal() <= 1e-9 * G0.diagonal().abs()).any()) or not bool(torch.isfinite(Gp).all()):
        return None
    L, info = torch.linalg.cholesky_ex(Gp)
    if int(info) != 0:
        return None
    T, amp = _orthonormalizer_q(torch.cat([Gp, CtC.diagonal()[None, :]], 0))
    if not (amp < float("inf")) or (ortho_tol > 0.0 and eps * amp >= ortho_tol):
        return None
    CT = C @ T
    GKraw = torch.zeros((pr, pr), dtype=GG.dtype)
    if ncl:
        GKraw[ny:ny + ncl, ny:ny + ncl] = torch.diag(lam_locked)
    GKraw[ny + ncl:w0, ny + ncl:w0] = Gxp
    GKraw[:w0, w0:] = GK[:w0]
    GKraw[w0:, :w0] = GK[:w0].transpose(0, 1)
    GKraw[w0:, w0:] = _sym(GK[w0:])
    Q = torch.zeros((pr, nxp + na), dtype=GG.dtype)
    Q[ny + ncl:w0, :nxp] = torch.eye(nxp, dtype=GG.dtype)
    Q[:w0, nxp:] = -CT
    Q[w0:, nxp:] = T
    return Q.transpose(0, 1) @ GKraw @ Q, Q


def _rr_step(GA, na):
    """Rayleigh-Ritz on the active basis: lowest na Ritz pairs and the coefficient block of the new P."""
    E_, Z = torch.linalg.eigh(_sym(GA))
    Z1 = Z[:, :na].contiguous()
    # P spans (I - Z1 Z1^T) E_x: the part of the old active X that left the new Ritz block - the same space
    # as the reference's S Z2 basis((Z[:b, b:])^T) (_lobpcg.py:466), but it needs only the wanted Ritz
    # vectors and a small Cholesky instead of a Householder QR of a (2b x b) matrix.
    Tm = -Z1 @ Z1[:na, :].transpose(0, 1)
    Tm[:na] += torch.eye(na, dtype=GA.dtype, device=GA.device)
    return E_[:na].contiguous(), Z1, _orthonormal_columns(Tm).contiguous()


def _stats(ops, name):
    """The two diagnostic counters ``name`` of an operator object (created on first use).  They live on the OPERATORS - one
    object per hypothesis lane, touched by that lane's thread only - not on a class: round 5 kept them as class attributes that
    eight lane threads incremented without a lock and that two pipelines in one process would have shared."""
    st = getattr(ops, name, None)
    if st is None:
        st = [0, 0]
        try:
            setattr(ops, name, st)
        except AttributeError:
            pass
    return st


class ChebyshevBlockJacobi:
    """W = p(T K) T R with T = inverse 3x3 diagonal blocks of K and p the degree-(d-1) Chebyshev
    polynomial that approximates 1/x on [lmax/ratio, lmax] (Saad, Iterative Methods, Alg. 12.1).
    Costs d-1 block-SpMMs with K per application; symmetric and fixed, as LOBPCG requires."""

    def __init__(self, ops, degree, ratio, power_iters=30, seed=0, safety=1.2, cap=0.0, warm_iters=None, warm_spread=None):
        """``warm_iters`` / ``warm_spread``: SolverConfig.warm_power_iters / warm_power_spread of the solve this preconditioner
        belongs to (None: the class defaults) - per solver, nothing process-wide (round 6: bench.py used to write the class
        attributes, which two pipelines in one process would have fought over)."""
        self.ops = ops
        self.degree = max(1, int(degree))
        # T of this polynomial: the operator object's group-block Jacobi (HipModalOps.group_jacobi, the corner-node level) or its
        # 3 x 3 node blocks.  The native bf16 cycle reads it off the level descriptor; here: the power iteration and the Python path.
        self.group = int(getattr(ops, "group_jacobi", 0) or 0)
        warm_iters = ChebyshevBlockJacobi.warm_power_iters if warm_iters is None else int(warm_iters)
        warm_spread = ChebyshevBlockJacobi.warm_spread if warm_spread is None else float(warm_spread)
        if power_iters <= 0:  # no estimate: the rigorous bound lambda_max(T K) <= nodes per element is the interval's end
            if cap <= 0.0:
                raise ValueError("ChebyshevBlockJacobi: power_iters = 0 needs a rigorous bound (lmax_cap)")
            self.lmax = float(cap)
            self.lmin = self.lmax / float(ratio)
            self._D = self._AD = None
            return
        n, dev, dt = ops.n, ops.device, ops.dtype
        # The dominant vectors of T K barely move when the material changes, so an ops object that already went
        # through a power iteration hands its block over and a few steps re-converge the bound (the
        # eigensolve itself still starts cold; only this spectral bound of the preconditioner is warm).
        x = getattr(ops, "_power_block", None)
        # (warm only on the geometry the block was iterated on: new coordinates take the full count again)
        pkey = getattr(ops, "norm_probe_key", None)
        pkey = None if pkey is None else pkey()
        warm = x is not None and x.shape == (n, 8) and x.dtype == dt and getattr(ops, "_power_block_key", None) == pkey
        if not warm:
            g = torch.Generator(device=dev).manual_seed(seed + 17)  # device-side RNG: no 100 MB host round trip
            x = torch.randn((n, 8), generator=g, dtype=torch.float32, device=dev).to(dt)
        y = torch.empty_like(x)
        z = torch.empty_like(x)
        lm = prev = None
        for i in range(power_iters):  # largest eigenvalue of T K by block power iteration
            ops.apply_K(x, y)
            if self.group:
                x = ops.group_T(y)
            else:
                ops.cheb_init(y, z, x, 1.0)  # x = T y
            nrm = torch.linalg.vector_norm(x.double(), dim=0)
            lm = nrm.max()
            x = (x / nrm.to(dt)[None, :]).contiguous()
            # A warm block stops as soon as the estimate has stopped moving: SUCCESSIVE estimates (the growth factor of step i
            # against step i - 1) agree to ``warm_spread``, after at least two steps and at least ``warm_iters`` unless they
            # agree earlier.  Round 5 took the agreement AMONG the 8 columns after one step as the sign of convergence - but the
            # columns, never orthogonalised, have all collapsed onto the previous material's dominant vector, and vectors that are
            # equal agree under ANY operator: a jump nu 0.45 -> 0.12 left the one-step estimate at 0.878 of
            # lambda_max, 1.05 x under the safety factor.  The growth factor of a block that the new material has left behind
            # keeps rising from step to step; it is compared with itself.  (The host reads one number per step.)
            if warm and warm_spread > 0.0:
                if i == 0:
                    first = lm  # (not read yet: the first two estimates travel to the host together - one wait instead of two)
                    continue
                if i == 1:
                    prev, cur = torch.stack((first, lm)).tolist()
                else:
                    cur = float(lm)
                if abs(cur - prev) < warm_spread * cur:
                    lm = cur
                    break
                prev = cur
            elif warm and i + 1 >= max(1, warm_iters):
                break
        if warm:  # (diagnostic counters of THIS operator object, read by bench.py: estimates from a warm block, steps they took)
            st = _stats(ops, "warm_stats")
            st[0] += 1
            st[1] += i + 1
        try:
            ops._power_block, ops._power_block_key = x, pkey
        except AttributeError:
            pass
        # power iteration under-estimates; an under-estimated lmax makes the polynomial blow up on
        # the top of the spectrum (measured: 2% low -> no convergence), an over-estimate costs little
        self.lmax = safety * float(lm)
        if cap > 0.0:
            self.lmax = min(self.lmax, cap)
        self.lmin = self.lmax / float(ratio)
        self._D = None
        self._AD = None

    _CHUNK = 80  # columns per fused launch (the fused kernel takes <= 84)
    # power iterations when the ops hand over the block of an earlier estimate (another material on the same mesh):
    # lambda_max(T K) depends on the Poisson ratio only, and mildly (3.1 ... 3.6 over nu = 0.12 ... 0.38 on the
    # benchmark mesh), the dominant vectors hardly at all
    warm_power_iters = 3  # steps of a warm estimate when ``warm_spread`` is 0 (7, 4, 2 and 1 give the same outer iteration counts on the benchmark)
    warm_spread = 0.003   # a warm estimate stops when two successive estimates agree to this (at least two steps); 0: warm_power_iters steps

    def apply(self, R, W, from_guess=False):
        """W <- p(T K) T R (R may be destroyed).  ``from_guess``: W holds an initial guess W_0 and the same
        number of terms of the Chebyshev ITERATION for K W = R is run from it (W <- W_0 + p(T K) T (R - K W_0));
        this is the post-smoother of the two-level cycle and needs the fused-term op."""
        ops = self.ops
        if self.group:
            return self._apply_group(R, W, from_guess)
        if hasattr(ops, "cheb_spmm") and (self.degree > 1 or from_guess):
            for c0 in range(0, R.shape[1], self._CHUNK):  # columns are independent: wide blocks go in chunks
                c1 = min(R.shape[1], c0 + self._CHUNK)
                self._fused(R[:, c0:c1], W[:, c0:c1], from_guess)
            return
        if from_guess:
            raise RuntimeError("Chebyshev iteration from an initial guess needs ops.cheb_spmm")
        theta = 0.5 * (self.lmax + self.lmin)
        delta = 0.5 * (self.lmax - self.lmin)
        D, AD = self._buffers(R)
        ops.cheb_init(R, D, W, 1.0 / theta)  # D = T R / theta ; W = D
        sigma1 = theta / delta
        rho = 1.0 / sigma1
        for _ in range(self.degree - 1):
            ops.apply_K(D, AD)
            rho_new = 1.0 / (2.0 * sigma1 - rho)
            ops.cheb_step(AD, R, D, W, rho_new * rho, 2.0 * rho_new / delta)  # R-=AD; D=c1 D+c2 T R; W+=D
            rho = rho_new

    def _apply_group(self, R, W, from_guess):
        """The polynomial in T_g K in torch on the level's fp32 product - the Python path of the group-block Jacobi (a tracker
        callback, fp32 storage, the tests' comparisons); the product path is the native bf16 cycle."""
        if from_guess:
            raise RuntimeError("the group-block Jacobi serves the polynomial from a zero guess (the corner-node level)")
        ops = self.ops
        theta, delta = 0.5 * (self.lmax + self.lmin), 0.5 * (self.lmax - self.lmin)
        z = ops.group_T(R.contiguous())
        x = z / theta
        d = x.clone()
        sigma1 = theta / delta
        rho = 1.0 / sigma1
        kx = torch.empty_like(x)
        for _ in range(self.degree - 1):
            ops.apply_K(x, kx)
            rho_new = 1.0 / (2.0 * sigma1 - rho)
            d = (rho_new * rho) * d + (2.0 * rho_new / delta) * (z - ops.group_T(kx))
            x = x + d
            rho = rho_new
        W.copy_(x)

    def _buffers(self, R):
        if self._D is None or self._D.shape != R.shape:
            self._D = torch.empty(R.shape, dtype=R.dtype, device=R.device)
            self._AD = torch.empty(R.shape, dtype=R.dtype, device=R.device)
        return self._D, self._AD

    def _fused(self, R, W, from_guess):
        # three-term form W_{k+1} = W_k + c1 (W_k - W_{k-1}) + c2 T (R0 - K W_k): one fused launch per term,
        # ping-ponging between W and a scratch block; R is only read
        ops = self.ops
        theta = 0.5 * (self.lmax + self.lmin)
        delta = 0.5 * (self.lmax - self.lmin)
        D, AD = self._buffers(R)
        sigma1 = theta / delta
        rho = 1.0 / sigma1
        if from_guess:
            terms = self.degree
            cur, oth = W, D
        else:
            terms = self.degree - 1
            cur, oth = (W, D) if terms % 2 == 0 else (D, W)  # so that the last term lands in W
            ops.cheb_init(R, AD, cur, 1.0 / theta)  # W_1 = T R0 / theta  (W_0 = 0)
        for k in range(terms):
            if from_guess and k == 0:
                ops.cheb_spmm(cur, oth, R, 0.0, 1.0 / theta, first=True)  # W_1 = W_0 + T (R0 - K W_0) / theta
            else:
                rho_new = 1.0 / (2.0 * sigma1 - rho)
                ops.cheb_spmm(cur, oth, R, rho_new * rho, 2.0 * rho_new / delta, first=(k == 0))
                rho = rho_new
            cur, oth = oth, cur
        if cur is not W:
            W.copy_(cur)


class TwoLevelChebyshev:
    """Symmetric two-level V-cycle for ord-2 meshes:
         W1 = S R ;  W2 = W1 + P C P^T (R - K W1) ;  W = W2 + S (R - K W2)
    S: Chebyshev block-Jacobi smoother on the fine level (damps [lmax/smooth_ratio, lmax]); P: embedding of the
    corner-node P1 space; C: Chebyshev block-Jacobi polynomial for the P1 operator P^T K P (an ord-1 assembly of
    the corner sub-mesh, ~14x fewer non-zeros).  The smooth modes that the one-level polynomial needs a high
    degree for are handled on the cheap level: ~6 fine SpMMs per application instead of ~48 at the same
    outer iteration count.  Fixed, symmetric positive definite, as LOBPCG requires."""

    use_native = True  # run the cycle through ops.twolevel_apply (one native call) when the ops offer it

    def __init__(self, ops, cfg):
        self.ops = ops
        self.storage = cfg.precond_storage
        self._buf16 = None
        args = (cfg.power_iters, cfg.seed, cfg.lmax_safety)
        warm = dict(warm_iters=getattr(cfg, "warm_power_iters", None), warm_spread=getattr(cfg, "warm_power_spread", None))
        self.smooth = ChebyshevBlockJacobi(ops, cfg.smooth_degree, cfg.smooth_ratio, *args, cap=cfg.lmax_cap, **warm)
        grp = bool(getattr(ops.coarse, "group_jacobi", 0))
        self.coarse = ChebyshevBlockJacobi(ops.coarse, cfg.group_degree if grp else cfg.coarse_degree,
                                           cfg.group_ratio if grp else cfg.coarse_ratio, *args,
                                           cap=min(cfg.lmax_cap, 4.0) if cfg.lmax_cap > 0 else 0.0, **warm)
        self.lmax = self.smooth.lmax
        self._buf = None

    def apply(self, R, W):
        ops = self.ops
        nc = ops.coarse.n
        ch = ChebyshevBlockJacobi._CHUNK
        for c0 in range(0, R.shape[1], ch):
            Rs, Ws = R[:, c0:c0 + ch], W[:, c0:c0 + ch]
            w = Rs.shape[1]
            if self._buf is None or self._buf[0].shape[1] != w:
                mk = lambda rows: torch.empty((rows, w), dtype=R.dtype, device=R.device)
                self._buf = (mk(R.shape[0]), mk(nc), mk(nc), mk(R.shape[0]))
            Rr, Rc, Ec, Wc = self._buf
            native = getattr(ops, "twolevel_apply", None)
            form = getattr(self, "cycle_form", "symmetric")
            if form != "symmetric":  # (experiment, Python path only: tools/experiments/cycle_forms.py)
                if form == "post":        # W = C' R, then post-smoothing from that guess
                    ops.restrict(Rs, Rc)
                    self.coarse.apply(Rc, Ec)
                    Ws.zero_()
                    ops.prolong_add(Ec, Ws)
                    self.smooth.apply(Rs, Ws, from_guess=True)
                elif form == "pre":       # pre-smoothing, then the corner-level correction of its residual
                    Rr.copy_(Rs)
                    self.smooth.apply(Rr, Ws)
                    ops.spmm_residual(Ws, Rs, Rr)
                    ops.restrict(Rr, Rc)
                    self.coarse.apply(Rc, Ec)
                    ops.prolong_add(Ec, Ws)
                elif form == "additive":  # W = S R + P C P^T R
                    ops.restrict(Rs, Rc)
                    self.coarse.apply(Rc, Ec)
                    Rr.copy_(Rs)
                    self.smooth.apply(Rr, Ws)
                    ops.prolong_add(Ec, Ws)
                else:
                    raise ValueError(form)
                continue
            if native is not None and self.use_native and self.storage == "bf16" and R.is_cuda:
                if self._buf16 is None or self._buf16[0].shape[2] != w:
                    mk = lambda cnt, rows: torch.empty((cnt, rows, w), dtype=torch.bfloat16, device=R.device)
                    self._buf16 = (mk(5, R.shape[0]), mk(4, nc))
                f16, c16 = self._buf16
                if native((self.smooth.degree, self.smooth.lmax, self.smooth.lmin),
                          (self.coarse.degree, self.coarse.lmax, self.coarse.lmin), Rs, Ws, f16[1], f16[2], f16[3], c16[0],
                          c16[1], c16[2], c16[3], f16[0], R16=f16[4]):
                    continue
            if native is not None and self.use_native and not self.coarse.group:  # (the fp32 cycle knows the node blocks only)
                D, AD = self.smooth._buffers(Rs)
                Dc, ADc = self.coarse._buffers(Rc)
                if native((self.smooth.degree, self.smooth.lmax, self.smooth.lmin),
                          (self.coarse.degree, self.coarse.lmax, self.coarse.lmin), Rs, Ws, D, AD, Rr, Rc, Ec, Dc, ADc,
                          Wc):
                    continue
            self.smooth.apply(Rs, Ws)
            ops.spmm_residual(Ws, Rs, Rr)
            ops.restrict(Rr, Rc)
            self.coarse.apply(Rc, Ec)
            ops.prolong_add(Ec, Ws)
            self.smooth.apply(Rs, Ws, from_guess=True)


class SolverState:
    """What a ``tracker`` callback sees after every iteration - the same fields the reference's worker
    exposes (src/lobpcg/_lobpcg.py:246-256, 335-342): ``ivars['istep']``, ``ivars['converged_count']``,
    ``tvars['rerr']``, ``E``, ``X`` and the writable ``bvars['force_stop']``."""

    def __init__(self, iparams, fparams, bparams):
        self.iparams, self.fparams, self.bparams = iparams, fparams, bparams
        self.ivars = {"istep": 0, "converged_count": 0, "iterations_left": iparams.get("niter", 0)}
        self.fvars = {}
        self.bvars = {"force_stop": False}
        self.tvars = {}
        self.E = None
        self.X = None


class ModalSolver:
    # ``ModalResult.block_vectors`` - the whole converged block, rotated to its Ritz basis: what a warm start of the next solve takes.
    # A caller that starts every solve cold switches it off and saves one (n x b) update per solve.
    keep_block = True

    def __init__(self, ops, cfg: Optional[SolverConfig] = None, precond=None, precond_object=None):
        """precond: optional callable (R, W) -> None writing the preconditioned residual into W
        (the ``iK`` argument of the reference API); default Chebyshev block-Jacobi.  precond_object: an already built
        ChebyshevBlockJacobi / TwoLevelChebyshev on these ops (its power iteration is then not repeated)."""
        self.ops = ops
        self.cfg = cfg or SolverConfig()
        self.ortho_log = []
        if precond_object is not None:
            self.precond = precond_object
            self.precond_apply = precond_object.apply
        elif precond is not None:
            self.precond_apply = precond
            self.precond = None
        else:
            two = self.cfg.precond == "twolevel" or (self.cfg.precond == "auto" and getattr(ops, "coarse", None) is not None)
            if two and getattr(ops, "coarse", None) is None:
                raise ValueError("precond='twolevel' needs ops with a coarse level (an ord-2 mesh)")
            if two:
                self.precond = TwoLevelChebyshev(ops, self.cfg)
            else:
                grp = bool(getattr(ops, "group_jacobi", 0))
                self.precond = ChebyshevBlockJacobi(ops, self.cfg.cheb_group_degree if grp else self.cfg.cheb_degree,
                                                    self.cfg.cheb_group_ratio if grp else self.cfg.cheb_ratio,
                                                    self.cfg.power_iters, self.cfg.seed, self.cfg.lmax_safety,
                                                    self.cfg.lmax_cap, warm_iters=self.cfg.warm_power_iters,
                                                    warm_spread=self.cfg.warm_power_spread)
            self.precond_apply = self.precond.apply

    # ------------------------------------------------------------------ helpers
    def _orthonormalize(self, W, V, MW, VW=None):
        """Make W M-orthogonal to the block V (may be None) and M-orthonormal (reference _get_ortho,
        _lobpcg.py:587-679, with a fixed number of passes instead of host-synchronising norms).

        VW: the contiguous block [V | W] when W directly follows V in memory (it does in the solver's basis
        buffer).  V is M-orthonormal, so ONE product M W and ONE Gram launch [V W]^T (M W) give both the
        projection coefficients C = V^T M W and, as G0 - C^T C, the Gram matrix of the projected block; the
        projection and the Cholesky-QR transform are then one update W <- [V W] [-C T; T].  The closed form is used
        for the first sweep only and abandoned when a column turns out to lie (numerically) in span(V); the repair
        sweep, when one is needed, is the explicit project / re-multiply / Cholesky-QR sequence."""
        ops, cfg = self.ops, self.cfg
        eps = 6e-8 if ops.dtype == torch.float32 else 1.1e-16
        nv_ = 0 if V is None else V.shape[1]
        for ip in range(cfg.ortho_passes):
            ops.apply_M(W, MW)
            done = False
            if nv_ > 0 and VW is not None and ip == 0:
                G = ops.gram(VW, MW)  # rows :nv_ = V^T M W, rows nv_: = W^T M W

                def transform(G_, nv=nv_):
                    C = G_[:nv]
                    CtC = C.transpose(0, 1) @ C
                    G0 = _sym(G_[nv:])
                    Gp = G0 - CtC
                    # a column (numerically) inside span(V) leaves nothing of itself in G0 - C^T C but cancellation
                    # noise: the closed form has broken down, take the explicit route for this sweep
                    if bool((Gp.diagonal() <= 1e-9 * G0.diagonal().abs()).any()) or not bool(torch.isfinite(Gp).all()):
                        return None, float("inf")
                    L, info = torch.linalg.cholesky_ex(Gp)
                    if int(info) != 0:
                        return None, float("inf")
                    T, amp = _orthonormalizer_q(torch.cat([Gp, CtC.diagonal()[None, :]], 0))
                    return torch.cat([-(C @ T), T], 0).contiguous(), amp

                coef, amp = _small(transform, ops.device, G)
                if coef is not None:
                    ops.mix(VW, coef, W)  # in place: W is the trailing column range of VW (ds_mix allows it)
                    done = True
            if not done:
                if nv_ > 0:
                    C = ops.gram(V, MW)
                    ops.mix(V, C, W, alpha=-1.0, beta=1.0)
                    rem = (C * C).sum(0)  # ||V C_j||_M^2 (V is M-orthonormal): what the projection removed
                    ops.apply_M(W, MW)
                G = ops.gram(W, MW, symmetric=True)
                if nv_ > 0:
                    G = torch.cat([G, rem[None, :].to(G.dtype)], 0)
                T, amp = _small(_orthonormalizer_q, ops.device, G)
                ops.mix_inplace(W, T)
            self.ortho_log.append(amp)
            # a further pass only repairs what this one lost to rounding: eps * amp in the orthogonality of W
            if cfg.ortho_tol > 0.0 and eps * amp < cfg.ortho_tol:
                break

    # ------------------------------------------------------------------ main entry
    def _nested_start(self, k, b, out=None):
        """Start block of the fine solve from a short solve on the corner-node level (see SolverConfig.nested_tol)."""
        ops, cfg = self.ops, self.cfg
        co = ops.coarse
        if co.rigid is None:
            co.rigid = co._rigid_basis()
        grp = bool(getattr(co, "group_jacobi", 0))
        ccfg = SolverConfig(block=b, guard=cfg.guard, tol=cfg.nested_tol, maxit=cfg.nested_maxit, seed=cfg.seed,
                            cheb_degree=cfg.group_degree if grp else cfg.nested_cheb_degree,
                            cheb_ratio=cfg.group_ratio if grp else cfg.nested_cheb_ratio,
                            cheb_group_degree=cfg.group_degree, cheb_group_ratio=cfg.group_ratio,
                            power_iters=cfg.power_iters, lmax_safety=cfg.lmax_safety,
                            lmax_cap=min(cfg.lmax_cap, 4.0) if cfg.lmax_cap > 0 else 0.0, precond="chebyshev",
                            raw_rr=cfg.raw_rr, raw_start=cfg.raw_start, warm_power_iters=cfg.warm_power_iters,
                            warm_power_spread=cfg.warm_power_spread, start_sweeps=cfg.start_sweeps,
                            start_sweeps_fp32=cfg.start_sweeps_fp32, start_sweeps_qr=cfg.start_sweeps_qr,
                            ritz_tol=cfg.nested_ritz_tol,
                            precond_sweeps=getattr(cfg, "nested_precond_sweeps", 1), native=cfg.native)
        pre = self.precond.coarse if isinstance(self.precond, TwoLevelChebyshev) else None
        if pre is not None and (pre.degree != ccfg.cheb_degree
                                or abs(pre.lmax / pre.lmin - ccfg.cheb_ratio) > 1e-6 * ccfg.cheb_ratio):
            pre = None  # the V-cycle's corner-level polynomial is reused when it is the one asked for
        cs = ModalSolver(co, ccfg, precond_object=pre)
        rc = cs.solve(k, polish=False)
        self.nested_iterations = rc.iterations
        if out is not None and hasattr(ops, "prolong") and out.dtype == ops.dtype:
            ops.prolong(rc.block_vectors, out)
            return out
        X0 = torch.zeros((ops.n, b), dtype=ops.dtype, device=ops.device)
        ops.prolong_add(rc.block_vectors, X0)
        return X0

    def solve(self, k: int, X0: Optional[torch.Tensor] = None, tracker: Optional[Callable] = None,
              state: Optional[SolverState] = None, polish: bool = True) -> ModalResult:
        ops, cfg = self.ops, self.cfg
        n, dev, dt = ops.n, ops.device, ops.dtype
        b = cfg.block or ((k + cfg.guard + 7) // 8) * 8
        if X0 is not None and X0.shape[1] > b:
            b = ((X0.shape[1] + 3) // 4) * 4
        self.nested_iterations = 0
        nested = (X0 is None and cfg.nested_tol > 0.0 and getattr(ops, "coarse", None) is not None
                  and hasattr(ops, "prolong_add") and ops.coarse.n >= 3 * b + 6)
        Y = ops.rigid
        nrigid = 0 if Y is None else 6
        if n < 3 * b + nrigid:
            raise ValueError(
                "LPBPCG algorithm is not applicable when the number of A rows (={})"
                " is smaller than 3 x the number of requested eigenpairs (={})".format(n, b))
        state = state or SolverState({"niter": cfg.maxit, "k": k, "n": b, "m": n}, {}, {})
        # One row-major buffer [Y | X | P | W]: the rigid basis rides in front of the search basis so the
        # projection against [Y, X, P] is ONE Gram + ONE update launch; the active basis S[:, ny:] is what
        # the stiffness SpMM and the Rayleigh-Ritz Gram see.
        ny = 0 if Y is None else Y.shape[1]
        # Round 5: on the device the rigid block takes 16 columns (its 6 vectors + zero columns) instead of 8, so that X, P and W
        # - 80-column blocks in the benchmark - start at byte offsets 64, 384 and 704 of a 1 KiB row: every 320-byte row piece
        # the neighbour-union products gather is then five whole 64-byte sectors (with 8 columns in front they started 32 bytes
        # into a sector and touched six: K W drew 1.31 x its algorithmic bytes from memory on these operands against 1.17 x
        # on compact blocks, profiles/r04_spmm_pmc_kx.json).  The zero columns cost the Gram / update kernels 3 % more columns.
        if ny and ny % 16 and dev.type == "cuda" and dt == torch.float32 and b % 16 == 0:
            ny = -(-ny // 16) * 16

        def wide(cols):
            """(n x cols) block inside a buffer whose rows are a multiple of 1 KiB apart (fp32 on the device): every 3-row
            panel of a column range then starts at the same offset inside a cache line - the neighbour-union products gather
            such panels, and on the benchmark mesh K X takes 186 us on an 80-column range of a 256-column buffer against 196 us
            with 248 columns (M X 152 against 164; profiles/r04_mb_kx_strided.txt)."""
            ld = cols if (dev.type != "cuda" or dt != torch.float32) else -(-cols // 256) * 256
            return torch.empty((n, ld), dtype=dt, device=dev)[:, :cols]

        S = wide(ny + 3 * b)
        S2 = wide(ny + 3 * b)
        if ny:
            for buf in (S, S2):
                buf[:, :Y.shape[1]].copy_(Y)
                if ny > Y.shape[1]:
                    buf[:, Y.shape[1]:ny].zero_()
        KS = wide(3 * b)
        R = torch.empty((n, b), dtype=dt, device=dev)
        MX = torch.empty((n, b), dtype=dt, device=dev)
        MW = torch.empty((n, b), dtype=dt, device=dev)

        X = S[:, ny:ny + b]
        g = torch.Generator(device=dev).manual_seed(cfg.seed)
        if nested:  # (the prolongated corner-level block goes straight into the basis buffer)
            X0 = self._nested_start(k, b, out=X)
        nx0 = 0 if X0 is None else X0.shape[1]
        if nx0 and X0 is not X:
            X[:, :nx0].copy_(X0.to(dt))
        if nx0 < b:
            X[:, nx0:].copy_(torch.randn((n, b - nx0), generator=g, dtype=torch.float32, device=dev).to(dt))
        # ``start_sweeps`` (round 6): the RANDOM start block is passed through the preconditioner before its first Ritz step - steps of
        # a preconditioned inverse subspace iteration without the Ritz algebra.  White noise holds every frequency alike; after two
        # sweeps the block is dominated by the low end of the spectrum and the corner-node level of a nested start reaches its
        # tolerance in 3 iterations instead of 5 (profiles/r06_start_sweeps.txt) - two sweeps are 0.7 ms of device work, two
        # iterations 1 ms of device work plus 3.2 ms of Rayleigh-Ritz on the host thread.
        # ONLY in the corner-node phase of a nested start (``polish`` False): the fine level projects and orthonormalises that phase's
        # result again.  A solve that nothing follows keeps its plain random start - on a small problem the swept block collapses onto
        # a few low modes, what is left of its other columns is rounding noise with rigid-body remnants in it, and the start block's
        # normalisation scales that up (two spurious low "eigenvalues" on a 4^3 ord-1 cube when this ran on one-level solves).
        for _ in range(cfg.start_sweeps if (X0 is None and not polish) else 0):
            R.copy_(X)
            native_sweep = getattr(ops, "chebyshev_apply16", None)
            if not (native_sweep is not None and isinstance(self.precond, ChebyshevBlockJacobi) and cfg.precond_storage == "bf16"
                    and dt == torch.float32 and not getattr(cfg, "start_sweeps_fp32", False) and native_sweep(self.precond, R, X)):
                self.precond_apply(R, X)
            # (every sweep scales the block by ~1 / ||K||: 1e-10 on the benchmark's stiffness - three of them would leave the range
            # the preconditioner's bf16 blocks can hold; back to unit size after each)
            X.mul_(1.0 / X.abs().max().clamp(min=1e-30))
            # ... and the rigid modes leave after each: the preconditioner approximates K^-1, so the null vectors of K are amplified by
            # p(0) - on a small mesh, whose lowest elastic modes lie INSIDE the polynomial's interval, ~30 x more per sweep than
            # anything wanted; two sweeps later the block is rigid motion with the elastic content in its last fp32 digits, and the
            # start block's projection in coefficients cannot recover it (six ~zero "eigenvalues" came back on a 4^3 ord-1 cube:
            # tests/test_api_gpu.py::test_shape_loop_on_one_object_matches_fresh_objects).  X <- X - Y (Y^T M X), Y M-orthonormal.
            if Y is not None:
                ops.apply_M(X, MW)
                ops.mix(Y, ops.gram(Y, MW), X, alpha=-1.0, beta=1.0)
            if getattr(cfg, "start_sweeps_qr", False):  # (experiment: M-orthonormalise between the sweeps - a true block inverse iteration)
                self._orthonormalize(X, S[:, :ny] if ny else None, MW, VW=S[:, :ny + b] if ny else None)
        # operator norm estimates with a random block, as the reference does (_lobpcg.py:280-285)
        # (The probe block is the same every time - same seed, same number of columns drawn before it - and ||M G0|| depends on
        # the geometry only: operators that can name their geometry's generation keep the block, its norm and ||M G0|| / ||G0||
        # from one solve to the next; a pass then multiplies the block by K alone.  Same numbers, bit for bit.)
        pkey = getattr(ops, "norm_probe_key", None)
        pkey = None if pkey is None else (pkey(), cfg.seed, b - nx0, n, str(dt))
        kept = getattr(ops, "_norm_probe", None)
        terms = None
        if pkey is not None and kept is not None and kept[0] == pkey:
            _, G0, gn, B_norm, terms = kept
        else:
            G0 = torch.randn((n, 8), generator=g, dtype=torch.float32, device=dev).to(dt)
            gn = torch.linalg.vector_norm(G0.double())
            # operators of the form K = sum c_i K_i with geometry-only terms (the linear material: lam K_lambda + mu K_mu) hand over
            # K_i G0 and M G0 of ONE walk; ||K G0|| of every material on this geometry is then a small vector operation
            prods = ops.probe_products(G0) if pkey is not None and hasattr(ops, "probe_products") else None
            if prods is not None:
                terms = (prods[0], prods[1])
                B_norm = torch.linalg.vector_norm(prods[2]) / gn
            else:
                G1 = torch.empty_like(G0)
                ops.apply_M(G0, G1)
                B_norm = torch.linalg.vector_norm(G1.double()) / gn
            if pkey is not None:
                ops._norm_probe = (pkey, G0, gn, B_norm, terms)
        if terms is not None:
            cl, cm = ops.lame
            A_norm = torch.linalg.vector_norm(torch.add(terms[0] * float(cl), terms[1], alpha=float(cm))) / gn
        else:
            G1 = torch.empty_like(G0)
            ops.apply_K(G0, G1)
            A_norm = torch.linalg.vector_norm(G1.double()) / gn
        state.fvars.update(A_norm=float(A_norm), B_norm=float(B_norm))
        tol = cfg.tol or (2e-6 if dt == torch.float32 else 1e-10)
        KS2 = wide(3 * b)
        lam = None
        if (cfg.raw_rr and cfg.raw_start and ny and getattr(ops, "apply_KM_ok", None) is not None and b % 4 == 0
                and ops.apply_KM_ok(X, KS[:, :b], KS[:, b:2 * b])):
            # The start block's projection against Y, its M-orthonormalisation and its first Ritz step IN COEFFICIENTS (round 5,
            # the raw-basis idea of the iteration applied to the start): K X0 and M X0 in ONE walk, [Y X0]^T [K X0 | M X0] in ONE
            # Gram launch, then on the host C = Y^T M X0, B = X0^T M X0 - C^T C, the Cholesky-QR transform T of B, the Ritz pairs
            # of T^T (X0^T K X0 - C^T (Y^T K X0) - (Y^T K X0)^T C) T, and ONE update X = [Y X0] [-C T Z; T Z] (K X = (K X0) T Z:
            # K Y = 0).  Before: M X0, a Gram, an update, K X, a Gram, two updates - and twice the first three when the block
            # was far from orthonormal.  A block too ill-conditioned for one sweep takes that explicit route as before.
            ops.apply_KM(X, KS[:, :b], KS[:, b:2 * b])
            eps_ = 6e-8 if dt == torch.float32 else 1.1e-16

            def start(G_, ny_=ny, b_=b):
                Gyk, Cy = G_[:ny_, :b_], G_[:ny_, b_:]
                A, B0 = _sym(G_[ny_:, :b_]), _sym(G_[ny_:, b_:])
                CtC = Cy.transpose(0, 1) @ Cy
                Bp = B0 - CtC
                if bool((Bp.diagonal() <= 1e-9 * B0.diagonal().abs()).any()) or not bool(torch.isfinite(Bp).all()):
                    return None
                T, amp = _orthonormalizer_q(torch.cat([Bp, CtC.diagonal()[None, :]], 0))
                if not (cfg.ortho_tol > 0.0 and eps_ * amp < cfg.ortho_tol):
                    return None  # (one sweep would leave eps * amp in the block's orthogonality: the explicit route repairs it)
                A1 = A - Cy.transpose(0, 1) @ Gyk - Gyk.transpose(0, 1) @ Cy
                E_, Z_ = torch.linalg.eigh(_sym(T.transpose(0, 1) @ A1 @ T))
                Cx = T @ Z_
                return E_, torch.cat([-(Cy @ Cx), Cx], 0).contiguous(), Cx.contiguous(), amp

            Gs = ops.gram(S[:, :ny + b], KS[:, :2 * b])
            if dev.type == "cuda" and cfg.native and dt == torch.float32:
                # (the same algebra on the host thread in ONE native call with the solver loop's LAPACK table - ds_host_start_block -
                # instead of ~30 torch calls on 80 x 80 CPU tensors: 0.65 -> 0.3 ms per start block, two per pass; round 6)
                from .. import _hip

                got = _hip.host_start_block(Gs.cpu(), ny, b, cfg.ortho_tol, eps_)
                if got is not None:
                    got = (got[0].to(dev), got[1].to(dev), got[2].to(dev), got[3])
            else:
                got = _small(start, dev, Gs)
            _stats(ops, "raw_start_stats")[0 if got is not None else 1] += 1  # (diagnostic counters: taken, handed to the explicit route)
            if got is not None:
                lam, coef, Cx, amp = got
                lam = lam.clone()
                self.ortho_log.append(amp)
                ops.mix(S[:, :ny + b], coef, S2[:, ny:ny + b])
                S, S2 = S2, S
                # K X of the new block (K Y = 0) - which nobody reads when the iteration forms its residuals in one walk of the
                # unions (fused_residual with kx_fresh: K X' is formed inside that kernel): the update is skipped then (round 6)
                if not (cfg.fused_residual and cfg.kx_fresh and cfg.rr_refresh > 0 and hasattr(ops, "residual_fused")
                        and ops.residual_fused_ok(S[:, ny:ny + b], R)):
                    ops.mix(KS[:, :b], Cx, KS2[:, :b])
                    KS, KS2 = KS2, KS
                X = S[:, ny:ny + b]
        if lam is None:
            self._orthonormalize(X, S[:, :ny], MW, VW=S[:, :ny + b] if ny else None)
            ops.apply_K(X, KS[:, :b])
            lam, Z = _small(lambda G: torch.linalg.eigh(_sym(G)), dev, ops.gram(X, KS[:, :b], symmetric=True))
            lam = lam.clone()
            ops.mix(X, Z, S2[:, ny:ny + b])
            S, S2 = S2, S
            ops.mix(KS[:, :b], Z, KS2[:, :b])  # K X of the rotated block
            KS, KS2 = KS2, KS
        history = []
        it = 0
        ncl = 0  # locked (converged) leading columns, kept a multiple of 4 for 16-byte aligned slices
        npc = 0  # columns of P
        k0 = 0  # first column of K X_active inside KS (columns locked since the last Ritz step are skipped)
        # Gram blocks of the part of the basis that the last Ritz step produced: [X_a P]^T K [X_a P] (host, fp64)
        Gxp = torch.diag(lam.detach().to(torch.float64).cpu()) if dev.type == "cuda" else torch.diag(lam.to(torch.float64))
        since_refresh = 0
        best_worst = float("inf")
        rel = torch.full((b,), float("inf"), dtype=torch.float64, device=dev)
        lam_prev = None  # Ritz values of the step before (cfg.ritz_tol)
        # The iteration as ONE native call (ds_lobpcg_iterate) when the ops offer it and nothing needs the interpreter
        # between iterations (no tracker callback, the built-in preconditioners): same kernels, same dense steps, but
        # a hypothesis lane then runs its whole solve without the interpreter lock.
        native = None
        if (cfg.native and tracker is None and self.precond is not None and dt == torch.float32
                and hasattr(ops, "native_lobpcg")):
            native = ops.native_lobpcg(self.precond, cfg, k, b, ny, S, S2, KS, KS2, R, MX, MW, lam, float(A_norm),
                                       float(B_norm), tol)
        if native is not None:
            it, in_s2, lam, rel, history = native
            if in_s2:
                S, S2 = S2, S
                KS, KS2 = KS2, KS
            state.ivars.update(istep=it, converged_count=int((rel[:k] < tol).sum()), iterations_left=cfg.maxit - it)
            state.tvars["rerr"] = rel[:k]
            state.E, state.X = lam, S[:, ny:ny + b]
        for it in (range(cfg.maxit + 1) if native is None else ()):
            na = b - ncl
            X = S[:, ny:ny + b]
            Xa = X[:, ncl:]
            fused = (cfg.fused_residual and cfg.kx_fresh and hasattr(ops, "residual_fused")
                     and ops.residual_fused_ok(Xa, R[:, :na]))
            if fused:
                rn2, xn2 = ops.residual_fused(Xa, lam[ncl:], R[:, :na])
            else:
                ops.apply_M(Xa, MX[:, :na])
                # R <- K X - M X lam on the active columns (K X_active sits at column k0 = 0 of KS here: the Ritz step
                # has just rewritten it), with ||R_j||^2 and ||X_j||^2 in fp64
                rn2, xn2 = ops.residual(R[:, :na], MX[:, :na], Xa, lam[ncl:], src=KS[:, k0:k0 + na])
            rel[ncl:] = torch.sqrt(rn2 / xn2) / (A_norm + lam[ncl:].abs() * B_norm)
            relk = rel[:k]
            conv = relk < tol
            if cfg.ritz_tol > 0.0:  # ... and settled: |theta - theta_before| <= ritz_tol |theta|  (never at the start block's own Ritz values)
                conv = (conv & ((lam[:k] - lam_prev[:k]).abs() <= cfg.ritz_tol * lam[:k].abs())) if lam_prev is not None else torch.zeros_like(conv)
            conv = conv.to(torch.int32)
            # leading converged pairs only, to keep strict ordering (reference _lobpcg.py:321-328)
            nconv = int(torch.cumprod(conv, 0).sum())
            history.append((it, float(relk.max())))
            state.ivars.update(istep=it, converged_count=nconv, iterations_left=cfg.maxit - it)
            state.tvars["rerr"] = relk
            state.E, state.X = lam, X
            if tracker is not None:
                tracker(state)
            if nconv >= k or it == cfg.maxit or state.bvars.get("force_stop", False):
                break
            # A tolerance below what the iterates' precision can reach never locks anything; the block then sits converged to
            # rounding while [X P W] degenerates (W and P are noise), the residuals creep up again and, a few iterations later,
            # the block collapses (seen on a random pencil with tol = 1e-6 in fp32: 4e-7 at iteration 22, 3e-4 at 28, garbage at
            # 29).  Stop at the first clear rise above the best residual reached - the block is still good to ~10 x that floor.
            best_worst = min(best_worst, history[-1][1])
            if it > 10 and history[-1][1] > 10.0 * best_worst and best_worst < 1e-3:
                state.bvars["stagnated"] = True
                break
            # hard locking as in the reference (S_ = S[:, nc:ns], _lobpcg.py:458): converged leading columns
            # leave the Rayleigh-Ritz problem, the residual block and the preconditioner; they stay in V
            new_ncl = (nconv // 4) * 4 if cfg.lock else 0
            if new_ncl > ncl:
                shift = new_ncl - ncl
                R[:, :na - shift].copy_(R[:, shift:na].clone())
                Gxp = Gxp[shift:, shift:]
                k0 += shift
                ncl = new_ncl
                na = b - ncl
            w0 = ny + b + npc
            W = S[:, w0:w0 + na]
            self.precond_apply(R[:, :na], W)
            for _ in range(max(0, cfg.precond_sweeps - 1)):  # (experiment: W <- W + B (R - K W))
                ops.apply_K(W, MW[:, :na])
                torch.sub(R[:, :na], MW[:, :na], out=MX[:, :na])
                self.precond_apply(MX[:, :na], MW[:, :na])
                W += MW[:, :na]
            sz = na + npc + na
            Sa = S[:, ny + ncl:ny + ncl + sz]
            KSa = KS[:, k0:k0 + sz]
            full = cfg.rr_refresh <= 0 or since_refresh >= cfg.rr_refresh
            rawQ = None  # Rayleigh-Ritz on the raw basis (SolverConfig.raw_rr): (G, Q) of _raw_basis_transform
            if (cfg.raw_rr and fused and not full and hasattr(ops, "apply_KM") and not getattr(ops, "gram_exact", False)
                    and ops.apply_KM_ok(W, KS[:, :na], KS[:, na:2 * na])):
                ops.apply_KM(W, KS[:, :na], KS[:, na:2 * na])
                GG = ops.gram(S[:, :w0 + na], KS[:, :2 * na])
                eps_ = 6e-8 if ops.dtype == torch.float32 else 1.1e-16
                lam_l = lam[:ncl].detach().to(torch.float64).cpu()
                if dev.type == "cuda":
                    with _one_thread():
                        rawQ = _raw_basis_transform(GG.cpu(), Gxp, lam_l, ny, ncl, na + npc, na, cfg.ortho_tol, eps_)
                else:
                    rawQ = _raw_basis_transform(GG, Gxp, lam_l, ny, ncl, na + npc, na, cfg.ortho_tol, eps_)
                if rawQ is not None:
                    since_refresh += 1
            if rawQ is None:
                self._orthonormalize(W, S[:, :w0], MW[:, :na], VW=S[:, :w0 + na])
            if rawQ is not None:
                GA = rawQ[0]
            elif full:
                ops.apply_K(Sa, KSa)
                GA = ops.gram(Sa, KSa, symmetric=True)
                since_refresh = 0
            else:
                # only the new columns meet K; the Gram blocks among X and P come from the last Ritz step
                ops.apply_K(W, KSa[:, na + npc:])
                GA = ops.gram(Sa, KSa[:, na + npc:])  # (sz x na) = [X P W]^T K W
                since_refresh += 1

            def ritz(GA_, Gxp_=Gxp, full_=full or rawQ is not None, na_=na, nxp=na + npc):
                if full_:
                    G = _sym(GA_)
                else:
                    G = torch.empty((GA_.shape[0], GA_.shape[0]), dtype=GA_.dtype)
                    G[:nxp, :nxp] = Gxp_
                    G[:, nxp:] = GA_
                    G[nxp:, :nxp] = GA_[:nxp].transpose(0, 1)
                    G = _sym(G)
                E_, Z1_, Zp_ = _rr_step(G, na_)
                ZZ = torch.cat([Z1_, Zp_], 1).contiguous()
                return E_, ZZ, _sym(ZZ.transpose(0, 1) @ G @ ZZ)  # [X' P']^T K [X' P'] of the new basis

            if dev.type == "cuda":
                host = GA.cpu()
                with _one_thread():
                    Ea, ZZ, Gxp = ritz(host)
                    if rawQ is not None:
                        ZZ = (rawQ[1] @ ZZ).contiguous()  # coefficients of [X' P'] in the raw basis [Y X P W]
                Ea, ZZ = Ea.to(dev, non_blocking=True), ZZ.to(dev, non_blocking=True)
            else:
                Ea, ZZ, Gxp = ritz(GA)
                if rawQ is not None:
                    ZZ = (rawQ[1] @ ZZ).contiguous()
            if cfg.ritz_tol > 0.0:
                lam_prev = lam.clone()
            lam[ncl:] = Ea
            if ncl:
                S2[:, ny:ny + ncl].copy_(S[:, ny:ny + ncl])
            if rawQ is not None:
                Sa = S[:, :w0 + na]  # the update reads the whole raw basis
            # X_new | P_new (and K X_new | K P_new) are adjacent column ranges: ONE update [X' P'] = [X P W] [Z1 Zp]
            # per product reads the 240-column operand once instead of twice (the LDS-staged mix kernel holds the
            # 240 x 160 coefficient image; with the first, register-only kernel one wide launch was slower than two)
            if 2 * na <= 160:
                ops.mix(Sa, ZZ, S2[:, ny + ncl:ny + b + na])
                if not cfg.kx_fresh:
                    ops.mix(KSa, ZZ, KS2[:, :2 * na])
            else:
                Z1, Zp = ZZ[:, :na], ZZ[:, na:]
                ops.mix(Sa, Z1, S2[:, ny + ncl:ny + b])
                ops.mix(Sa, Zp, S2[:, ny + b:ny + b + na])
                if not cfg.kx_fresh:
                    ops.mix(KSa, Z1, KS2[:, :na])  # K X_new
                    ops.mix(KSa, Zp, KS2[:, na:2 * na])  # K P_new
            if cfg.kx_fresh and not fused:
                ops.apply_K(S2[:, ny + ncl:ny + b], KS2[:, :na])  # K X_new, fresh (K P_new is never needed)
            S, S2 = S2, S
            KS, KS2 = KS2, KS
            k0 = 0
            npc = na

        if native is not None:
            it = native[0]
        X = S[:, ny:ny + b]
        if not polish:  # (the corner-level phase of a nested start: the rotated fp32 block is all that is wanted)
            return ModalResult(lam[:k].clone(), X[:, :k], None, None, None, iterations=it, rerr=rel[:k].clone(),
                               history=history, block_vectors=X.contiguous())
        res = self._polish(X, k, it, rel[:k].clone(), history)
        res.coarse_iterations = self.nested_iterations
        if cfg.refine_tol > 0.0:
            try:
                res = self.refine64(res, k, float(A_norm), float(B_norm))
            finally:  # (a step that raises must not leave the combined fp64 K array cached for a later material)
                if hasattr(self.ops, "combined_k64"):
                    self.ops.combined_k64(False)
        return res

    # ------------------------------------------------------------------ the reference's "basic" method
    def solve_basic(self, k, X0=None, tracker=None, state=None):
        """``method='basic'`` of the reference API (src/lobpcg/_lobpcg.py:390-431, ``_update_basic``): NO explicit orthogonalisation of
        the search directions - every step's basis S = [X_active | P | W] goes through the Rayleigh-Ritz transform
        Ri = D^-1/2 chol(D^-1/2 S^T B S D^-1/2)^-T (``_get_rayleigh_ritz_transform``, :479-525) and the eigenvectors of
        Ri^T (S^T A S) Ri give X' = S Ri Z[:, :n - nc] and P' = S Ri Z[:, n : 2n - nc].  Same convergence test and hard locking of the
        leading converged pairs as the ortho iteration.  It is the textbook LOBPCG: cheaper per step than 'ortho' and less robust
        (the Gram matrix of a nearly dependent [X P W] loses its Cholesky factor close to convergence: the step is then repeated
        without P, and without progress the iteration stops) - the product's own solves use 'ortho'; this exists so that a caller of
        the reference API who asks for 'basic' gets that iteration (round 6; rounds 1-5 served it by the ortho iteration).  Large
        operations through ``ops`` (HIP kernels), the <= 3n x 3n dense steps in fp64 on the host.  No rigid-mode deflation."""
        ops, cfg = self.ops, self.cfg
        n, dev, dt = ops.n, ops.device, ops.dtype
        b = cfg.block or ((k + 3) // 4) * 4
        if X0 is not None and X0.shape[1] > b:
            b = ((X0.shape[1] + 3) // 4) * 4
        if n < 3 * b:
            raise ValueError(
                "LPBPCG algorithm is not applicable when the number of A rows (={})"
                " is smaller than 3 x the number of requested eigenpairs (={})".format(n, b))
        state = state or SolverState({"niter": cfg.maxit, "k": k, "n": b, "m": n}, {}, {})
        g = torch.Generator(device=dev).manual_seed(cfg.seed)
        S = torch.empty((n, 3 * b), dtype=dt, device=dev)
        AS, BS = torch.empty_like(S), torch.empty_like(S)
        S2 = torch.empty((n, 2 * b), dtype=dt, device=dev)
        R = torch.empty((n, b), dtype=dt, device=dev)
        nx0 = 0 if X0 is None else X0.shape[1]
        if nx0:
            S[:, :nx0].copy_(X0.to(dt))
        if nx0 < b:
            S[:, nx0:b].copy_(torch.randn((n, b - nx0), generator=g, dtype=torch.float32, device=dev).to(dt))
        G0 = torch.randn((n, 8), generator=g, dtype=torch.float32, device=dev).to(dt)
        G1 = torch.empty_like(G0)
        gn = torch.linalg.vector_norm(G0.double())
        ops.apply_K(G0, G1)
        A_norm = float(torch.linalg.vector_norm(G1.double()) / gn)
        ops.apply_M(G0, G1)
        B_norm = float(torch.linalg.vector_norm(G1.double()) / gn)
        state.fvars.update(A_norm=A_norm, B_norm=B_norm)
        tol = cfg.tol or (2e-6 if dt == torch.float32 else 1e-10)

        eps = 6e-8 if dt == torch.float32 else 1.1e-16

        def transform(GB):
            """Ri of the reference: None when the scaled Gram matrix has no Cholesky factor - or one so ill-conditioned that
            S Ri would come out of the fp32 update visibly non-orthonormal (eps x cond(S^T B S) >= 1e-3: the vectors are stored in
            ``dt``; the reference, which has no such test, then iterates on a basis that is no longer one)."""
            GB = _sym(GB)
            d = torch.rsqrt(torch.clamp(GB.diagonal(), min=1e-300))
            L, info = torch.linalg.cholesky_ex(GB * d[:, None] * d[None, :])
            if int(info) != 0 or not bool(torch.isfinite(L).all()):
                return None
            if eps / max(float(L.diagonal().min()), 1e-300) ** 2 >= 1e-3:
                return None
            Li = torch.linalg.solve_triangular(L, torch.eye(L.shape[0], dtype=L.dtype), upper=False)
            return d[:, None] * Li.transpose(0, 1)

        def ritz(GA, GB, keep=None):
            keep = GA.shape[0] if keep is None else keep
            Ri = transform(GB)
            if Ri is None:
                return None
            E_, Z = torch.linalg.eigh(_sym(Ri.transpose(0, 1) @ _sym(GA) @ Ri))
            return E_, (Ri @ Z[:, :keep]).contiguous()

        lam = torch.zeros(b, dtype=torch.float64, device=dev)
        rel = torch.full((b,), float("inf"), dtype=torch.float64, device=dev)
        nc, npc, ns = 0, 0, b  # converged leading columns, columns of P, columns of S in use
        history = []
        it = 0
        for it in range(cfg.maxit + 1):
            Sa = S[:, nc:ns]
            w = ns - nc
            ops.apply_K(Sa, AS[:, :w])
            ops.apply_M(Sa, BS[:, :w])
            GA, GB = ops.gram(Sa, AS[:, :w], exact=True), ops.gram(Sa, BS[:, :w], exact=True)
            na = b - nc
            keep = na if it == 0 else min(w, 2 * na)
            out = _small(lambda a_, b_, keep_=keep: ritz(a_, b_, keep_), dev, GA, GB)
            if out is None and npc:  # [X P W] numerically dependent: the step without P
                idx = torch.cat([torch.arange(0, na, device=dev), torch.arange(na + npc, w, device=dev)])
                keep = na
                out = _small(lambda a_, b_, keep_=keep: ritz(a_, b_, keep_), dev, GA[idx][:, idx].contiguous(), GB[idx][:, idx].contiguous())
                if out is not None:
                    Zf = torch.zeros((w, keep), dtype=torch.float64, device=dev)
                    Zf[idx] = out[1]
                    out = (out[0], Zf)
            if out is None:
                break  # (no factorisation even without P: the block has collapsed; what has converged so far is returned)
            E_, C = out
            lam[nc:] = E_[:na]
            # X' | P' = S_ (Ri Z): one update into the other buffer, then back (the locked columns stay where they are)
            ops.mix(Sa, C, S2[:, :keep])
            npc = keep - na
            S[:, nc:nc + keep].copy_(S2[:, :keep])  # (X' over the active X, P' behind it: S = [X_locked | X' | P' | W])
            Xa = S[:, nc:b]
            ops.apply_K(Xa, AS[:, :na])
            ops.apply_M(Xa, BS[:, :na])
            rn2, xn2 = ops.residual(R[:, :na], BS[:, :na], Xa, lam[nc:], src=AS[:, :na])
            rel[nc:] = torch.sqrt(rn2 / xn2) / (A_norm + lam[nc:].abs() * B_norm)
            relk = rel[:k]
            nconv = int(torch.cumprod((relk < tol).to(torch.int32), 0).sum())
            history.append((it, float(relk.max())))
            state.ivars.update(istep=it, converged_count=nconv, iterations_left=cfg.maxit - it)
            state.tvars["rerr"] = relk
            state.E, state.X = lam, S[:, :b]
            if tracker is not None:
                tracker(state)
            if nconv >= k or it == cfg.maxit or state.bvars.get("force_stop", False):
                break
            new_nc = (nconv // 4) * 4 if cfg.lock else 0
            shift = new_nc - nc
            if shift > 0:  # newly converged leading columns leave the active set (X_active = S[:, nc:b]; P stays at S[:, b:])
                nc = new_nc
                na = b - nc
            Wc = S[:, b + npc:b + npc + na]
            if shift > 0:
                Rn = R[:, shift:shift + na].contiguous()
            else:
                Rn = R[:, :na]
            self.precond_apply(Rn, Wc)
            ns = b + npc + na
            # layout for the next step: S[:, nc:ns] = [X_active (na) | P (npc) | W (na)]
        X = S[:, :b]
        res = self._polish(X, k, it, rel[:k].clone(), history)
        return res

    # ------------------------------------------------------------------ fp64 refinement
    def refine64(self, res, k, A_norm, B_norm):
        """Continue from the converged fp32 block with fp64 vectors (see SolverConfig.refine_tol): LOBPCG steps in
        fp64.  One step = K W, M W for the new block (fp64 SpMM), one Rayleigh-Ritz on S = [Y | X | P | W] through the
        generalised (3b + 6)-dimensional pencil (S^T K S, S^T M S) - the rigid modes come out as its six ~zero
        eigenvalues and are dropped - and X, P and their products with K and M for the next step by linearity.
        RAYLEIGH-RITZ BY RECURRENCE, as in the fp32 iteration: only the NEW columns meet the vectors - the Gram blocks
        S^T K W and S^T M W (ONE pass over the rows of S, K W and M W per step: ds_gram64_blocks) - while the blocks among Y, X and P follow
        from the last step's eigenvector matrix by (3b + 6)-dimensional algebra; every ``refine_refresh``-th step
        recomputes all blocks from the vectors.  The n x b updates are ``ops.mix64`` (ds_mix64, fp64 MFMA) over the LIST
        of blocks of S: one pass per result, no temporaries of the size of a block."""
        ops, cfg = self.ops, self.cfg
        dev = ops.device
        X = res.block_vectors.double()
        n, b = X.shape
        Y = ops.rigid64()
        ny = 0 if Y is None else 6
        f64 = dict(dtype=torch.float64, device=dev)
        KX, MX = torch.empty((n, b), **f64), torch.empty((n, b), **f64)
        if hasattr(ops, "combined_k64"):
            ops.combined_k64(True)  # one fp64 block array for K during the steps (released below)
        ops.apply_K64(X, KX)
        ops.apply_M64(X, MX)
        if ny:
            MY, KY = torch.empty_like(Y), torch.empty_like(Y)
            ops.apply_M64(Y, MY)
            ops.apply_K64(Y, KY)  # ~ eps ||K|| (the rigid modes are null vectors of K); kept, not assumed zero
            Y6, KY6, MY6 = Y[:, :6], KY[:, :6], MY[:, :6]

        def gen_eigh(GA_, GB_):
            L = torch.linalg.cholesky(_sym(GB_))
            Li = torch.linalg.solve_triangular(L, torch.eye(L.shape[0], dtype=L.dtype), upper=False)
            E_, Zt = torch.linalg.eigh(_sym(Li @ _sym(GA_) @ Li.transpose(0, 1)))
            return E_, (Li.transpose(0, 1) @ Zt).contiguous()

        def full_grams(blocks, kblocks, mblocks):
            # all pairs of blocks of S = [Y | X | P | W] against K S and M S: one pass over the rows each
            return ops.gram_blocks(blocks, kblocks, symmetric=True), ops.gram_blocks(blocks, mblocks, symmetric=True)

        # Rayleigh-Ritz on the fp32 block alone: the pairs whose residual is tested first, and an X that is
        # K-diagonal / M-orthonormal - which every later step's X is by construction
        GA, GB = ops.gram(X, KX), ops.gram(X, MX)
        lam, C = _small(gen_eigh, dev, GA, GB)
        X, KX, MX = ops.mix64([X], C), ops.mix64([KX], C), ops.mix64([MX], C)
        P = KP = MP = None
        # Every n-sized block of a step lives in a POOL of (n x b) buffers allocated once and used through column views (round 5):
        # the number of active columns changes from step to step, and blocks of ever new sizes - 2 to 4.5 GB each at configs[4] -
        # made the caching allocator release and re-request device memory in the middle of the steps (0.3 s of run-to-run spread)
        pool = {}

        def buf(name, cols, dtype=torch.float64):
            t = pool.get(name)
            if t is None:
                t = pool[name] = torch.empty((n, b), dtype=dtype, device=dev)
            return t[:, :cols]

        G0A = G0B = None  # Gram blocks among [Y | X | P] of the current basis (fp64, m0 x m0), by recurrence
        refresh = max(1, int(getattr(cfg, "refine_refresh", 8)))
        since = refresh  # the first step forms everything from the vectors
        hist = []
        for it in range(cfg.refine_maxit + 1):
            fused64 = hasattr(ops, "residual64")  # (the HIP operators: norms in one pass, no fp64 residual block)
            if fused64:
                rn2, xn2 = ops.residual64(KX, MX, X, lam)
                rn, R = torch.sqrt(rn2), None
                rel = rn / (torch.sqrt(xn2) * (A_norm + lam.abs() * B_norm))
            else:
                R = torch.addcmul(KX, MX, lam[None, :], value=-1.0)
                rn = torch.linalg.vector_norm(R, dim=0)
                rel = rn / (torch.linalg.vector_norm(X, dim=0) * (A_norm + lam.abs() * B_norm))
            worst = float(rel[:k].max())
            hist.append(worst)
            if worst < cfg.refine_tol or it == cfg.refine_maxit:
                break
            # soft locking: only the pairs that still miss the tolerance (with a margin) and the guard columns get a new
            # search direction; the others stay in X and take part in every Rayleigh-Ritz step, nothing else
            act = rel >= 0.3 * cfg.refine_tol
            act[k:] = True
            idx = torch.nonzero(act).reshape(-1)
            if idx.numel() % 4:  # (the preconditioner's kernels take multiples of 4 columns)
                rest_ = torch.nonzero(~act).reshape(-1)
                pad = rest_[torch.argsort(rel[rest_], descending=True)[:4 - idx.numel() % 4]]
                idx = torch.sort(torch.cat([idx, pad])).values
            # W = B R in fp32 (columns scaled to unit norm: the preconditioner is linear), promoted to fp64
            nact = int(idx.numel())
            R32 = buf("R32", nact, torch.float32)
            if fused64:
                ops.residual64_scaled(KX, MX, lam, 1.0 / rn.clamp(min=1e-300), idx, out=R32)
            else:
                R32.copy_(R[:, idx] / rn[idx].clamp(min=1e-300)[None, :])
            W32 = buf("W32", nact, torch.float32)
            self.precond_apply(R32, W32)
            for _ in range(max(0, int(getattr(cfg, "refine_sweeps", 1)) - 1)):
                # one more sweep of the preconditioned Richardson iteration: W <- W + B (R - K W), all fp32.  A step of
                # the fp64 phase is dominated by its dense n x b products, not by the preconditioner: a stronger
                # correction per step buys fewer steps
                T32, D32 = buf("T32", nact, torch.float32), buf("D32", nact, torch.float32)
                ops.apply_K(W32, T32)
                torch.sub(R32, T32, out=T32)
                self.precond_apply(T32, D32)
                W32 += D32
            W = buf("W", nact)
            W.copy_(W32)
            del R
            KW, MW = buf("KW", nact), buf("MW", nact)
            ops.apply_M64(W, MW)
            ops.apply_K64(W, KW)
            # (unit M-norm columns of W - a well scaled pencil - without touching the vectors: the scale wn comes off the
            # diagonal of W^T M W below, goes into the small matrices there and into the coefficients of the update)
            # pencil on S = [Y | X | P | W]  (P: the previous step's update directions - the locally optimal 3-term
            # recurrence; without it the pairs next to the guard vectors crawl); should the Gram matrix of S be
            # numerically singular (P and W nearly dependent close to convergence), the step is repeated without P
            head = ([(Y6, KY6, MY6)] if ny else []) + [(X, KX, MX)] + ([(P, KP, MP)] if P is not None else [])
            blocks = [h[0] for h in head] + [W]
            offs = [0]
            for blk in blocks:
                offs.append(offs[-1] + blk.shape[1])
            m, m0, na = offs[-1], offs[-2], W.shape[1]
            if since >= refresh or G0A is None or G0A.shape[0] != m0:
                GA, GB = full_grams(blocks, [h[1] for h in head] + [KW], [h[2] for h in head] + [MW])
                wn = torch.rsqrt(torch.diagonal(GB)[m0:].clamp(min=1e-300))
                for Gm in (GA, GB):
                    Gm[:, m0:] *= wn[None, :]
                    Gm[m0:, :] *= wn[:, None]
                since = 1
            else:  # only the new columns meet the vectors: S^T [K W | M W] in one pass over the rows
                GA, GB = torch.zeros((m, m), **f64), torch.zeros((m, m), **f64)
                GA[:m0, :m0], GB[:m0, :m0] = G0A, G0B
                Gw = ops.gram_blocks(blocks, [KW, MW])
                wn = torch.rsqrt(torch.diagonal(Gw[m0:, na:]).clamp(min=1e-300))
                for Gm, G in ((GA, Gw[:, :na]), (GB, Gw[:, na:])):
                    G = G * wn[None, :]
                    G[m0:] *= wn[:, None]
                    Gm[:, m0:] = G
                    Gm[m0:, :m0] = G[:m0].transpose(0, 1)
                since += 1
            GA, GB = _sym(GA), _sym(GB)
            use_p = P is not None
            try:
                E_, Z = _small(gen_eigh, dev, GA, GB)
            except torch.linalg.LinAlgError:
                if not use_p:
                    raise
                keep = torch.cat([torch.arange(0, offs[-3], device=dev), torch.arange(m0, m, device=dev)])  # without P
                E_, Zk = _small(gen_eigh, dev, GA[keep][:, keep].contiguous(), GB[keep][:, keep].contiguous())
                Z = torch.zeros((m, Zk.shape[1]), **f64)
                Z[keep] = Zk
            Zs = Z[:, ny:ny + b].contiguous()  # the six lowest pairs are the rigid modes
            lam = E_[ny:ny + b].contiguous()
            xo = ny  # row offset of X in S
            # update directions: everything of the new X that is not the old X.  Their scale (unit M-norm) comes from the
            # small algebra (Pn = S Zr), so the coefficients of the scaled directions in S are known before any n-sized
            # work, and every result is ONE pass over the blocks of S (ds_mix64: each block read once, the result
            # written once, no n x b temporaries)
            Zr = Zs.clone()
            Zr[xo:xo + b] = 0.0
            pn2 = ((Zr.transpose(0, 1) @ GB) * Zr.transpose(0, 1)).sum(1)[idx].clamp(min=1e-300)
            sc = torch.rsqrt(pn2)
            Tp = (Zr[:, idx] * sc[None, :]).contiguous()
            xi = 1 if ny else 0  # position of X in the head
            Zu, Tu = Zs.clone(), Tp.clone()  # the same coefficients for the W held in memory (not scaled)
            Zu[m0:] *= wn[:, None]
            Tu[m0:] *= wn[:, None]
            news = []
            for which in range(3):  # the vectors, their K-products, their M-products
                parts = [h[which] for h in head] + [(W, KW, MW)[which]]
                # (two sets of pool buffers, alternating: a step reads the set the previous one wrote)
                Xn = ops.mix64(parts, Zu, out=buf(f"X{which}_{it & 1}", b))
                Pd = ops.mix64([(blk, offs[i_]) for i_, blk in enumerate(parts) if i_ != xi], Tu,
                               out=buf(f"P{which}_{it & 1}", nact))  # (X's rows of Tu are zero)
                news.append((Xn, Pd))
            (X, P), (KX, KP), (MX, MP) = news
            del news
            # Gram blocks among [Y | X_new | P_new] for the next step: T^T G T with T the coefficients of that basis in S
            T = torch.zeros((m, ny + b + idx.numel()), **f64)
            if ny:
                T[:ny, :ny] = torch.eye(ny, **f64)
            T[:, ny:ny + b] = Zs
            T[:, ny + b:] = Zr[:, idx] * sc[None, :]
            G0A, G0B = _sym(T.transpose(0, 1) @ GA @ T), _sym(T.transpose(0, 1) @ GB @ T)
            del W, KW, MW
        if hasattr(ops, "combined_k64"):
            ops.combined_k64(False)
        U = X[:, :k].contiguous()
        kparts = ops.apply_K64(U, torch.empty_like(U), terms=True)
        MU = torch.empty_like(U)
        ops.apply_M64(U, MU)
        a = (U * kparts[0]).sum(0)
        bq = (U * kparts[1]).sum(0) if len(kparts) > 1 else None
        m_ = (U * MU).sum(0)
        out = ModalResult(lam[:k].clone(), U, a, bq, m_, iterations=res.iterations, rerr=rel[:k].clone(),
                          history=res.history, block_vectors=X)
        out.coarse_iterations = res.coarse_iterations
        out.refine_iterations = len(hist) - 1
        out.refine_history = hist
        return out

    # ------------------------------------------------------------------ fp64 Rayleigh-Ritz polish
    def _polish(self, X, k, it, rerr, history):
        ops = self.ops
        GK, coef, GM = ops.polish_products(X)  # fp64 (b x b) Gram matrices of the terms of K, and of M
        coef = [float(c) for c in coef]

        def small(GM_, *GK_):
            # everything (b x b) on the host, fp64, one LAPACK thread: the generalised Ritz problem and the quadratic forms
            # u^T K_i u, u^T M u of the wanted pairs.  (Until round 5 the quadratic forms were torch matmuls on the device: eight
            # more tiny launches per pass, and rocBLAS is free to sum a split-K product with atomics - the one place of a pass whose
            # last bits were not tied down; tests/test_fullsize_gpu.py compares two 8-lane runs bit for bit.)
            GA_ = _sym(sum(c * G for c, G in zip(coef, GK_)))
            GB_ = _sym(GM_)
            L = torch.linalg.cholesky(GB_)
            Li = torch.linalg.solve_triangular(L, torch.eye(L.shape[0], dtype=L.dtype), upper=False)
            E_, Zt = torch.linalg.eigh(_sym(Li @ GA_ @ Li.transpose(0, 1)))
            C_ = (Li.transpose(0, 1) @ Zt).contiguous()  # generalized eigenvectors, C^T GB C = I
            Ck_ = C_[:, :k].contiguous()
            quad = lambda G: ((Ck_.transpose(0, 1) @ _sym(G)) * Ck_.transpose(0, 1)).sum(1)
            qs = torch.stack([quad(G) for G in GK_] + [quad(GB_)])
            return E_[:k].clone(), C_, Ck_, qs

        if ops.device.type == "cuda" and self.cfg.native and ops.dtype == torch.float32:
            # (ds_host_polish: the same algebra in one native call on the host thread, 1.1 -> 0.5 ms per pass; round 6)
            from .. import _hip

            try:
                E, C, qs = _hip.host_polish([G.cpu() for G in GK], coef, GM.cpu(), k)
            except RuntimeError as ex:  # (X^T M X not positive definite: the error the torch form raises)
                raise torch.linalg.LinAlgError(str(ex)) from ex
            Ck = C[:, :k].contiguous()
            dev_ = ops.device
            E, C, Ck, qs = E.to(dev_), C.to(dev_), Ck.to(dev_), qs.to(dev_)
        else:
            E, C, Ck, qs = _small(small, ops.device, GM, *GK)
        U = torch.empty((ops.n, k), dtype=ops.dtype, device=ops.device)
        ops.mix(X, Ck, U)
        a = qs[0]
        bq = qs[1] if len(GK) > 1 else None
        m = qs[-1]
        Xb = None
        if self.keep_block or self.cfg.refine_tol > 0.0:  # (the whole rotated block: a warm start's or the refinement's input)
            Xb = torch.empty_like(X)
            ops.mix(X, C, Xb)
        return ModalResult(E, U, a, bq, m, iterations=it, rerr=rerr, history=history, block_vectors=Xb)
