"""``lobpcg`` / ``lobpcg_func`` - the reference's solver API (src/lobpcg/_lobpcg.py:8-25,123-140) on
top of the MI355X block eigensolver.

Same 17-argument signatures and return values.  Semantics kept: ``B`` is required by
``lobpcg_func`` and defines dtype/device/size; ``X`` (m x n) fixes the block size; ``largest``
defaults to True (the reference's default; ``LOBPCG_solver_freq`` passes False); ``tracker(worker)``
is called after every iteration with ``worker.ivars['istep'|'converged_count']``,
``worker.tvars['rerr']``, ``worker.E``, ``worker.X`` and may set ``worker.bvars['force_stop']``;
``ValueError`` when m < 3n; the convergence test is the backward-stable one of
``update_converged_count`` (:307-333).  Deliberate deviations (documented in DESIGN.md): fp32
iterates with fp64 Rayleigh-Ritz (the reference can only run fp32), default
``tol`` = 2e-6 instead of the reference's unreachable 1.5e-8 (dtype-table bug :35-38), a
Chebyshev block-Jacobi preconditioner when ``iK`` is None (the reference uses none and does not
converge), ``profiler`` (a TensorBoard log dir in the reference) is accepted and ignored - use rocprofv3.
Round 6: ANY pencil is served - a row count that is not a multiple of 3 is padded with decoupled rows whose eigenvalue sits
above the spectrum (the kernels work on 3 x 3 node blocks) and the pad rows are cut off the result; dense ``A`` / ``B`` are
re-blocked like sparse ones; ``method='basic'`` runs the reference's basic iteration (ModalSolver.solve_basic), any other
method name is a ``ValueError``.
"""
from typing import Dict, Optional, Tuple

import torch
from torch import Tensor

from .modal_solver import ModalSolver, SolverConfig, SolverState


class _CallableOps:
    """Wraps ops so that every product with A calls a user callable A(X) -> AX (reference _linalg_utils.py:34-35).
    The base ops are built on the pencil (B, B) for the pattern and the mass products only: nothing of A is known
    but its action, so the block-Jacobi blocks are the identity (no preconditioner, as in the reference when iK is
    None) and the fp64 polish forms X^T (A X) from the callable's own product."""

    def __init__(self, base, fn, sign):
        self._base, self._fn, self._sign = base, fn, sign
        eye = torch.eye(3, dtype=torch.float32, device=base.device).reshape(1, 9)
        base.dinv = eye.repeat(base.nv, 1).contiguous()

    def __getattr__(self, name):
        return getattr(self._base, name)

    def apply_K(self, X, out):
        out.copy_(self._fn(X.contiguous()) * self._sign)

    def polish_products(self, X):
        b = self._base
        Xc = X.contiguous()
        AX = (self._fn(Xc) * self._sign).to(torch.float64).contiguous()
        GA = b.gram(Xc, AX, exact=True)
        _, (mkind, mvals) = b.polish_terms()
        MX = b._scratch("polish", Xc.shape, torch.float64)
        b._spmm(mkind, mvals, Xc, MX)
        return [GA], [1.0], b.gram(Xc, MX)


def pad_value(A, B):
    """Eigenvalue of the decoupled pad rows: ||A||_inf / min diag(B) - at least every Rayleigh quotient of a unit vector, i.e. at
    the top of the spectrum, out of the way of the wanted (lowest) pairs.  A None: the pencil (B, B) of a callable A -> 2."""
    def coo(T):
        T = T.to_sparse_coo().coalesce() if T.layout != torch.sparse_coo else T.coalesce()
        return T.indices(), T.values().double()

    bi, bv = coo(B)
    bdiag = bv[bi[0] == bi[1]]
    bmin = float(bdiag.abs().min()) if bdiag.numel() else 1.0
    if A is None:
        return 1.0 / max(bmin, 1e-300)
    ai, av = coo(A)
    rows = torch.zeros(A.shape[0], dtype=torch.float64, device=av.device).index_add_(0, ai[0], av.abs())
    return float(rows.max()) / max(bmin, 1e-300)


def _pad_pencil(A, B, pad):
    """(A, B) as sparse COO with ``pad`` extra decoupled rows: diag(A, big B_min I), diag(B, B_min I) - eigenvalue ``big`` each."""
    def coo(T):
        return T.to_sparse_coo().coalesce() if T.layout != torch.sparse_coo else T.coalesce()

    A, B = coo(A), coo(B)
    if not pad:
        return A, B
    m = B.shape[-1]
    big = pad_value(A if A is not B else None, B)
    bi, bv = B.indices(), B.values()
    bdiag = bv[bi[0] == bi[1]]
    bmin = bdiag.abs().min() if bdiag.numel() else torch.ones((), dtype=bv.dtype, device=bv.device)
    extra = torch.arange(m, m + pad, device=B.device)
    ex = torch.stack([extra, extra])

    def grow(T, val):
        return torch.sparse_coo_tensor(torch.cat([T.indices(), ex], 1), torch.cat([T.values(), val.to(T.values().dtype).expand(pad)]),
                                       (m + pad, m + pad)).coalesce()

    Bp = grow(B, bmin)
    Ap = Bp if A is B else grow(A, bmin.double() * big)
    return Ap, Bp


def _solve(A, B, k, X, E, n, iK, niter, tol, largest, method, tracker, ortho_iparams, ortho_fparams,
           ortho_bparams, return_rerr):
    from ..modal_ops import HipSparseOps

    method = "ortho" if method is None else method
    if method not in ("ortho", "basic"):
        raise ValueError(f"lobpcg_func: unknown method {method!r} (the reference implements 'ortho' and 'basic', _lobpcg.py:366-369)")
    if not isinstance(B, torch.Tensor):
        raise TypeError("lobpcg_func: B must be a torch tensor (sparse or dense)")
    if not B.is_cuda:
        raise RuntimeError("diffsound_amd.lobpcg: tensors must live on the HIP device (there is no CPU fallback)")
    m = B.shape[-1]
    k = (1 if X is None else X.shape[-1]) if k is None else k
    n = (k if n is None else n) if X is None else X.shape[-1]
    if m < 3 * n:
        raise ValueError(
            "LPBPCG algorithm is not applicable when the number of A rows (={})"
            " is smaller than 3 x the number of requested eigenpairs (={})".format(m, n))
    largest = True if largest is None else largest
    sign = -1.0 if largest else 1.0
    a_callable = callable(A) and not isinstance(A, torch.Tensor)
    pad = (-m) % 3  # the kernels work on 3 x 3 node blocks: a pencil of any other row count gets `pad` decoupled rows
    if a_callable:
        Bp, _ = _pad_pencil(B, B, pad)
        ops = HipSparseOps(Bp, Bp)  # pattern/diagonal from B; K products come from the callable
        fn = A
        if pad:  # the callable sees the caller's m rows; the pad rows carry a value above the spectrum
            gprobe = torch.randn((m, 4), dtype=torch.float32, device=B.device)
            big = 4.0 * float(torch.linalg.vector_norm(A(gprobe).double()) / torch.linalg.vector_norm(gprobe.double())) * pad_value(None, B)

            def fn(Xp, A=A, m=m, big=big, sign=sign):
                out = torch.empty_like(Xp)
                out[:m] = A(Xp[:m].contiguous())
                out[m:] = Xp[m:] * (big * sign)
                return out
        ops = _CallableOps(ops, fn, sign)
    else:
        Ae = A if not largest else -A
        Ap, Bp = _pad_pencil(Ae, B, pad)
        ops = HipSparseOps(Ap, Bp)
    if X is not None and pad:
        X = torch.cat([X, torch.zeros((pad, X.shape[1]), dtype=X.dtype, device=X.device)], 0)
    cfg = SolverConfig(block=((n + 3) // 4) * 4, maxit=1000 if niter is None else niter, tol=tol or 0.0)
    if largest or a_callable:
        cfg.cheb_degree = 1  # the polynomial preconditioner targets the low end of an SPD spectrum only
    else:  # the polynomial's interval ends at a RIGOROUS bound of lambda_max(T A) (HipSparseOps.lmax_bound), not at an estimate
        cfg.lmax_cap, cfg.power_iters = float(ops.lmax_bound), 0
    precond = None
    if iK is not None:
        if callable(iK) and not isinstance(iK, torch.Tensor):
            apply_ik = iK
        elif iK.layout in (torch.sparse_coo, torch.sparse_csr):
            apply_ik = lambda R: torch.sparse.mm(iK.to(R.dtype), R)
        else:
            apply_ik = lambda R: iK.to(R.dtype) @ R
        if pad:  # the caller's preconditioner sees the caller's m rows; the decoupled pad rows pass through
            def precond(R, W, m=m):
                W[:m].copy_(apply_ik(R[:m].contiguous()))
                W[m:].copy_(R[m:])
        else:
            precond = lambda R, W: W.copy_(apply_ik(R))
    iparams = {"m": m, "n": n, "k": k, "niter": cfg.maxit}
    if ortho_iparams:
        iparams.update(ortho_iparams)
    fparams = {"tol": cfg.tol}
    if ortho_fparams:
        fparams.update(ortho_fparams)
    bparams = {"largest": largest}
    if ortho_bparams:
        bparams.update(ortho_bparams)
    cfg.ortho_passes = max(2, min(3, int(iparams.get("ortho_i_max", 3))))
    state = SolverState(iparams, fparams, bparams)
    state.E = E
    solver = ModalSolver(ops, cfg, precond=precond)
    if tracker is not None:
        tracker(state)  # the reference calls the tracker once before the first update (:350-351)
    if pad and tracker is not None:
        inner = tracker

        def tracker(st, inner=inner, m=m):  # the tracker sees the caller's m rows
            Xfull = st.X
            st.X = None if Xfull is None else Xfull[:m]
            try:
                inner(st)
            finally:
                st.X = Xfull
    if method == "basic":
        res = solver.solve_basic(k, X0=X, tracker=tracker, state=state)
    else:
        res = solver.solve(k, X0=X, tracker=tracker, state=state)
    Eo = res.eigenvalues * sign
    out_dtype = torch.float32 if B.dtype not in (torch.float32, torch.float64) else B.dtype
    Eo, Xo = Eo.to(out_dtype), res.vectors[:m].to(out_dtype)
    if return_rerr:
        return Eo, Xo, res.rerr
    return Eo, Xo


def lobpcg(A: Tensor, k: Optional[int] = None, B: Optional[Tensor] = None, X: Optional[Tensor] = None, E=None,
           n: Optional[int] = None, iK: Optional[Tensor] = None, niter: Optional[int] = None,
           tol: Optional[float] = None, largest: Optional[bool] = None, method: Optional[str] = None,
           tracker: Optional[None] = None, ortho_iparams: Optional[Dict[str, int]] = None,
           ortho_fparams: Optional[Dict[str, float]] = None, ortho_bparams: Optional[Dict[str, bool]] = None,
           return_rerr=False, profiler=None) -> Tuple[Tensor, Tensor]:
    """reference _lobpcg.py:8-121 (matrix A, optional B; B=None means the standard problem)."""
    assert A.shape[-2] == A.shape[-1], A.shape
    if B is None:
        nn_ = A.shape[-1]
        idx = torch.arange(nn_, device=A.device)
        B = torch.sparse_coo_tensor(torch.stack([idx, idx]), torch.ones(nn_, dtype=A.dtype, device=A.device),
                                    (nn_, nn_))
    else:
        assert A.shape == B.shape, (A.shape, B.shape)
    return _solve(A, B, k, X, E, n, iK, niter, tol, largest, method, tracker, ortho_iparams, ortho_fparams,
                  ortho_bparams, return_rerr)


def lobpcg_func(A, B: Tensor, k: Optional[int] = None, X: Optional[Tensor] = None, E=None, n: Optional[int] = None,
                iK: Optional[Tensor] = None, niter: Optional[int] = None, tol: Optional[float] = None,
                largest: Optional[bool] = None, method: Optional[str] = None, tracker: Optional[None] = None,
                ortho_iparams: Optional[Dict[str, int]] = None, ortho_fparams: Optional[Dict[str, float]] = None,
                ortho_bparams: Optional[Dict[str, bool]] = None, return_rerr=False, profiler=None
                ) -> Tuple[Tensor, Tensor]:
    """reference _lobpcg.py:123-212 (A may be a tensor, a sparse tensor or a callable X -> A X)."""
    return _solve(A, B, k, X, E, n, iK, niter, tol, largest, method, tracker, ortho_iparams, ortho_fparams,
                  ortho_bparams, return_rerr)
