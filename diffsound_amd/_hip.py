"""ctypes binding of libdiffsound_hip.so (the C ABI declared in include/diffsound_hip.h).

There is NO fallback: if the library is missing or a call fails, a RuntimeError is raised
(the reference's native op surfaces errors the same way, as RuntimeError through pybind:
reference src/include/macro.h:75-83).  Tensors are passed as raw ``data_ptr()`` values and the
kernels are enqueued on torch's current HIP stream.
"""
import ctypes
import threading
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DS_EXP_LIB") or os.path.join(_HERE, "csrc", "libdiffsound_hip.so")  # (DS_EXP_LIB: A/B builds, experiments)
ABI_VERSION = 33  # DS_ABI_VERSION of include/diffsound_hip.h

c_i32p = ctypes.POINTER(ctypes.c_int32)
_P = ctypes.c_void_p
_I64 = ctypes.c_int64
_I = ctypes.c_int
_D = ctypes.c_double
_F = ctypes.c_float

_SIGNATURES = {
    "ds_last_error": (ctypes.c_char_p, []),
    "ds_abi_version": (_I, []),
    "ds_pattern_build": (_I, [_P, _I64, _I, _I64, _I, ctypes.POINTER(_P)]),
    "ds_pattern_sizes": (_I, [_P, ctypes.POINTER(_I64), ctypes.POINTER(_I64), ctypes.POINTER(_I64)]),
    "ds_pattern_export": (_I, [_P, _P, _P, _P, _P, _P]),
    "ds_pattern_free": (None, [_P]),
    "ds_dpattern_build": (_I, [_P, _I64, _I, _I64, _I, _P, ctypes.POINTER(_P)]),
    "ds_dpattern_sizes": (_I, [_P, ctypes.POINTER(_I64), ctypes.POINTER(_I64), ctypes.POINTER(_I64), ctypes.POINTER(_I64),
                               ctypes.POINTER(_I64)]),
    "ds_dpattern_export": (_I, [_P] * 13),
    "ds_dpattern_free": (None, [_P]),
    "ds_spmm_f64_polish": (_I, [_P, _P, _P, _P, _P, _I64, _P, _I64, _P, _P, _P, _I64, _I, _P]),
    "ds_spmm_f64_union": (_I, [_I, _I, _P, _P, _I64, _I, _P, _P, _I64, _I64, _P, _I64, _P, _I64, _I, _P]),
    "ds_spmm_f64_polish_f32out": (_I, [_P, _P, _P, _P, _P, _I64, _P, _I64, _P, _P, _P, _I64, _I, _P]),
    "ds_pack_kc": (_I, [_P, _P, _I64, _P, _P]),
    "ds_spmm_union16m": (_I, [_I, _I, _I, _P, _P, _P, _P, _P, _P, _I64, _I64, _I, _I, _I64, _P, _I64, _P, _I64, _I, _P, _I64, _P, _I, _F,
                              _F, _I, _P, _I64, _P]),
    "ds_spmm_union32m": (_I, [_I, _I, _P, _P, _P, _P, _P, _I64, _I64, _I64, _I, _I, _I64, _P, _I64, _P, _I64, _I, _P]),
    "ds_edge_table": (_I, [_P, _I64, _I64, _P, _P, _P, ctypes.POINTER(_I64), _P]),
    "ds_unique_rows3": (_I, [_P, _I64, _P, _P, ctypes.POINTER(_I64), _P]),
    "ds_assemble_kml": (_I, [_P, _P, _I64, _I, _I64, _P, _P, _I64, _P, _P, _P, _P, _P, _P, _P]),
    "ds_combine_material": (_I, [_P, _P, _P, _I64, _P, _I64, _D, _D, _P, _P, _P, _P, _P]),
    "ds_spmm_bsr3": (_I, [_I, _P, _P, _P, _P, _I64, _P, _I64, _P, _I64, _I, _P]),
    "ds_gram_workspace_bytes": (_I64, [_I64, _I, _I]),
    "ds_gram": (_I, [_P, _I, _I64, _I, _P, _I, _I64, _I, _I64, _I, _P, _P, _I64, _P]),
    "ds_residual": (_I, [_P, _I64, _P, _I64, _P, _I64, _P, _I64, _P, _I64, _I, _P, _P, _P]),
    "ds_cheb_init": (_I, [_P, _I64, _P, _I64, _P, _I64, _P, _I64, _I, _F, _P]),
    "ds_cheb_step": (_I, [_P, _I64, _P, _I64, _P, _I64, _P, _I64, _P, _I64, _I, _F, _F, _P]),
    "ds_cheb_spmm": (_I, [_P, _P, _P, _I64, _P, _I64, _P, _I64, _P, _I64, _P, _I, _F, _F, _I, _P]),
    "ds_spmm_residual": (_I, [_P, _P, _P, _I64, _P, _I64, _P, _I64, _P, _I64, _I, _P]),
    "ds_scalar_csr_spmm": (_I, [_P, _P, _P, _I64, _P, _I64, _P, _I64, _I, _F, _P]),
    "ds_groups_build": (_I, [_P, _P, _I64, ctypes.POINTER(_P)]),
    "ds_groups_sizes": (_I, [_P, ctypes.POINTER(_I64), ctypes.POINTER(_I64)]),
    "ds_groups_export": (_I, [_P, _P, _P, _P, _P]),
    "ds_groups_free": (None, [_P]),
    "ds_pack_groups": (_I, [_P, _P, _I64, _P, _P]),
    "ds_geometry_grad": (_I, [_P, _I64, _I, _I64, _P, _P, _I64, _I, _P, _P, _D, _D, _P, _P, _I, _P, _P, _P]),
    "ds_spmm_union": (_I, [_I, _I, _P, _P, _I64, _I, _P, _P, _I64, _I64, _P, _I64, _P, _I64, _P, _I64, _P, _I, _F, _F, _I, _P,
                          _I64, _P]),
    "ds_union_residual_workspace_bytes": (_I64, [_I64, _I]),
    "ds_spmm_union_narrow": (_I, [_I, _I, _P, _P, _I64, _P, _P, _I64, _I64, _P, _I64, _P, _I64, _I, _P]),
    "ds_spmm_union_km": (_I, [_I, _P, _P, _I64, _I, _P, _P, _P, _I64, _I64, _P, _I64, _P, _I64, _P, _I64, _I, _P]),
    "ds_union_residual": (_I, [_I, _P, _P, _I64, _I, _P, _P, _P, _I64, _I64, _P, _I64, _P, _P, _I64, _I, _P, _I64, _P, _P, _P]),
    "ds_mix": (_I, [_P, _I64, _I, _P, _I, _P, _I64, _I64, _F, _F, _P]),
    "ds_gram64_blocks": (_I, [_I, _P, _I, _P, _I64, _I, _P, _P, _I64, _P]),
    "ds_residual64_workspace_doubles": (_I64, [_I]),
    "ds_residual64_norms": (_I, [_P, _I64, _P, _I64, _P, _I64, _P, _I64, _I, _P, _I64, _P, _P, _P]),
    "ds_residual64_scaled": (_I, [_P, _I64, _P, _I64, _P, _P, _P, _I, _P, _I64, _I64, _P]),
    "ds_mix64": (_I, [_I, _P, _P, _I64, _I, _P, _I64, _I64, _D, _D, _P]),
    "ds_osc_bank_fwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _D, _P, _P]),
    "ds_osc_bank_bwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _D, _P, _P, _P, _P, _P]),
    "ds_readout_pass": (_I, [_P, _P, _P, _P, _I, _D, _D, _D, _D, _D, _D, _D, _D, _P, _I, _I, _D, _P, _I, _P, _P, _P, _P, _P, _P]),
    "ds_osc_tv_workspace_floats": (_I64, [_I, _I, _I]),
    "ds_osc_tv_fwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _D, _P, _P, _P]),
    "ds_osc_tv_bwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _D, _P, _P, _P, _P, _P]),
    "ds_stream_triad": (_I, [_P, _P, _P, _I64, _F, _P]),
    "ds_profile_stream": (_I, [_P, _I64]),
    "ds_profile_collect": (_I64, [_P, _P, _P, _P, _P, _I64]),
    "ds_profile_kinds": (_I, [ctypes.c_uint]),
    "ds_stft_power": (_I, [_P, _I, _I, _I, _I, _P, _P, _P, _P]),
    "ds_spec_loss": (_I, [_I, _P, _P, _I, _I, _I, _F, _F, _I, _P, _P, _P]),
    "ds_stft_power_bwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _F, _P, _P, _P]),
}

class Block64(ctypes.Structure):
    """ds_block64_t of include/diffsound_hip.h."""
    _fields_ = [("a", _P), ("lda", _I64), ("p", ctypes.c_int32), ("offset", ctypes.c_int32)]


class LevelDesc(ctypes.Structure):
    """ds_level_t of include/diffsound_hip.h."""
    _fields_ = [("utab", _P), ("ctab", _P), ("ngroups", _I64), ("cap_blocks", ctypes.c_int32),
                ("degree", ctypes.c_int32), ("gent", _P), ("kgrp", _P), ("nnzb", _I64), ("nv", _I64), ("dinv", _P),
                ("lmax", _D), ("lmin", _D), ("mf_group_nodes", ctypes.c_int32), ("mf_max_entries", ctypes.c_int32),
                ("mf_max_batch_blocks", ctypes.c_int32), ("level_tag", ctypes.c_int32),
                ("mf_gptr", _P), ("mf_gcol", _P), ("mf_gmeta", _P), ("mf_gbase", _P), ("mf_ghead", _P), ("mf_kc", _P),
                ("m32_max_entries", ctypes.c_int32), ("m32_max_batch_blocks", ctypes.c_int32),
                ("m32_gptr", _P), ("m32_gcol", _P), ("m32_gmeta", _P), ("m32_gbase", _P), ("m32_k", _P), ("m32_m", _P),
                ("tgrp", _P), ("mf_nblocks", _I64)]


class TwoLevelDesc(ctypes.Structure):
    """ds_twolevel_t of include/diffsound_hip.h."""
    _fields_ = [("fine", LevelDesc), ("coarse", LevelDesc), ("rptr", _P), ("rcol", _P), ("rw", _P), ("pptr", _P),
                ("pcol", _P), ("pw", _P), ("R", _P), ("ldr", _I64), ("W", _P), ("ldw", _I64), ("D", _P), ("ldd", _I64),
                ("AD", _P), ("lda", _I64), ("Rr", _P), ("ldrr", _I64), ("Rc", _P), ("Ec", _P), ("Dc", _P), ("ADc", _P),
                ("ldc", _I64), ("ncols", ctypes.c_int32), ("Wc", _P), ("ldwc", _I64), ("storage", ctypes.c_int32),
                ("R16", _P), ("ldr16", _I64)]


class LapackTable(ctypes.Structure):
    """ds_lapack_t: Fortran-convention dsyevd / dgemm entry points, and (ABI 31, optional) the stages of dsyevd."""
    _fields_ = [("dsyevd", _P), ("dgemm", _P), ("dsytrd", _P), ("dstedc", _P), ("dormtr", _P)]


class LobpcgDesc(ctypes.Structure):
    """ds_lobpcg_t of include/diffsound_hip.h."""
    _i32 = ctypes.c_int32
    _fields_ = [("n", _I64), ("nv", _I64), ("b", _i32), ("k", _i32), ("ny", _i32), ("maxit", _i32), ("lock", _i32),
                ("ortho_passes", _i32), ("rr_refresh", _i32), ("gram_exact", _i32), ("kx_fresh", _i32), ("raw_rr", _i32),
                ("tol", _D), ("ortho_tol", _D),
                ("A_norm", _D), ("B_norm", _D), ("S", _P), ("S2", _P), ("KS", _P), ("KS2", _P), ("R", _P), ("MX", _P),
                ("MW", _P), ("lds", _I64), ("ldks", _I64), ("ldr", _I64), ("level", LevelDesc), ("mgrp", _P),
                ("rowptr", _P), ("colidx", _P), ("k32", _P), ("k32t", _P), ("twolevel", ctypes.POINTER(TwoLevelDesc)),
                ("pa", _P), ("pb", _P), ("ldp", _I64), ("pr16", _P), ("gbuf", _P), ("cbuf", _P), ("nrm", _P), ("lam_dev", _P),
                ("res_work", _P), ("res_work_bytes", _I64), ("gram_work", _P), ("gram_work_bytes", _I64), ("lam", ctypes.POINTER(_D)), ("rerr", ctypes.POINTER(_D)),
                ("history", ctypes.POINTER(_D)), ("history_cap", _i32), ("iterations", _i32), ("result_in_s2", _i32),
                ("wait_mode", _i32), ("ritz_tol", ctypes.c_double)]


_lapack = None


def lapack_table(source=None):
    """ds_lapack_t for ds_lobpcg_iterate: Fortran-convention dsyevd / dgemm entry points and, where the library has them, the
    stages dsytrd / dstedc / dormtr.  Sources, in order of preference (``source`` or the environment variable DS_LAPACK names
    one: "scipy" | "torch"):
      scipy  scipy.linalg.cython_lapack / cython_blas export every routine as a PyCapsule (OpenBLAS: dsyevd 240 x 240 in
             2.35 ms on the GPU box's host, and the stages, so the Ritz step back-transforms only the wanted third);
      torch  the MKL inside libtorch_cpu.so exports dsyevd_ / dgemm_ as plain symbols (2.9 ms; no stages) - so the product does
             not DEPEND on SciPy's private capsule table: without SciPy, or with a SciPy whose capsules have moved,
             the solve still runs, 20 % slower on the host side.
    The table changes nothing in the process: the BLAS thread count is the caller's business (``blas_one_thread`` below is what
    the native solve is wrapped in)."""
    global _lapack
    want = source or os.environ.get("DS_LAPACK") or None
    if want not in (None, "scipy", "torch"):
        raise ValueError(f"lapack_table: unknown source {want!r} (scipy | torch)")
    if _lapack is not None and (want is None or _lapack_source == want):
        return _lapack
    errors = []
    for name in ((want,) if want else ("scipy", "torch")):
        try:
            table = _lapack_from_scipy() if name == "scipy" else _lapack_from_torch()
        except Exception as ex:  # (ImportError, KeyError on a capsule, OSError / AttributeError on a symbol)
            errors.append(f"{name}: {type(ex).__name__}: {ex}")
            continue
        if want is None or _lapack is None:
            _set_lapack(table, name)
        return table
    raise RuntimeError("diffsound_amd: no LAPACK for the eigensolver's dense steps (" + "; ".join(errors) + ")")


_lapack_source = None


def _set_lapack(table, name):
    global _lapack, _lapack_source
    _lapack, _lapack_source = table, name


def lapack_source():
    """Which library the process-wide table came from ("scipy" | "torch"), None before the first call of lapack_table()."""
    return _lapack_source


def _lapack_from_scipy():
    import scipy.linalg.cython_blas as cb
    import scipy.linalg.cython_lapack as cl

    get = ctypes.pythonapi.PyCapsule_GetPointer
    name = ctypes.pythonapi.PyCapsule_GetName
    get.restype, get.argtypes = ctypes.c_void_p, [ctypes.py_object, ctypes.c_char_p]
    name.restype, name.argtypes = ctypes.c_char_p, [ctypes.py_object]
    addr = lambda cap: get(cap, name(cap))
    stages = [addr(cl.__pyx_capi__[nm]) if nm in cl.__pyx_capi__ else None for nm in ("dsytrd", "dstedc", "dormtr")]
    if not all(stages):
        stages = [None, None, None]
    return LapackTable(addr(cl.__pyx_capi__["dsyevd"]), addr(cb.__pyx_capi__["dgemm"]), *stages)


def _lapack_from_torch():
    lib = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libtorch_cpu.so"))
    addr = lambda nm: ctypes.cast(getattr(lib, nm), ctypes.c_void_p).value
    stages = [None, None, None]
    try:
        stages = [addr(nm) for nm in ("dsytrd_", "dstedc_", "dormtr_")]
    except AttributeError:
        pass
    return LapackTable(addr("dsyevd_"), addr("dgemm_"), *stages)


class blas_one_thread:
    """The <= 3b x 3b problems of the native eigensolver loop are fastest on one BLAS thread and are solved by several
    hypothesis lanes at once.  Context manager, re-entrant across threads: the first solve that enters limits the BLAS
    behind SciPy to one thread (threadpoolctl), the last one that leaves restores what was there - nothing stays changed
    in the host process (round 2 left the limit in place for good, which also skewed the CPU baseline)."""

    _lock = threading.Lock()
    _depth = 0
    _ctl = None
    _controller = False  # threadpoolctl's view of the loaded BLAS libraries, built once (a scan of every loaded library:
                         # ~1 ms, too much to pay on every solve)

    def __enter__(self):
        cls = blas_one_thread
        with cls._lock:
            if cls._depth == 0:
                if cls._controller is False:
                    lapack_table()  # (loads SciPy's BLAS: the controller only knows the libraries loaded when it is built)
                    try:
                        from threadpoolctl import ThreadpoolController

                        cls._controller = ThreadpoolController()
                    except Exception:  # threadpoolctl missing: the BLAS keeps its own thread count
                        cls._controller = None
                cls._ctl = None if cls._controller is None else cls._controller.limit(limits=1, user_api="blas")
            cls._depth += 1

    def __exit__(self, *a):
        cls = blas_one_thread
        with cls._lock:
            cls._depth -= 1
            if cls._depth == 0 and cls._ctl is not None:
                cls._ctl.restore_original_limits()
                cls._ctl = None


_SIGNATURES["ds_twolevel_apply"] = (_I, [ctypes.POINTER(TwoLevelDesc), _P])
_SIGNATURES["ds_chebyshev_apply"] = (_I, [ctypes.POINTER(LevelDesc), _P, _I64, _P, _I64, _P, _P, _I64, _I, _P])
_SIGNATURES["ds_chebyshev_apply16"] = (_I, [ctypes.POINTER(LevelDesc), _P, _I64, _P, _I64, _P, _P, _P, _I64, _I, _P])
_SIGNATURES["ds_spmm_union16"] = (_I, [_I, _P, _P, _I64, _I, _P, _P, _I64, _I64, _P, _I64, _P, _I64, _I, _P, _I64, _P, _I, _F,
                                      _F, _I, _P, _I64, _P])
_SIGNATURES["ds_group_inverse"] = (_I, [_P, _P, _P, _I64, _I, _P, _P])
_SIGNATURES["ds_group_pack_kc"] = (_I, [_P, _P, _P, _P, _P, _P, _I, _I64, _P, _P])
_SIGNATURES["ds_group_apply16"] = (_I, [_P, _I, _P, _I, _I64, _P, _I, _I64, _I64, _I, _P])
_SIGNATURES["ds_cheb_init16"] = (_I, [_P, _I, _I64, _P, _I64, _P, _I64, _P, _I64, _I, _F, _P])
_SIGNATURES["ds_scalar_csr_spmm16"] = (_I, [_P, _P, _P, _I64, _P, _I64, _P, _I64, _I, _F, _P])
_SIGNATURES["ds_lobpcg_iterate"] = (_I, [ctypes.POINTER(LobpcgDesc), ctypes.POINTER(LapackTable), _P])
_SIGNATURES["ds_host_wait_mode"] = (_I, [_I])
_SIGNATURES["ds_selftest_dense"] = (_I, [ctypes.POINTER(LapackTable), _I, _I, ctypes.c_uint, ctypes.POINTER(_D)])
_SIGNATURES["ds_host_start_block"] = (_I, [ctypes.POINTER(LapackTable), _P, _I, _I, _D, _D, _P, _P, _P, ctypes.POINTER(_D), ctypes.POINTER(_I)])
_SIGNATURES["ds_host_polish"] = (_I, [ctypes.POINTER(LapackTable), _I, _P, _P, _P, _I, _I, _P, _P, _P])
EXPORTED_SYMBOLS = tuple(_SIGNATURES)
_lib = None


def lib():
    """Load (once) and return the ctypes handle; RuntimeError if the HIP extension is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"diffsound_amd: HIP extension {LIB_PATH} is missing - build it with "
                "`python -c 'import __graft_entry__ as g; g.build()'` or `make -C diffsound_amd/csrc`. "
                "There is no CPU fallback.")
        handle = ctypes.CDLL(LIB_PATH)
        handle.ds_abi_version.restype = _I
        got = handle.ds_abi_version()
        if got != ABI_VERSION:  # a stale build resolves every symbol and would be called with shifted arguments
            raise RuntimeError(f"diffsound_amd: {LIB_PATH} has ABI version {got}, this package needs {ABI_VERSION} - "
                               "rebuild it with `make -C diffsound_amd/csrc`")
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def host_start_block(G, ny, b, ortho_tol, eps):
    """ds_host_start_block on a host fp64 tensor G ((ny + b) x 2 b): (lam, coef, cx, amp) as host tensors, or None when the caller
    has to take the explicit route."""
    G = G.contiguous()
    lam = torch.empty(b, dtype=torch.float64)
    coef = torch.empty((ny + b, b), dtype=torch.float64)
    cx = torch.empty((b, b), dtype=torch.float64)
    amp, route = _D(0.0), _I(0)
    with blas_one_thread():
        check(lib().ds_host_start_block(ctypes.byref(lapack_table()), G.data_ptr(), ny, b, float(ortho_tol), float(eps), lam.data_ptr(),
                                        coef.data_ptr(), cx.data_ptr(), ctypes.byref(amp), ctypes.byref(route)), "ds_host_start_block")
    return None if route.value else (lam, coef, cx, amp.value)


def host_polish(GK, coefs, GM, k):
    """ds_host_polish on host fp64 tensors: GK a list of (b x b) matrices, coefs their weights, GM (b x b): (E (k), C (b x b),
    qs ((len(GK) + 1) x k))."""
    b = GM.shape[0]
    gk = torch.stack([g_.contiguous() for g_ in GK]).contiguous()
    cf = torch.tensor([float(c) for c in coefs], dtype=torch.float64)
    GM = GM.contiguous()
    E = torch.empty(k, dtype=torch.float64)
    C = torch.empty((b, b), dtype=torch.float64)
    qs = torch.empty((len(GK) + 1, k), dtype=torch.float64)
    with blas_one_thread():
        check(lib().ds_host_polish(ctypes.byref(lapack_table()), len(GK), gk.data_ptr(), cf.data_ptr(), GM.data_ptr(), b, k, E.data_ptr(),
                                   C.data_ptr(), qs.data_ptr()), "ds_host_polish")
    return E, C, qs


def check(status, what):
    if status != 0:
        msg = lib().ds_last_error()
        raise RuntimeError(f"{what} failed (status {status}): {msg.decode() if msg else ''}")


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_ptr():
    """The calling thread's current HIP stream as a pointer-sized int (ctypes converts it for the c_void_p slot).
    The raw-stream query is the one compiled-kernel launchers use: ~0.3 us against ~9 us for building a
    torch.cuda.Stream object on every launch (a pass makes ~1000 launches per lane, under the interpreter lock)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def ptr(t):
    return None if t is None else t.data_ptr()


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("diffsound_amd: tensors must live on the HIP device (no CPU fallback)")


class Pattern:
    """Host-side symbolic BSR-3 pattern (see ds_pattern_build in include/diffsound_hip.h)."""

    def __init__(self, tets_cpu_i32, nv, nthreads=0):
        t = tets_cpu_i32
        if t.device.type != "cpu" or t.dtype != torch.int32 or not t.is_contiguous() or t.dim() != 2:
            raise ValueError("Pattern: tets must be a contiguous (T, N) int32 CPU tensor")
        handle = ctypes.c_void_p()
        check(lib().ds_pattern_build(ptr(t), t.shape[0], t.shape[1], nv, nthreads, ctypes.byref(handle)),
              "ds_pattern_build")
        try:
            a, b, c = _I64(), _I64(), _I64()
            check(lib().ds_pattern_sizes(handle, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)),
                  "ds_pattern_sizes")
            self.nv, self.nnzb, self.ncontrib = a.value, b.value, c.value
            self.rowptr = torch.empty(self.nv + 1, dtype=torch.int32)
            self.colidx = torch.empty(self.nnzb, dtype=torch.int32)
            self.diagidx = torch.empty(self.nv, dtype=torch.int32)
            self.cptr = torch.empty(self.nnzb + 1, dtype=torch.int32)
            self.clist = torch.empty(self.ncontrib, dtype=torch.int32)
            check(lib().ds_pattern_export(handle, ptr(self.rowptr), ptr(self.colidx), ptr(self.diagidx),
                                          ptr(self.cptr), ptr(self.clist)), "ds_pattern_export")
        finally:
            lib().ds_pattern_free(handle)


class DevicePattern:
    """Symbolic phase on the device (ds_dpattern_build): every table as an int32 tensor on ``tets``' device.
    ``utab`` (ngroups, 2) = chunk range of each group, ``ctab`` (nchunks, 4) = (e0, e1, b0, b1) per chunk."""

    def __init__(self, tets_i32, nv, cap):
        t = tets_i32
        if not t.is_cuda or t.dtype != torch.int32 or not t.is_contiguous() or t.dim() != 2:
            raise ValueError("DevicePattern: tets must be a contiguous (T, N) int32 HIP tensor")
        handle = ctypes.c_void_p()
        check(lib().ds_dpattern_build(ptr(t), t.shape[0], t.shape[1], nv, cap, stream_ptr(), ctypes.byref(handle)),
              "ds_dpattern_build")
        try:
            a, b, c, d, e = _I64(), _I64(), _I64(), _I64(), _I64()
            check(lib().ds_dpattern_sizes(handle, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c), ctypes.byref(d),
                                          ctypes.byref(e)), "ds_dpattern_sizes")
            self.nv, self.nnzb, self.ncontrib, self.ne, self.ngroups, self.nchunks = nv, a.value, b.value, c.value, d.value, e.value
            self.single = self.nchunks == self.ngroups  # every group is one chunk
            mk = lambda n: torch.empty(n, dtype=torch.int32, device=t.device)
            self.rowptr, self.colidx, self.diagidx = mk(nv + 1), mk(self.nnzb), mk(nv)
            self.cptr, self.clist = mk(self.nnzb + 1), mk(self.ncontrib)
            self.gptr, self.gent, self.goff, self.kperm = mk(self.ngroups + 1), mk(self.ne), mk(self.ne + 1), mk(self.nnzb)
            self.utab = torch.empty((self.ngroups, 2), dtype=torch.int32, device=t.device)
            self.ctab = torch.empty((self.nchunks, 4), dtype=torch.int32, device=t.device)
            check(lib().ds_dpattern_export(handle, ptr(self.rowptr), ptr(self.colidx), ptr(self.diagidx), ptr(self.cptr),
                                           ptr(self.clist), ptr(self.gptr), ptr(self.gent), ptr(self.goff),
                                           ptr(self.kperm), ptr(self.utab), ptr(self.ctab), stream_ptr()), "ds_dpattern_export")
            torch.cuda.current_stream(t.device).synchronize()  # the handle's arrays are freed below
        finally:
            lib().ds_dpattern_free(handle)


def edge_table(tets4, nv):
    """ds_edge_table: (T, 4) long HIP tensor -> (ea, eb, tet_edge): the distinct undirected edges (ea < eb, sorted)
    and the (T, 6) edge ids of every element's local edges, all long tensors on the same device."""
    t = tets4
    if not t.is_cuda or t.dtype != torch.int64 or t.dim() != 2 or t.shape[1] != 4:
        raise ValueError("edge_table: tets must be a (T, 4) int64 HIP tensor")
    t = t.contiguous()
    T = t.shape[0]
    ea = torch.empty(6 * T, dtype=torch.int64, device=t.device)
    eb = torch.empty(6 * T, dtype=torch.int64, device=t.device)
    te = torch.empty((T, 6), dtype=torch.int64, device=t.device)
    ne = _I64()
    with torch.cuda.device(t.device):
        check(lib().ds_edge_table(ptr(t), T, nv, ptr(ea), ptr(eb), ptr(te), ctypes.byref(ne), stream_ptr()), "ds_edge_table")
    return ea[:ne.value], eb[:ne.value], te


def unique_rows3(xyz):
    """ds_unique_rows3: (n, 3) fp32 HIP tensor -> (inv (n,), first (n_unique,)) as torch.unique(dim=0,
    return_inverse=True) numbers the rows; first[i] = lowest row index holding the i-th distinct coordinate."""
    x = xyz
    if not x.is_cuda or x.dtype != torch.float32 or x.dim() != 2 or x.shape[1] != 3:
        raise ValueError("unique_rows3: need an (n, 3) float32 HIP tensor")
    x = x.contiguous()
    n = x.shape[0]
    inv = torch.empty(n, dtype=torch.int64, device=x.device)
    first = torch.empty(n, dtype=torch.int64, device=x.device)
    nu = _I64()
    with torch.cuda.device(x.device):
        check(lib().ds_unique_rows3(ptr(x), n, ptr(inv), ptr(first), ctypes.byref(nu), stream_ptr()), "ds_unique_rows3")
    return inv, first[:nu.value]


class Groups:
    """Node groups of the neighbour-union SpMM (ds_groups_build): host arrays gptr, gent, goff, kperm (int32)."""

    def __init__(self, rowptr_cpu, colidx_cpu, nv):
        handle = ctypes.c_void_p()
        check(lib().ds_groups_build(ptr(rowptr_cpu), ptr(colidx_cpu), nv, ctypes.byref(handle)), "ds_groups_build")
        try:
            a, b = _I64(), _I64()
            check(lib().ds_groups_sizes(handle, ctypes.byref(a), ctypes.byref(b)), "ds_groups_sizes")
            self.ngroups, self.ne = a.value, b.value
            self.gptr = torch.empty(self.ngroups + 1, dtype=torch.int32)
            self.gent = torch.empty(self.ne, dtype=torch.int32)
            self.goff = torch.empty(self.ne + 1, dtype=torch.int32)
            self.kperm = torch.empty(colidx_cpu.numel(), dtype=torch.int32)
            check(lib().ds_groups_export(handle, ptr(self.gptr), ptr(self.gent), ptr(self.goff), ptr(self.kperm)),
                  "ds_groups_export")
        finally:
            lib().ds_groups_free(handle)


def union_chunks(gptr_cpu, goff_cpu, cap):
    """Chunk tables of the neighbour-union SpMM (ds_spmm_union): every group of 4 nodes (entries gptr[g]..gptr[g+1],
    blocks goff[e]..) is cut into chunks of whole entries with at most ``cap`` entries and ``cap`` blocks.  Returns
    (utab (ngroups, 2) int32 = chunk range of each group, ctab (nchunks, 4) int32 = (e0, e1, b0, b1)) as CPU tensors,
    or (None, None) if a single entry does not fit."""
    import numpy as np

    gp, go = gptr_cpu.numpy().astype(np.int64), goff_cpu.numpy().astype(np.int64)
    ng = gp.shape[0] - 1
    e0, e1 = gp[:-1], gp[1:]
    single = ((e1 - e0) <= cap) & ((go[e1] - go[e0]) <= cap)
    per_group = [None] * ng if not single.all() else None
    counts = np.ones(ng, dtype=np.int64)
    for gi in np.nonzero(~single)[0]:  # rare: a group whose union or block count exceeds the image
        e, rows = int(e0[gi]), []
        while e < e1[gi]:
            lim = min(int(e1[gi]), e + cap)  # furthest entry end with <= cap blocks and <= cap entries
            nxt = max(int(np.searchsorted(go[e:lim + 1], go[e] + cap, side="right")) - 1 + e, e + 1)
            rows.append((e, nxt, go[e], go[nxt]))
            e = nxt
        per_group[gi] = rows
        counts[gi] = len(rows)
    start = np.concatenate([[0], np.cumsum(counts)])
    ctab = np.empty((int(start[-1]), 4), dtype=np.int64)
    s_idx = start[:-1][single]
    ctab[s_idx] = np.stack([e0[single], e1[single], go[e0[single]], go[e1[single]]], 1)
    for gi in np.nonzero(~single)[0]:
        ctab[start[gi]:start[gi + 1]] = np.asarray(per_group[gi], dtype=np.int64)
    if ctab.shape[0] == 0 or int((ctab[:, 3] - ctab[:, 2]).max()) > cap:
        return None, None
    utab = np.stack([start[:-1], start[1:]], 1)
    return torch.from_numpy(utab.astype(np.int32)), torch.from_numpy(ctab.astype(np.int32))
