"""The C-ABI library loads on a CPU-only box and exports every symbol the header declares; the
host-side symbolic phase (ds_pattern_*) is exercised for real (it needs no GPU)."""
import os
import re

import numpy as np
import pytest
import scipy.sparse as sp
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _lib():
    from diffsound_amd import _hip

    if not os.path.exists(_hip.LIB_PATH):
        import __graft_entry__ as g

        g.build()
    return _hip


def test_header_symbols_exported():
    _hip = _lib()
    header = open(os.path.join(ROOT, "include", "diffsound_hip.h")).read()
    declared = set(re.findall(r"\b(ds_[a-z0-9_]+)\s*\(", header))
    declared = {d for d in declared if not d.endswith("_t")}  # type names followed by a parenthesis in comments
    assert declared, "no declarations parsed"
    lib = _hip.lib()
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/diffsound_hip.h but not exported"
    assert declared == set(_hip.EXPORTED_SYMBOLS), (declared ^ set(_hip.EXPORTED_SYMBOLS))
    assert lib.ds_abi_version() == _hip.ABI_VERSION
    assert lib.ds_last_error() is not None


def _clique_pattern(t, nv):
    N = t.shape[1]
    pairs = np.unique(np.stack([np.repeat(t, N, axis=1).reshape(-1), np.tile(t, (1, N)).reshape(-1)], 1), axis=0)
    ref = sp.csr_matrix((np.ones(len(pairs)), (pairs[:, 0], pairs[:, 1])), shape=(nv, nv))
    ref.sort_indices()
    return ref


@pytest.mark.parametrize("order", [1, 2])
def test_pattern_build(order):
    _hip = _lib()
    from diffsound_amd import meshgen
    from oracle import fem

    v, t = meshgen.kuhn_box(3)
    v, t = fem.to_high_order(torch.from_numpy(v), torch.from_numpy(t).long(), order)
    nv = v.shape[0]
    for threads in (1, 3):
        pat = _hip.Pattern(t.to(torch.int32).contiguous(), nv, nthreads=threads)
        ref = _clique_pattern(t.numpy(), nv)
        assert pat.nnzb == ref.nnz
        assert np.array_equal(pat.rowptr.numpy(), ref.indptr)
        assert np.array_equal(pat.colidx.numpy(), ref.indices)
        N = t.shape[1]
        cl = pat.clist.numpy().astype(np.int64)
        slot = np.repeat(np.arange(pat.nnzb), np.diff(pat.cptr.numpy()))
        te, a, b = cl // (N * N), (cl % (N * N)) // N, cl % N
        rows = np.repeat(np.arange(nv), np.diff(pat.rowptr.numpy()))
        tn = t.numpy()
        assert np.array_equal(tn[te, a], rows[slot]) and np.array_equal(tn[te, b], pat.colidx.numpy()[slot])
        assert np.array_equal(np.sort(cl), np.arange(len(cl)))  # every contribution exactly once
        assert all(np.all(np.diff(cl[s:e]) > 0) for s, e in zip(pat.cptr.numpy()[:50], pat.cptr.numpy()[1:51]))
        assert np.array_equal(pat.colidx.numpy()[pat.diagidx.numpy()], np.arange(nv))


def test_pattern_errors_are_loud():
    _hip = _lib()
    bad = torch.zeros((2, 5), dtype=torch.int32)
    with pytest.raises(RuntimeError, match="N must be 4 or 10"):
        _hip.Pattern(bad, 4)
    oob = torch.tensor([[0, 1, 2, 9]], dtype=torch.int32)
    with pytest.raises(RuntimeError, match="out of range"):
        _hip.Pattern(oob, 4)
    with pytest.raises(ValueError):
        _hip.Pattern(torch.zeros((2, 4), dtype=torch.int64), 4)


def test_no_cpu_fallback():
    """Product entry points refuse CPU tensors instead of silently computing on the host."""
    from diffsound_amd.diffelastic.diff_model import DiffSoundObj
    from diffsound_amd.modal_ops import TetSystem

    v = torch.rand((8, 3))
    t = torch.tensor([[0, 1, 2, 3]])
    with pytest.raises(RuntimeError, match="HIP"):
        TetSystem(v, t, 1, 1000.0)
    with pytest.raises(RuntimeError, match="HIP"):
        DiffSoundObj(vertices=v, tets=t, mode_num=2)


def _gfx950_code_objects(lib_path):
    """The gfx950 code objects inside a HIP shared library: the .hip_fatbin section is a sequence of clang offload
    bundles (one per translation unit: magic, entry count, then (offset, size, triple) records)."""
    import struct

    data = open(lib_path, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    pos, out = data.find(magic), []
    while pos >= 0:
        (n,) = struct.unpack_from("<Q", data, pos + len(magic))
        p = pos + len(magic) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", data, p)
            triple = data[p + 24:p + 24 + tlen].decode()
            p += 24 + tlen
            if "gfx950" in triple and size:
                out.append(data[pos + off:pos + off + size])
        pos = data.find(magic, pos + len(magic))
    return out


def test_library_issues_no_packed_fp32_and_no_double_rate_bf16_mfma(tmp_path):
    """gfx950: a v_mfma_f32_16x16x32_bf16 in flight in another wave of a compute unit changes results of packed-FP32
    instructions (tests/probes/mfma_probe.hip, profiles/r03_mfma_interference_matrix.txt).  The library is therefore
    built without packed FP32 (so that nothing another stream or process runs can corrupt it) and issues the 16x16x16
    form of the bf16 MFMA (so that it corrupts nobody else).  Checked on the ISA of the shipped binary."""
    import shutil
    import subprocess

    objdump = shutil.which("llvm-objdump") or "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("llvm-objdump not found")
    from diffsound_amd import _hip

    objs = _gfx950_code_objects(_hip.LIB_PATH)
    assert len(objs) >= 8  # one per .hip translation unit
    counts = {"v_pk_fma_f32": 0, "v_pk_mul_f32": 0, "v_pk_add_f32": 0, "v_mfma_f32_16x16x32_bf16": 0,
              "v_mfma_f32_16x16x16_bf16": 0, "v_fma_f32": 0}
    for i, blob in enumerate(objs):
        path = tmp_path / f"co{i}.o"
        path.write_bytes(blob)
        text = subprocess.run([objdump, "-d", str(path)], capture_output=True, text=True, check=True).stdout
        for k in counts:
            counts[k] += text.count(k + " ")
    assert counts["v_pk_fma_f32"] == counts["v_pk_mul_f32"] == counts["v_pk_add_f32"] == 0, counts
    assert counts["v_mfma_f32_16x16x32_bf16"] == 0 and counts["v_mfma_f32_16x16x16_bf16"] > 100, counts
    assert counts["v_fma_f32"] > 1000, counts


def test_mfma_spmm_window_registers_do_not_move_between_issue_sites(tmp_path):
    """The bf16 term kernel (csrc/spmm_mfma.inc) keeps a batch's panel loads in flight across its loop: the window registers are
    written by loads issued in front of the loop (first batch) and inside it (next batch), and read behind counted waits.  That
    only works if BOTH issue sites name the same registers - a compiler that peels a trip off the loop, or joins the sites
    through copies, moves values whose loads have not landed (seen in round 5 with the tail form at 84 columns: wrong results, no
    fault).  Checked on the ISA of the shipped binary: per instantiation one loop's worth of sites, identical destinations."""
    import re
    import shutil
    import subprocess

    objdump = shutil.which("llvm-objdump") or "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("llvm-objdump not found")
    from diffsound_amd import _hip

    seen = 0
    for i, blob in enumerate(_gfx950_code_objects(_hip.LIB_PATH)):
        if b"spmm_union_mfma_kernel" not in blob:
            continue
        path = tmp_path / f"co{i}.o"
        path.write_bytes(blob)
        text = subprocess.run([objdump, "-d", str(path)], capture_output=True, text=True, check=True).stdout
        for m in re.finditer(r"<(_ZN12_GLOBAL__N_122spmm_union_mfma_kernelILi8ELi(\d)ELi(\d)ELb([01])ELi(\d+)ELi([01])ELi(\d)E[^>]*)>:\n(.*?)s_endpgm",
                             text, re.S):
            batch, tail, body = int(m.group(5)), int(m.group(7)), m.group(8)
            dst = re.findall(r"buffer_load_dwordx2 (v\[\d+:\d+\])", body)
            win = batch + tail
            assert len(dst) >= 2 * win, (m.group(1), len(dst))
            assert dst[:win] == dst[win:2 * win], (m.group(1), dst[:win], dst[win:2 * win])
            assert len(set(dst[:win])) == win, m.group(1)
            seen += 1
    assert seen >= 36, seen  # 6 widths x 3 epilogue forms x 2 levels, plus the tail forms


@pytest.mark.parametrize("n,m", [(240, 80), (160, 56), (96, 32), (20, 8)])
def test_host_dense_steps_of_the_native_solver_loop(n, m):
    """ds_selftest_dense (ABI 31): the dense steps ds_lobpcg_iterate runs on the host between its launches - the staged
    eigensolver that back-transforms only the wanted third of the Ritz vectors (dsytrd + dstedc + dormtr on m columns) against
    dsyevd, the Cholesky factor and its inverse on four-accumulator dot products, the twice-applied Cholesky-QR - checked on the
    CPU with SciPy's LAPACK, the table the product hands in."""
    import ctypes

    _hip = _lib()
    errs = (ctypes.c_double * 6)()
    with _hip.blas_one_thread():
        _hip.check(_hip.lib().ds_selftest_dense(ctypes.byref(_hip.lapack_table()), n, m, 7, errs), "ds_selftest_dense")
    e = list(errs)
    assert e[0] < 1e-13 and e[1] < 1e-13, e     # the same eigenvalues as dsyevd; G z = w z for the m lowest
    assert e[2] < 1e-13 and e[3] < 1e-11, e     # L L^T = G, L^-1 L = I
    assert e[4] < 1e-12, e                      # Q^T Q = I
    assert e[5] == (1.0 if n >= 32 else 0.0)    # SciPy's table carries the stages; tiny problems call dsyevd
    tbl = _hip.lapack_table()
    partial = _hip.LapackTable(tbl.dsyevd, tbl.dgemm, tbl.dsytrd, None, None)
    assert _hip.lib().ds_selftest_dense(ctypes.byref(_hip.LapackTable(tbl.dsyevd, tbl.dgemm, None, None, None)), n, m, 7, errs) == 0
    assert errs[5] == 0.0 and errs[1] < 1e-13   # no stages in the table: dsyevd, the same answer
    del partial


def test_lapack_table_has_a_source_without_scipy():
    """the product must not DEPEND on SciPy's private capsule table.  _hip.lapack_table() prefers SciPy's OpenBLAS
    (faster, and it has the stages of dsyevd) and falls back to the MKL entry points libtorch_cpu.so exports; both tables pass
    the dense self-check, and the process-wide table can be chosen by name (DS_LAPACK)."""
    import ctypes

    _hip = _lib()
    for src in ("scipy", "torch"):
        tbl = _hip.lapack_table(src)
        assert tbl.dsyevd and tbl.dgemm
        errs = (ctypes.c_double * 6)()
        with _hip.blas_one_thread():
            _hip.check(_hip.lib().ds_selftest_dense(ctypes.byref(tbl), 96, 32, 11, errs), "ds_selftest_dense")
        assert errs[1] < 1e-13 and errs[2] < 1e-13 and errs[4] < 1e-12, (src, list(errs))
        assert errs[5] == (1.0 if src == "scipy" else 0.0)  # (MKL's symbols in libtorch carry no stages: dsyevd then)
    assert _hip.lapack_table() is not None and _hip.lapack_source() in ("scipy", "torch")
    with pytest.raises(ValueError, match="unknown source"):
        _hip.lapack_table("netlib")


def test_native_start_block_and_polish_match_the_python_forms():
    """ds_host_start_block / ds_host_polish (ABI 31) against the torch forms they replace on the device path
    (lobpcg/modal_solver.py: `start` in ModalSolver.solve, `small` in ModalSolver._polish): the same Ritz values, the same
    coefficient matrices up to the sign of an eigenvector, the same quadratic forms - on a synthetic start block with a rigid
    block in front, and the explicit-route signal for a block that one sweep cannot orthonormalise."""
    import torch

    from diffsound_amd.lobpcg.modal_solver import _orthonormalizer_q, _sym

    _hip = _lib()
    rng = np.random.default_rng(5)
    n, ny, b, k = 400, 16, 24, 20
    Bq = rng.standard_normal((n, n))
    M = Bq @ Bq.T / n + np.eye(n)
    Kq = rng.standard_normal((n, n))
    K = Kq @ Kq.T + 5 * np.eye(n)
    Y = np.linalg.qr(rng.standard_normal((n, 6)))[0]
    Y = Y @ np.linalg.inv(np.linalg.cholesky(Y.T @ M @ Y)).T           # M-orthonormal rigid block, 6 vectors + 10 zero columns
    Y = np.concatenate([Y, np.zeros((n, ny - 6))], 1)
    K = K - K @ Y[:, :6] @ np.linalg.solve(Y[:, :6].T @ K @ Y[:, :6], Y[:, :6].T @ K)  # K Y = 0
    K = 0.5 * (K + K.T)
    X0 = rng.standard_normal((n, b))
    S = np.concatenate([Y, X0], 1)
    G = torch.from_numpy(np.concatenate([S.T @ K @ X0, S.T @ M @ X0], 1))
    got = _hip.host_start_block(G, ny, b, 2e-6, 1.1e-16)
    assert got is not None
    lam, coef, cx, amp = got
    # the Python form (as in ModalSolver.solve)
    Gyk, Cy = G[:ny, :b], G[:ny, b:]
    A, B0 = _sym(G[ny:, :b]), _sym(G[ny:, b:])
    CtC = Cy.T @ Cy
    T, amp_py = _orthonormalizer_q(torch.cat([B0 - CtC, CtC.diagonal()[None, :]], 0))
    A1 = A - Cy.T @ Gyk - Gyk.T @ Cy
    E_, Z_ = torch.linalg.eigh(_sym(T.T @ A1 @ T))
    assert torch.allclose(lam, E_, rtol=1e-11) and abs(amp / amp_py - 1) < 1e-12
    Cx_py = T @ Z_
    sign = torch.sign((cx * Cx_py).sum(0))
    assert torch.allclose(cx * sign[None, :], Cx_py, atol=1e-9 * float(Cx_py.abs().max()))
    assert torch.allclose(coef[ny:], cx) and torch.allclose(coef[:ny], -(Cy @ cx), atol=1e-12)
    Xn = S @ coef.numpy()                                                # the new block: M-orthonormal, M-orthogonal to Y, K-diagonal
    assert np.abs(Xn.T @ M @ Xn - np.eye(b)).max() < 1e-9 and np.abs(Y.T @ M @ Xn).max() < 1e-9
    assert np.abs(Xn.T @ K @ Xn - np.diag(lam.numpy())).max() < 1e-8 * float(lam.max())
    # a block with two equal columns: the explicit route
    Xd = X0.copy()
    Xd[:, 1] = Xd[:, 0]
    Gd = torch.from_numpy(np.concatenate([np.concatenate([Y, Xd], 1).T @ K @ Xd, np.concatenate([Y, Xd], 1).T @ M @ Xd], 1))
    assert _hip.host_start_block(Gd, ny, b, 2e-6, 6e-8) is None
    # polish: two terms of K, the mass Gram matrix
    Xc = Xn[:, :b]
    K1 = 0.3 * K + np.diag(rng.uniform(0, 1, n))
    K2 = K - 0.5 * K1
    GK = [torch.from_numpy(Xc.T @ K1 @ Xc), torch.from_numpy(Xc.T @ K2 @ Xc)]
    GM = torch.from_numpy(Xc.T @ M @ Xc)
    cf = [0.5, 1.0]
    E, C, qs = _hip.host_polish(GK, cf, GM, k)
    GA = _sym(cf[0] * GK[0] + cf[1] * GK[1])
    L = torch.linalg.cholesky(_sym(GM))
    Li = torch.linalg.solve_triangular(L, torch.eye(b, dtype=torch.float64), upper=False)
    Ep, Zt = torch.linalg.eigh(_sym(Li @ GA @ Li.T))
    assert torch.allclose(E, Ep[:k], rtol=1e-11)
    Cp = Li.T @ Zt
    sg = torch.sign((C * Cp).sum(0))
    assert torch.allclose(C * sg[None, :], Cp, atol=1e-9 * float(Cp.abs().max()))
    Ck = C[:, :k]
    for t_, Gt in enumerate(GK + [GM]):
        assert torch.allclose(qs[t_], ((Ck.T @ _sym(Gt)) * Ck.T).sum(1), rtol=1e-10)
    assert torch.allclose(C.T @ _sym(GM) @ C, torch.eye(b, dtype=torch.float64), atol=1e-10)
    with pytest.raises(RuntimeError, match="not positive definite"):
        _hip.host_polish(GK, cf, GM - 10 * torch.eye(b, dtype=torch.float64), k)
