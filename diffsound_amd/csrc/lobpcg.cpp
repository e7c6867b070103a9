// Native driver of the block eigensolver's iteration - the loop of ModalSolver.solve (lobpcg/modal_solver.py) as ONE
// call per eigensolve: residual test, hard locking, preconditioner (ds_twolevel_apply or the one-level Chebyshev
// polynomial), single-sweep projected Cholesky-QR, Rayleigh-Ritz by recurrence, the [X' P'] = [X P W] [Z1 Zp] updates.
// It issues the same kernels (ds_spmm_union, ds_gram, ds_mix, ds_residual, ...) on the caller's stream and does the
// <= 3b x 3b dense steps on the calling host thread (LAPACK dsyevd / BLAS dgemm supplied by the embedding application
// through ds_lapack_t - the Python binding passes SciPy's), so a hypothesis lane spends its whole solve outside the
// interpreter: no interpreter lock between its launches, none between the lanes.
//
// Reference: the iteration of src/lobpcg/_lobpcg.py:344-376, 433-477 (LOBPCG.run / _update_ortho), whose per-iteration
// host synchronisations (float(torch.norm(..)) :655,663; the Python loop over a device tensor :323-328) are the two
// small device-to-host copies of this loop (residual norms, Gram blocks).  Semantics are those of the Python loop it
// replaces, statement for statement; tests/test_modal_gpu.py runs both and compares.
#include <atomic>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <vector>
#include <chrono>
#include <cstdlib>

#include "ds_common.h"

namespace {
// EXPERIMENT: host-time breakdown of one solve (DS_EXP_TIMING=1)
struct Tm {
    double sync = 0, eigh = 0, dense = 0, total = 0;
    int nsync = 0, neigh = 0;
};
thread_local Tm g_tm;

inline double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// How the host thread of a solve waits for its stream (ds_host_wait_mode; process-wide).  0: hipStreamSynchronize - under the
// runtime's default policy a SPIN whenever the host has more cores than devices.  1: a short poll, then a sleep on an event created
// with hipEventBlockingSync.  Eight hypothesis lanes spend four fifths of their time waiting for the device: spinning, they hold
// eight cores for nothing, and on a host that has only a few to give (a container under CPU quota beside busy neighbours) the
// lanes' dense steps queue behind the spinners - 52 passes/s against 60 with sleeping waits on four cores; with cores to spare
// the wake-up latency of a sleep costs 2 % (profiles/r05_host_wait_mode.txt).  One solve at a time keeps the spin.
std::atomic<int> g_wait_mode{0};

// One blocking event per (host thread, device), created on the device of the stream it first records and destroyed when the
// thread ends (a single thread_local event belonged to whatever device was current at its creation - a thread that
// later solved on a stream of another device got hipErrorInvalidResourceHandle - and was never destroyed: one leaked event per
// replaced lane pool).
struct WaitEvents {
    static constexpr int MAX_DEV = 64;
    hipEvent_t ev[MAX_DEV] = {};
    ~WaitEvents() {
        for (hipEvent_t e : ev)
            if (e) (void)hipEventDestroy(e);
    }
};
thread_local WaitEvents g_wait_events;

thread_local int g_solve_wait_mode = -1;  // wait mode of the solve running on this thread (ds_lobpcg_t.wait_mode); -1: the process default

hipError_t wait_for_stream(hipStream_t st) {
    const int mode = g_solve_wait_mode >= 0 ? g_solve_wait_mode : g_wait_mode.load(std::memory_order_relaxed);
    if (mode == 0) return hipStreamSynchronize(st);
    int dev = 0;
    hipError_t e = hipStreamGetDevice(st, &dev);  // (the stream's own device, not the thread's current one)
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= WaitEvents::MAX_DEV) return hipStreamSynchronize(st);
    hipEvent_t& ev = g_wait_events.ev[dev];
    if (!ev) {
        int cur = 0;
        if ((e = hipGetDevice(&cur)) != hipSuccess) return e;
        if (cur != dev && (e = hipSetDevice(dev)) != hipSuccess) return e;
        e = hipEventCreateWithFlags(&ev, hipEventBlockingSync | hipEventDisableTiming);
        if (cur != dev) (void)hipSetDevice(cur);
        if (e != hipSuccess) return e;
    }
    e = hipEventRecord(ev, st);
    if (e != hipSuccess) return e;
    const double t0 = now_s();
    while ((e = hipEventQuery(ev)) == hipErrorNotReady && now_s() - t0 < 20e-6) {}  // (what is about to finish needs no sleep)
    if (e != hipErrorNotReady) return e;
    return hipEventSynchronize(ev);
}

// ---------------------------------------------------------------------------------------------------------------
// small dense algebra, row-major double
struct Mat {
    int r = 0, c = 0;
    std::vector<double> a;
    Mat() = default;
    Mat(int r_, int c_) : r(r_), c(c_), a((size_t)r_ * c_, 0.0) {}
    double& operator()(int i, int j) { return a[(size_t)i * c + j]; }
    double operator()(int i, int j) const { return a[(size_t)i * c + j]; }
};

void symmetrize(Mat& G) {
    for (int i = 0; i < G.r; ++i)
        for (int j = i + 1; j < G.c; ++j) {
            const double v = 0.5 * (G(i, j) + G(j, i));
            G(i, j) = G(j, i) = v;
        }
}

bool all_finite(const Mat& G) {
    for (double v : G.a)
        if (!std::isfinite(v)) return false;
    return true;
}

// C = op(A) op(B), row-major, through column-major dgemm: C^T = op(B)^T op(A)^T
Mat gemm(const ds_lapack_t& la, const Mat& A, bool ta, const Mat& B, bool tb) {
    const int m = ta ? A.c : A.r, k = ta ? A.r : A.c, n = tb ? B.r : B.c;
    Mat C(m, n);
    if (m == 0 || n == 0 || k == 0) return C;
    struct T_ { double t0 = now_s(); ~T_() { g_tm.dense += now_s() - t0; } } t_;
    char opb = tb ? 'T' : 'N', opa = ta ? 'T' : 'N';
    int mm = n, nn = m, kk = k, ldb = B.c, lda = A.c, ldc = n;
    double one = 1.0, zero = 0.0;
    la.dgemm(&opb, &opa, &mm, &nn, &kk, &one, const_cast<double*>(B.a.data()), &ldb, const_cast<double*>(A.a.data()), &lda,
             &zero, C.a.data(), &ldc);
    return C;
}

// sum of a[k] * b[k * sb], k < n, on four independent accumulators: the compiler keeps a floating-point reduction in its
// source order, i.e. one dependent chain of 4-cycle multiply-adds - the factorisations below spent 0.07 ms each on an 80 x 80
// block that way, seven of them per iteration: most of the "rest" of profiles/r05_host_time_one_lane.txt (0.4 ms per iteration)
inline double dot4(const double* a, const double* b, int sb, int n) {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int k = 0;
    for (; k + 4 <= n; k += 4) {
        s0 += a[k] * b[(size_t)k * sb];
        s1 += a[k + 1] * b[(size_t)(k + 1) * sb];
        s2 += a[k + 2] * b[(size_t)(k + 2) * sb];
        s3 += a[k + 3] * b[(size_t)(k + 3) * sb];
    }
    for (; k < n; ++k) s0 += a[k] * b[(size_t)k * sb];
    return (s0 + s1) + (s2 + s3);
}

// lower Cholesky factor of a symmetric matrix (row-major L, L L^T = A); false on breakdown
bool cholesky(const Mat& A, Mat& L) {
    const int n = A.r;
    L = Mat(n, n);
    double* l = L.a.data();
    for (int j = 0; j < n; ++j) {
        const double d = A(j, j) - dot4(l + (size_t)j * n, l + (size_t)j * n, 1, j);
        if (!(d > 0.0) || !std::isfinite(d)) return false;
        const double ljj = std::sqrt(d);
        L(j, j) = ljj;
        for (int i = j + 1; i < n; ++i) L(i, j) = (A(i, j) - dot4(l + (size_t)i * n, l + (size_t)j * n, 1, j)) / ljj;
    }
    return all_finite(L);
}

Mat lower_inverse(const Mat& L) {
    const int n = L.r;
    Mat X(n, n);
    const double* l = L.a.data();
    double* x = X.a.data();
    for (int j = 0; j < n; ++j) {
        X(j, j) = 1.0 / L(j, j);
        for (int i = j + 1; i < n; ++i)  // (X's column j is walked with stride n: 80 x 80 doubles sit in the L1)
            X(i, j) = -dot4(l + (size_t)i * n + j, x + (size_t)j * n + j, n, i - j) / L(i, i);
    }
    return X;
}

// eigenvalues ascending, Z columns = eigenvectors (row-major Z); returns false when LAPACK reports failure
bool eigh(const ds_lapack_t& la, const Mat& Gsym, std::vector<double>& w, Mat& Z) {
    struct T_ { double t0 = now_s(); ~T_() { g_tm.eigh += now_s() - t0; ++g_tm.neigh; } } t_;
    const int n = Gsym.r;
    std::vector<double> A = Gsym.a;  // symmetric: row-major == column-major
    w.assign(n, 0.0);
    char jobz = 'V', uplo = 'L';
    int nn = n, lda = n, info = 0, lwork = -1, liwork = -1, iq = 0;
    double wq = 0.0;
    la.dsyevd(&jobz, &uplo, &nn, A.data(), &lda, w.data(), &wq, &lwork, &iq, &liwork, &info);
    if (info != 0) return false;
    lwork = (int)wq;
    liwork = iq;
    std::vector<double> work((size_t)std::max(1, lwork));
    std::vector<int> iwork((size_t)std::max(1, liwork));
    la.dsyevd(&jobz, &uplo, &nn, A.data(), &lda, w.data(), work.data(), &lwork, iwork.data(), &liwork, &info);
    if (info != 0) return false;
    Z = Mat(n, n);  // column-major eigenvector j = A[j * n + i]  ->  Z(i, j)
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i) Z(i, j) = A[(size_t)j * n + i];
    return true;
}

// The LOWEST m eigenpairs of a symmetric matrix: w = all n eigenvalues ascending, Zm (n x m, row-major) = the first m eigenvectors.
// The Ritz step wants a third of the vectors of its 3 na x 3 na problem.  dsyevd is tridiagonalisation + divide and conquer on the
// tridiagonal matrix + back-transformation of ALL n vectors (dormtr, 2 n^3 flops); with the three stages called one by one only
// the wanted m columns are back-transformed: 13 % of the call at n = 240, m = 80 (the divide-and-conquer stage, which has no
// subset form, is half of it).  (dsyevr / dsyevx on the index range and the MRRR tridiagonal solver dstemr are slower in SciPy's
// OpenBLAS: profiles/r05_host_eigh_probe.txt, profiles/r06_host_eigh_stages.txt.)  Falls back to dsyevd when the table has no stages.
bool eigh_lowest(const ds_lapack_t& la, const Mat& Gsym, int m, std::vector<double>& w, Mat& Zm) {
    const int n = Gsym.r;
    if (!la.dsytrd || !la.dstedc || !la.dormtr || m >= n || n < 32) {
        Mat Z;
        if (!eigh(la, Gsym, w, Z)) return false;
        Zm = Mat(n, std::min(m, n));
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < Zm.c; ++j) Zm(i, j) = Z(i, j);
        return true;
    }
    struct T_ { double t0 = now_s(); ~T_() { g_tm.eigh += now_s() - t0; ++g_tm.neigh; } } t_;
    thread_local std::vector<double> A, e, tau, work, Zt;
    thread_local std::vector<int> iwork;
    A = Gsym.a;  // symmetric: row-major == column-major
    w.assign(n, 0.0);
    e.assign(n, 0.0), tau.assign(n, 0.0);
    int nn = n, lda = n, info = 0;
    int lwork = std::max(64 * n, 1 + 4 * n + n * n), liwork = 3 + 5 * n;
    if ((int)work.size() < lwork) work.resize(lwork);
    if ((int)iwork.size() < liwork) iwork.resize(liwork);
    if (Zt.size() < (size_t)n * n) Zt.resize((size_t)n * n);
    char lo = 'L', compz = 'I', side = 'L', notr = 'N';
    la.dsytrd(&lo, &nn, A.data(), &lda, w.data(), e.data(), tau.data(), work.data(), &lwork, &info);
    if (info != 0) return false;
    la.dstedc(&compz, &nn, w.data(), e.data(), Zt.data(), &lda, work.data(), &lwork, iwork.data(), &liwork, &info);
    if (info != 0) return false;
    int mm = m;
    la.dormtr(&side, &lo, &notr, &nn, &mm, A.data(), &lda, tau.data(), Zt.data(), &lda, work.data(), &lwork, &info);
    if (info != 0) return false;
    Zm = Mat(n, m);
    for (int j = 0; j < m; ++j)
        for (int i = 0; i < n; ++i) Zm(i, j) = Zt[(size_t)j * n + i];
    return true;
}

// T with (W T)^T M (W T) = I given G = W^T M W, clamped-eigenvalue form (ModalSolver._svqb_transform)
bool svqb_transform(const ds_lapack_t& la, const Mat& G, Mat& T) {
    const int n = G.r;
    std::vector<double> d(n);
    for (int i = 0; i < n; ++i) d[i] = 1.0 / std::sqrt(std::max(G(i, i), 1e-300));
    Mat Gs = G;
    symmetrize(Gs);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) Gs(i, j) *= d[i] * d[j];
    std::vector<double> E;
    Mat Z;
    if (!eigh(la, Gs, E, Z)) return false;
    double emax = 0.0;
    for (double e : E) emax = std::max(emax, std::fabs(e));
    T = Mat(n, n);
    for (int j = 0; j < n; ++j) {
        const double e = std::max(E[j], 1e-12 * emax);
        const double s = 1.0 / std::sqrt(e);
        for (int i = 0; i < n; ++i) T(i, j) = d[i] * Z(i, j) * s;
    }
    return true;
}

// (T, amp) of ModalSolver._orthonormalizer_q; rem: optional squared M-norms removed by the preceding projection
bool orthonormalizer_q(const ds_lapack_t& la, const Mat& Gin, const std::vector<double>* rem, Mat& T, double& amp) {
    Mat G = Gin;
    symmetrize(G);
    const int n = G.r;
    std::vector<double> diag(n), d(n);
    for (int i = 0; i < n; ++i) {
        diag[i] = std::max(G(i, i), 1e-300);
        d[i] = 1.0 / std::sqrt(diag[i]);
    }
    Mat Gs = G;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) Gs(i, j) *= d[i] * d[j];
    Mat L;
    if (!cholesky(Gs, L)) {
        amp = std::numeric_limits<double>::infinity();
        return svqb_transform(la, G, T);
    }
    double lmin = std::numeric_limits<double>::infinity();
    for (int i = 0; i < n; ++i) lmin = std::min(lmin, L(i, i));
    amp = 1.0 / std::max(lmin, 1e-300);
    if (rem)
        for (int i = 0; i < n; ++i) amp = std::max(amp, std::sqrt((*rem)[i] / diag[i]));
    const Mat Li = lower_inverse(L);
    T = Mat(n, n);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) T(i, j) = d[i] * Li(j, i);
    return true;
}

// orthonormal basis (Euclidean) of the columns of Tm: scaled Cholesky-QR twice (ModalSolver._orthonormal_columns)
bool orthonormal_columns(const ds_lapack_t& la, const Mat& Tm, Mat& Q) {
    Q = Tm;
    for (int pass = 0; pass < 2; ++pass) {
        Mat G = gemm(la, Q, true, Q, false);
        const int n = G.r;
        std::vector<double> d(n);
        for (int i = 0; i < n; ++i) d[i] = 1.0 / std::sqrt(std::max(G(i, i), 1e-300));
        symmetrize(G);
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) G(i, j) *= d[i] * d[j];
        Mat L;
        if (!cholesky(G, L)) return false;  // (the Python loop falls back to Householder QR: the caller does, too)
        const Mat Li = lower_inverse(L);
        Mat X(n, n);
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) X(i, j) = d[i] * Li(j, i);
        Q = gemm(la, Q, false, X, false);
    }
    return true;
}

// column ranges of at most 84 columns (multiples of 4, as equal as possible) that tile a c-column block: what one launch of
// the neighbour-union kernels takes (the same rule as _HipBlockOps.col_slices of the Python binding)
template <typename F>
int for_col_slices(int c, F&& fn) {
    if (c <= 84) return fn(0, c);
    const int k = (c + 83) / 84;
    const int w = (((c + k - 1) / k) + 3) / 4 * 4;
    for (int c0 = 0; c0 < c; c0 += w) {
        const int rc = fn(c0, std::min(c, c0 + w));
        if (rc != DS_OK) return rc;
    }
    return DS_OK;
}

// ---------------------------------------------------------------------------------------------------------------
constexpr int COEF_SLOTS = 8;
constexpr int LAM_SLOT = 256;  // doubles of the pinned staging area kept for the Ritz values (>= the widest block, 160)

struct PinnedRing {  // per host thread, grown on demand, released when the thread ends
    ~PinnedRing() {
        if (host) (void)hipHostFree(host);
        if (back) (void)hipHostFree(back);
    }
    float* host = nullptr;
    size_t slot_floats = 0;
    double* back = nullptr;  // device-to-host staging (Gram blocks, residual norms): a pageable destination would
    size_t back_doubles = 0; // send the "asynchronous" copy through the runtime's shared staging path
    int reserve_back(size_t doubles) {
        if (doubles <= back_doubles) return DS_OK;
        if (back) (void)hipHostFree(back);
        back = nullptr, back_doubles = 0;
        int rc = ds::check_hip(hipHostMalloc(reinterpret_cast<void**>(&back), doubles * sizeof(double), 0),
                               "hipHostMalloc(result staging)");
        if (rc == DS_OK) back_doubles = doubles;
        return rc;
    }
    int reserve(size_t floats_per_slot) {
        if (floats_per_slot <= slot_floats) return DS_OK;
        if (host) (void)hipHostFree(host);
        host = nullptr, slot_floats = 0;
        int rc = ds::check_hip(hipHostMalloc(reinterpret_cast<void**>(&host), floats_per_slot * COEF_SLOTS * sizeof(float), 0),
                               "hipHostMalloc(coefficient staging)");
        if (rc == DS_OK) slot_floats = floats_per_slot;
        return rc;
    }
};
thread_local PinnedRing g_ring;

struct Ctx {
    ds_lobpcg_t* p;
    hipStream_t st;
    ds_stream_t stream;
    const ds_lapack_t& la;
    float *S, *S2, *KS, *KS2;

    int hip(hipError_t e, const char* what) { return ds::check_hip(e, what); }

    int gram(const float* A, int64_t lda, int pc, const float* B, int64_t ldb, int qc, bool sym, Mat& out) {
        const int flags = (sym ? DS_GRAM_SYMMETRIC : 0) | (p->gram_exact ? DS_GRAM_EXACT : 0);
        int rc = ds_gram(A, DS_F32, lda, pc, B, DS_F32, ldb, qc, p->n, flags, p->gbuf, p->gram_work, p->gram_work_bytes,
                         stream);
        if (rc != DS_OK) return rc;
        out = Mat(pc, qc);
        double* stage = ring->back + 2048 + LAM_SLOT;
        rc = hip(hipMemcpyAsync(stage, p->gbuf, sizeof(double) * (size_t)pc * qc, hipMemcpyDeviceToHost, st),
                 "ds_lobpcg_iterate: Gram block to host");
        if (rc != DS_OK) return rc;
        { const double t0 = now_s(); rc = hip(wait_for_stream(st), "ds_lobpcg_iterate: stream synchronise"); g_tm.sync += now_s() - t0; ++g_tm.nsync; }
        if (rc == DS_OK) std::memcpy(out.a.data(), stage, sizeof(double) * (size_t)pc * qc);
        return rc;
    }

    // coefficient matrices travel through a ring of pinned host slots and matching device slots: an asynchronous copy
    // may read its source after the call returns, and a slot is reused only after >= 8 further updates - every
    // iteration synchronises the stream (Gram blocks) at least twice in between
    int mix(const float* A, int64_t lda, int pc, const Mat& C, float* Out, int64_t ldo, float alpha = 1.f, float beta = 0.f) {
        const size_t cnt = (size_t)C.r * C.c;
        const size_t stride = (size_t)(p->ny + 3 * p->b) * 2 * p->b;  // slot size of THIS solve (the ring may be larger)
        float* h = ring->host + (size_t)slot * stride;
        float* dv = p->cbuf + (size_t)slot * stride;
        slot = (slot + 1) % COEF_SLOTS;
        for (size_t i = 0; i < cnt; ++i) h[i] = (float)C.a[i];
        int rc = hip(hipMemcpyAsync(dv, h, sizeof(float) * cnt, hipMemcpyHostToDevice, st),
                     "ds_lobpcg_iterate: coefficients to device");
        if (rc != DS_OK) return rc;
        return ds_mix(A, lda, pc, dv, C.c, Out, ldo, p->n, alpha, beta, stream);
    }

    // (blocks wider than one launch of the neighbour-union kernels takes - configs[4]'s 136-column block, the periodic refresh
    // K [X P W] - go in column slices through the same kernels: round 5; until then they fell to the wave-per-node kernels)
    int apply_K(const float* X, int64_t ldx, float* Y, int64_t ldy, int ncols) {
        const ds_level_t& L = p->level;
        if (ncols <= 84 && ncols % 4 == 0 && L.m32_gptr && L.m32_k && 3 * L.nv * ldx * 4 < (int64_t)0x7f000000)
            return ds_spmm_union32m(0, L.level_tag, L.m32_gptr, L.m32_gcol, L.m32_gmeta, L.m32_gbase, L.m32_k, L.nnzb * 36 + 16,
                                    L.nnzb, (L.nv + 3) / 4, L.m32_max_entries, L.m32_max_batch_blocks, L.nv, X, ldx, Y, ldy,
                                    ncols, stream);
        if (ncols % 4 == 0)
            return for_col_slices(ncols, [&](int c0, int c1) {
                return ds_spmm_union(0, L.level_tag, L.utab, L.ctab, L.ngroups, L.cap_blocks, L.gent, L.kgrp, L.nnzb, L.nv, X + c0, ldx,
                                     Y + c0, ldy, nullptr, 0, nullptr, c1 - c0, 0.f, 0.f, 0, nullptr, 0, stream);
            });
        for (int c0 = 0; c0 < ncols; c0 += 256) {  // odd widths: wave-per-node kernels
            const int c1 = std::min(ncols, c0 + 256);
            int rc = ds_spmm_bsr3(0, p->rowptr, p->colidx, p->k32, p->k32t, p->nv, X + c0, ldx, Y + c0, ldy, c1 - c0, stream);
            if (rc != DS_OK) return rc;
        }
        return DS_OK;
    }

    int apply_M(const float* X, int64_t ldx, float* Y, int64_t ldy, int ncols) {
        const ds_level_t& L = p->level;
        if (ncols <= 84 && L.m32_gptr && L.m32_m && 3 * L.nv * ldx * 4 < (int64_t)0x7f000000)
            return ds_spmm_union32m(3, L.level_tag, L.m32_gptr, L.m32_gcol, L.m32_gmeta, L.m32_gbase, L.m32_m, L.nnzb * 4 + 16,
                                    L.nnzb, (L.nv + 3) / 4, L.m32_max_entries, L.m32_max_batch_blocks, L.nv, X, ldx, Y, ldy,
                                    ncols, stream);
        return for_col_slices(ncols, [&](int c0, int c1) {
            return ds_spmm_union(3, L.level_tag, L.utab, L.ctab, L.ngroups, L.cap_blocks, L.gent, p->mgrp, L.nnzb, L.nv, X + c0, ldx,
                                 Y + c0, ldy, nullptr, 0, nullptr, c1 - c0, 0.f, 0.f, 0, nullptr, 0, stream);
        });
    }

    // K X -> KX and M X -> MX of one block in one walk of the unions
    int apply_KM(const float* X, int64_t ldx, float* KX, int64_t ldk, float* MX, int64_t ldm, int ncols) {
        const ds_level_t& L = p->level;
        return for_col_slices(ncols, [&](int c0, int c1) {
            return ds_spmm_union_km(L.level_tag, L.utab, L.ctab, L.ngroups, L.cap_blocks, L.gent, L.kgrp, p->mgrp, L.nnzb, L.nv,
                                    X + c0, ldx, KX + c0, ldk, MX + c0, ldm, c1 - c0, stream);
        });
    }

    // R = K X - (M X) diag(lam) and the column norms (nrm, nrm + 1024) in one walk of the unions per column slice
    int residual_fused(const float* X, int64_t ldx, float* R, int64_t ldr_, int ncols) {
        const ds_level_t& L = p->level;
        return for_col_slices(ncols, [&](int c0, int c1) {
            return ds_union_residual(L.level_tag, L.utab, L.ctab, L.ngroups, L.cap_blocks, L.gent, L.kgrp, p->mgrp, L.nnzb, L.nv,
                                     X + c0, ldx, p->lam_dev + c0, R + c0, ldr_, c1 - c0, p->res_work, p->res_work_bytes, p->nrm + c0,
                                     p->nrm + 1024 + c0, stream);
        });
    }

    int copy_cols(float* dst, int64_t ldd, const float* src, int64_t lds_, int ncols) {
        if (ncols <= 0) return DS_OK;
        return hip(hipMemcpy2DAsync(dst, (size_t)ldd * 4, src, (size_t)lds_ * 4, (size_t)ncols * 4, (size_t)p->n,
                                    hipMemcpyDeviceToDevice, st),
                   "ds_lobpcg_iterate: column copy");
    }

    // W <- B R  (two-level V-cycle, or the one-level Chebyshev polynomial W = p(T K) T R); columns are independent: blocks wider
    // than the fused kernels take go in slices that share the scratch blocks (stream order)
    int precond(float* R, int na, float* W, int64_t ldw) {
        return for_col_slices(na, [&](int c0, int c1) {
            if (p->twolevel) {
                ds_twolevel_t d = *p->twolevel;
                d.R = R + c0, d.ldr = p->ldr, d.W = W + c0, d.ldw = ldw, d.ncols = c1 - c0;
                return ds_twolevel_apply(&d, stream);
            }
            if (p->pr16)
                return ds_chebyshev_apply16(&p->level, R + c0, p->ldr, W + c0, ldw, p->pa, p->pb, p->pr16, p->ldp, c1 - c0, stream);
            return ds_chebyshev_apply(&p->level, R + c0, p->ldr, W + c0, ldw, p->pa, p->pb, p->ldp, c1 - c0, stream);
        });
    }

    PinnedRing* ring = &g_ring;
    int slot = 0;

    // W (n x na, column range [w0, w0 + na) of S) M-orthogonal to V = S[:, :w0] and M-orthonormal
    int orthonormalize(int w0, int na) {
        const double eps = 6e-8;
        float* W = S + w0;
        const int64_t ld = p->lds;
        for (int ip = 0; ip < p->ortho_passes; ++ip) {
            int rc = apply_M(W, ld, p->MW, p->ldr, na);
            if (rc != DS_OK) return rc;
            bool done = false;
            double amp = 0.0;
            if (w0 > 0 && ip == 0) {
                Mat G;
                if ((rc = gram(S, ld, w0 + na, p->MW, p->ldr, na, false, G)) != DS_OK) return rc;
                Mat C(w0, na), G0(na, na);
                for (int i = 0; i < w0; ++i)
                    for (int j = 0; j < na; ++j) C(i, j) = G(i, j);
                for (int i = 0; i < na; ++i)
                    for (int j = 0; j < na; ++j) G0(i, j) = G(w0 + i, j);
                symmetrize(G0);
                Mat CtC = gemm(la, C, true, C, false);
                Mat Gp(na, na);
                bool ok = true;
                for (int i = 0; i < na; ++i)
                    for (int j = 0; j < na; ++j) Gp(i, j) = G0(i, j) - CtC(i, j);
                for (int i = 0; i < na; ++i)
                    if (Gp(i, i) <= 1e-9 * std::fabs(G0(i, i))) ok = false;
                Mat L;
                if (ok && all_finite(Gp) && cholesky(Gp, L)) {
                    std::vector<double> rem(na);
                    for (int i = 0; i < na; ++i) rem[i] = CtC(i, i);
                    Mat T;
                    if (!orthonormalizer_q(la, Gp, &rem, T, amp)) {
                        ds::set_error("ds_lobpcg_iterate: dsyevd failed in the orthonormalisation");
                        return DS_ERR_ARG;
                    }
                    const Mat CT = gemm(la, C, false, T, false);
                    Mat coef(w0 + na, na);
                    for (int i = 0; i < w0; ++i)
                        for (int j = 0; j < na; ++j) coef(i, j) = -CT(i, j);
                    for (int i = 0; i < na; ++i)
                        for (int j = 0; j < na; ++j) coef(w0 + i, j) = T(i, j);
                    if ((rc = mix(S, ld, w0 + na, coef, W, ld)) != DS_OK) return rc;  // in place (W trails V in S)
                    done = true;
                }
            }
            if (!done) {
                std::vector<double> rem;
                if (w0 > 0) {
                    Mat C;
                    if ((rc = gram(S, ld, w0, p->MW, p->ldr, na, false, C)) != DS_OK) return rc;
                    if ((rc = mix(S, ld, w0, C, W, ld, -1.f, 1.f)) != DS_OK) return rc;
                    rem.assign(na, 0.0);
                    for (int i = 0; i < w0; ++i)
                        for (int j = 0; j < na; ++j) rem[j] += C(i, j) * C(i, j);
                    if ((rc = apply_M(W, ld, p->MW, p->ldr, na)) != DS_OK) return rc;
                }
                Mat G;
                if ((rc = gram(W, ld, na, p->MW, p->ldr, na, true, G)) != DS_OK) return rc;
                Mat T;
                if (!orthonormalizer_q(la, G, w0 > 0 ? &rem : nullptr, T, amp)) {
                    ds::set_error("ds_lobpcg_iterate: dsyevd failed in the orthonormalisation");
                    return DS_ERR_ARG;
                }
                if ((rc = mix(W, ld, na, T, W, ld)) != DS_OK) return rc;  // in place: na <= b <= 160 columns, what ds_mix takes aliased
            }
            if (p->ortho_tol > 0.0 && eps * amp < p->ortho_tol) break;
        }
        return DS_OK;
    }
};

}  // namespace

// Self-check of the host-side dense steps (no device): on a seeded random symmetric positive definite n x n matrix,
// errs[0] = max |w_staged - w_dsyevd| / |w|_max and errs[1] = max |G Z - Z diag(w)| / |G|_max of the lowest m pairs from the staged
// eigensolver (eigh_lowest), errs[2] = max |L L^T - G| / |G|_max of the Cholesky factor, errs[3] = max |L^-1 L - I|,
// errs[4] = max |Q^T Q - I| of orthonormal_columns on n x m random columns; errs[5] = 1 when the staged path ran (the table has
// the three stages), else 0.
extern "C" int ds_selftest_dense(const ds_lapack_t* lapack, int n, int m, unsigned seed, double* errs) {
    DS_REQUIRE(lapack && lapack->dsyevd && lapack->dgemm && errs && n >= 2 && m >= 1 && m <= n, "ds_selftest_dense: bad arguments");
    unsigned long long st = seed * 2654435761ull + 12345ull;
    auto rnd = [&]() {
        st = st * 6364136223846793005ull + 1442695040888963407ull;
        return ((st >> 11) & ((1ull << 53) - 1)) / double(1ull << 53) - 0.5;
    };
    Mat B(n, n);
    for (double& v : B.a) v = rnd();
    Mat G = gemm(*lapack, B, false, B, true);
    for (int i = 0; i < n; ++i) G(i, i) += 0.05 * (i + 1);
    symmetrize(G);
    double gmax = 0.0;
    for (double v : G.a) gmax = std::max(gmax, std::fabs(v));
    std::vector<double> w, wf;
    Mat Zm, Zf;
    if (!eigh_lowest(*lapack, G, m, w, Zm) || !eigh(*lapack, G, wf, Zf)) {
        ds::set_error("ds_selftest_dense: LAPACK reported failure");
        return DS_ERR_ARG;
    }
    errs[0] = errs[1] = errs[2] = errs[3] = errs[4] = 0.0;
    for (int j = 0; j < n; ++j) errs[0] = std::max(errs[0], std::fabs(w[j] - wf[j]) / std::fabs(wf[n - 1]));
    const Mat GZ = gemm(*lapack, G, false, Zm, false);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < m; ++j) errs[1] = std::max(errs[1], std::fabs(GZ(i, j) - Zm(i, j) * w[j]) / gmax);
    Mat L;
    if (!cholesky(G, L)) {
        ds::set_error("ds_selftest_dense: Cholesky broke down on a positive definite matrix");
        return DS_ERR_ARG;
    }
    const Mat LLt = gemm(*lapack, L, false, L, true);
    for (size_t i = 0; i < G.a.size(); ++i) errs[2] = std::max(errs[2], std::fabs(LLt.a[i] - G.a[i]) / gmax);
    const Mat Li = lower_inverse(L), I1 = gemm(*lapack, Li, false, L, false);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) errs[3] = std::max(errs[3], std::fabs(I1(i, j) - (i == j ? 1.0 : 0.0)));
    Mat Tm(n, m), Q;
    for (double& v : Tm.a) v = rnd();
    if (!orthonormal_columns(*lapack, Tm, Q)) {
        ds::set_error("ds_selftest_dense: orthonormal_columns broke down");
        return DS_ERR_ARG;
    }
    const Mat QtQ = gemm(*lapack, Q, true, Q, false);
    for (int i = 0; i < m; ++i)
        for (int j = 0; j < m; ++j) errs[4] = std::max(errs[4], std::fabs(QtQ(i, j) - (i == j ? 1.0 : 0.0)));
    errs[5] = (lapack->dsytrd && lapack->dstedc && lapack->dormtr && m < n && n >= 32) ? 1.0 : 0.0;
    return DS_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// The two small dense steps that frame the iteration - the start block's first Ritz step in coefficients and the fp64 polish of
// the converged block - on the host with the caller's LAPACK table (round 6: they were ~30 torch calls on tiny CPU tensors each,
// 0.65 ms and 1.1 ms per call on the host thread of a solve; one hypothesis alone waits for every one of them).  Pure host
// functions: the CPU test suite checks them against the Python forms they replace (lobpcg/modal_solver.py: `start`, `small`).

// G = [Y X0]^T [K X0 | M X0], (ny + b) x 2 b row-major.  Status DS_OK with *route = 0: lam (b), coef ((ny + b) x b: X = [Y X0] coef),
// cx (b x b: K X = (K X0) cx), *amp filled; *route = 1: the block is too ill-conditioned for one sweep (or its projected Gram matrix
// broke down) - the caller takes the explicit route; nothing else is written.
extern "C" int ds_host_start_block(const ds_lapack_t* lapack, const double* Gin, int ny, int b, double ortho_tol, double eps,
                                   double* lam, double* coef, double* cx, double* amp_out, int* route) {
    DS_REQUIRE(lapack && lapack->dsyevd && lapack->dgemm && Gin && lam && coef && cx && amp_out && route && ny >= 0 && b > 0,
               "ds_host_start_block: bad arguments");
    const ds_lapack_t& la = *lapack;
    *route = 1;
    Mat Gyk(ny, b), Cy(ny, b), A(b, b), B0(b, b);
    const int ld = 2 * b;
    for (int i = 0; i < ny; ++i)
        for (int j = 0; j < b; ++j) Gyk(i, j) = Gin[(size_t)i * ld + j], Cy(i, j) = Gin[(size_t)i * ld + b + j];
    for (int i = 0; i < b; ++i)
        for (int j = 0; j < b; ++j) A(i, j) = Gin[(size_t)(ny + i) * ld + j], B0(i, j) = Gin[(size_t)(ny + i) * ld + b + j];
    symmetrize(A), symmetrize(B0);
    const Mat CtC = ny ? gemm(la, Cy, true, Cy, false) : Mat(b, b);
    Mat Bp(b, b);
    for (int i = 0; i < b; ++i)
        for (int j = 0; j < b; ++j) Bp(i, j) = B0(i, j) - CtC(i, j);
    for (int i = 0; i < b; ++i)
        if (Bp(i, i) <= 1e-9 * std::fabs(B0(i, i))) return DS_OK;
    if (!all_finite(Bp)) return DS_OK;
    std::vector<double> rem(b);
    for (int i = 0; i < b; ++i) rem[i] = CtC(i, i);
    Mat T;
    double amp = 0.0;
    if (!orthonormalizer_q(la, Bp, &rem, T, amp)) {
        ds::set_error("ds_host_start_block: dsyevd failed in the orthonormalisation");
        return DS_ERR_ARG;
    }
    if (!(ortho_tol > 0.0 && eps * amp < ortho_tol)) return DS_OK;  // (one sweep would leave eps * amp: the explicit route repairs it)
    Mat A1 = A;
    if (ny) {
        const Mat CtG = gemm(la, Cy, true, Gyk, false);
        for (int i = 0; i < b; ++i)
            for (int j = 0; j < b; ++j) A1(i, j) -= CtG(i, j) + CtG(j, i);
    }
    Mat H = gemm(la, gemm(la, T, true, A1, false), false, T, false);
    symmetrize(H);
    std::vector<double> E;
    Mat Z;
    if (!eigh(la, H, E, Z)) {
        ds::set_error("ds_host_start_block: dsyevd failed in the first Ritz step");
        return DS_ERR_ARG;
    }
    const Mat Cx = gemm(la, T, false, Z, false);
    const Mat CyCx = ny ? gemm(la, Cy, false, Cx, false) : Mat(0, b);
    for (int j = 0; j < b; ++j) lam[j] = E[j];
    for (int i = 0; i < ny; ++i)
        for (int j = 0; j < b; ++j) coef[(size_t)i * b + j] = -CyCx(i, j);
    for (int i = 0; i < b; ++i)
        for (int j = 0; j < b; ++j) coef[(size_t)(ny + i) * b + j] = cx[(size_t)i * b + j] = Cx(i, j);
    *amp_out = amp;
    *route = 0;
    return DS_OK;
}

// fp64 polish of a converged block: GK = nterms (b x b) Gram matrices X^T K_i X (row-major, one after the other), coef their
// weights, GM = X^T M X.  The generalised Ritz problem (sum c_i GK_i) z = e GM z through the Cholesky factor of GM: E (k lowest
// values), C (b x b generalised eigenvectors, C^T GM C = I, columns ascending), qs ((nterms + 1) x k: the quadratic forms
// c_j^T GK_i c_j and c_j^T GM c_j of the k wanted vectors).  DS_ERR_ARG with a message when GM is not positive definite.
extern "C" int ds_host_polish(const ds_lapack_t* lapack, int nterms, const double* GK, const double* coefs, const double* GMin, int b, int k,
                              double* Eout, double* Cout, double* qs) {
    DS_REQUIRE(lapack && lapack->dsyevd && lapack->dgemm && GK && coefs && GMin && Eout && Cout && qs && nterms >= 1 && b > 0 && k > 0 && k <= b,
               "ds_host_polish: bad arguments");
    const ds_lapack_t& la = *lapack;
    Mat GA(b, b), GB(b, b);
    for (int t = 0; t < nterms; ++t)
        for (size_t i = 0; i < GA.a.size(); ++i) GA.a[i] += coefs[t] * GK[(size_t)t * b * b + i];
    for (size_t i = 0; i < GB.a.size(); ++i) GB.a[i] = GMin[i];
    symmetrize(GA), symmetrize(GB);
    Mat L;
    if (!cholesky(GB, L)) {
        ds::set_error("ds_host_polish: X^T M X of the converged block is not positive definite");
        return DS_ERR_ARG;
    }
    const Mat Li = lower_inverse(L);
    Mat H = gemm(la, gemm(la, Li, false, GA, false), false, Li, true);
    symmetrize(H);
    std::vector<double> E;
    Mat Zt;
    if (!eigh(la, H, E, Zt)) {
        ds::set_error("ds_host_polish: dsyevd failed");
        return DS_ERR_ARG;
    }
    const Mat C = gemm(la, Li, true, Zt, false);
    for (int j = 0; j < k; ++j) Eout[j] = E[j];
    for (size_t i = 0; i < C.a.size(); ++i) Cout[i] = C.a[i];
    Mat Ck(b, k);
    for (int i = 0; i < b; ++i)
        for (int j = 0; j < k; ++j) Ck(i, j) = C(i, j);
    for (int t = 0; t <= nterms; ++t) {
        Mat Gt(b, b);
        if (t < nterms) {
            for (size_t i = 0; i < Gt.a.size(); ++i) Gt.a[i] = GK[(size_t)t * b * b + i];
            symmetrize(Gt);
        } else {
            Gt = GB;
        }
        const Mat GC = gemm(la, Gt, false, Ck, false);
        for (int j = 0; j < k; ++j) {
            double sacc = 0.0;
            for (int i = 0; i < b; ++i) sacc += Ck(i, j) * GC(i, j);
            qs[(size_t)t * k + j] = sacc;
        }
    }
    return DS_OK;
}

extern "C" int ds_host_wait_mode(int mode) {
    DS_REQUIRE(mode == 0 || mode == 1, "ds_host_wait_mode: 0 (the runtime's stream synchronisation) or 1 (poll, then sleep on a blocking event)");
    g_wait_mode.store(mode, std::memory_order_relaxed);
    return DS_OK;
}

extern "C" int ds_lobpcg_iterate(ds_lobpcg_t* p, const ds_lapack_t* lapack, ds_stream_t stream) {
    DS_REQUIRE(p && lapack && lapack->dsyevd && lapack->dgemm, "ds_lobpcg_iterate: null descriptor or LAPACK table");
    DS_REQUIRE((!lapack->dsytrd) == (!lapack->dstedc) && (!lapack->dsytrd) == (!lapack->dormtr),
               "ds_lobpcg_iterate: the staged eigensolver needs dsytrd, dstedc and dormtr together (or none of them)");
    DS_REQUIRE(p->S && p->S2 && p->KS && p->KS2 && p->R && p->MX && p->MW && p->lam && p->rerr && p->gbuf && p->cbuf &&
                   p->nrm && p->lam_dev && p->gram_work && p->mgrp,
               "ds_lobpcg_iterate: null buffer");
    // (<= 160: the in-place updates of the orthonormalisation go through ds_mix, which takes an aliased result only up to 160
    // columns - blocks of 164 / 168 columns stay on the Python loop, which splits that update: ADVICE r05)
    DS_REQUIRE(p->b > 0 && p->b <= 160 && p->b % 4 == 0 && p->k > 0 && p->k <= p->b && (p->ny == 0 || p->ny % 4 == 0),
               "ds_lobpcg_iterate: block width must be a multiple of 4 <= 160 (got %d)", p->b);
    DS_REQUIRE(p->n == 3 * p->nv && p->n >= 3 * p->b + p->ny, "ds_lobpcg_iterate: bad problem size");
    DS_REQUIRE(p->twolevel || (p->pa && p->pb), "ds_lobpcg_iterate: no preconditioner scratch");
    // (the periodic full refresh multiplies K by [X P W], up to 3 b columns, in column slices through the union kernels: the
    // plain BSR arrays are no longer needed for it - every width the loop forms is a multiple of 4)
    Ctx c{p, ds::as_stream(stream), stream, *lapack, p->S, p->S2, p->KS, p->KS2};
    DS_REQUIRE(p->wait_mode >= -1 && p->wait_mode <= 1, "ds_lobpcg_iterate: wait_mode is -1 (the process default), 0 or 1");
    struct WaitScope {  // this solve's own way of waiting for its stream (round 6: per solve, i.e. per hypothesis lane)
        int saved;
        explicit WaitScope(int m) : saved(g_solve_wait_mode) { g_solve_wait_mode = m; }
        ~WaitScope() { g_solve_wait_mode = saved; }
    } wait_scope(p->wait_mode);
    const int b = p->b, k = p->k, ny = p->ny;
    {   // cbuf holds COEF_SLOTS slots of (ny + 3 b) x 2 b floats
        int rcr = g_ring.reserve((size_t)(ny + 3 * b) * 2 * b);
        if (rcr != DS_OK) return rcr;
        rcr = g_ring.reserve_back(2048 + LAM_SLOT + (size_t)(ny + 3 * b) * 3 * b);  // norms | Ritz values | Gram block
        if (rcr != DS_OK) return rcr;
    }
    const int64_t n = p->n, lds = p->lds, ldks = p->ldks, ldr = p->ldr;
    int rc;
    g_tm = Tm();
    const double t_begin = now_s();
    std::vector<double> lam(p->lam, p->lam + b), rel(b, std::numeric_limits<double>::infinity());
    // Ritz values of the step before (p->ritz_tol > 0: a pair also has to have SETTLED to count as converged - see the header)
    std::vector<double> lam_prev(b, std::numeric_limits<double>::infinity());
    Mat Gxp(b, b);
    for (int i = 0; i < b; ++i) Gxp(i, i) = lam[i];
    int ncl = 0, npc = 0, k0 = 0, since_refresh = 0, it = 0;
    // the fused residual needs the operands the neighbour-union kernel takes (one descriptor per operand block) and replaces
    // the fresh K X' of kx_fresh - without kx_fresh K X' comes out of the recurrence and is there anyway
    const bool fused_res = p->res_work && p->kx_fresh &&
                           p->res_work_bytes >= ds_union_residual_workspace_bytes(p->level.ngroups, std::min(b, 84));
    // Rayleigh-Ritz on the raw basis: K X' must not be needed from K [X P W] (kx_fresh) and comes from the fused residual
    const bool raw = p->raw_rr && fused_res && !p->gram_exact;
    double worst = std::numeric_limits<double>::infinity();
    // The residual R = K X - (M X) diag(lam) of the CURRENT Ritz block and its norms: launched at the end of an iteration, right
    // behind the update that wrote the block (and once before the loop) - the host algebra that only the NEXT Ritz step needs
    // ([X' P']^T K [X' P'], below) then runs while the device works on the update and this residual, instead of in front of them
    // (round 6: one lane alone is bound by the host's serial steps).
    auto launch_residual = [&]() -> int {
        const int na = b - ncl;
        float* Xa = c.S + ny + ncl;
        double* lam_pin = g_ring.back + 2048;  // (the previous iteration's copy completed before its last synchronise)
        std::memcpy(lam_pin, lam.data() + ncl, sizeof(double) * na);
        int rc_ = c.hip(hipMemcpyAsync(p->lam_dev, lam_pin, sizeof(double) * na, hipMemcpyHostToDevice, c.st),
                        "ds_lobpcg_iterate: Ritz values to device");
        if (rc_ != DS_OK) return rc_;
        if (fused_res) {  // R = K X - (M X) diag(lam) and the norms in one walk of the unions; K X, M X never reach memory
            if ((rc_ = c.residual_fused(Xa, lds, p->R, ldr, na)) != DS_OK) return rc_;
        } else {
            if ((rc_ = c.apply_M(Xa, lds, p->MX, ldr, na)) != DS_OK) return rc_;
            if ((rc_ = ds_residual(c.KS + k0, ldks, p->R, ldr, p->MX, ldr, Xa, lds, p->lam_dev, n, na, p->nrm, p->nrm + 1024,
                                   stream)) != DS_OK)
                return rc_;
        }
        return c.hip(hipMemcpyAsync(g_ring.back, p->nrm, sizeof(double) * 2048, hipMemcpyDeviceToHost, c.st),
                     "ds_lobpcg_iterate: residual norms to host");
    };
    if ((rc = launch_residual()) != DS_OK) return rc;
    for (it = 0; it <= p->maxit; ++it) {
        int na = b - ncl;
        double* nrm = g_ring.back;
        { const double t0 = now_s(); rc = c.hip(wait_for_stream(c.st), "ds_lobpcg_iterate: stream synchronise"); g_tm.sync += now_s() - t0; ++g_tm.nsync; }
        if (rc != DS_OK) return rc;
        for (int j = 0; j < na; ++j)
            rel[ncl + j] = std::sqrt(nrm[j] / nrm[1024 + j]) / (p->A_norm + std::fabs(lam[ncl + j]) * p->B_norm);
        int nconv = 0;
        // leading converged pairs only (reference _lobpcg.py:321-328)
        while (nconv < k && rel[nconv] < p->tol &&
               (p->ritz_tol <= 0.0 || std::fabs(lam[nconv] - lam_prev[nconv]) <= p->ritz_tol * std::fabs(lam[nconv])))
            ++nconv;
        worst = 0.0;
        for (int j = 0; j < k; ++j) worst = std::max(worst, rel[j]);
        if (p->history && it < p->history_cap) p->history[it] = worst;
        if (nconv >= k || it == p->maxit) break;
        // hard locking: converged leading columns leave the Ritz problem, the residual block and the preconditioner
        const int new_ncl = p->lock ? (nconv / 4) * 4 : 0;
        float* Ract = p->R;  // the residual columns of the pairs that stay active
        if (new_ncl > ncl) {
            const int shift = new_ncl - ncl;
            // the newly locked columns leave the residual block by a pointer offset (a multiple of 4 columns: 16-byte aligned)
            // and enter the OTHER basis buffer once, here - from then on both buffers hold them (they used to be copied in
            // front of the active ones in every later iteration, and the residual block was shifted through a scratch block)
            Ract = p->R + shift;
            if ((rc = c.copy_cols(c.S2 + ny + ncl, lds, c.S + ny + ncl, lds, shift)) != DS_OK) return rc;
            Mat G2(Gxp.r - shift, Gxp.c - shift);
            for (int i = 0; i < G2.r; ++i)
                for (int j = 0; j < G2.c; ++j) G2(i, j) = Gxp(i + shift, j + shift);
            Gxp = G2;
            k0 += shift;
            ncl = new_ncl;
            na = b - ncl;
        }
        const int w0 = ny + b + npc;
        float* W = c.S + w0;
        if ((rc = c.precond(Ract, na, W, lds)) != DS_OK) return rc;
        const int sz = na + npc + na, nxp = na + npc;
        float* Sa = c.S + ny + ncl;
        float* KSa = c.KS + k0;
        const bool full = p->rr_refresh <= 0 || since_refresh >= p->rr_refresh;
        Mat G(sz, sz);
        // ---- Rayleigh-Ritz on the RAW basis (round 5, p->raw_rr): W stays as the preconditioner left it.  K W and M W come out of
        // one walk of the unions, [Y X P W]^T [K W | M W] out of ONE Gram launch; [Y X P] is M-orthonormal, so C = V^T M W and
        // G0 = W^T M W give the projected Cholesky-QR transform W_o = [V W] [-C T; T] in coefficients only, and every block of
        // S_a^T K S_a for S_a = [X_a P W_o] follows from the Gram rows and the recurrence's [X_a P]^T K [X_a P] (K Y = 0;
        // X_l^T K X_l = diag(lam_l), X_l^T K [X_a P] = 0 for the locked - converged - columns).  The Ritz coefficients go back
        // to the raw basis, Z_raw = Q Z, and ONE update [X' P'] = [Y X P W] Z_raw writes the new basis: the explicit
        // orthonormalisation's product M W, its Gram and its in-place update of W are gone (an iteration was
        // M W, Gram, update, K W, Gram, update; it is [K W | M W], Gram, update).  An iteration whose W is too ill-conditioned for a
        // single sweep (eps x amplification >= ortho_tol, or a breakdown of the factorisation) takes the explicit route below.
        Mat Qraw;  // (w0 + na) x sz: the active orthonormal basis in coordinates of the raw one; empty = explicit route
        if (raw && !full) {
            const int pr = w0 + na;
            if ((rc = c.apply_KM(W, lds, c.KS, ldks, c.KS + na, ldks, na)) != DS_OK) return rc;
            Mat GG;
            if ((rc = c.gram(c.S, lds, pr, c.KS, ldks, 2 * na, false, GG)) != DS_OK) return rc;
            Mat C(w0, na), G0(na, na);
            for (int i = 0; i < w0; ++i)
                for (int j = 0; j < na; ++j) C(i, j) = GG(i, na + j);
            for (int i = 0; i < na; ++i)
                for (int j = 0; j < na; ++j) G0(i, j) = GG(w0 + i, na + j);
            symmetrize(G0);
            const Mat CtC = gemm(*lapack, C, true, C, false);
            Mat Gp(na, na);
            bool ok = true;
            for (int i = 0; i < na; ++i)
                for (int j = 0; j < na; ++j) Gp(i, j) = G0(i, j) - CtC(i, j);
            for (int i = 0; i < na; ++i)
                if (Gp(i, i) <= 1e-9 * std::fabs(G0(i, i))) ok = false;
            Mat T;
            double amp = 0.0;
            // (the factorisation of the diagonally scaled block inside orthonormalizer_q is the breakdown test: a block that is not
            // positive definite comes back with amp = inf and takes the explicit route - a separate unscaled Cholesky in front of
            // it, as the Python loop has, was 0.07 ms of every iteration for the same answer)
            if (ok && all_finite(Gp)) {
                std::vector<double> rem(na);
                for (int i = 0; i < na; ++i) rem[i] = CtC(i, i);
                ok = orthonormalizer_q(*lapack, Gp, &rem, T, amp) && std::isfinite(amp) && !(p->ortho_tol > 0.0 && 6e-8 * amp >= p->ortho_tol);
            } else {
                ok = false;
            }
            if (ok) {
                const Mat CT = gemm(*lapack, C, false, T, false);
                Mat GKraw(pr, pr);  // [Y X P W]^T K [Y X P W]: known blocks among Y, X, P; measured columns of W
                for (int i = 0; i < ncl; ++i) GKraw(ny + i, ny + i) = lam[i];
                for (int i = 0; i < nxp; ++i)
                    for (int j = 0; j < nxp; ++j) GKraw(ny + ncl + i, ny + ncl + j) = Gxp(i, j);
                for (int i = 0; i < w0; ++i)
                    for (int j = 0; j < na; ++j) GKraw(i, w0 + j) = GKraw(w0 + j, i) = GG(i, j);
                for (int i = 0; i < na; ++i)
                    for (int j = 0; j < na; ++j) GKraw(w0 + i, w0 + j) = 0.5 * (GG(w0 + i, j) + GG(w0 + j, i));
                // Q = [E | Qw]: E picks the rows of [X_a P] (unit columns), Qw = [-C T; T] are W_o's coordinates.  G = Q^T GKraw Q
                // block by block - the unit columns cost nothing: a third of the flops of the two full products (one lane alone is
                // bound by this host algebra, not by the kernels)
                Qraw = Mat(pr, na);  // Qw only; the unit part is implicit
                for (int i = 0; i < w0; ++i)
                    for (int j = 0; j < na; ++j) Qraw(i, j) = -CT(i, j);
                for (int i = 0; i < na; ++i)
                    for (int j = 0; j < na; ++j) Qraw(w0 + i, j) = T(i, j);
                const Mat H = gemm(*lapack, GKraw, false, Qraw, false);    // (pr x na) = GKraw Qw
                const Mat Gww = gemm(*lapack, Qraw, true, H, false);       // (na x na) = Qw^T GKraw Qw
                for (int i = 0; i < nxp; ++i)
                    for (int j = 0; j < nxp; ++j) G(i, j) = Gxp(i, j);
                for (int i = 0; i < nxp; ++i)
                    for (int j = 0; j < na; ++j) G(i, nxp + j) = G(nxp + j, i) = H(ny + ncl + i, j);
                for (int i = 0; i < na; ++i)
                    for (int j = 0; j < na; ++j) G(nxp + i, nxp + j) = Gww(i, j);
                ++since_refresh;
            }
        }
        if (Qraw.r == 0 && (rc = c.orthonormalize(w0, na)) != DS_OK) return rc;
        if (Qraw.r != 0) {
            // (G is complete)
        } else if (full) {
            if ((rc = c.apply_K(Sa, lds, KSa, ldks, sz)) != DS_OK) return rc;
            if ((rc = c.gram(Sa, lds, sz, KSa, ldks, sz, true, G)) != DS_OK) return rc;
            since_refresh = 0;
        } else {
            if ((rc = c.apply_K(W, lds, KSa + nxp, ldks, na)) != DS_OK) return rc;
            Mat GA;
            if ((rc = c.gram(Sa, lds, sz, KSa + nxp, ldks, na, false, GA)) != DS_OK) return rc;
            for (int i = 0; i < nxp; ++i)
                for (int j = 0; j < nxp; ++j) G(i, j) = Gxp(i, j);
            for (int i = 0; i < sz; ++i)
                for (int j = 0; j < na; ++j) G(i, nxp + j) = GA(i, j);
            for (int i = 0; i < na; ++i)
                for (int j = 0; j < nxp; ++j) G(nxp + i, j) = GA(j, i);
            ++since_refresh;
        }
        symmetrize(G);
        // Rayleigh-Ritz: lowest na pairs, and P = the part of the old active X that left the new Ritz block
        // (dsyevr on the index range 1..na - the same tridiagonalisation, a third of the vectors - was measured: 3.3 ms per
        // 240 x 240 problem with SciPy's OpenBLAS against 1.45 ms for the full dsyevd; not used)
        std::vector<double> E;
        Mat Z1;  // (sz x na): only the wanted third of the vectors is back-transformed (eigh_lowest)
        if (!eigh_lowest(*lapack, G, na, E, Z1)) {
            ds::set_error("ds_lobpcg_iterate: dsyevd failed in the Rayleigh-Ritz step");
            return DS_ERR_ARG;
        }
        Mat Z1top(na, na);
        for (int i = 0; i < na; ++i)
            for (int j = 0; j < na; ++j) Z1top(i, j) = Z1(i, j);
        Mat Tm = gemm(*lapack, Z1, false, Z1top, true);  // Z1 Z1[:na]^T
        for (double& v : Tm.a) v = -v;
        for (int i = 0; i < na; ++i) Tm(i, i) += 1.0;
        Mat Zp;
        if (!orthonormal_columns(*lapack, Tm, Zp)) {  // rank-deficient P block: clamped-eigenvalue basis instead
            Mat GT = gemm(*lapack, Tm, true, Tm, false), Tq;
            if (!svqb_transform(*lapack, GT, Tq)) {
                ds::set_error("ds_lobpcg_iterate: dsyevd failed on the P block");
                return DS_ERR_ARG;
            }
            Zp = gemm(*lapack, Tm, false, Tq, false);
        }
        Mat ZZ(sz, 2 * na);
        for (int i = 0; i < sz; ++i)
            for (int j = 0; j < na; ++j) {
                ZZ(i, j) = Z1(i, j);
                ZZ(i, na + j) = Zp(i, j);
            }
        for (int j = 0; j < na; ++j) { lam_prev[ncl + j] = lam[ncl + j]; lam[ncl + j] = E[j]; }
        if (Qraw.r != 0) {  // the new basis straight from the raw one: [X' P'] = [Y X P W] (Q [Z1 Zp]), Q = [E | Qw]
            Mat Zbot(na, 2 * na);
            for (int i = 0; i < na; ++i)
                for (int j = 0; j < 2 * na; ++j) Zbot(i, j) = ZZ(nxp + i, j);
            Mat Zr = gemm(*lapack, Qraw, false, Zbot, false);  // Qw Z_w ...
            for (int i = 0; i < nxp; ++i)                       // ... + E Z_xp
                for (int j = 0; j < 2 * na; ++j) Zr(ny + ncl + i, j) += ZZ(i, j);
            const int pr = w0 + na;
            if (2 * na <= 160) {
                if ((rc = c.mix(c.S, lds, pr, Zr, c.S2 + ny + ncl, lds)) != DS_OK) return rc;
            } else {
                Mat Za(pr, na), Zb(pr, na);
                for (int i = 0; i < pr; ++i)
                    for (int j = 0; j < na; ++j) Za(i, j) = Zr(i, j), Zb(i, j) = Zr(i, na + j);
                if ((rc = c.mix(c.S, lds, pr, Za, c.S2 + ny + ncl, lds)) != DS_OK) return rc;
                if ((rc = c.mix(c.S, lds, pr, Zb, c.S2 + ny + b, lds)) != DS_OK) return rc;
            }
        } else if (2 * na <= 160) {
            if ((rc = c.mix(Sa, lds, sz, ZZ, c.S2 + ny + ncl, lds)) != DS_OK) return rc;
            if (!p->kx_fresh && (rc = c.mix(KSa, ldks, sz, ZZ, c.KS2, ldks)) != DS_OK) return rc;
        } else {
            if ((rc = c.mix(Sa, lds, sz, Z1, c.S2 + ny + ncl, lds)) != DS_OK) return rc;
            if ((rc = c.mix(Sa, lds, sz, Zp, c.S2 + ny + b, lds)) != DS_OK) return rc;
            if (!p->kx_fresh) {
                if ((rc = c.mix(KSa, ldks, sz, Z1, c.KS2, ldks)) != DS_OK) return rc;
                if ((rc = c.mix(KSa, ldks, sz, Zp, c.KS2 + na, ldks)) != DS_OK) return rc;
            }
        }
        // K X' fresh: one b-column product instead of the 3b -> 2b column update of K [X P W] (K P' is never needed)
        // (with the fused residual the next iteration forms K X' inside its residual kernel and nobody else reads it)
        if (p->kx_fresh && !fused_res && (rc = c.apply_K(c.S2 + ny + ncl, lds, c.KS2, ldks, na)) != DS_OK) return rc;
        std::swap(c.S, c.S2);
        std::swap(c.KS, c.KS2);
        k0 = 0;
        npc = na;
        if ((rc = launch_residual()) != DS_OK) return rc;  // (of the block just written: the next iteration's first wait)
        // ---- behind the launches, while the device runs the update and the residual:
        // [X' P']^T K [X' P'] = ZZ^T G ZZ of the new basis.  The columns of Z1 are eigenvectors of G (Z1^T G Z1 = diag(E) to the
        // rounding of dsyevd), so only the products with Zp are formed: half the flops of the two full products
        {
            const Mat GZp = gemm(*lapack, G, false, Zp, false);
            const Mat Gxz = gemm(*lapack, Z1, true, GZp, false), Gpp = gemm(*lapack, Zp, true, GZp, false);
            Gxp = Mat(2 * na, 2 * na);
            for (int i = 0; i < na; ++i) Gxp(i, i) = E[i];
            for (int i = 0; i < na; ++i)
                for (int j = 0; j < na; ++j) {
                    Gxp(i, na + j) = Gxp(na + j, i) = Gxz(i, j);
                    Gxp(na + i, na + j) = Gpp(i, j);
                }
        }
        symmetrize(Gxp);
    }
    if (getenv("DS_EXP_TIMING")) {
        g_tm.total = now_s() - t_begin;
        fprintf(stderr, "lobpcg n=%lld b=%d: %d iterations, host total %.2f ms: waiting for the stream %.2f ms (%d syncs), dsyevd %.2f ms (%d calls), dgemm %.2f ms, rest (launch calls, Cholesky, copies) %.2f ms\n",
                (long long)p->n, b, it, g_tm.total * 1e3, g_tm.sync * 1e3, g_tm.nsync, g_tm.eigh * 1e3, g_tm.neigh, g_tm.dense * 1e3,
                (g_tm.total - g_tm.sync - g_tm.eigh - g_tm.dense) * 1e3);
    }
    p->iterations = it;
    p->result_in_s2 = (c.S == p->S) ? 0 : 1;
    std::memcpy(p->lam, lam.data(), sizeof(double) * b);
    std::memcpy(p->rerr, rel.data(), sizeof(double) * b);
    return DS_OK;
}
