// Host-side driver of the two-level V-cycle preconditioner: ONE call issues the ~45 kernel launches of
//   W1 = S R ;  W2 = W1 + P C P^T (R - K W1) ;  W = W2 + S (R - K W2)
// (S: Chebyshev block-Jacobi smoother on the fine level, C: Chebyshev polynomial on the corner-node level, every term
// one fused ds_spmm_union launch) on the caller's stream.  It replaces the same sequence issued launch by launch
// from the Python solver (lobpcg/modal_solver.py, TwoLevelChebyshev.apply / ChebyshevBlockJacobi._fused): with three
// hypothesis lanes per GPU those ~450 interpreter round trips per eigensolve and lane all compete for one
// interpreter lock.  The reference has no counterpart (its LOBPCG takes the preconditioner as an opaque callable
// `iK`, src/lobpcg/_lobpcg.py:441).
#include <hip/hip_runtime.h>

#include "ds_common.h"

namespace {

// Wout <- p(T K) T R by the three-term Chebyshev recurrence W_{k+1} = W_k + c1 (W_k - W_{k-1}) + c2 T (R - K W_k)
// (Saad, Iterative Methods, Alg. 12.1).  The iterates ping-pong between the COMPACT scratch blocks A and B (a neighbour
// panel of a compact block is 960 contiguous bytes; the same panel of a column range of the 248-column basis buffer
// spans a quarter more cache lines: 0.287 against 0.260 ms per fused term), and the LAST term reads its W_{k-1} from
// scratch but writes into Wout (out-of-place form of ds_spmm_union).  from_guess: A holds W_0 and `degree` terms of
// the ITERATION for K W = R are run from it.  B doubles as the unused second output of ds_cheb_init.  Mirrors
// ChebyshevBlockJacobi._fused.
int chebyshev(const ds_level_t& L, const float* R, int64_t ldr, float* Wout, int64_t ldw, float* A, float* B,
              int64_t lds, int ncols, bool from_guess, ds_stream_t stream) {
    const double theta = 0.5 * (L.lmax + L.lmin), delta = 0.5 * (L.lmax - L.lmin);
    const double sigma1 = theta / delta;
    double rho = 1.0 / sigma1;
    const int terms = from_guess ? L.degree : L.degree - 1;
    float *cur = A, *oth = B;
    if (L.tgrp) {
        ds::set_error("chebyshev: the group-block Jacobi lives on the bf16 cycle's matrix-core tables only");
        return DS_ERR_ARG;
    }
    if (!from_guess) {  // W_1 = T R / theta (straight into Wout when no term follows)
        int rc = ds_cheb_init(R, ldr, B, lds, terms == 0 ? Wout : cur, terms == 0 ? ldw : lds, L.dinv, L.nv, ncols,
                              (float)(1.0 / theta), stream);
        if (rc != DS_OK) return rc;
    }
    for (int k = 0; k < terms; ++k) {
        float c1, c2;
        if (from_guess && k == 0) {
            c1 = 0.f, c2 = (float)(1.0 / theta);
        } else {
            const double rho_new = 1.0 / (2.0 * sigma1 - rho);
            c1 = (float)(rho_new * rho), c2 = (float)(2.0 * rho_new / delta);
            rho = rho_new;
        }
        const bool last = k == terms - 1;
        // W_{k+1} overwrites W_{k-1} (oth) - except the last one, which reads oth and lands in Wout
        int rc = ds_spmm_union(1, L.level_tag, L.utab, L.ctab, L.ngroups, L.cap_blocks, L.gent, L.kgrp, L.nnzb, L.nv, cur, lds,
                               last ? Wout : oth, last ? ldw : lds, R, ldr, L.dinv, ncols, c1, c2, k == 0 ? 1 : 0,
                               last ? oth : nullptr, last ? lds : 0, stream);
        if (rc != DS_OK) return rc;
        float* t = cur;
        cur = oth, oth = t;
    }
    return DS_OK;
}

// The level's bf16 term: the MFMA form when the level carries its tables, else the VALU kernel.
int union16(const ds_level_t& L, int epilogue, const void* X, int64_t ldx, void* Y, int64_t ldy, int y_f32, const void* R0,
            int64_t ldr, const float* dinv, int ncols, float c1, float c2, int first, const void* Wprev, int64_t ldp,
            ds_stream_t stream) {
    if (L.mf_group_nodes)
        return ds_spmm_union16m(epilogue, L.mf_group_nodes, L.level_tag, L.mf_gptr, L.mf_gcol, L.mf_gmeta, L.mf_gbase, L.mf_ghead, L.mf_kc,
                                L.mf_nblocks > 0 ? L.mf_nblocks : L.nnzb,
                                (L.nv + L.mf_group_nodes - 1) / L.mf_group_nodes, L.mf_max_entries, L.mf_max_batch_blocks, L.nv, X, ldx, Y,
                                ldy, y_f32,
                                R0, ldr, dinv, ncols, c1, c2, first, Wprev, ldp, stream);
    return ds_spmm_union16(epilogue, L.utab, L.ctab, L.ngroups, L.cap_blocks, L.gent, L.kgrp, L.nnzb, L.nv, X, ldx, Y, ldy, y_f32,
                           R0, ldr, dinv, ncols, c1, c2, first, Wprev, ldp, stream);
}

// The same recurrence on bf16 blocks (ds_spmm_union16): R16 is the bf16 right-hand side every term reads; the iterates
// ping-pong between the bf16 scratch blocks A and B; the LAST term writes Wout - bf16, or fp32 when out32 (the result
// that goes back to the solver).  Not from a guess: W_1 = T Rinit / theta, Rinit fp32 (rinit_f32: the solver's residual
// block, whose bf16 copy is written to R16 on the way) or bf16 (then Rinit == R16).
int chebyshev16(const ds_level_t& L, const void* Rinit, int rinit_f32, int64_t ldri, void* R16, int64_t ldr, void* Wout,
                int64_t ldw, int out32, void* A, void* B, int64_t lds, int ncols, bool from_guess, ds_stream_t stream) {
    const double theta = 0.5 * (L.lmax + L.lmin), delta = 0.5 * (L.lmax - L.lmin);
    const double sigma1 = theta / delta;
    double rho = 1.0 / sigma1;
    const int terms = from_guess ? L.degree : L.degree - 1;
    if (terms == 0 && out32) {
        ds::set_error("chebyshev16: a degree-1 polynomial has no fused term to convert its result");
        return DS_ERR_ARG;
    }
    void *cur = A, *oth = B;
    if (L.tgrp) {  // group-block Jacobi: the right-hand side of every term is T_g R (bf16, in R16); L.dinv is an identity
        if (from_guess || !L.mf_group_nodes) {
            ds::set_error("chebyshev16: the group-block Jacobi serves the polynomial from a zero guess on the matrix-core tables");
            return DS_ERR_ARG;
        }
        int rc = ds_group_apply16(L.tgrp, L.mf_group_nodes, Rinit, rinit_f32, ldri, R16, 0, ldr, L.nv, ncols, stream);
        if (rc != DS_OK) return rc;
        Rinit = R16, rinit_f32 = 0, ldri = ldr;
    }
    if (!from_guess) {
        int rc = ds_cheb_init16(Rinit, rinit_f32, ldri, terms == 0 ? Wout : cur, terms == 0 ? ldw : lds,
                                rinit_f32 ? R16 : nullptr, ldr, L.dinv, L.nv, ncols, (float)(1.0 / theta), stream);
        if (rc != DS_OK) return rc;
    }
    for (int k = 0; k < terms; ++k) {
        float c1, c2;
        if (from_guess && k == 0) {
            c1 = 0.f, c2 = (float)(1.0 / theta);
        } else {
            const double rho_new = 1.0 / (2.0 * sigma1 - rho);
            c1 = (float)(rho_new * rho), c2 = (float)(2.0 * rho_new / delta);
            rho = rho_new;
        }
        const bool last = k == terms - 1;
        int rc = union16(L, 1, cur, lds, last ? Wout : oth, last ? ldw : lds, last ? out32 : 0, R16, ldr, L.dinv, ncols, c1, c2,
                         k == 0 ? 1 : 0, last ? oth : nullptr, last ? lds : 0, stream);
        if (rc != DS_OK) return rc;
        void* t = cur;
        cur = oth, oth = t;
    }
    return DS_OK;
}

int twolevel16(const ds_twolevel_t* p, ds_stream_t stream) {
    const int c = p->ncols;
    if (p->fine.tgrp) {  // (its residual launch and its post-smoothing from a guess need K itself)
        ds::set_error("ds_twolevel_apply16: the group-block Jacobi is the corner-node level's");
        return DS_ERR_ARG;
    }
    // W1 = S R (bf16 iterates in Wc / D / AD; R16 = bf16 copy of R, written by the first step)
    int rc = chebyshev16(p->fine, p->R, 1, p->ldr, p->R16, p->ldr16, p->Wc, p->ldwc, 0, p->D, p->AD, p->ldd, c, false, stream);
    if (rc != DS_OK) return rc;
    rc = union16(p->fine, 2, p->Wc, p->ldwc, p->Rr, p->ldrr, 0, p->R16, p->ldr16, nullptr, c, 0.f, 0.f, 0, nullptr, 0, stream);
    if (rc != DS_OK) return rc;
    rc = ds_scalar_csr_spmm16(p->rptr, p->rcol, p->rw, p->coarse.nv, p->Rr, p->ldrr, p->Rc, p->ldc, c, 0.f, stream);
    if (rc != DS_OK) return rc;
    rc = chebyshev16(p->coarse, p->Rc, 0, p->ldc, p->Rc, p->ldc, p->Ec, p->ldc, 0, p->Dc, p->ADc, p->ldc, c, false, stream);
    if (rc != DS_OK) return rc;
    rc = ds_scalar_csr_spmm16(p->pptr, p->pcol, p->pw, p->fine.nv, p->Ec, p->ldc, p->Wc, p->ldwc, c, 1.f, stream);
    if (rc != DS_OK) return rc;
    // W = W2 + S (R - K W2): from the guess W2 (in Wc); the last term writes the fp32 result into W
    return chebyshev16(p->fine, nullptr, 0, 0, p->R16, p->ldr16, p->W, p->ldw, 1, p->Wc, p->D, p->ldd, c, true, stream);
}

}  // namespace

extern "C" int ds_chebyshev_apply16(const ds_level_t* level, const float* R, int64_t ldr, float* W, int64_t ldw, void* a,
                                    void* b, void* r16, int64_t lds, int ncols, ds_stream_t stream) {
    DS_REQUIRE(level && R && W && a && b && r16, "ds_chebyshev_apply16: null pointer");
    DS_REQUIRE(level->degree >= 2 && level->lmax > level->lmin && level->lmin > 0,
               "ds_chebyshev_apply16: needs a polynomial of degree >= 2");
    return chebyshev16(*level, R, 1, ldr, r16, lds, W, ldw, 1, a, b, lds, ncols, false, stream);
}

extern "C" int ds_chebyshev_apply(const ds_level_t* level, const float* R, int64_t ldr, float* W, int64_t ldw, float* a,
                                  float* b, int64_t lds, int ncols, ds_stream_t stream) {
    DS_REQUIRE(level && R && W && a && b, "ds_chebyshev_apply: null pointer");
    DS_REQUIRE(level->degree >= 1 && level->lmax > level->lmin && level->lmin > 0, "ds_chebyshev_apply: bad polynomial");
    return chebyshev(*level, R, ldr, W, ldw, a, b, lds, ncols, false, stream);
}

extern "C" int ds_twolevel_apply(const ds_twolevel_t* p, ds_stream_t stream) {
    DS_REQUIRE(p, "ds_twolevel_apply: null descriptor");
    DS_REQUIRE(p->R && p->W && p->Wc && p->D && p->AD && p->Rr && p->Rc && p->Ec && p->Dc && p->ADc,
               "ds_twolevel_apply: null block pointer");
    DS_REQUIRE(p->rptr && p->rcol && p->rw && p->pptr && p->pcol && p->pw, "ds_twolevel_apply: null transfer operator");
    DS_REQUIRE(p->ncols > 0 && p->ncols % 4 == 0 && p->ncols <= 84, "ds_twolevel_apply: ncols must be a multiple of 4 <= 84");
    DS_REQUIRE(p->ldwc == p->ldd && p->lda == p->ldd, "ds_twolevel_apply: Wc, D and AD must share one leading dimension");
    DS_REQUIRE(p->fine.degree >= 1 && p->coarse.degree >= 1 && p->fine.lmax > p->fine.lmin && p->fine.lmin > 0.0 &&
                   p->coarse.lmax > p->coarse.lmin && p->coarse.lmin > 0.0,
               "ds_twolevel_apply: bad polynomial degrees / spectral intervals");
    if (p->storage == 1) {
        DS_REQUIRE(p->R16 && p->ldr16 >= p->ncols && p->fine.degree >= 1, "ds_twolevel_apply: bf16 storage needs the R16 block");
        return twolevel16(p, stream);
    }
    DS_REQUIRE(p->storage == 0, "ds_twolevel_apply: storage must be 0 (fp32) or 1 (bf16)");
    const int c = p->ncols;
    // fine-level iterates live in the compact blocks Wc / D / AD; W is written once, by the last term of the cycle
    int rc = chebyshev(p->fine, p->R, p->ldr, p->Wc, p->ldwc, p->D, p->AD, p->ldd, c, false, stream);  // W1 = S R
    if (rc != DS_OK) return rc;
    // Rr = R - K W1 ;  Rc = P^T Rr
    rc = ds_spmm_union(2, p->fine.level_tag, p->fine.utab, p->fine.ctab, p->fine.ngroups, p->fine.cap_blocks, p->fine.gent, p->fine.kgrp,
                       p->fine.nnzb, p->fine.nv, p->Wc, p->ldwc, p->Rr, p->ldrr, p->R, p->ldr, nullptr, c, 0.f, 0.f, 0,
                       nullptr, 0, stream);
    if (rc != DS_OK) return rc;
    rc = ds_scalar_csr_spmm(p->rptr, p->rcol, p->rw, p->coarse.nv, p->Rr, p->ldrr, p->Rc, p->ldc, c, 0.f, stream);
    if (rc != DS_OK) return rc;
    // Ec = C Rc ;  W2 = W1 + P Ec
    rc = chebyshev(p->coarse, p->Rc, p->ldc, p->Ec, p->ldc, p->Dc, p->ADc, p->ldc, c, false, stream);
    if (rc != DS_OK) return rc;
    rc = ds_scalar_csr_spmm(p->pptr, p->pcol, p->pw, p->fine.nv, p->Ec, p->ldc, p->Wc, p->ldwc, c, 1.f, stream);
    if (rc != DS_OK) return rc;
    // W = W2 + S (R - K W2): the smoother's iteration started from W2 (in Wc), its last term lands in W
    return chebyshev(p->fine, p->R, p->ldr, p->W, p->ldw, p->Wc, p->D, p->ldd, c, true, stream);
}
