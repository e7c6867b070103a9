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

// W <- p(T K) T R by the three-term Chebyshev recurrence W_{k+1} = W_k + c1 (W_k - W_{k-1}) + c2 T (R - K W_k)
// (Saad, Iterative Methods, Alg. 12.1), ping-ponging between W and D; from_guess: W holds W_0 and `degree` terms
// of the ITERATION for K W = R are run from it.  Mirrors ChebyshevBlockJacobi._fused.
int chebyshev(const ds_level_t& L, const float* R, int64_t ldr, float* W, int64_t ldw, float* D, int64_t ldd,
              float* AD, int64_t lda, int ncols, bool from_guess, ds_stream_t stream) {
    const double theta = 0.5 * (L.lmax + L.lmin), delta = 0.5 * (L.lmax - L.lmin);
    const double sigma1 = theta / delta;
    double rho = 1.0 / sigma1;
    int terms;
    float *cur, *oth;
    int64_t ldcur, ldoth;
    if (from_guess) {
        terms = L.degree;
        cur = W, ldcur = ldw, oth = D, ldoth = ldd;
    } else {
        terms = L.degree - 1;
        const bool w_first = terms % 2 == 0;  // so that the last term lands in W
        cur = w_first ? W : D, ldcur = w_first ? ldw : ldd;
        oth = w_first ? D : W, ldoth = w_first ? ldd : ldw;
        int rc = ds_cheb_init(R, ldr, AD, lda, cur, ldcur, L.dinv, L.nv, ncols, (float)(1.0 / theta), stream);
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
        int rc = ds_spmm_union(1, L.utab, L.ctab, L.ngroups, L.cap_blocks, L.gent, L.kgrp, L.nnzb, L.nv, cur, ldcur, oth,
                               ldoth, R, ldr, L.dinv, ncols, c1, c2, k == 0 ? 1 : 0, stream);
        if (rc != DS_OK) return rc;
        float* t = cur;
        cur = oth, oth = t;
        const int64_t tl = ldcur;
        ldcur = ldoth, ldoth = tl;
    }
    if (cur != W) {
        int rc = ds::check_hip(hipMemcpy2DAsync(W, (size_t)ldw * 4, cur, (size_t)ldcur * 4, (size_t)ncols * 4,
                                                (size_t)(3 * L.nv), hipMemcpyDeviceToDevice, ds::as_stream(stream)),
                               "ds_twolevel_apply: hipMemcpy2DAsync");
        if (rc != DS_OK) return rc;
    }
    return DS_OK;
}

}  // namespace

extern "C" int ds_twolevel_apply(const ds_twolevel_t* p, ds_stream_t stream) {
    DS_REQUIRE(p, "ds_twolevel_apply: null descriptor");
    DS_REQUIRE(p->R && p->W && p->D && p->AD && p->Rr && p->Rc && p->Ec && p->Dc && p->ADc,
               "ds_twolevel_apply: null block pointer");
    DS_REQUIRE(p->rptr && p->rcol && p->rw && p->pptr && p->pcol && p->pw, "ds_twolevel_apply: null transfer operator");
    DS_REQUIRE(p->ncols > 0 && p->ncols % 4 == 0 && p->ncols <= 84, "ds_twolevel_apply: ncols must be a multiple of 4 <= 84");
    DS_REQUIRE(p->fine.degree >= 1 && p->coarse.degree >= 2 && p->fine.lmax > p->fine.lmin && p->fine.lmin > 0.0 &&
                   p->coarse.lmax > p->coarse.lmin && p->coarse.lmin > 0.0,
               "ds_twolevel_apply: bad polynomial degrees / spectral intervals");
    const int c = p->ncols;
    int rc;
    // W1 = S R
    if (p->fine.degree > 1)
        rc = chebyshev(p->fine, p->R, p->ldr, p->W, p->ldw, p->D, p->ldd, p->AD, p->lda, c, false, stream);
    else
        rc = ds_cheb_init(p->R, p->ldr, p->D, p->ldd, p->W, p->ldw, p->fine.dinv, p->fine.nv, c,
                          (float)(1.0 / (0.5 * (p->fine.lmax + p->fine.lmin))), stream);
    if (rc != DS_OK) return rc;
    // Rr = R - K W1 ;  Rc = P^T Rr
    rc = ds_spmm_union(2, p->fine.utab, p->fine.ctab, p->fine.ngroups, p->fine.cap_blocks, p->fine.gent, p->fine.kgrp,
                       p->fine.nnzb, p->fine.nv, p->W, p->ldw, p->Rr, p->ldrr, p->R, p->ldr, nullptr, c, 0.f, 0.f, 0, stream);
    if (rc != DS_OK) return rc;
    rc = ds_scalar_csr_spmm(p->rptr, p->rcol, p->rw, p->coarse.nv, p->Rr, p->ldrr, p->Rc, p->ldc, c, 0.f, stream);
    if (rc != DS_OK) return rc;
    // Ec = C Rc ;  W2 = W1 + P Ec
    rc = chebyshev(p->coarse, p->Rc, p->ldc, p->Ec, p->ldc, p->Dc, p->ldc, p->ADc, p->ldc, c, false, stream);
    if (rc != DS_OK) return rc;
    rc = ds_scalar_csr_spmm(p->pptr, p->pcol, p->pw, p->fine.nv, p->Ec, p->ldc, p->W, p->ldw, c, 1.f, stream);
    if (rc != DS_OK) return rc;
    // W = W2 + S (R - K W2): the smoother's iteration started from W2
    return chebyshev(p->fine, p->R, p->ldr, p->W, p->ldw, p->D, p->ldd, p->AD, p->lda, c, true, stream);
}
