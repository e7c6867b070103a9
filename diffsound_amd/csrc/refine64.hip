// Element-wise passes of the fp64 refinement (configs[4]) fused - gfx950.
//
// A refinement step of lobpcg/modal_solver.py::refine64 needs, from the fp64 blocks K X, M X, X (n x b, b <= 168): the residual
// norms ||K x_j - lam_j M x_j|| and ||x_j|| of every column, and then the residual columns of the pairs that are still active,
// scaled to unit norm, as an fp32 block (the input of the fp32 / bf16 preconditioner).  As torch operations that was addcmul,
// two norms, a column gather, a division, a cast and a copy: ~10 passes over 4.5 GB blocks at configs[4] (14 ms of a 117 ms
// step).  Here: ONE pass over the three blocks for the norms (no residual block is written), ONE pass over two of them that
// forms the residual again, scales it and writes the fp32 columns.  (reference: update_residual / update_converged_count of
// src/lobpcg/_lobpcg.py:301-333, in fp64.)  Sums in a fixed order: reproducible.
#include <algorithm>

#include "ds_common.h"

namespace {

using d2 = __attribute__((ext_vector_type(2))) double;
using f4 = __attribute__((ext_vector_type(4))) float;
constexpr int R64_BLOCKS = 1024;  // row shares of the norm pass

// partial[blk][0:b] = sum over the block's rows of r^2, partial[blk][b:2b] = of x^2; a thread owns 2 columns and every
// (256 / (b / 2))-th row of its block's share; the row lanes are added through LDS in lane order
__global__ void __launch_bounds__(256)
    residual64_norms_kernel(const double* __restrict__ KX, int64_t ldk, const double* __restrict__ MX, int64_t ldm,
                            const double* __restrict__ X, int64_t ldx, const double* __restrict__ lam, int64_t n, int b,
                            double* __restrict__ partial) {
    __shared__ double sm[256 * 4];
    const int cg = b / 2, nrl = 256 / cg;
    const int rl = threadIdx.x / cg, c2 = (threadIdx.x - rl * cg) * 2;
    const int64_t per = (n + R64_BLOCKS - 1) / R64_BLOCKS;
    const int64_t r0 = (int64_t)blockIdx.x * per, r1 = min(n, r0 + per);
    double sr0 = 0.0, sr1 = 0.0, sx0 = 0.0, sx1 = 0.0;
    if (rl < nrl) {
        const double l0 = lam[c2], l1 = lam[c2 + 1];
        for (int64_t r = r0 + rl; r < r1; r += nrl) {
            const d2 k = *reinterpret_cast<const d2*>(KX + r * ldk + c2);
            const d2 m = *reinterpret_cast<const d2*>(MX + r * ldm + c2);
            const d2 x = *reinterpret_cast<const d2*>(X + r * ldx + c2);
            const double a = fma(-m[0], l0, k[0]), c = fma(-m[1], l1, k[1]);
            sr0 = fma(a, a, sr0), sr1 = fma(c, c, sr1);
            sx0 = fma(x[0], x[0], sx0), sx1 = fma(x[1], x[1], sx1);
        }
    }
    sm[threadIdx.x * 4 + 0] = sr0, sm[threadIdx.x * 4 + 1] = sr1, sm[threadIdx.x * 4 + 2] = sx0, sm[threadIdx.x * 4 + 3] = sx1;
    __syncthreads();
    for (int j = threadIdx.x; j < 2 * b; j += blockDim.x) {
        const int col = j < b ? j : j - b, which = (j < b ? 0 : 2) + (col & 1);
        double t = 0.0;
        for (int q = 0; q < nrl; ++q) t += sm[(q * cg + col / 2) * 4 + which];
        partial[(size_t)blockIdx.x * 2 * b + j] = t;
    }
}

__global__ void __launch_bounds__(256)
    residual64_final_kernel(const double* __restrict__ partial, int b, double* __restrict__ rn2, double* __restrict__ xn2) {
    // one workgroup per output column: R64_BLOCKS partials, four per thread, added by a fixed tree
    __shared__ double sm[256];
    const int j = blockIdx.x;
    double t = 0.0;
    for (int q = threadIdx.x; q < R64_BLOCKS; q += 256) t += partial[(size_t)q * 2 * b + j];
    sm[threadIdx.x] = t;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) sm[threadIdx.x] += sm[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (j < b) rn2[j] = sm[0];
        else xn2[j - b] = sm[0];
    }
}

// R[row][j .. j+3] = (float)((K x - lam M x)[row][cols[j .. j+3]] * scale[cols[...]]): a thread writes 16 bytes
__global__ void __launch_bounds__(256)
    residual64_scaled_kernel(const double* __restrict__ KX, int64_t ldk, const double* __restrict__ MX, int64_t ldm,
                             const double* __restrict__ lam, const double* __restrict__ scale, const int32_t* __restrict__ cols,
                             int nact, float* __restrict__ R, int64_t ldr, int64_t n) {
    const int q4 = nact / 4;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t row = t / q4;
    if (row >= n) return;
    const int j = (int)(t - row * q4) * 4;
    f4 out;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int c = cols[j + u];
        const double r = fma(-MX[row * ldm + c], lam[c], KX[row * ldk + c]);
        out[u] = (float)(r * scale[c]);
    }
    *reinterpret_cast<f4*>(R + row * ldr + j) = out;
}

}  // namespace

extern "C" int64_t ds_residual64_workspace_doubles(int b) { return (int64_t)R64_BLOCKS * 2 * b; }

extern "C" int ds_residual64_norms(const double* KX, int64_t ldk, const double* MX, int64_t ldm, const double* X, int64_t ldx,
                                   const double* lam, int64_t n, int b, double* work, int64_t work_doubles, double* rn2,
                                   double* xn2, ds_stream_t stream) {
    DS_REQUIRE(KX && MX && X && lam && work && rn2 && xn2, "ds_residual64_norms: null pointer");
    DS_REQUIRE(n > 0 && b > 0 && b % 2 == 0 && b <= 512, "ds_residual64_norms: b must be even and <= 512 (got %d)", b);
    DS_REQUIRE(ldk >= b && ldm >= b && ldx >= b, "ds_residual64_norms: leading dimension smaller than b");
    const uintptr_t al = reinterpret_cast<uintptr_t>(KX) | reinterpret_cast<uintptr_t>(MX) | reinterpret_cast<uintptr_t>(X) |
                         (uintptr_t)(ldk * 8) | (uintptr_t)(ldm * 8) | (uintptr_t)(ldx * 8);
    DS_REQUIRE((al & 15) == 0, "ds_residual64_norms: rows must be 16-byte aligned");
    DS_REQUIRE(work_doubles >= ds_residual64_workspace_doubles(b), "ds_residual64_norms: workspace of %lld doubles needed",
               (long long)ds_residual64_workspace_doubles(b));
    hipStream_t st = ds::as_stream(stream);
    residual64_norms_kernel<<<R64_BLOCKS, 256, 0, st>>>(KX, ldk, MX, ldm, X, ldx, lam, n, b, work);
    DS_LAUNCH_CHECK("residual64_norms_kernel");
    residual64_final_kernel<<<(unsigned)(2 * b), 256, 0, st>>>(work, b, rn2, xn2);
    DS_LAUNCH_CHECK("residual64_final_kernel");
    return DS_OK;
}

extern "C" int ds_residual64_scaled(const double* KX, int64_t ldk, const double* MX, int64_t ldm, const double* lam,
                                    const double* scale, const int32_t* cols, int nact, float* R, int64_t ldr, int64_t n,
                                    ds_stream_t stream) {
    DS_REQUIRE(KX && MX && lam && scale && cols && R, "ds_residual64_scaled: null pointer");
    DS_REQUIRE(n > 0 && nact > 0 && nact % 4 == 0 && ldr >= nact, "ds_residual64_scaled: nact must be a multiple of 4 <= ldr");
    DS_REQUIRE(((reinterpret_cast<uintptr_t>(R) | (uintptr_t)(ldr * 4)) & 15) == 0, "ds_residual64_scaled: rows of R must be 16-byte aligned");
    const int64_t threads = n * (nact / 4);
    residual64_scaled_kernel<<<(unsigned)ds::ceil_div(threads, 256), 256, 0, ds::as_stream(stream)>>>(KX, ldk, MX, ldm, lam, scale,
                                                                                                  cols, nact, R, ldr, n);
    DS_LAUNCH_CHECK("residual64_scaled_kernel");
    return DS_OK;
}
