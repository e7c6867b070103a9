// Tall-skinny Gram  G = A^T B  (p x q) with fp64 MFMA accumulation - gfx950.
//
// Replaces the dense (k x n)(n x k) torch.matmul products that the reference's Rayleigh-Ritz,
// svqb and ortho steps perform (src/lobpcg/_linalg_utils.py:64-73, _lobpcg.py:459,516-525,
// 543-547,627-650).  n is 1e4..4e6, p,q <= 3*block <= ~400: a split-K GEMM whose K dimension is
// the long mesh dimension.
//
// v_mfma_f64_16x16x4_f64: lane l supplies A[i = l&15][k = l>>4] and B[k = l>>4][j = l&15] and holds
// D[row = (l>>4) + 4*reg][col = l&15].  With "k" = 4 consecutive mesh rows, both operands are read
// as 16 consecutive columns of one row per 16-lane group: coalesced 64-byte segments straight from
// the row-major blocks, no transposition, fp32 -> fp64 conversion in registers (products are then
// exact, accumulation is fp64: the Rayleigh-Ritz matrix keeps ~1e-16 relative accuracy although
// the iterates are fp32).
//
// Decomposition: a wave owns a (16*TI x 16*TJ) output tile in registers; a workgroup = 2x2 waves
// streams the same rows (shared through L1/L2); grid.y splits the rows; partial tiles go to a
// workspace [nsplit][p][q] that a second kernel sums in fixed order (deterministic, no atomics).
#include <algorithm>

#include "ds_common.h"

namespace {

using d4 = __attribute__((ext_vector_type(4))) double;

constexpr int TI = 3, TJ = 3;          // MFMA tiles per wave in i / j
constexpr int WT = 16 * TI;            // wave tile edge (48)
constexpr int BT = 2 * WT;             // workgroup tile edge (96)
constexpr int ROWS_PER_STEP = 4;

template <typename TB>
__global__ void __launch_bounds__(256)
    gram_partial_kernel(const float* __restrict__ A, int64_t lda, int p, const TB* __restrict__ B, int64_t ldb, int q,
                        int64_t n, int64_t rows_per_split, int tiles_j, double* __restrict__ ws) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int ti = blockIdx.x / tiles_j, tj = blockIdx.x - ti * tiles_j;
    const int i0 = ti * BT + (wave >> 1) * WT;
    const int j0 = tj * BT + (wave & 1) * WT;
    const int64_t r_begin = (int64_t)blockIdx.y * rows_per_split;
    const int64_t r_end = min(n, r_begin + rows_per_split);
    const int lc = lane & 15, lr = lane >> 4;

    d4 acc[TI][TJ];
#pragma unroll
    for (int a = 0; a < TI; ++a)
#pragma unroll
        for (int b = 0; b < TJ; ++b) acc[a][b] = d4{0.0, 0.0, 0.0, 0.0};

    bool ia[TI], jb[TJ];
#pragma unroll
    for (int a = 0; a < TI; ++a) ia[a] = (i0 + a * 16 + lc) < p;
#pragma unroll
    for (int b = 0; b < TJ; ++b) jb[b] = (j0 + b * 16 + lc) < q;
    const bool wave_active = (i0 < p) && (j0 < q);

    if (wave_active) {
#pragma unroll 4
        for (int64_t r = r_begin + lr; r < r_end + lr; r += ROWS_PER_STEP) {
            // r - lr is wave-uniform; rows past r_end contribute zeros
            const bool rv = r < r_end;
            const float* ap = A + r * lda + i0 + lc;
            const TB* bp = B + r * ldb + j0 + lc;
            double av[TI], bv[TJ];
#pragma unroll
            for (int a = 0; a < TI; ++a) av[a] = (rv && ia[a]) ? (double)ap[a * 16] : 0.0;
#pragma unroll
            for (int b = 0; b < TJ; ++b) bv[b] = (rv && jb[b]) ? (double)bp[b * 16] : 0.0;
#pragma unroll
            for (int a = 0; a < TI; ++a)
#pragma unroll
                for (int b = 0; b < TJ; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[a], bv[b], acc[a][b], 0, 0, 0);
        }
    }
    // partial tile -> workspace (every element of [p][q] is written by exactly one lane per split)
    double* w = ws + (int64_t)blockIdx.y * p * q;
#pragma unroll
    for (int a = 0; a < TI; ++a)
#pragma unroll
        for (int b = 0; b < TJ; ++b)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int row = i0 + a * 16 + lr + 4 * g;
                const int col = j0 + b * 16 + lc;
                if (row < p && col < q) w[(int64_t)row * q + col] = acc[a][b][g];
            }
}

__global__ void gram_reduce_kernel(const double* __restrict__ ws, int nsplit, int64_t pq, double* __restrict__ G) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= pq) return;
    double s = 0.0;
    for (int k = 0; k < nsplit; ++k) s += ws[(int64_t)k * pq + i];
    G[i] = s;
}

struct Plan {
    int tiles_i, tiles_j, nsplit;
    int64_t rows_per_split;
};

Plan make_plan(int64_t n, int p, int q) {
    Plan pl;
    pl.tiles_i = (int)ds::ceil_div(p, BT);
    pl.tiles_j = (int)ds::ceil_div(q, BT);
    const int64_t tiles = (int64_t)pl.tiles_i * pl.tiles_j;
    int64_t nsplit = ds::ceil_div(1024, tiles);               // ~4 workgroups per CU
    nsplit = std::min<int64_t>(nsplit, ds::ceil_div(n, 256));  // at least 256 rows per split
    nsplit = std::max<int64_t>(nsplit, 1);
    int64_t rps = ds::ceil_div(n, nsplit);
    rps = ds::ceil_div(rps, ROWS_PER_STEP) * ROWS_PER_STEP;
    pl.rows_per_split = rps;
    pl.nsplit = (int)ds::ceil_div(n, rps);
    return pl;
}

}  // namespace

extern "C" int64_t ds_gram_workspace_bytes(int64_t n, int p, int q) {
    if (n <= 0 || p <= 0 || q <= 0) return 0;
    const Plan pl = make_plan(n, p, q);
    return (int64_t)pl.nsplit * p * q * (int64_t)sizeof(double);
}

extern "C" int ds_gram(const float* A, int64_t lda, int p, const void* B, int b_dtype, int64_t ldb, int q, int64_t n,
                       double* G, void* work, int64_t work_bytes, ds_stream_t stream) {
    DS_REQUIRE(A && B && G && work, "ds_gram: null pointer");
    DS_REQUIRE(n > 0 && p > 0 && q > 0, "ds_gram: empty problem");
    DS_REQUIRE(lda >= p && ldb >= q, "ds_gram: leading dimension smaller than the block width");
    DS_REQUIRE(b_dtype == DS_F32 || b_dtype == DS_F64, "ds_gram: bad dtype code %d", b_dtype);
    const Plan pl = make_plan(n, p, q);
    DS_REQUIRE(work_bytes >= (int64_t)pl.nsplit * p * q * (int64_t)sizeof(double),
               "ds_gram: workspace too small (%lld bytes given)", (long long)work_bytes);
    hipStream_t st = ds::as_stream(stream);
    dim3 grid((unsigned)(pl.tiles_i * pl.tiles_j), (unsigned)pl.nsplit);
    double* ws = static_cast<double*>(work);
    if (b_dtype == DS_F32)
        gram_partial_kernel<float><<<grid, 256, 0, st>>>(A, lda, p, static_cast<const float*>(B), ldb, q, n,
                                                         pl.rows_per_split, pl.tiles_j, ws);
    else
        gram_partial_kernel<double><<<grid, 256, 0, st>>>(A, lda, p, static_cast<const double*>(B), ldb, q, n,
                                                          pl.rows_per_split, pl.tiles_j, ws);
    DS_LAUNCH_CHECK("gram_partial_kernel");
    const int64_t pq = (int64_t)p * q;
    gram_reduce_kernel<<<(unsigned)ds::ceil_div(pq, 256), 256, 0, st>>>(ws, pl.nsplit, pq, G);
    DS_LAUNCH_CHECK("gram_reduce_kernel");
    return DS_OK;
}
