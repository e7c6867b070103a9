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
// Decomposition: a wave owns a (48 x 48) output tile = 3x3 MFMA accumulators (72 VGPRs); a workgroup
// = 2x2 waves streams the same 16-row batches (shared through L1/L2); grid.y splits the rows.  The
// row loop is software-pipelined by hand: the 24 operand loads of batch t+1 are in flight while the
// 36 MFMAs of batch t issue (the first version waited on every 4-row step and ran 10x below the
// MFMA rate).  Partial tiles go to a workspace [nsplit][p][q] that a second kernel sums in fixed
// order: deterministic, no atomics.  With `symmetric`, only workgroup tiles on or above the
// diagonal are computed and the reduction mirrors them (A^T (K A) for symmetric K).
#include <algorithm>
#include <cstdlib>

#include "ds_common.h"

namespace {

using d4 = __attribute__((ext_vector_type(4))) double;

constexpr int TI = 3, TJ = 3;  // MFMA tiles per wave in i / j
constexpr int WT = 16 * TI;    // wave tile edge (48)
constexpr int KS = 4;          // MFMA k-steps (of 4 rows) per pipelined batch
constexpr int RB = 4 * KS;     // rows per batch (16)

template <typename TB>
struct Batch {
    float a[KS][TI];
    TB b[KS][TJ];
};

// which of the wave tile's 3x3 MFMA tiles exist (wave-uniform): a < na, b < nb, and on a diagonal wave tile of
// a symmetric product only a <= b (the reduction mirrors the rest)
struct Active {
    int na, nb, diag;
    __device__ __forceinline__ bool operator()(int a, int b) const { return a < na && b < nb && (!diag || a <= b); }
};

template <typename TB>
__device__ __forceinline__ void load_batch(Batch<TB>& t, const float* __restrict__ A, int64_t lda,
                                           const TB* __restrict__ B, int64_t ldb, int64_t r0, int64_t r_end, int lr,
                                           int acol, int bcol, const bool (&ia)[TI], const bool (&jb)[TJ],
                                           const Active act) {
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int64_t r = r0 + 4 * s + lr;
        const bool rv = r < r_end;
        const float* ap = A + r * lda + acol;
        const TB* bp = B + r * ldb + bcol;
#pragma unroll
        for (int a = 0; a < TI; ++a)
            if (a < act.na) t.a[s][a] = (rv && ia[a]) ? ap[a * 16] : 0.f;
#pragma unroll
        for (int b = 0; b < TJ; ++b)
            if (b < act.nb) t.b[s][b] = (rv && jb[b]) ? bp[b * 16] : (TB)0;
    }
}

template <typename TB>
__device__ __forceinline__ void mfma_batch(const Batch<TB>& t, d4 (&acc)[TI][TJ], const Active act) {
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int a = 0; a < TI; ++a)
#pragma unroll
            for (int b = 0; b < TJ; ++b)
                if (act(a, b))
                    acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64((double)t.a[s][a], (double)t.b[s][b], acc[a][b], 0, 0, 0);
}

// One wave per (48 x 48) wave tile of the NEEDED part of G: all tiles for a general product, the block-upper
// triangle for a symmetric one; ragged edges and the diagonal tiles skip the MFMA tiles that are out of range
// or mirrored (the fp64 MFMA pipe is the bound of this kernel, idle tiles cost as much as useful ones).
// blockIdx.x: groups of 4 consecutive wave tiles (they share row batches through L1/L2); blockIdx.y: row split.
template <typename TB>
__global__ void __launch_bounds__(256)
    gram_partial_kernel(const float* __restrict__ A, int64_t lda, int p, const TB* __restrict__ B, int64_t ldb, int q,
                        int64_t n, int64_t rows_per_split, int ntj, int ntiles, int symmetric,
                        double* __restrict__ ws) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wt = blockIdx.x * 4 + wave;
    if (wt >= ntiles) return;  // wave-uniform; the kernel has no workgroup barrier
    int ti, tj;
    if (symmetric) {  // row-major walk of the upper triangle
        int rem = wt;
        ti = 0;
        while (rem >= ntj - ti) {
            rem -= ntj - ti;
            ++ti;
        }
        tj = ti + rem;
    } else {
        ti = wt / ntj;
        tj = wt - ti * ntj;
    }
    const int i0 = ti * WT, j0 = tj * WT;
    Active act;
    act.na = min(TI, (p - i0 + 15) >> 4);
    act.nb = min(TJ, (q - j0 + 15) >> 4);
    act.diag = symmetric && ti == tj;
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

    Batch<TB> cur, nxt;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
#pragma unroll
        for (int a = 0; a < TI; ++a) cur.a[s][a] = nxt.a[s][a] = 0.f;
#pragma unroll
        for (int b = 0; b < TJ; ++b) cur.b[s][b] = nxt.b[s][b] = (TB)0;
    }
    load_batch(cur, A, lda, B, ldb, r_begin, r_end, lr, i0 + lc, j0 + lc, ia, jb, act);
    for (int64_t r0 = r_begin; r0 < r_end; r0 += RB) {
        const int64_t rn = r0 + RB;
        if (rn < r_end) load_batch(nxt, A, lda, B, ldb, rn, r_end, lr, i0 + lc, j0 + lc, ia, jb, act);
        mfma_batch(cur, acc, act);
        cur = nxt;
    }
    // partial tile -> workspace (every element of the computed MFMA tiles is written by exactly one lane)
    double* w = ws + (int64_t)blockIdx.y * p * q;
#pragma unroll
    for (int a = 0; a < TI; ++a)
#pragma unroll
        for (int b = 0; b < TJ; ++b)
            if (act(a, b)) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int row = i0 + a * 16 + lr + 4 * g;
                    const int col = j0 + b * 16 + lc;
                    if (row < p && col < q) w[(int64_t)row * q + col] = acc[a][b][g];
                }
            }
}

// G[i][j] = sum_s ws[s][i][j]: 16 lanes share the split loop of one element (strided, so every lane has many
// independent loads in flight - the first version walked the splits with 4 lanes per element and took 94 us
// of pure load latency) and merge by a fixed-order shuffle tree: deterministic, no atomics.  With
// symmetric != 0 an element whose MFMA tile lies below the diagonal reads its mirror.
__global__ void __launch_bounds__(256)
    gram_reduce_kernel(const double* __restrict__ ws, int nsplit, int p, int q, int symmetric,
                       double* __restrict__ G) {
    const int e = threadIdx.x & 15, sub = threadIdx.x >> 4;  // 16 elements x 16 split lanes per workgroup
    const int64_t pq = (int64_t)p * q;
    const int64_t idx = (int64_t)blockIdx.x * 16 + e;
    double s0 = 0.0, s1 = 0.0;
    if (idx < pq) {
        int i = (int)(idx / q), j = (int)(idx - (int64_t)i * q);
        if (symmetric && (j >> 4) < (i >> 4)) {  // MFMA tiles below the diagonal are not computed
            const int t = i;
            i = j;
            j = t;
        }
        const double* src = ws + (int64_t)i * q + j;
        int k = sub;
        for (; k + 16 < nsplit; k += 32) {
            s0 += src[(int64_t)k * pq];
            s1 += src[(int64_t)(k + 16) * pq];
        }
        if (k < nsplit) s0 += src[(int64_t)k * pq];
    }
    double s = s0 + s1;
    // lanes of one element are 16 apart: threadIdx = sub * 16 + e  ->  xor 16, 32 inside the wave, then LDS
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    __shared__ double part[4][16];
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) < 16) part[wave][e] = s;
    __syncthreads();
    if (threadIdx.x < 16 && idx < pq) G[idx] = (part[0][e] + part[1][e]) + (part[2][e] + part[3][e]);
}

struct Plan {
    int ntj, ntiles, groups, nsplit;
    int64_t rows_per_split;
};

Plan make_plan(int64_t n, int p, int q, int symmetric) {
    Plan pl;
    const int nti = (int)ds::ceil_div(p, WT);
    pl.ntj = (int)ds::ceil_div(q, WT);
    pl.ntiles = symmetric ? nti * (nti + 1) / 2 : nti * pl.ntj;
    pl.groups = (int)ds::ceil_div(pl.ntiles, 4);
    // The kernel holds 152 VGPRs: 3 workgroups per CU, 768 resident on the chip.  The grid is sized to fill the
    // chip an integer number of times - a 1026-workgroup grid ran one full round and a second one at a third of
    // the occupancy, i.e. in the time of two (DS_GRAM_WGS overrides the target for experiments).
    static const int64_t target = getenv("DS_GRAM_WGS") ? atoll(getenv("DS_GRAM_WGS")) : 768;
    int64_t nsplit = std::max<int64_t>(1, target / pl.groups);
    nsplit = std::min<int64_t>(nsplit, ds::ceil_div(n, 512));   // at least 512 rows per split
    nsplit = std::max<int64_t>(nsplit, 1);
    int64_t rps = ds::ceil_div(n, nsplit);
    rps = ds::ceil_div(rps, RB) * RB;
    pl.rows_per_split = rps;
    pl.nsplit = (int)ds::ceil_div(n, rps);
    return pl;
}

}  // namespace

extern "C" int64_t ds_gram_workspace_bytes(int64_t n, int p, int q) {
    if (n <= 0 || p <= 0 || q <= 0) return 0;
    // the symmetric plan never needs more splits than the general one
    const Plan pl = make_plan(n, p, q, 0), ps = make_plan(n, p, q, p == q);
    return (int64_t)std::max(pl.nsplit, ps.nsplit) * p * q * (int64_t)sizeof(double);
}

extern "C" int ds_gram(const float* A, int64_t lda, int p, const void* B, int b_dtype, int64_t ldb, int q, int64_t n,
                       int symmetric, double* G, void* work, int64_t work_bytes, ds_stream_t stream) {
    DS_REQUIRE(A && B && G && work, "ds_gram: null pointer");
    DS_REQUIRE(n > 0 && p > 0 && q > 0, "ds_gram: empty problem");
    DS_REQUIRE(lda >= p && ldb >= q, "ds_gram: leading dimension smaller than the block width");
    DS_REQUIRE(b_dtype == DS_F32 || b_dtype == DS_F64, "ds_gram: bad dtype code %d", b_dtype);
    DS_REQUIRE(!symmetric || p == q, "ds_gram: symmetric needs p == q");
    const Plan pl = make_plan(n, p, q, symmetric);
    DS_REQUIRE(work_bytes >= (int64_t)pl.nsplit * p * q * (int64_t)sizeof(double),
               "ds_gram: workspace too small (%lld bytes given)", (long long)work_bytes);
    hipStream_t st = ds::as_stream(stream);
    dim3 grid((unsigned)pl.groups, (unsigned)pl.nsplit);
    double* ws = static_cast<double*>(work);
    if (b_dtype == DS_F32)
        gram_partial_kernel<float><<<grid, 256, 0, st>>>(A, lda, p, static_cast<const float*>(B), ldb, q, n,
                                                         pl.rows_per_split, pl.ntj, pl.ntiles, symmetric, ws);
    else
        gram_partial_kernel<double><<<grid, 256, 0, st>>>(A, lda, p, static_cast<const double*>(B), ldb, q, n,
                                                          pl.rows_per_split, pl.ntj, pl.ntiles, symmetric, ws);
    DS_LAUNCH_CHECK("gram_partial_kernel");
    const int64_t pq = (int64_t)p * q;
    gram_reduce_kernel<<<(unsigned)ds::ceil_div(pq, 16), 256, 0, st>>>(ws, pl.nsplit, p, q, symmetric, G);
    DS_LAUNCH_CHECK("gram_reduce_kernel");
    return DS_OK;
}
