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

// One wave tile with its shape as template arguments: NA x NB MFMA tiles, DIAG = a diagonal wave tile of a symmetric
// product (tiles a > b are mirrors: the fp64 MFMA pipe is the bound of this kernel, idle tiles cost as much as useful
// ones).  Every lane's three columns of A and of B come as (pointer to the lane's first row, row stride) pairs - a plain
// (n x p) operand and a LIST of blocks (ds_gram64_blocks) look the same from here.  With the shape known at compile time
// the row loop has no branch, and no load is predicated (the compiler turns a select behind a load into a branch around
// it with a vmcnt(0) behind it), so the compiler's wait counts are exact: the loads of batch t + 1 fly under the MFMAs of
// batch t.  (The first version of this kernel kept the tile counts in registers: every MFMA sat behind a branch and a
// full wait, and it ran at 40 % of what this one does.)  Full 16-row batches need no mask; the one ragged batch at the
// end of the rows reads clamped rows and zeroes A's values there.
template <typename TA, typename TB, int NA, int NB, bool DIAG>
__device__ __forceinline__ void gram_tile(const TA* const (&pa)[TI], const TB* const (&pb)[TJ], const int64_t (&sa)[TI],
                                          const int64_t (&sb)[TJ], int64_t nrows, int lr, int lc, int i0, int j0, int p,
                                          int q, double* __restrict__ w) {
    d4 acc[NA][NB];
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[a][b] = d4{0.0, 0.0, 0.0, 0.0};
    // (global address space, spelled out: behind a pointer selection the compiler no longer knows it and emits flat
    // loads, whose completion it can only wait for with vmcnt(0))
    using gpa = const __attribute__((address_space(1))) TA*;
    using gpb = const __attribute__((address_space(1))) TB*;
    gpa qa[NA];
    gpb qb[NB];
#pragma unroll
    for (int a = 0; a < NA; ++a) qa[a] = (gpa)pa[a];
#pragma unroll
    for (int b = 0; b < NB; ++b) qb[b] = (gpb)pb[b];
    struct Rows {
        TA a[KS][NA];
        TB b[KS][NB];
    };
    // a full batch (16 existing rows) at the pointers, which then move on by `adv` batches (0 at the end of the rows:
    // the prefetch behind the last batch re-reads it)
    auto load_full = [&](Rows& t, int adv) {
#pragma unroll
        for (int s = 0; s < KS; ++s) {
#pragma unroll
            for (int a = 0; a < NA; ++a) t.a[s][a] = qa[a][4 * s * sa[a]];
#pragma unroll
            for (int b = 0; b < NB; ++b) t.b[s][b] = qb[b][4 * s * sb[b]];
        }
#pragma unroll
        for (int a = 0; a < NA; ++a) qa[a] += adv * RB * sa[a];
#pragma unroll
        for (int b = 0; b < NB; ++b) qb[b] += adv * RB * sb[b];
    };
    auto mfma = [&](const Rows& t) {
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int a = 0; a < NA; ++a)
#pragma unroll
                for (int b = 0; b < NB; ++b)
                    if (!DIAG || a <= b)
                        acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64((double)t.a[s][a], (double)t.b[s][b], acc[a][b], 0, 0, 0);
    };
    const int64_t nfull = nrows / RB;
    Rows t0, t1;
    if (nfull > 0) {
        load_full(t0, nfull > 1 ? 1 : 0);
        __builtin_amdgcn_sched_barrier(0);
        for (int64_t k = 0; k < nfull; k += 2) {
            load_full(t1, k + 2 < nfull ? 1 : 0);  // batch k + 1 (or, behind the last batch, that one again: unused)
            __builtin_amdgcn_sched_barrier(0);
            mfma(t0);
            __builtin_amdgcn_sched_barrier(0);
            load_full(t0, k + 3 < nfull ? 1 : 0);  // batch k + 2
            __builtin_amdgcn_sched_barrier(0);
            if (k + 1 < nfull) mfma(t1);
            __builtin_amdgcn_sched_barrier(0);
        }
        // the pointers stand on the last full batch: move behind it
#pragma unroll
        for (int a = 0; a < NA; ++a) qa[a] += RB * sa[a];
#pragma unroll
        for (int b = 0; b < NB; ++b) qb[b] += RB * sb[b];
    }
    if (nfull * RB < nrows) {  // the ragged batch: rows past the end read the last existing row, A is zero there
        const int64_t left = nrows - nfull * RB;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int64_t rr = 4 * s + lr;  // row of the batch
            const int64_t back = rr < left ? 0 : rr - (left - 1);
#pragma unroll
            for (int a = 0; a < NA; ++a) {
                const TA v = qa[a][(4 * s - back) * sa[a]];
                t0.a[s][a] = rr < left ? v : (TA)0;
            }
#pragma unroll
            for (int b = 0; b < NB; ++b) t0.b[s][b] = qb[b][(4 * s - back) * sb[b]];
        }
        mfma(t0);
    }
    // partial tile -> workspace (every element of the computed MFMA tiles is written by exactly one lane);
    // fp64 C/D map: lane (lc, lr) holds rows lr + 4 g, column lc of each tile
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b)
            if (!DIAG || a <= b) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int row = i0 + a * 16 + lr + 4 * g;
                    const int col = j0 + b * 16 + lc;
                    if (row < p && col < q) w[(int64_t)row * q + col] = acc[a][b][g];
                }
            }
}

// the tile shape of a wave, from run-time counts to template arguments (wave-uniform)
#define DS_GRAM_DISPATCH(TA_, TB_, na_, nb_, diag_, ...)                                    \
    do {                                                                                    \
        if (diag_) {                                                                        \
            if ((na_) == 3) gram_tile<TA_, TB_, 3, 3, true>(__VA_ARGS__);                    \
            else if ((na_) == 2) gram_tile<TA_, TB_, 2, 2, true>(__VA_ARGS__);               \
            else gram_tile<TA_, TB_, 1, 1, true>(__VA_ARGS__);                               \
        } else {                                                                            \
            switch ((na_) * 4 + (nb_)) {                                                    \
                case 3 * 4 + 3: gram_tile<TA_, TB_, 3, 3, false>(__VA_ARGS__); break;        \
                case 3 * 4 + 2: gram_tile<TA_, TB_, 3, 2, false>(__VA_ARGS__); break;        \
                case 3 * 4 + 1: gram_tile<TA_, TB_, 3, 1, false>(__VA_ARGS__); break;        \
                case 2 * 4 + 3: gram_tile<TA_, TB_, 2, 3, false>(__VA_ARGS__); break;        \
                case 2 * 4 + 2: gram_tile<TA_, TB_, 2, 2, false>(__VA_ARGS__); break;        \
                case 2 * 4 + 1: gram_tile<TA_, TB_, 2, 1, false>(__VA_ARGS__); break;        \
                case 1 * 4 + 3: gram_tile<TA_, TB_, 1, 3, false>(__VA_ARGS__); break;        \
                case 1 * 4 + 2: gram_tile<TA_, TB_, 1, 2, false>(__VA_ARGS__); break;        \
                default: gram_tile<TA_, TB_, 1, 1, false>(__VA_ARGS__); break;               \
            }                                                                               \
        }                                                                                   \
    } while (0)

// which (48 x 48) wave tile of the NEEDED part of G a wave owns: all tiles for a general product, the block-upper
// triangle (row-major) for a symmetric one.  1-D grid, split-major: the groups of one row split are consecutive
// logical workgroups of ONE XCD and fetch the rows they share into one L2.
struct WaveTile {
    int split, ti, tj;
    bool any;
};
__device__ __forceinline__ WaveTile wave_tile(int groups, int ntj, int ntiles, int symmetric) {
    WaveTile t;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned logical = ds::xcd_remap(blockIdx.x, gridDim.x);
    t.split = (int)(logical / (unsigned)groups);
    const int wt = (int)(logical - (unsigned)t.split * groups) * 4 + wave;
    t.any = wt < ntiles;
    if (symmetric) {
        int rem = wt;
        t.ti = 0;
        while (t.any && rem >= ntj - t.ti) {
            rem -= ntj - t.ti;
            ++t.ti;
        }
        t.tj = t.ti + rem;
    } else {
        t.ti = wt / ntj;
        t.tj = wt - t.ti * ntj;
    }
    return t;
}

// One wave per (48 x 48) wave tile; workgroups of 4 consecutive wave tiles of one row split (they share its rows
// through L1/L2).
template <typename TA, typename TB>
__global__ void __launch_bounds__(256)
    gram_partial_kernel(const TA* __restrict__ A, int64_t lda, int p, const TB* __restrict__ B, int64_t ldb, int q,
                        int64_t n, int64_t rows_per_split, int ntj, int ntiles, int groups, int symmetric,
                        double* __restrict__ ws) {
    const WaveTile t = wave_tile(groups, ntj, ntiles, symmetric);
    if (!t.any) return;  // wave-uniform; the kernel has no workgroup barrier
    const int lane = threadIdx.x & 63;
    const int i0 = t.ti * WT, j0 = t.tj * WT;
    const int na = min(TI, (p - i0 + 15) >> 4), nb = min(TJ, (q - j0 + 15) >> 4);
    const bool diag = symmetric && t.ti == t.tj;
    const int64_t r_begin = (int64_t)t.split * rows_per_split;
    const int64_t nrows = min(n, r_begin + rows_per_split) - r_begin;
    const int lc = lane & 15, lr = lane >> 4;
    // a lane past p or q reads the last column: its products land in elements that are never stored
    const TA* pa[TI];
    const TB* pb[TJ];
    int64_t sa[TI], sb[TJ];
#pragma unroll
    for (int a = 0; a < TI; ++a) pa[a] = A + (r_begin + lr) * lda + min(i0 + a * 16 + lc, p - 1), sa[a] = lda;
#pragma unroll
    for (int b = 0; b < TJ; ++b) pb[b] = B + (r_begin + lr) * ldb + min(j0 + b * 16 + lc, q - 1), sb[b] = ldb;
    double* w = ws + (int64_t)t.split * p * q;
    DS_GRAM_DISPATCH(TA, TB, na, nb, diag, pa, pb, sa, sb, nrows, lr, lc, i0, j0, p, q, w);
}

// ---------------------------------------------------------------------------------------------------------------
// fp32 operands: the fast path.  v_mfma_f32_16x16x4_f32 runs at twice the fp64 rate and needs no conversions; its
// fp32 accumulators are folded into fp64 ones every 48 mesh rows, so a partial sum never grows past 12 chained
// MFMAs: measured error 9e-10 of |A_i|.|B_j| at n = 4.5e5 (4e-8 at n = 225: it is relative to the 48-row partials and
// averages out over the folds) - below the fp32 rounding already present in the operands.
//
// Loads are buffer loads of NA / NB consecutive dwords per lane (16 lanes x 12 B = the 48 columns of a wave tile in
// one instruction instead of three).  Lane lc then holds columns NA*lc + c, so MFMA tile c of the wave tile is the
// INTERLEAVED column set {i0 + NA*i + c}: an MFMA does not care which columns its 16 indices stand for, the
// write-out undoes the permutation.  Range checks come from the descriptor: it ends after the last wanted column
// of the split's last row, so rows past the split read as zero; columns past p / q inside a row read the
// neighbouring block's data, which only reaches output elements that are never written.
using f4 = __attribute__((ext_vector_type(4))) float;
using u3 = __attribute__((ext_vector_type(3))) unsigned;
using u2 = __attribute__((ext_vector_type(2))) unsigned;


__device__ __forceinline__ float as_f32(unsigned u) { return __builtin_bit_cast(float, u); }

template <int NW>
__device__ __forceinline__ void load_cols(float (&dst)[3], __amdgpu_buffer_rsrc_t rs, unsigned voff) {
    if constexpr (NW == 3) {
        const u3 v = __builtin_amdgcn_raw_buffer_load_b96(rs, voff, 0, 0);
        const unsigned a = v[0], b = v[1], c = v[2];
        dst[0] = as_f32(a), dst[1] = as_f32(b), dst[2] = as_f32(c);
    } else if constexpr (NW == 2) {
        const u2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, 0, 0);
        const unsigned a = v[0], b = v[1];
        dst[0] = as_f32(a), dst[1] = as_f32(b);
    } else {
        dst[0] = as_f32(__builtin_amdgcn_raw_buffer_load_b32(rs, voff, 0, 0));
    }
}

template <int NB2>
struct Batch32 {
    float a[KS][3], b[KS][3], c[KS][NB2 ? 3 : 1];
};

// One wave tile: NA MFMA tiles in i, NB1 + NB2 in j (B is read as two pieces of NB1 and NB2 interleaved columns:
// 48 + 32 columns cover the eigensolver's 80-column blocks in one wave tile).  DIAG: a symmetric product's diagonal
// wave tile, tiles a > b are mirrors.
template <int NA, int NB1, int NB2, bool DIAG>
__device__ __forceinline__ void gram32_tile(__amdgpu_buffer_rsrc_t ra, __amdgpu_buffer_rsrc_t rb, unsigned offa,
                                            unsigned offb, unsigned offc, unsigned stepa, unsigned stepb, int nbatch,
                                            int lr, int lc, int i0, int j0, int p, int q, double* __restrict__ w) {
    constexpr int NB = NB1 + NB2;
    f4 acc[NA][NB];
    double sum[NA][NB][4];
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            acc[a][b] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int g = 0; g < 4; ++g) sum[a][b][g] = 0.0;
        }
    auto load = [&](Batch32<NB2>& t, int batch) {
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            // + lr is part of offa / offb.  The row term stays a scalar added at the load (readfirstlane keeps the
            // compiler from turning the offsets of the ring into as many induction VGPR pairs); it has to go through
            // the VGPR offset, the descriptor's range check does not see an SGPR offset
            const unsigned row = (unsigned)__builtin_amdgcn_readfirstlane(batch * RB + 4 * s);
            load_cols<NA>(t.a[s], ra, offa + row * stepa);
            load_cols<NB1>(t.b[s], rb, offb + row * stepb);
            if constexpr (NB2 > 0) load_cols<NB2>(t.c[s], rb, offc + row * stepb);
        }
    };
    auto mfma = [&](const Batch32<NB2>& t) {
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int a = 0; a < NA; ++a)
#pragma unroll
                for (int b = 0; b < NB; ++b)
                    if (!DIAG || a <= b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(t.a[s][a], b < NB1 ? t.b[s][b] : t.c[s][b - NB1],
                                                                         acc[a][b], 0, 0, 0);
    };
    auto fold = [&]() {
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
            for (int b = 0; b < NB; ++b)
                if (!DIAG || a <= b) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const float e = acc[a][b][g];
                        sum[a][b][g] += (double)e;
                    }
                    acc[a][b] = f4{0.f, 0.f, 0.f, 0.f};
                }
    };
    // three batches in the ring: two are in flight while one feeds the MFMAs (a batch is ~40 MFMAs ~ 0.5 us, HBM
    // latency under load is several times that); batches past the split read zeros
    Batch32<NB2> t0, t1, t2;
    load(t0, 0);
    load(t1, 1);
    // The scheduling barriers keep each batch's loads together in program order, so that every wait is "all but
    // the two youngest batches" on the loop-entry path and on the back edge alike (left alone, the compiler slid
    // the reloads in between the MFMAs and the entry/back-edge merge then produced vmcnt(8): the ring drained
    // once per turn, 75 TF/s).
    __builtin_amdgcn_sched_barrier(0);
    // (knock-out builds of round 3 - no folds / no operand loads / no MFMAs, profiles/r03_gram_knockout.txt, made from the
    // sources of commit 81379de - at 240 x 80: whole 0.225-0.234 ms, MFMAs alone 0.168, loads alone 0.208: the operand loads
    // (every wave reads its own copy of B: 896 floats per mesh row against 320) are the bound, the folds are hidden; keeping
    // the four waves of a workgroup in step with a barrier per ring turn changed nothing)
    for (int t = 0; t < nbatch; t += 3) {
        load(t2, t + 2);
        __builtin_amdgcn_sched_barrier(0);
        mfma(t0);
        __builtin_amdgcn_sched_barrier(0);
        load(t0, t + 3);
        __builtin_amdgcn_sched_barrier(0);
        mfma(t1);
        __builtin_amdgcn_sched_barrier(0);
        load(t1, t + 4);
        __builtin_amdgcn_sched_barrier(0);
        mfma(t2);
        __builtin_amdgcn_sched_barrier(0);
        fold();  // every 48 rows
        __builtin_amdgcn_sched_barrier(0);
    }
    // fp32 C/D map: lane holds rows 4*(lane>>4) + g, column lane&15 of each MFMA tile
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b)
            if (!DIAG || a <= b) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int row = i0 + NA * (4 * lr + g) + a;
                    const int col = b < NB1 ? j0 + NB1 * lc + b : j0 + 16 * NB1 + NB2 * lc + (b - NB1);
                    if (row < p && col < q) w[(int64_t)row * q + col] = sum[a][b][g];
                }
            }
}

// Wave tiles are (16 tiw) x (16 tjw) outputs: 48 x 48 for a symmetric product (block-upper triangle only), 32 x 80
// otherwise - the eigensolver's products are (<= 240) x 80, and 48-wide tiles cut 80 columns into 48 + 32: waves of
// 9 and 6 MFMA tiles per step on SIMDs that cannot trade work (measured 70 TF/s against 157 peak).
__global__ void __launch_bounds__(256)
    gram32_partial_kernel(const float* __restrict__ A, int64_t lda, int p, const float* __restrict__ B, int64_t ldb,
                          int q, int64_t n, int64_t rows_per_split, int tiw, int tjw, int ntj, int ntiles, int groups,
                          int symmetric, double* __restrict__ ws) {
    const int lane = threadIdx.x & 63;
    // 1-D grid, split-major: the groups of one row split are consecutive logical workgroups of ONE XCD, so the rows
    // they share are fetched into one L2 (a 2-D grid dealt them to different XCDs: B was read once per group)
    const unsigned logical = ds::xcd_remap(blockIdx.x, gridDim.x);
    const int split = (int)(logical / (unsigned)groups), group = (int)(logical - (unsigned)split * groups);
    // the wave tiles of a group differ in cost (ragged edges, diagonal tiles): rotate which SIMD gets which
    const int wave = __builtin_amdgcn_readfirstlane((int)((threadIdx.x >> 6) + logical) & 3);
    const int wt = group * 4 + wave;
    if (wt >= ntiles) return;  // wave-uniform; the kernel has no workgroup barrier
    int ti, tj;
    if (symmetric) {
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
    const int i0 = ti * 16 * tiw, j0 = tj * 16 * tjw;
    const int na = min(tiw, (p - i0 + 15) >> 4), nbt = min(tjw, (q - j0 + 15) >> 4);
    const int nb1 = min(3, nbt), nb2 = nbt - nb1;
    const bool diag = symmetric && ti == tj;
    const int64_t r_begin = (int64_t)split * rows_per_split;
    const int64_t rows = min(n, r_begin + rows_per_split) - r_begin;
    const int lc = lane & 15, lr = lane >> 4;
    const auto ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A + r_begin * lda), 0,
                                                      (int)(((rows - 1) * lda + p) * 4), 0x00020000);
    const auto rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(B + r_begin * ldb), 0,
                                                      (int)(((rows - 1) * ldb + q) * 4), 0x00020000);
    const unsigned stepa = (unsigned)lda * 4u, stepb = (unsigned)ldb * 4u;
    const unsigned offa = (unsigned)lr * stepa + (unsigned)(i0 + na * lc) * 4u;
    const unsigned offb = (unsigned)lr * stepb + (unsigned)(j0 + nb1 * lc) * 4u;
    const unsigned offc = (unsigned)lr * stepb + (unsigned)(j0 + 16 * nb1 + nb2 * lc) * 4u;
    const int nbatch = (int)((rows + RB - 1) / RB);
    double* w = ws + (int64_t)split * p * q;
#define DS_G32(NA_, NB1_, NB2_, DG_) \
    gram32_tile<NA_, NB1_, NB2_, DG_>(ra, rb, offa, offb, offc, stepa, stepb, nbatch, lr, lc, i0, j0, p, q, w)
    if (diag) {
        if (na == 3) DS_G32(3, 3, 0, true);
        else if (na == 2) DS_G32(2, 2, 0, true);
        else DS_G32(1, 1, 0, true);
    } else {
        switch (na * 16 + nb1 * 4 + nb2) {
            case 3 * 16 + 3 * 4: DS_G32(3, 3, 0, false); break;
            case 3 * 16 + 2 * 4: DS_G32(3, 2, 0, false); break;
            case 3 * 16 + 1 * 4: DS_G32(3, 1, 0, false); break;
            case 2 * 16 + 3 * 4 + 2: DS_G32(2, 3, 2, false); break;
            case 2 * 16 + 3 * 4 + 1: DS_G32(2, 3, 1, false); break;
            case 2 * 16 + 3 * 4: DS_G32(2, 3, 0, false); break;
            case 2 * 16 + 2 * 4: DS_G32(2, 2, 0, false); break;
            case 2 * 16 + 1 * 4: DS_G32(2, 1, 0, false); break;
            case 1 * 16 + 3 * 4 + 2: DS_G32(1, 3, 2, false); break;
            case 1 * 16 + 3 * 4 + 1: DS_G32(1, 3, 1, false); break;
            case 1 * 16 + 3 * 4: DS_G32(1, 3, 0, false); break;
            case 1 * 16 + 2 * 4: DS_G32(1, 2, 0, false); break;
            default: DS_G32(1, 1, 0, false); break;  // (1, 1, 0); (3, 3, 1..2) cannot occur: tjw = 5 comes with tiw = 2
        }
    }
#undef DS_G32
}

// G[i][j] = sum_s ws[s][i][j]: 16 lanes share the split loop of one element (strided, so every lane has many
// independent loads in flight - the first version walked the splits with 4 lanes per element and took 94 us
// of pure load latency) and merge by a fixed-order shuffle tree: deterministic, no atomics.  With
// symmetric != 0 an element whose MFMA tile lies below the diagonal reads its mirror.
__global__ void __launch_bounds__(256)
    gram_reduce_kernel(const double* __restrict__ ws, int nsplit, int p, int q, int symmetric, int interleaved,
                       double* __restrict__ G) {
    const int e = threadIdx.x & 15, sub = threadIdx.x >> 4;  // 16 elements x 16 split lanes per workgroup
    const int64_t pq = (int64_t)p * q;
    const int64_t idx = (int64_t)blockIdx.x * 16 + e;
    double s0 = 0.0, s1 = 0.0;
    if (idx < pq) {
        int i = (int)(idx / q), j = (int)(idx - (int64_t)i * q);
        bool mirror = false;  // MFMA tiles below the diagonal are not computed
        if (symmetric && !interleaved) mirror = (j >> 4) < (i >> 4);
        if (symmetric && interleaved) {  // gram32: tile c of a wave tile = its columns c mod (tiles in the wave tile)
            const int ti = i / WT, tj = j / WT;
            const int nt = min(TI, (p - ti * WT + 15) >> 4);
            mirror = tj < ti || (tj == ti && (i - ti * WT) % nt > (j - ti * WT) % nt);
        }
        if (mirror) {
            const int t = i;
            i = j;
            j = t;
        }
        const double* src = ws + (int64_t)i * q + j;
        int k = sub;
        for (; k + 112 < nsplit; k += 128) {  // eight loads in flight (two per trip met a full wait each: 8 trips of latency)
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = src[(int64_t)(k + 16 * u) * pq];
#pragma unroll
            for (int u = 0; u < 8; u += 2) s0 += v[u], s1 += v[u + 1];
        }
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

// ---------------------------------------------------------------------------------------------------------------
// Both operands as LISTS of fp64 blocks (ds_gram64_blocks): every lane's three columns of A and of B are resolved to
// (pointer, row stride) once - a wave tile may straddle blocks.  The blocks tile [0, P) and [0, Q) (the entry point
// checks); a lane past P or Q reads the last column.
struct GramBlocksArgs {
    ds_block64_t a[DS_MIX64_MAX_BLOCKS], b[DS_MIX64_MAX_BLOCKS];
    int na, nb;
};

__device__ __forceinline__ void resolve_column(const ds_block64_t* blk, int nblk, int col, const double*& ptr, int64_t& ld) {
    ptr = blk[nblk - 1].a + (blk[nblk - 1].p - 1);  // past the last block: its last column
    ld = blk[nblk - 1].lda;
    for (int k = 0; k < nblk; ++k) {
        const int c = col - blk[k].offset;
        if (c >= 0 && c < blk[k].p) {
            ptr = blk[k].a + c;
            ld = blk[k].lda;
        }
    }
}

__global__ void __launch_bounds__(256)
    gram_blocks_kernel(const GramBlocksArgs args, int p, int q, int64_t n, int64_t rows_per_split, int ntj, int ntiles,
                       int groups, int symmetric, double* __restrict__ ws) {
    const WaveTile t = wave_tile(groups, ntj, ntiles, symmetric);
    if (!t.any) return;  // wave-uniform; the kernel has no workgroup barrier
    const int lane = threadIdx.x & 63;
    const int i0 = t.ti * WT, j0 = t.tj * WT;
    const int na = min(TI, (p - i0 + 15) >> 4), nb = min(TJ, (q - j0 + 15) >> 4);
    const bool diag = symmetric && t.ti == t.tj;
    const int64_t r_begin = (int64_t)t.split * rows_per_split;
    const int64_t nrows = min(n, r_begin + rows_per_split) - r_begin;
    const int lc = lane & 15, lr = lane >> 4;
    const double* pa[TI];
    const double* pb[TJ];
    int64_t sa[TI], sb[TJ];
#pragma unroll
    for (int a = 0; a < TI; ++a) {
        resolve_column(args.a, args.na, i0 + a * 16 + lc, pa[a], sa[a]);
        pa[a] += (r_begin + lr) * sa[a];
    }
#pragma unroll
    for (int b = 0; b < TJ; ++b) {
        resolve_column(args.b, args.nb, j0 + b * 16 + lc, pb[b], sb[b]);
        pb[b] += (r_begin + lr) * sb[b];
    }
    double* w = ws + (int64_t)t.split * p * q;
    DS_GRAM_DISPATCH(double, double, na, nb, diag, pa, pb, sa, sb, nrows, lr, lc, i0, j0, p, q, w);
}

struct Plan {
    int tiw, tjw;  // MFMA tiles per wave tile (fast kernel: 3 x 3 symmetric, 2 x 5 otherwise; fp64 kernel 3 x 3)
    int ntj, ntiles, groups, nsplit;
    int64_t rows_per_split;
};

// workgroups resident on the chip at once for a kernel (its register budget decides)
template <typename K>
int resident_workgroups(K kernel) {
    int per_cu = 0, dev = 0, cus = 256;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, 0) != hipSuccess || per_cu <= 0) per_cu = 2;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    return per_cu * cus;
}

Plan make_plan(int64_t n, int p, int q, int symmetric, bool fast, int64_t resident = 0) {
    Plan pl;
    pl.tiw = (fast && !symmetric) ? 2 : TI;
    pl.tjw = (fast && !symmetric) ? 5 : TJ;
    const int nti = (int)ds::ceil_div(p, 16 * pl.tiw);
    pl.ntj = (int)ds::ceil_div(q, 16 * pl.tjw);
    pl.ntiles = symmetric ? nti * (nti + 1) / 2 : nti * pl.ntj;
    pl.groups = (int)ds::ceil_div(pl.ntiles, 4);
    // The grid is sized to fill the chip an integer number of times: the fp64 kernel holds 152 VGPRs = 3 workgroups
    // per CU, 768 on the chip, and a 1026-workgroup grid ran one full round and a second one at a third of the
    // occupancy, i.e. in the time of two.
    static const int64_t res32 = resident_workgroups(gram32_partial_kernel);
    const int64_t target = resident > 0 ? resident : (fast ? res32 : 512);
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
    // (the row splits follow from how many workgroups of the kernel taken are resident: the SAME occupancy queries the launch
    // paths use, the largest over the kernels - a register-allocation change that lets a fifth workgroup share a CU must not
    // turn into "workspace too small")
    static const int64_t resident = std::max(
        std::max(resident_workgroups(gram_partial_kernel<float, float>), resident_workgroups(gram_partial_kernel<float, double>)),
        std::max(resident_workgroups(gram_partial_kernel<double, double>), resident_workgroups(gram_blocks_kernel)));
    int ns = 1;
    for (int sym = 0; sym <= (p == q ? 1 : 0); ++sym) {
        ns = std::max(ns, make_plan(n, p, q, sym, true).nsplit);
        ns = std::max(ns, make_plan(n, p, q, sym, false, resident).nsplit);
    }
    return (int64_t)ns * p * q * (int64_t)sizeof(double);
}

extern "C" int ds_gram(const void* A, int a_dtype, int64_t lda, int p, const void* B, int b_dtype, int64_t ldb, int q,
                       int64_t n, int flags, double* G, void* work, int64_t work_bytes, ds_stream_t stream) {
    DS_REQUIRE(A && B && G && work, "ds_gram: null pointer");
    DS_REQUIRE(n > 0 && p > 0 && q > 0, "ds_gram: empty problem");
    DS_REQUIRE(lda >= p && ldb >= q, "ds_gram: leading dimension smaller than the block width");
    DS_REQUIRE((a_dtype == DS_F32 || a_dtype == DS_F64) && (b_dtype == DS_F32 || b_dtype == DS_F64),
               "ds_gram: bad dtype codes %d, %d", a_dtype, b_dtype);
    DS_REQUIRE(a_dtype == DS_F32 || b_dtype == DS_F64, "ds_gram: an f64 A needs an f64 B");
    DS_REQUIRE((flags & ~(DS_GRAM_SYMMETRIC | DS_GRAM_EXACT)) == 0, "ds_gram: unknown flag bits %d", flags);
    const int symmetric = (flags & DS_GRAM_SYMMETRIC) ? 1 : 0;
    DS_REQUIRE(!symmetric || p == q, "ds_gram: symmetric needs p == q");
    const bool fast = a_dtype == DS_F32 && b_dtype == DS_F32 && !(flags & DS_GRAM_EXACT);
    static const int64_t res_ff = resident_workgroups(gram_partial_kernel<float, float>),
                         res_fd = resident_workgroups(gram_partial_kernel<float, double>),
                         res_dd = resident_workgroups(gram_partial_kernel<double, double>);
    const Plan pl = make_plan(n, p, q, symmetric, fast, fast ? 0 : (b_dtype == DS_F32 ? res_ff : (a_dtype == DS_F32 ? res_fd : res_dd)));
    DS_REQUIRE(work_bytes >= (int64_t)pl.nsplit * p * q * (int64_t)sizeof(double),
               "ds_gram: workspace too small (%lld bytes given)", (long long)work_bytes);
    // the fast kernel addresses a row split with 32-bit byte offsets (prefetched batches overshoot it by up to 5 batches)
    DS_REQUIRE(!fast || (pl.rows_per_split + 6 * RB) * std::max(lda, ldb) * 4 < (int64_t)1 << 31,
               "ds_gram: row split of %lld rows too large for 32-bit offsets", (long long)pl.rows_per_split);
    hipStream_t st = ds::as_stream(stream);
    const unsigned grid = (unsigned)(pl.groups * pl.nsplit);
    double* ws = static_cast<double*>(work);
    const float* Af = static_cast<const float*>(A);
    ds::ProfScope prof(stream, DS_PROF_GRAM, p, n, q, symmetric | (fast ? 2 : 0));  // (partial products + reduction)
    if (fast)
        gram32_partial_kernel<<<grid, 256, 0, st>>>(Af, lda, p, static_cast<const float*>(B), ldb, q, n, pl.rows_per_split, pl.tiw,
                                                    pl.tjw, pl.ntj, pl.ntiles, pl.groups, symmetric, ws);
    else if (b_dtype == DS_F32)
        gram_partial_kernel<float, float><<<grid, 256, 0, st>>>(Af, lda, p, static_cast<const float*>(B), ldb, q, n,
                                                                pl.rows_per_split, pl.ntj, pl.ntiles, pl.groups, symmetric, ws);
    else if (a_dtype == DS_F32)
        gram_partial_kernel<float, double><<<grid, 256, 0, st>>>(Af, lda, p, static_cast<const double*>(B), ldb, q, n,
                                                                 pl.rows_per_split, pl.ntj, pl.ntiles, pl.groups, symmetric, ws);
    else
        gram_partial_kernel<double, double><<<grid, 256, 0, st>>>(static_cast<const double*>(A), lda, p,
                                                                  static_cast<const double*>(B), ldb, q, n, pl.rows_per_split,
                                                                  pl.ntj, pl.ntiles, pl.groups, symmetric, ws);
    DS_LAUNCH_CHECK("gram_partial_kernel");
    const int64_t pq = (int64_t)p * q;
    gram_reduce_kernel<<<(unsigned)ds::ceil_div(pq, 16), 256, 0, st>>>(ws, pl.nsplit, p, q, symmetric, fast ? 1 : 0, G);
    DS_LAUNCH_CHECK("gram_reduce_kernel");
    return DS_OK;
}

extern "C" int ds_gram64_blocks(int na, const ds_block64_t* A, int nb, const ds_block64_t* B, int64_t n, int flags, double* G,
                                void* work, int64_t work_bytes, ds_stream_t stream) {
    DS_REQUIRE(A && B && G && work, "ds_gram64_blocks: null pointer");
    DS_REQUIRE(na >= 1 && na <= DS_MIX64_MAX_BLOCKS && nb >= 1 && nb <= DS_MIX64_MAX_BLOCKS,
               "ds_gram64_blocks: %d x %d blocks (1..%d per side)", na, nb, DS_MIX64_MAX_BLOCKS);
    DS_REQUIRE(n > 0, "ds_gram64_blocks: empty problem");
    DS_REQUIRE((flags & ~DS_GRAM_SYMMETRIC) == 0, "ds_gram64_blocks: unknown flag bits %d", flags);
    GramBlocksArgs args;
    args.na = na, args.nb = nb;
    int p = 0, q = 0;
    for (int k = 0; k < na; ++k) {
        DS_REQUIRE(A[k].a && A[k].p > 0 && A[k].lda >= A[k].p, "ds_gram64_blocks: block %d of A", k);
        DS_REQUIRE(A[k].offset == p, "ds_gram64_blocks: the blocks of A tile the rows of G in order (block %d at %d, expected %d)",
                   k, A[k].offset, p);
        args.a[k] = A[k];
        p += A[k].p;
    }
    for (int k = 0; k < nb; ++k) {
        DS_REQUIRE(B[k].a && B[k].p > 0 && B[k].lda >= B[k].p, "ds_gram64_blocks: block %d of B", k);
        DS_REQUIRE(B[k].offset == q, "ds_gram64_blocks: the blocks of B tile the columns of G in order (block %d at %d, expected %d)",
                   k, B[k].offset, q);
        args.b[k] = B[k];
        q += B[k].p;
    }
    const int symmetric = (flags & DS_GRAM_SYMMETRIC) ? 1 : 0;
    DS_REQUIRE(!symmetric || p == q, "ds_gram64_blocks: symmetric needs P == Q");
    static const int64_t resident = resident_workgroups(gram_blocks_kernel);
    const Plan pl = make_plan(n, p, q, symmetric, false, resident);
    DS_REQUIRE(work_bytes >= (int64_t)pl.nsplit * p * q * (int64_t)sizeof(double),
               "ds_gram64_blocks: workspace too small (%lld bytes given)", (long long)work_bytes);
    hipStream_t st = ds::as_stream(stream);
    double* ws = static_cast<double*>(work);
    gram_blocks_kernel<<<(unsigned)(pl.groups * pl.nsplit), 256, 0, st>>>(args, p, q, n, pl.rows_per_split, pl.ntj, pl.ntiles,
                                                                         pl.groups, symmetric, ws);
    DS_LAUNCH_CHECK("gram_blocks_kernel");
    const int64_t pq = (int64_t)p * q;
    gram_reduce_kernel<<<(unsigned)ds::ceil_div(pq, 16), 256, 0, st>>>(ws, pl.nsplit, p, q, symmetric, 0, G);
    DS_LAUNCH_CHECK("gram_reduce_kernel");
    return DS_OK;
}
