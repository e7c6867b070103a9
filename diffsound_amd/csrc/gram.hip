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

template <typename TA, typename TB>
struct Batch {
    TA a[KS][TI];
    TB b[KS][TJ];
};

// which of the wave tile's 3x3 MFMA tiles exist (wave-uniform): a < na, b < nb, and on a diagonal wave tile of
// a symmetric product only a <= b (the reduction mirrors the rest)
struct Active {
    int na, nb, diag;
    __device__ __forceinline__ bool operator()(int a, int b) const { return a < na && b < nb && (!diag || a <= b); }
};

template <typename TA, typename TB>
__device__ __forceinline__ void load_batch(Batch<TA, TB>& t, const TA* __restrict__ A, int64_t lda,
                                           const TB* __restrict__ B, int64_t ldb, int64_t r0, int64_t r_end, int lr,
                                           int acol, int bcol, const bool (&ia)[TI], const bool (&jb)[TJ],
                                           const Active act) {
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int64_t r = r0 + 4 * s + lr;
        const bool rv = r < r_end;
        const TA* ap = A + r * lda + acol;
        const TB* bp = B + r * ldb + bcol;
#pragma unroll
        for (int a = 0; a < TI; ++a)
            if (a < act.na) t.a[s][a] = (rv && ia[a]) ? ap[a * 16] : (TA)0;
#pragma unroll
        for (int b = 0; b < TJ; ++b)
            if (b < act.nb) t.b[s][b] = (rv && jb[b]) ? bp[b * 16] : (TB)0;
    }
}

template <typename TA, typename TB>
__device__ __forceinline__ void mfma_batch(const Batch<TA, TB>& t, d4 (&acc)[TI][TJ], const Active act) {
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
template <typename TA, typename TB>
__global__ void __launch_bounds__(256)
    gram_partial_kernel(const TA* __restrict__ A, int64_t lda, int p, const TB* __restrict__ B, int64_t ldb, int q,
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

    Batch<TA, TB> cur, nxt;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
#pragma unroll
        for (int a = 0; a < TI; ++a) cur.a[s][a] = nxt.a[s][a] = (TA)0;
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

Plan make_plan(int64_t n, int p, int q, int symmetric, bool fast) {
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
    static const int64_t res64 = resident_workgroups(gram_partial_kernel<float, float>);
    static const int64_t res32 = resident_workgroups(gram32_partial_kernel);
    const int64_t target = fast ? res32 : res64;
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
    int ns = 1;
    for (int sym = 0; sym <= (p == q ? 1 : 0); ++sym)
        for (int fast = 0; fast <= 1; ++fast) ns = std::max(ns, make_plan(n, p, q, sym, fast != 0).nsplit);
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
    const Plan pl = make_plan(n, p, q, symmetric, fast);
    DS_REQUIRE(work_bytes >= (int64_t)pl.nsplit * p * q * (int64_t)sizeof(double),
               "ds_gram: workspace too small (%lld bytes given)", (long long)work_bytes);
    // the fast kernel addresses a row split with 32-bit byte offsets (prefetched batches overshoot it by up to 5 batches)
    DS_REQUIRE(!fast || (pl.rows_per_split + 6 * RB) * std::max(lda, ldb) * 4 < (int64_t)1 << 31,
               "ds_gram: row split of %lld rows too large for 32-bit offsets", (long long)pl.rows_per_split);
    hipStream_t st = ds::as_stream(stream);
    dim3 grid((unsigned)pl.groups, (unsigned)pl.nsplit);
    double* ws = static_cast<double*>(work);
    const float* Af = static_cast<const float*>(A);
    if (fast)
        gram32_partial_kernel<<<(unsigned)(pl.groups * pl.nsplit), 256, 0, st>>>(
            Af, lda, p, static_cast<const float*>(B), ldb, q, n, pl.rows_per_split, pl.tiw, pl.tjw, pl.ntj, pl.ntiles,
            pl.groups,
            symmetric, ws);
    else if (b_dtype == DS_F32)
        gram_partial_kernel<float, float><<<grid, 256, 0, st>>>(Af, lda, p, static_cast<const float*>(B), ldb, q, n,
                                                                pl.rows_per_split, pl.ntj, pl.ntiles, symmetric, ws);
    else if (a_dtype == DS_F32)
        gram_partial_kernel<float, double><<<grid, 256, 0, st>>>(Af, lda, p, static_cast<const double*>(B), ldb, q, n,
                                                                 pl.rows_per_split, pl.ntj, pl.ntiles, symmetric, ws);
    else
        gram_partial_kernel<double, double><<<grid, 256, 0, st>>>(static_cast<const double*>(A), lda, p,
                                                                  static_cast<const double*>(B), ldb, q, n,
                                                                  pl.rows_per_split, pl.ntj, pl.ntiles, symmetric, ws);
    DS_LAUNCH_CHECK("gram_partial_kernel");
    const int64_t pq = (int64_t)p * q;
    gram_reduce_kernel<<<(unsigned)ds::ceil_div(pq, 16), 256, 0, st>>>(ws, pl.nsplit, p, q, symmetric, fast ? 1 : 0, G);
    DS_LAUNCH_CHECK("gram_reduce_kernel");
    return DS_OK;
}
