// Out <- alpha * sum_b A_b C_b + beta * Out in fp64 on v_mfma_f64_16x16x4_f64 - gfx950.
//
// The dense n x b updates of the fp64 refinement (X' = [Y X P W] Z and the new directions P' = [Y P W] Zr, and the
// same for the products with K and M; reference: the basis update X <- S Z of src/lobpcg/_lobpcg.py:457-477).  The
// basis S is never formed: its blocks (the rigid modes, X, P, W - separate arrays of different widths) are handed
// over as a short list, and C is the matching stack of coefficient rows, so that one launch reads every block once
// and writes the result once (a chain of GEMM calls re-read and re-wrote the n x b accumulator once per block).
//
// Lane (i = l & 15, kq = l >> 4) of a wave loads the four values A[row i][k0 + 4 kq + s], s = 0..3, of its row as
// two 16-byte loads; MFMA step s then reduces over the index set {k0 + 4 kq + s} and its B operand is row
// k0 + 4 (l >> 4) + s of C: 16 consecutive doubles per 16-lane group, from L1/L2 (C is at most 0.5 MB and every wave
// of the launch reads the same rows; the fp64 MFMA takes 64 cycles, so that one 512-byte operand read per two MFMAs
// is a quarter of what the CU's vector memory pipe delivers - no LDS image, no limit on the depth of C).
// A wave owns 32 rows x (16 JT) columns: 2 x JT accumulators of 8 registers.
#include <algorithm>

#include "ds_common.h"

namespace {

using d4 = __attribute__((ext_vector_type(4))) double;
using d2 = __attribute__((ext_vector_type(2))) double;

constexpr int RT64 = 2;  // 16-row tiles per wave

struct Mix64Args {
    ds_block64_t blk[DS_MIX64_MAX_BLOCKS];
    int nblk;
};

// nothing moves across: neither in the IR (memory clobber) nor in the machine scheduler
#define DS_PIN64()                          \
    do {                                    \
        asm volatile("" ::: "memory");        \
        __builtin_amdgcn_sched_barrier(0);  \
    } while (0)

template <int JT>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
    mix64_kernel(const Mix64Args args, const double* __restrict__ C, int64_t ldc, int q, double* __restrict__ Out,
                 int64_t ldo, int64_t n, double alpha, double beta) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 15, lq = lane >> 4;
    const int64_t row0 = ((int64_t)ds::xcd_remap(blockIdx.x, gridDim.x) * 4 + wave) * (RT64 * 16);
    if (row0 >= n) return;  // wave-uniform
    d4 acc[RT64][JT];
#pragma unroll
    for (int t = 0; t < RT64; ++t)
#pragma unroll
        for (int j = 0; j < JT; ++j) acc[t][j] = d4{0.0, 0.0, 0.0, 0.0};
    // No predicated loads (each would be a branch around the load): a column past q reads column q - 1 and a row past
    // n reads row n - 1 - results that are never stored -, a row of C past the block's p reads row p - 1 under a zero
    // of A.
    // (only the last tile can be partial: the others' column offsets are immediates of the loads)
    const int clast = min((JT - 1) * 16 + li, q - 1);
    int64_t ar[RT64];
#pragma unroll
    for (int t = 0; t < RT64; ++t) ar[t] = min(row0 + t * 16 + li, n - 1);

    for (int b = 0; b < args.nblk; ++b) {
        const double* __restrict__ A = args.blk[b].a;
        const int64_t lda = args.blk[b].lda;
        const int p = args.blk[b].p;
        const double* __restrict__ Cb = C + (int64_t)args.blk[b].offset * ldc;
        const bool vec = ((reinterpret_cast<uintptr_t>(A) | (uintptr_t)(lda * 8)) & 15) == 0 && (p & 3) == 0;  // wave-uniform
        const int nstep = (p + 15) >> 4;
        // the four values A[row][16 step + 4 lq + s] of the lane's rows
        auto load_a = [&](double (&a)[RT64][4], int step) {
            const int kk = 16 * step + 4 * lq;
            if (vec) {
                const int kc = min(kk, p - 4);
#pragma unroll
                for (int t = 0; t < RT64; ++t) {
                    const d2* ap = reinterpret_cast<const d2*>(A + ar[t] * lda + kc);
                    const d2 lo = ap[0], hi = ap[1];
                    a[t][0] = lo[0], a[t][1] = lo[1], a[t][2] = hi[0], a[t][3] = hi[1];
                }
            } else {
#pragma unroll
                for (int t = 0; t < RT64; ++t)
#pragma unroll
                    for (int s = 0; s < 4; ++s) a[t][s] = A[ar[t] * lda + min(kk + s, p - 1)];
            }
        };
        // (at the use, not at the load: a select behind the load would wait for it)
        auto mask_a = [&](double (&a)[RT64][4], int step) {
            const int kk = 16 * step + 4 * lq;
#pragma unroll
            for (int t = 0; t < RT64; ++t)
#pragma unroll
                for (int s = 0; s < 4; ++s) a[t][s] = kk + s < p ? a[t][s] : 0.0;
        };
        auto load_b = [&](double (&bv)[JT], int step, int s) {
            const double* cp = Cb + (int64_t)min(16 * step + 4 * lq + s, p - 1) * ldc;
#pragma unroll
            for (int j = 0; j < JT - 1; ++j) bv[j] = cp[j * 16 + li];
            bv[JT - 1] = cp[clast];
        };
        auto mfma = [&](const double (&a)[RT64][4], const double (&bv)[JT], int s) {
#pragma unroll
            for (int j = 0; j < JT; ++j)
#pragma unroll
                for (int t = 0; t < RT64; ++t)
                    acc[t][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[t][s], bv[j], acc[t][j], 0, 0, 0);
        };
        // Software pipeline: the B operands of sub-step s + 1 and the A values of the next step are requested before
        // the 2 JT MFMAs of sub-step s issue (one wave per SIMD at JT = 9: nothing else hides the latency).  A fetch
        // past the last step re-reads it and is not used.
        double a0[RT64][4], a1[RT64][4], b0[JT], b1[JT];
        load_a(a0, 0);
        load_b(b0, 0, 0);
        DS_PIN64();
        for (int step = 0; step < nstep; ++step) {
            const int nx = min(step + 1, nstep - 1);
            mask_a(a0, step);
            load_b(b1, step, 1);
            DS_PIN64();
            mfma(a0, b0, 0);
            DS_PIN64();
            load_b(b0, step, 2);
            load_a(a1, nx);
            DS_PIN64();
            mfma(a0, b1, 1);
            DS_PIN64();
            load_b(b1, step, 3);
            DS_PIN64();
            mfma(a0, b0, 2);
            DS_PIN64();
            load_b(b0, nx, 0);
            DS_PIN64();
            mfma(a0, b1, 3);
            DS_PIN64();
#pragma unroll
            for (int t = 0; t < RT64; ++t)
#pragma unroll
                for (int s = 0; s < 4; ++s) a0[t][s] = a1[t][s];
        }
    }
    // fp64 C/D map: lane (li, lq) holds rows lq + 4 g, column li of each tile
#pragma unroll
    for (int t = 0; t < RT64; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int64_t r = row0 + t * 16 + lq + 4 * g;
            if (r >= n) continue;
            double* o = Out + r * ldo + li;
#pragma unroll
            for (int j = 0; j < JT; ++j)
                if (j * 16 + li < q) {
                    const double v = alpha * acc[t][j][g];
                    o[j * 16] = beta == 0.0 ? v : __builtin_fma(beta, o[j * 16], v);
                }
        }
}
#undef DS_PIN64

template <int JT>
int launch_mix64(const Mix64Args& args, const double* C, int64_t ldc, int q, double* Out, int64_t ldo, int64_t n,
                 double alpha, double beta, hipStream_t st) {
    const unsigned grid = (unsigned)ds::ceil_div(n, 4 * RT64 * 16);
    mix64_kernel<JT><<<grid, 256, 0, st>>>(args, C, ldc, q, Out, ldo, n, alpha, beta);
    DS_LAUNCH_CHECK("mix64_kernel");
    return DS_OK;
}

}  // namespace

extern "C" int ds_mix64(int nblocks, const ds_block64_t* blocks, const double* C, int64_t ldc, int q, double* Out,
                        int64_t ldo, int64_t n, double alpha, double beta, ds_stream_t stream) {
    DS_REQUIRE(blocks && C && Out, "ds_mix64: null pointer");
    DS_REQUIRE(nblocks >= 1 && nblocks <= DS_MIX64_MAX_BLOCKS, "ds_mix64: %d blocks (1..%d)", nblocks, DS_MIX64_MAX_BLOCKS);
    DS_REQUIRE(n > 0 && q > 0, "ds_mix64: empty problem");
    DS_REQUIRE(ldc >= q && ldo >= q, "ds_mix64: leading dimension smaller than the result's width");
    Mix64Args args;
    args.nblk = nblocks;
    const char* o0 = reinterpret_cast<const char*>(Out);
    const char* o1 = o0 + ((n - 1) * ldo + q) * 8;
    for (int b = 0; b < nblocks; ++b) {
        const ds_block64_t& k = blocks[b];
        DS_REQUIRE(k.a && k.p > 0 && k.lda >= k.p && k.offset >= 0, "ds_mix64: block %d: bad pointer, width or offset", b);
        // (other waves read the rows a wave writes only if Out and a block share memory)
        const char* a0 = reinterpret_cast<const char*>(k.a);
        const char* a1 = a0 + ((n - 1) * k.lda + k.p) * 8;
        DS_REQUIRE(o1 <= a0 || a1 <= o0, "ds_mix64: Out overlaps block %d", b);
        args.blk[b] = k;
    }
    {   // every wave reads C while other waves write Out: the two must not share memory (rows of C used: up to the
        // last block's offset + width)
        int crow = 0;
        for (int b = 0; b < nblocks; ++b) crow = std::max(crow, blocks[b].offset + blocks[b].p);
        const char* c0 = reinterpret_cast<const char*>(C);
        const char* c1 = c0 + ((int64_t)(crow - 1) * ldc + q) * 8;
        DS_REQUIRE(o1 <= c0 || c1 <= o0, "ds_mix64: Out overlaps the coefficient matrix C");
    }
    hipStream_t st = ds::as_stream(stream);
    // column chunks of at most 9 MFMA tiles (144 columns): the accumulators of a chunk stay in registers
    constexpr int CH = 16 * 9;
    int rc = DS_OK;
    for (int jb = 0; jb < q && rc == DS_OK; jb += CH) {
        const int qc = std::min(q - jb, CH);
        const double* Cc = C + jb;
        double* Oc = Out + jb;
        switch ((qc + 15) / 16) {
            case 1: rc = launch_mix64<1>(args, Cc, ldc, qc, Oc, ldo, n, alpha, beta, st); break;
            case 2: rc = launch_mix64<2>(args, Cc, ldc, qc, Oc, ldo, n, alpha, beta, st); break;
            case 3: rc = launch_mix64<3>(args, Cc, ldc, qc, Oc, ldo, n, alpha, beta, st); break;
            case 4: rc = launch_mix64<4>(args, Cc, ldc, qc, Oc, ldo, n, alpha, beta, st); break;
            case 5: rc = launch_mix64<5>(args, Cc, ldc, qc, Oc, ldo, n, alpha, beta, st); break;
            case 6: rc = launch_mix64<6>(args, Cc, ldc, qc, Oc, ldo, n, alpha, beta, st); break;
            case 7: rc = launch_mix64<7>(args, Cc, ldc, qc, Oc, ldo, n, alpha, beta, st); break;
            case 8: rc = launch_mix64<8>(args, Cc, ldc, qc, Oc, ldo, n, alpha, beta, st); break;
            default: rc = launch_mix64<9>(args, Cc, ldc, qc, Oc, ldo, n, alpha, beta, st); break;
        }
    }
    return rc;
}
