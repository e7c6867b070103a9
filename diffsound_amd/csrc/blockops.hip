// Fused block-vector updates of the eigensolver - gfx950.
//
//  ds_residual   R <- K X - (M X) diag(lam) and the column norms the convergence test needs
//                (reference update_residual / update_converged_count, src/lobpcg/_lobpcg.py:301-333:
//                 three torch ops + two torch.norm + a Python loop over a device tensor)
//  ds_cheb_init / ds_cheb_step   one fused pass per term of the Chebyshev block-Jacobi polynomial
//                preconditioner (the reference has no preconditioner: iK=None, _linalg_utils.py:32-33)
//  ds_mix        Out <- alpha A C + beta Out, the "tall-skinny times small" update GEMM
//                (reference X <- S Z, P <- S Z_p, U <- U - V (V^T B U): _lobpcg.py:463-466, 629)
//
// All blocks are row-major (n x ncols) f32 with explicit leading dimensions; one lane owns 4
// consecutive columns of the 3 rows of one node (16-byte accesses), which is also the natural unit
// of the 3x3 block-Jacobi solve.
#include <algorithm>

#include <cstdlib>

#include "ds_common.h"

namespace {

using f4 = __attribute__((ext_vector_type(4))) float;

__device__ __forceinline__ f4 ld4(const float* p) { return *reinterpret_cast<const f4*>(p); }
__device__ __forceinline__ void st4(float* p, f4 v) { *reinterpret_cast<f4*>(p) = v; }

// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
    residual_kernel(const float* KX, int64_t ldk, float* R, int64_t ldr, const float* __restrict__ MX, int64_t ldm,
                    const float* __restrict__ X, int64_t ldx, const double* __restrict__ lam, int64_t n, int ncols,
                    int cgroups, double* __restrict__ rn2, double* __restrict__ xn2) {
    extern __shared__ __attribute__((aligned(16))) double sm[];  // [2][ncols]
    for (int i = threadIdx.x; i < 2 * ncols; i += blockDim.x) sm[i] = 0.0;
    __syncthreads();
    const int cg = threadIdx.x % cgroups;
    const int rl = threadIdx.x / cgroups;
    const int rows_per_block = blockDim.x / cgroups;
    const int c0 = cg * 4;
    double pr[4] = {0, 0, 0, 0}, px[4] = {0, 0, 0, 0};
    if (rl < rows_per_block) {
        float l4[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) l4[v] = (float)lam[c0 + v];
        for (int64_t r = (int64_t)blockIdx.x * rows_per_block + rl; r < n; r += (int64_t)gridDim.x * rows_per_block) {
            f4 rv = ld4(KX + r * ldk + c0);  // KX may be R itself (in place): a lane reads a piece before writing it
            const f4 mv = ld4(MX + r * ldm + c0);
            const f4 xv = ld4(X + r * ldx + c0);
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                rv[v] = fmaf(-mv[v], l4[v], rv[v]);
                pr[v] += (double)rv[v] * (double)rv[v];
                px[v] += (double)xv[v] * (double)xv[v];
            }
            st4(R + r * ldr + c0, rv);
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            atomicAdd(&sm[c0 + v], pr[v]);
            atomicAdd(&sm[ncols + c0 + v], px[v]);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < ncols; i += blockDim.x) {
        atomicAdd(&rn2[i], sm[i]);
        atomicAdd(&xn2[i], sm[ncols + i]);
    }
}

// ---------------------------------------------------------------------------------------------
// T r for one node: 3x3 block (row-major dinv) times the 3 rows r0,r1,r2 (4 columns each)
__device__ __forceinline__ void bj_apply(const float* __restrict__ d, const f4& r0, const f4& r1, const f4& r2, f4& t0,
                                         f4& t1, f4& t2) {
    t0 = d[0] * r0 + d[1] * r1 + d[2] * r2;
    t1 = d[3] * r0 + d[4] * r1 + d[5] * r2;
    t2 = d[6] * r0 + d[7] * r1 + d[8] * r2;
}

template <bool STEP>
__global__ void __launch_bounds__(256)
    cheb_kernel(const float* __restrict__ AD, int64_t lda, float* __restrict__ R, int64_t ldr, float* __restrict__ D,
                int64_t ldd, float* __restrict__ W, int64_t ldw, const float* __restrict__ dinv, int64_t nv,
                int cgroups, float c1, float c2) {
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t node = tid / cgroups;
    const int c0 = (int)(tid - node * cgroups) * 4;
    if (node >= nv) return;
    float d[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) d[k] = dinv[node * 9 + k];
    const int64_t r = node * 3;
    f4 r0 = ld4(R + r * ldr + c0), r1 = ld4(R + (r + 1) * ldr + c0), r2 = ld4(R + (r + 2) * ldr + c0);
    f4 t0, t1, t2;
    if (STEP) {
        r0 -= ld4(AD + r * lda + c0);
        r1 -= ld4(AD + (r + 1) * lda + c0);
        r2 -= ld4(AD + (r + 2) * lda + c0);
        st4(R + r * ldr + c0, r0);
        st4(R + (r + 1) * ldr + c0, r1);
        st4(R + (r + 2) * ldr + c0, r2);
        bj_apply(d, r0, r1, r2, t0, t1, t2);
        const f4 d0 = c1 * ld4(D + r * ldd + c0) + c2 * t0;
        const f4 d1 = c1 * ld4(D + (r + 1) * ldd + c0) + c2 * t1;
        const f4 d2 = c1 * ld4(D + (r + 2) * ldd + c0) + c2 * t2;
        st4(D + r * ldd + c0, d0);
        st4(D + (r + 1) * ldd + c0, d1);
        st4(D + (r + 2) * ldd + c0, d2);
        st4(W + r * ldw + c0, ld4(W + r * ldw + c0) + d0);
        st4(W + (r + 1) * ldw + c0, ld4(W + (r + 1) * ldw + c0) + d1);
        st4(W + (r + 2) * ldw + c0, ld4(W + (r + 2) * ldw + c0) + d2);
    } else {
        bj_apply(d, r0, r1, r2, t0, t1, t2);
        t0 *= c2;
        t1 *= c2;
        t2 *= c2;
        st4(D + r * ldd + c0, t0);
        st4(D + (r + 1) * ldd + c0, t1);
        st4(D + (r + 2) * ldd + c0, t2);
        st4(W + r * ldw + c0, t0);
        st4(W + (r + 1) * ldw + c0, t1);
        st4(W + (r + 2) * ldw + c0, t2);
    }
}

// ---------------------------------------------------------------------------------------------
// Out <- alpha A C + beta Out with v_mfma_f32_16x16x4_f32 (exact f32 FMA chain).
// A wave owns RT*16 rows x JT*16 columns of Out.  One 16-byte load gives lane (i = l&15, kq = l>>4)
// the four values A[row i][k0 + 4 kq + s], s = 0..3; MFMA step s therefore reduces over the index set
// {k0 + 4 kq + s}, and the C operand for that step is row k0 + 4 (l>>4) + s of C (L1/L2 resident).
using f4acc = __attribute__((ext_vector_type(4))) float;
constexpr int RT = 2;

template <int JT, bool VECA>
__global__ void __launch_bounds__(256)
    mix_kernel(const float* A, int64_t lda, int p, const float* __restrict__ C, int q, float* Out, int64_t ldo, int64_t n, int j_base, float alpha, float beta) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int li = lane & 15, lq = lane >> 4;
    const int64_t row0 = ((int64_t)blockIdx.x * 4 + wave) * (RT * 16);
    if (row0 >= n) return;  // wave-uniform
    f4acc acc[RT][JT];
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int j = 0; j < JT; ++j) acc[t][j] = f4acc{0.f, 0.f, 0.f, 0.f};
    bool jv[JT];
#pragma unroll
    for (int j = 0; j < JT; ++j) jv[j] = (j_base + j * 16 + li) < q;

    for (int k0 = 0; k0 < p; k0 += 16) {
        f4 a[RT];
        const int kk = k0 + 4 * lq;
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            const int64_t r = row0 + t * 16 + li;
            if (VECA) {
                a[t] = (r < n && kk < p) ? ld4(A + r * lda + kk) : f4{0.f, 0.f, 0.f, 0.f};  // p % 4 == 0 here
            } else {
#pragma unroll
                for (int s = 0; s < 4; ++s) a[t][s] = (r < n && kk + s < p) ? A[r * lda + kk + s] : 0.f;
            }
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int kr = kk + s;
            const float* cp = C + (int64_t)kr * q + j_base + li;
#pragma unroll
            for (int j = 0; j < JT; ++j) {
                const float b = (kr < p && jv[j]) ? cp[j * 16] : 0.f;
#pragma unroll
                for (int t = 0; t < RT; ++t) acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t][s], b, acc[t][j], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int j = 0; j < JT; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int64_t r = row0 + t * 16 + lq * 4 + g;
                const int col = j_base + j * 16 + li;
                if (r < n && col < q) {
                    float* o = Out + r * ldo + col;
                    const float v = alpha * acc[t][j][g];
                    *o = (beta == 0.f) ? v : fmaf(beta, *o, v);
                }
            }
}

template <int JT>
int launch_mix(const float* A, int64_t lda, int p, const float* C, int q, float* Out, int64_t ldo, int64_t n,
               int j_base, float alpha, float beta, bool veca, hipStream_t st) {
    const unsigned grid = (unsigned)ds::ceil_div(n, 4 * RT * 16);
    if (veca)
        mix_kernel<JT, true><<<grid, 256, 0, st>>>(A, lda, p, C, q, Out, ldo, n, j_base, alpha, beta);
    else
        mix_kernel<JT, false><<<grid, 256, 0, st>>>(A, lda, p, C, q, Out, ldo, n, j_base, alpha, beta);
    DS_LAUNCH_CHECK("mix_kernel");
    return DS_OK;
}

// Out <- alpha A C + beta Out for the eigensolver's shapes (p <= 256, q <= 80, 16-byte aligned rows of A): the
// coefficient matrix C (<= 80 KB) is staged ONCE per workgroup in LDS and the workgroups are persistent over row
// tiles (2 per CU), so the MFMA B operands come from LDS instead of twenty 4-byte global loads per 16-deep k-step.
//
// LDS layout [k / 4][position][k % 4]: the four k values a lane feeds to four consecutive MFMAs are one
// ds_read_b128 (five per k-step; the first version issued twenty predicated ds_read_b32, each followed by a wait
// and two MFMAs: 73 TF/s).  "position" j * 16 + li holds column JT * li + j: an MFMA does not care which column its
// index stands for, and with this interleave a lane ends up owning JT CONSECUTIVE output columns per row, so the
// result is stored (and, for beta != 0, read) as 16-lane x 20-byte = 320-byte row segments instead of 64-byte ones.
// A and the B operands of the next k-step are fetched before the current step's 40 MFMAs issue.
// The generic kernel above ran at 54 TFLOP/s on (n x 240)(240 x 80); the floor is the 0.57 GB it has to move.
// nothing moves across: neither in the IR (memory clobber) nor in the machine scheduler
#define DS_PIN_ORDER()                      \
    do {                                    \
        asm volatile("" ::: "memory");        \
        __builtin_amdgcn_sched_barrier(0);  \
    } while (0)

// (knock-out builds of round 3 - no loads of A / no LDS reads of the coefficients / no MFMAs, profiles/r03_mix_knockout.txt,
// made from the sources of commit 81379de: 240 -> 80 whole 0.207 ms, MFMAs alone 0.194, loads alone 0.160; 240 -> 160 whole
// 0.370, MFMAs alone 0.337, loads alone 0.222 - the kernel runs at the rate its MFMA loop issues alone, 88-102 TF/s; the
// instruction sustains 134-138 TF/s in tools/mfma_rate_probe.hip)
constexpr int MIX_NW = 8;  // waves per workgroup sharing one LDS copy of C (2 workgroups per CU -> 4 waves per SIMD)

using u4 = __attribute__((ext_vector_type(4))) unsigned;

template <int N>
struct alignas(4) FloatPack {
    float v[N];
};

template <int JT>
__global__ void __launch_bounds__(64 * MIX_NW)
    mix_lds_kernel(const float* A, int64_t lda, int p, const float* __restrict__ C, int ldc, int q, float* Out, int64_t ldo,
                   int64_t n, float alpha, float beta) {
    extern __shared__ __attribute__((aligned(16))) float s_c[];  // [p16 / 4][JT * 16][4], rows >= p and columns >= q zero
    constexpr int W = JT * 16;
    const int p16 = (p + 15) & ~15;
    // (eight independent loads per thread and round: one load per trip cost a workgroup ~25 k cycles before its first MFMA)
    constexpr int NTHR = 64 * MIX_NW, STG = 8;
    for (int t0 = threadIdx.x; t0 < p16 * W; t0 += NTHR * STG) {
        float v[STG];
#pragma unroll
        for (int u = 0; u < STG; ++u) {
            const int t = t0 + u * NTHR;
            const int kq = t / (W * 4), rem = t - kq * (W * 4), pos = rem >> 2;
            const int r = kq * 4 + (rem & 3), c = (pos & 15) * JT + (pos >> 4);
            v[u] = (t < p16 * W && r < p && c < q) ? C[r * ldc + c] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < STG; ++u)
            if (t0 + u * NTHR < p16 * W) s_c[t0 + u * NTHR] = v[u];
    }
    __syncthreads();
    const f4* sc4 = reinterpret_cast<const f4*>(s_c);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 15, lq = lane >> 4;
    const int64_t ntile = (n + RT * 16 - 1) / (RT * 16);
    const int nstep = p16 >> 4;
    for (int64_t tile = (int64_t)blockIdx.x * MIX_NW + wave; tile < ntile; tile += (int64_t)gridDim.x * MIX_NW) {
        const int64_t row0 = tile * (RT * 16);
        f4acc acc[RT][JT];
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
            for (int j = 0; j < JT; ++j) acc[t][j] = f4acc{0.f, 0.f, 0.f, 0.f};
        // A through a per-tile buffer descriptor that ends after column p of the tile's last row: rows past n read
        // zeros without a branch; a k-step past p (only the last one can be partial; p % 4 == 0) reads the
        // neighbouring block's columns, which may hold anything, and is replaced by zeros
        const int64_t trows = min((int64_t)(RT * 16), n - row0);
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A + row0 * lda), 0,
                                                          (int)(((trows - 1) * lda + p) * 4), 0x00020000);
        unsigned voff[RT];
#pragma unroll
        for (int t = 0; t < RT; ++t) voff[t] = (unsigned)((t * 16 + li) * lda + 4 * lq) * 4u;
        auto fetch = [&](f4 (&a)[RT], f4 (&b)[JT], int step) {
#pragma unroll
            for (int t = 0; t < RT; ++t) {
                const u4 raw = __builtin_amdgcn_raw_buffer_load_b128(rs, voff[t] + (unsigned)step * 64u, 0, 0);
                a[t] = __builtin_bit_cast(f4, raw);
            }
            const int sb = min(step, nstep - 1);  // the fetch past the last step re-reads it (never used)
#pragma unroll
            for (int j = 0; j < JT; ++j) {
                b[j] = sc4[(sb * 4 + lq) * W + j * 16 + li];
            }
        };
        // the zeroing of a partial last k-step happens here, at the use: done at the fetch it put a wait for the
        // prefetched operands in front of the current step's MFMAs
        auto mfma = [&](const f4 (&a)[RT], const f4 (&b)[JT], int step) {
            const bool kv = step * 16 + 4 * lq < p;
            f4 am[RT];
#pragma unroll
            for (int t = 0; t < RT; ++t) am[t] = kv ? a[t] : f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s_ = 0; s_ < 4; ++s_)
#pragma unroll
                for (int j = 0; j < JT; ++j)
#pragma unroll
                    for (int t = 0; t < RT; ++t)
                        acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(am[t][s_], b[j][s_], acc[t][j], 0, 0, 0);
        };
        f4 a0[RT], a1[RT], b0[JT], b1[JT];
        fetch(a0, b0, 0);
        // scheduling barriers: the operands of step + 1 are requested BEFORE the MFMAs of step issue and waited
        // for after them (left alone, the compiler gathered the requests of two steps behind the MFMAs and met
        // them with vmcnt(0) at the loop head)
        DS_PIN_ORDER();
        for (int step = 0; step + 1 < nstep; step += 2) {  // both halves unconditional: a fetch whose use sits
            fetch(a1, b1, step + 1);                       // under a condition is sunk to that use
            DS_PIN_ORDER();
            mfma(a0, b0, step);
            DS_PIN_ORDER();
            fetch(a0, b0, step + 2);
            DS_PIN_ORDER();
            mfma(a1, b1, step + 1);
            DS_PIN_ORDER();
        }
        if (nstep & 1) mfma(a0, b0, nstep - 1);
        // lane (li, lq) owns rows 4 lq + g and the JT consecutive columns JT li .. JT li + JT - 1
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int64_t r = row0 + t * 16 + lq * 4 + g;
                const int c0 = JT * li;
                if (r >= n || c0 >= q) continue;
                float* o = Out + r * ldo + c0;
                FloatPack<JT> v;
                if (c0 + JT <= q) {
                    if (beta != 0.f) v = *reinterpret_cast<const FloatPack<JT>*>(o);
#pragma unroll
                    for (int j = 0; j < JT; ++j) {
                        const float x = alpha * acc[t][j][g];
                        v.v[j] = (beta == 0.f) ? x : fmaf(beta, v.v[j], x);
                    }
                    *reinterpret_cast<FloatPack<JT>*>(o) = v;
                } else {
#pragma unroll
                    for (int j = 0; j < JT; ++j)
                        if (c0 + j < q) {
                            const float x = alpha * acc[t][j][g];
                            o[j] = (beta == 0.f) ? x : fmaf(beta, o[j], x);
                        }
                }
            }
    }
}

template <int JT>
int launch_mix_lds(const float* A, int64_t lda, int p, const float* C, int ldc, int q, float* Out, int64_t ldo, int64_t n,
                   float alpha, float beta, hipStream_t st) {
    const int p16 = (p + 15) & ~15;
    const size_t lds = (size_t)p16 * JT * 16 * sizeof(float);
    static size_t attr_bytes = 0;
    if (lds > 48 * 1024 && lds > attr_bytes) {  // opt in to large dynamic LDS (per instantiation, grows only)
        int rc = ds::check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(&mix_lds_kernel<JT>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds),
                               "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
        if (rc != DS_OK) return rc;
        attr_bytes = lds;
    }
    const int64_t ntile = ds::ceil_div(n, RT * 16);
    // persistent: 2 workgroups per CU while two coefficient images fit its LDS, else 1
    const unsigned grid = (unsigned)std::min<int64_t>(ds::ceil_div(ntile, MIX_NW), 2 * lds <= 160 * 1024 ? 512 : 256);
    mix_lds_kernel<JT><<<grid, 64 * MIX_NW, lds, st>>>(A, lda, p, C, ldc, q, Out, ldo, n, alpha, beta);
    DS_LAUNCH_CHECK("mix_lds_kernel");
    return DS_OK;
}

bool aligned16(const void* p, int64_t ld_elems) {
    return ((reinterpret_cast<uintptr_t>(p) | (uintptr_t)(ld_elems * 4)) & 15) == 0;
}

}  // namespace

extern "C" int ds_residual(const float* KX, int64_t ldk, float* R, int64_t ldr, const float* MX, int64_t ldm,
                           const float* X, int64_t ldx, const double* lam, int64_t n, int ncols, double* rn2,
                           double* xn2, ds_stream_t stream) {
    DS_REQUIRE(KX && R && MX && X && lam && rn2 && xn2, "ds_residual: null pointer");
    DS_REQUIRE(ldk >= ncols && ldr >= ncols, "ds_residual: leading dimension smaller than ncols");
    DS_REQUIRE(aligned16(KX, ldk), "ds_residual: rows must be 16-byte aligned");
    DS_REQUIRE(n > 0 && ncols > 0 && ncols % 4 == 0 && ncols <= 1024, "ds_residual: ncols must be a multiple of 4 <= 1024");
    DS_REQUIRE(aligned16(R, ldr) && aligned16(MX, ldm) && aligned16(X, ldx), "ds_residual: rows must be 16-byte aligned");
    hipStream_t st = ds::as_stream(stream);
    int rc = ds::check_hip(hipMemsetAsync(rn2, 0, sizeof(double) * ncols, st), "ds_residual memset");
    if (rc) return rc;
    rc = ds::check_hip(hipMemsetAsync(xn2, 0, sizeof(double) * ncols, st), "ds_residual memset");
    if (rc) return rc;
    const int cgroups = ncols / 4;
    const int rows_per_block = 256 / cgroups;
    const int64_t nblk = std::min<int64_t>(1024, ds::ceil_div(n, rows_per_block));
    residual_kernel<<<(unsigned)nblk, 256, 2 * ncols * sizeof(double), st>>>(KX, ldk, R, ldr, MX, ldm, X, ldx, lam, n,
                                                                             ncols, cgroups, rn2, xn2);
    DS_LAUNCH_CHECK("residual_kernel");
    return DS_OK;
}

extern "C" int ds_cheb_init(const float* R, int64_t ldr, float* D, int64_t ldd, float* W, int64_t ldw,
                            const float* dinv, int64_t nv, int ncols, float c, ds_stream_t stream) {
    DS_REQUIRE(R && D && W && dinv, "ds_cheb_init: null pointer");
    DS_REQUIRE(nv > 0 && ncols > 0 && ncols % 4 == 0, "ds_cheb_init: ncols must be a multiple of 4");
    DS_REQUIRE(aligned16(R, ldr) && aligned16(D, ldd) && aligned16(W, ldw), "ds_cheb_init: rows must be 16-byte aligned");
    const int cgroups = ncols / 4;
    const int64_t nthreads = nv * cgroups;
    cheb_kernel<false><<<(unsigned)ds::ceil_div(nthreads, 256), 256, 0, ds::as_stream(stream)>>>(
        nullptr, 0, const_cast<float*>(R), ldr, D, ldd, W, ldw, dinv, nv, cgroups, 0.f, c);
    DS_LAUNCH_CHECK("cheb_kernel<init>");
    return DS_OK;
}

extern "C" int ds_cheb_step(const float* AD, int64_t lda, float* R, int64_t ldr, float* D, int64_t ldd, float* W,
                            int64_t ldw, const float* dinv, int64_t nv, int ncols, float c1, float c2,
                            ds_stream_t stream) {
    DS_REQUIRE(AD && R && D && W && dinv, "ds_cheb_step: null pointer");
    DS_REQUIRE(nv > 0 && ncols > 0 && ncols % 4 == 0, "ds_cheb_step: ncols must be a multiple of 4");
    DS_REQUIRE(aligned16(AD, lda) && aligned16(R, ldr) && aligned16(D, ldd) && aligned16(W, ldw),
               "ds_cheb_step: rows must be 16-byte aligned");
    const int cgroups = ncols / 4;
    const int64_t nthreads = nv * cgroups;
    cheb_kernel<true><<<(unsigned)ds::ceil_div(nthreads, 256), 256, 0, ds::as_stream(stream)>>>(
        AD, lda, R, ldr, D, ldd, W, ldw, dinv, nv, cgroups, c1, c2);
    DS_LAUNCH_CHECK("cheb_kernel<step>");
    return DS_OK;
}

extern "C" int ds_mix(const float* A, int64_t lda, int p, const float* C, int q, float* Out, int64_t ldo, int64_t n,
                      float alpha, float beta, ds_stream_t stream) {
    DS_REQUIRE(A && C && Out, "ds_mix: null pointer");
    DS_REQUIRE(n > 0 && p > 0 && q > 0, "ds_mix: empty problem");
    DS_REQUIRE(lda >= p && ldo >= q, "ds_mix: leading dimension smaller than the block width");
    // Out may alias A (e.g. be a column range of it) when q <= 160: a wave reads its row tile completely before it
    // writes those rows, and no other wave touches them.  Wider results take several launches over A.
    {
        const char* a0 = reinterpret_cast<const char*>(A);
        const char* a1 = a0 + ((n - 1) * lda + p) * 4;
        const char* o0 = reinterpret_cast<const char*>(Out);
        const char* o1 = o0 + ((n - 1) * ldo + q) * 4;
        DS_REQUIRE(q <= 160 || o1 <= a0 || a1 <= o0, "ds_mix: Out overlaps A and q = %d > 160", q);
    }
    hipStream_t st = ds::as_stream(stream);
    const bool veca = aligned16(A, lda) && (p % 4 == 0);
    int rc = DS_OK;
    ds::ProfScope prof(stream, DS_PROF_MIX, p, n, q, 0);
    // LDS-staged path: the coefficient image ((p rounded to 16) x (q rounded to 16) floats) has to fit the CU's 160 KB;
    // up to 80 columns two workgroups share a CU, wider results (the fused [X' P'] = [X P W] [Z1 Zp] update) take it whole
    auto lds_path = [&](const float* A_, int p_, const float* C_, int q_, float* O_, float beta_) {
        switch ((q_ + 15) / 16) {
            case 1: return launch_mix_lds<1>(A_, lda, p_, C_, q, q_, O_, ldo, n, alpha, beta_, st);
            case 2: return launch_mix_lds<2>(A_, lda, p_, C_, q, q_, O_, ldo, n, alpha, beta_, st);
            case 3: return launch_mix_lds<3>(A_, lda, p_, C_, q, q_, O_, ldo, n, alpha, beta_, st);
            case 4: return launch_mix_lds<4>(A_, lda, p_, C_, q, q_, O_, ldo, n, alpha, beta_, st);
            case 5: return launch_mix_lds<5>(A_, lda, p_, C_, q, q_, O_, ldo, n, alpha, beta_, st);
            case 6: return launch_mix_lds<6>(A_, lda, p_, C_, q, q_, O_, ldo, n, alpha, beta_, st);
            case 7: return launch_mix_lds<7>(A_, lda, p_, C_, q, q_, O_, ldo, n, alpha, beta_, st);
            case 8: return launch_mix_lds<8>(A_, lda, p_, C_, q, q_, O_, ldo, n, alpha, beta_, st);
            case 9: return launch_mix_lds<9>(A_, lda, p_, C_, q, q_, O_, ldo, n, alpha, beta_, st);
            default: return launch_mix_lds<10>(A_, lda, p_, C_, q, q_, O_, ldo, n, alpha, beta_, st);
        }
    };
    const size_t image = (size_t)((p + 15) & ~15) * ((q + 15) / 16) * 16 * sizeof(float);
    if (veca && q <= 160 && p <= 256 && n >= 4096 && image <= 160 * 1024) return lds_path(A, p, C, q, Out, beta);
    // A deeper basis or a wider result (configs[4]: [X' P'] = [X P W] Z with a 136-column block is 408 -> 272 columns)
    // in slices of at most 160 columns whose coefficient image fits the LDS, the slices of the basis accumulated into
    // Out - unless Out overlaps A (a later slice would read what an earlier one wrote)
    {
        const char* a0 = reinterpret_cast<const char*>(A);
        const char* a1 = a0 + ((n - 1) * lda + p) * 4;
        const char* o0 = reinterpret_cast<const char*>(Out);
        const char* o1 = o0 + ((n - 1) * ldo + q) * 4;
        if (veca && n >= 4096 && (o1 <= a0 || a1 <= o0) && (((uintptr_t)Out | (uintptr_t)(ldo * 4)) & 3) == 0) {
            for (int j0 = 0; j0 < q && rc == DS_OK; j0 += 160) {
                const int qc = std::min(160, q - j0);
                const int slice = std::min(256, (int)(160 * 1024 / (((qc + 15) / 16) * 16 * sizeof(float))) & ~15);
                for (int k0 = 0; k0 < p && rc == DS_OK; k0 += slice)
                    rc = lds_path(A + k0, std::min(slice, p - k0), C + (int64_t)k0 * q + j0, qc, Out + j0, k0 == 0 ? beta : 1.f);
            }
            return rc;
        }
    }
    // column chunks of at most 10 MFMA tiles (160 columns) so the accumulators stay in registers
    for (int j_base = 0; j_base < q && rc == DS_OK; j_base += 160) {
        const int tiles = (int)ds::ceil_div(std::min(q - j_base, 160), 16);
        switch (tiles) {
            case 1: rc = launch_mix<1>(A, lda, p, C, q, Out, ldo, n, j_base, alpha, beta, veca, st); break;
            case 2: rc = launch_mix<2>(A, lda, p, C, q, Out, ldo, n, j_base, alpha, beta, veca, st); break;
            case 3: rc = launch_mix<3>(A, lda, p, C, q, Out, ldo, n, j_base, alpha, beta, veca, st); break;
            case 4: rc = launch_mix<4>(A, lda, p, C, q, Out, ldo, n, j_base, alpha, beta, veca, st); break;
            case 5: rc = launch_mix<5>(A, lda, p, C, q, Out, ldo, n, j_base, alpha, beta, veca, st); break;
            case 6: rc = launch_mix<6>(A, lda, p, C, q, Out, ldo, n, j_base, alpha, beta, veca, st); break;
            case 7:
            case 8: rc = launch_mix<8>(A, lda, p, C, q, Out, ldo, n, j_base, alpha, beta, veca, st); break;
            default: rc = launch_mix<10>(A, lda, p, C, q, Out, ldo, n, j_base, alpha, beta, veca, st); break;
        }
    }
    return rc;
}
