// Block SpMM  Y = A X  on the BSR-3 node pattern - gfx950.  The HBM-roofline kernel of the path.
//
// Replaces torch.sparse.mm(A, X) of the reference (src/lobpcg/_linalg_utils.py:36-37; also
// src/diffelastic/diff_model.py:385, 395-397) where A is a scalar COO matrix.  Exploits what the
// reference's matrices guarantee: K is made of full 3x3 node blocks and M = M_s (x) I3, so the
// index stream shrinks 9x (one int32 per block) and M needs one scalar per block.
//
// Mapping: X, Y are row-major (3 nv x ncols).  A group of LPN = ceil(ncols/VEC) lanes owns one
// node (3 rows of Y); each lane owns VEC consecutive columns, so every X access is a contiguous
// 16-byte (f32) load and a wave reads whole 3-row panels of a neighbour node (coalesced).  64/LPN
// node groups share a wave.  Block values are fetched by all lanes of a group from the same
// address (hardware broadcast).  Accumulation in registers, one store per output element.
// Algorithmic bytes per launch: nnzb*(vals + 4) + (nv+1)*4 + 2*3nv*ncols*sizeof(T).
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <mutex>
#include <type_traits>
#include <vector>

#include "ds_common.h"
#include "ds_diag.h"

namespace {

template <typename T, int V>
struct Vec;
template <>
struct Vec<float, 4> {
    using type = float4;
};
template <>
struct Vec<float, 2> {
    using type = float2;
};
template <>
struct Vec<double, 2> {
    using type = double2;
};

// KIND: 0 = 3x3 blocks, 1 = scalar (x I3).  TV value type, TX input type, TY output/accumulator.
// LPN_CT: lanes per node known at compile time (0 = run-time value).  The solver's block widths
// (b = 72/80 columns and 3b) get their own instantiations: the lane->(node, column) split becomes
// constant arithmetic, and rocprofv3 reports those launches under their own kernel names.
template <int KIND, typename TV, typename TX, typename TY, int VEC, int LPN_CT>
__global__ void __launch_bounds__(256)
    spmm_bsr3_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colidx,
                     const TV* __restrict__ vals, int64_t nv, const TX* __restrict__ X, int64_t ldx,
                     TY* __restrict__ Y, int64_t ldy, int ncols, int lpn_rt, int npw_rt, unsigned nblk) {
    const int lpn = LPN_CT ? LPN_CT : lpn_rt;
    const int npw = LPN_CT ? 64 / LPN_CT : npw_rt;
    const unsigned bid = ds::xcd_remap(blockIdx.x, nblk);
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int sub = lane / lpn;
    const int cl = lane - sub * lpn;
    const int64_t node = ((int64_t)bid * 4 + wave) * npw + sub;
    const int c0 = cl * VEC;
    if (sub >= npw || node >= nv || c0 >= ncols) return;
    // ncols is a multiple of VEC (checked on the host), so a lane is either fully in or out
    TY acc[3][VEC];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[r][v] = (TY)0;

    const int kb = rowptr[node], ke = rowptr[node + 1];
    using XV = typename Vec<TX, VEC>::type;
#pragma unroll 2
    for (int k = kb; k < ke; ++k) {
        const int64_t col = colidx[k];
        const TX* xp = X + (col * 3) * ldx + c0;
        const XV x0 = *reinterpret_cast<const XV*>(xp);
        const XV x1 = *reinterpret_cast<const XV*>(xp + ldx);
        const XV x2 = *reinterpret_cast<const XV*>(xp + 2 * ldx);
        const TX* x0p = reinterpret_cast<const TX*>(&x0);
        const TX* x1p = reinterpret_cast<const TX*>(&x1);
        const TX* x2p = reinterpret_cast<const TX*>(&x2);
        if (KIND == 0) {
            const TV* a = vals + (int64_t)k * 9;
            TY av[9];
#pragma unroll
            for (int q = 0; q < 9; ++q) av[q] = (TY)a[q];
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                const TY xa = (TY)x0p[v], xb = (TY)x1p[v], xc = (TY)x2p[v];
#pragma unroll
                for (int r = 0; r < 3; ++r)
                    acc[r][v] = fma(av[r * 3 + 0], xa, fma(av[r * 3 + 1], xb, fma(av[r * 3 + 2], xc, acc[r][v])));
            }
        } else {
            const TY m = (TY)vals[k];
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                acc[0][v] = fma(m, (TY)x0p[v], acc[0][v]);
                acc[1][v] = fma(m, (TY)x1p[v], acc[1][v]);
                acc[2][v] = fma(m, (TY)x2p[v], acc[2][v]);
            }
        }
    }
    using YV = typename Vec<TY, VEC>::type;
    TY* yp = Y + (node * 3) * ldy + c0;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        YV o;
        TY* op = reinterpret_cast<TY*>(&o);
#pragma unroll
        for (int v = 0; v < VEC; ++v) op[v] = acc[r][v];
        *reinterpret_cast<YV*>(yp + r * ldy) = o;
    }
}

template <int KIND, typename TV, typename TX, typename TY, int VEC, int LPN_CT>
int launch_ct(const int32_t* rowptr, const int32_t* colidx, const void* vals, int64_t nv, const void* X, int64_t ldx,
              void* Y, int64_t ldy, int ncols, int lpn, hipStream_t st) {
    const int npw = 64 / lpn;  // nodes per wave
    const int64_t nblk = ds::ceil_div(nv, (int64_t)npw * 4);
    spmm_bsr3_kernel<KIND, TV, TX, TY, VEC, LPN_CT><<<(unsigned)nblk, 256, 0, st>>>(
        rowptr, colidx, static_cast<const TV*>(vals), nv, static_cast<const TX*>(X), ldx, static_cast<TY*>(Y), ldy,
        ncols, lpn, npw, (unsigned)nblk);
    DS_LAUNCH_CHECK("spmm_bsr3_kernel");
    return DS_OK;
}

template <int KIND, typename TV, typename TX, typename TY, int VEC>
int launch(const int32_t* rowptr, const int32_t* colidx, const void* vals, int64_t nv, const void* X, int64_t ldx,
           void* Y, int64_t ldy, int ncols, hipStream_t st) {
    const int lpn = (ncols + VEC - 1) / VEC;  // lanes per node
    if (KIND == 0 && VEC == 4) {              // the eigensolver's stiffness products
        switch (lpn) {
            case 18: return launch_ct<KIND, TV, TX, TY, VEC, 18>(rowptr, colidx, vals, nv, X, ldx, Y, ldy, ncols, lpn, st);
            case 20: return launch_ct<KIND, TV, TX, TY, VEC, 20>(rowptr, colidx, vals, nv, X, ldx, Y, ldy, ncols, lpn, st);
            case 54: return launch_ct<KIND, TV, TX, TY, VEC, 54>(rowptr, colidx, vals, nv, X, ldx, Y, ldy, ncols, lpn, st);
            case 60: return launch_ct<KIND, TV, TX, TY, VEC, 60>(rowptr, colidx, vals, nv, X, ldx, Y, ldy, ncols, lpn, st);
            default: break;
        }
    }
    return launch_ct<KIND, TV, TX, TY, VEC, 0>(rowptr, colidx, vals, nv, X, ldx, Y, ldy, ncols, lpn, st);
}

// ------------------------------------------------------------------------------------------------
// Fast path for the f32 products of the eigensolver: ONE WAVE PER NODE, block metadata on the scalar
// path.  The node index is wave-uniform, so rowptr / colidx / the 3x3 values come through s_load
// (scalar cache, no vector-memory instructions, no dependent vector round trip before the X loads).
// RS = 3 (ncols <= 84): lane (cl, r) loads the 16 bytes [cl*4, cl*4+4) of input row r of the
//   neighbour, i.e. ONE 16-byte load instruction per block fetches the whole 3 x ncols panel; each
//   lane accumulates its row's contribution to all three output rows and the three lane groups are
//   merged by two cross-lane shuffles at the end (KIND 1, M = M_s (x) I3, needs no merge: row r only
//   feeds row r).
// RS = 1 (ncols <= 256): lane cl loads its 16 bytes of all three rows (three loads per block).
// Row metadata is fetched COOPERATIVELY once per row: lane l loads neighbour id l of the row (one
// coalesced load) and the row's 3x3 values stream into a per-wave LDS slab with coalesced loads; per
// block, the id comes back with v_readlane (scalar address arithmetic) and the coefficients with LDS
// broadcast reads.  The vector-memory pipe then carries only the X panels.  (Two earlier versions:
// s_load per block -> nine dependent scalar round trips per four blocks; per-lane vector loads of the
// coefficients -> 720 redundant bytes per block through the texture-addresser, as much as the X panel.
// Both sat at 1.3 TB/s.)
constexpr int WN_CHUNK = 64;  // blocks of a row staged per pass

// X panels are fetched with buffer loads: the lane's byte offset inside a 3-row panel is loop-invariant
// (voffset) and the neighbour's panel start is a wave-uniform scalar (soffset = id * panel bytes), so the
// per-block address arithmetic is one s_mul_i32 instead of a 64-bit multiply-add per lane.
using f4v = __attribute__((ext_vector_type(4))) float;
__device__ __forceinline__ f4v buf_load4(__amdgpu_buffer_rsrc_t rsrc, int voff, int soff) {
    return __builtin_bit_cast(f4v, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0));
}

// Optional fused epilogue (EPI = 1, KIND 0, RS 3): one term of the Chebyshev block-Jacobi polynomial,
//   Y_i <- X_i + c1 (X_i - Y_i) + c2 T_i (R0_i - (K X)_i)        (Y holds W_{k-1} on entry, W_{k+1} on exit)
// so a preconditioner term is ONE launch and the product K X never goes to HBM.
struct ChebEpilogue {
    const float* r0;
    int64_t ldr;
    const float* dinv;
    float c1, c2;
    int first;  // W_{k-1} = 0: Y is not read
    // neighbour-union kernel only: W_{k-1} read from here instead of from Y (nullptr: from Y, the in-place form) - the
    // last term of a polynomial run on compact scratch blocks can then land in a column range of a wider buffer
    const float* wprev = nullptr;
    int64_t ldp = 0;
    // neighbour-union kernel, epilogue 4 (fused residual R = K X - (M X) diag(lam)): node-scalar mass values in group order,
    // the Ritz values (device, fp64, one per column) and the per-group partial column norms (ngroups x 2 x ncols floats)
    const float* mvals = nullptr;
    const double* lam = nullptr;
    float* nwork = nullptr;
};

template <int KIND, int RS, int LPN_CT, int EPI>
__global__ void __launch_bounds__(256)
    spmm_wave_node_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colidx,
                          const float* __restrict__ vals, const float* __restrict__ vals_t, int64_t nv,
                          const float* __restrict__ X, int64_t ldx, float* __restrict__ Y, int64_t ldy, int lpn_rt,
                          unsigned nblk, ChebEpilogue epi, int big) {
    using f4 = __attribute__((ext_vector_type(4))) float;
    __shared__ float s_vals[4][WN_CHUNK * 9];
    const int lpn = LPN_CT ? LPN_CT : lpn_rt;
    const unsigned bid = ds::xcd_remap(blockIdx.x, nblk);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t node = (int64_t)bid * 4 + wave;
    if (node >= nv) return;  // wave-uniform
    const int r_raw = RS == 3 ? lane / lpn : 0;
    const int cl_raw = lane - r_raw * lpn;
    const bool active = r_raw < RS && cl_raw < lpn;
    // idle lanes shadow a valid lane (no exec-mask branches inside the loop); they never store
    const int r = active ? r_raw : 0;
    const int cl = active ? cl_raw : 0;
    const int c0 = cl * 4;
    const float* xbase = X + (int64_t)r * ldx + c0;
    const int64_t ldx3 = 3 * ldx;
    // whole X block as one buffer (host guarantees 3 nv ldx 4 < 2^32)
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(X), 0, big ? 0 : (int)(unsigned)(3 * nv * ldx * 4), 0x00020000);
    const int xvoff = (int)(((int64_t)r * ldx + c0) * 4);
    const int panel_bytes = (int)(ldx3 * 4);
    const int kb = rowptr[node], ke = rowptr[node + 1];
    float* sv = s_vals[wave];
    f4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0, acc2 = acc0;
    for (int kc = kb; kc < ke; kc += WN_CHUNK) {
        const int cnt = min(WN_CHUNK, ke - kc);  // wave-uniform
        const int colreg = lane < cnt ? colidx[kc + lane] : 0;
        float mreg = 0.f;
        if (KIND == 1) {
            mreg = lane < cnt ? vals[kc + lane] : 0.f;
        } else if (RS == 3) {
            const float* vsrc = vals + (int64_t)kc * 9;
            for (int t = lane; t < cnt * 9; t += 64) sv[t] = vsrc[t];
            // same-wave LDS write -> read: order them without a workgroup barrier (rows differ in length)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        int u = 0;
        // UX panel loads in flight per wave before the first use: the kernel is latency-bound (72 % of wave
        // cycles parked in s_waitcnt with 4 in flight), so depth, not bandwidth, is what to raise
        constexpr int UX = RS == 3 ? 8 : 4;
        for (; u + UX <= cnt; u += UX) {
            f4 xa[UX], xb[UX], xc[UX];
#pragma unroll
            for (int q = 0; q < UX; ++q) {
                const int j = __builtin_amdgcn_readlane(colreg, u + q);
                if (!big) {  // wave-uniform: blocks under 4 GiB go through the buffer descriptor
                    const int soff = j * panel_bytes;
                    xa[q] = buf_load4(xrsrc, xvoff, soff);
                    if (RS == 1) {
                        xb[q] = buf_load4(xrsrc, xvoff + (int)(ldx * 4), soff);
                        xc[q] = buf_load4(xrsrc, xvoff + (int)(ldx * 8), soff);
                    }
                } else {
                    const float* p = xbase + (int64_t)j * ldx3;
                    xa[q] = *reinterpret_cast<const f4*>(p);
                    if (RS == 1) {
                        xb[q] = *reinterpret_cast<const f4*>(p + ldx);
                        xc[q] = *reinterpret_cast<const f4*>(p + 2 * ldx);
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < UX; ++q) {
                if (KIND == 1) {
                    const float m = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mreg), u + q));
                    acc0 += m * xa[q];
                    if (RS == 1) {
                        acc1 += m * xb[q];
                        acc2 += m * xc[q];
                    }
                } else if (RS == 3) {
                    const float* a = sv + (u + q) * 9 + r;  // column r of the block
                    acc0 += a[0] * xa[q];
                    acc1 += a[3] * xa[q];
                    acc2 += a[6] * xa[q];
                } else {
                    const float* a = vals + (int64_t)(kc + u + q) * 9;  // same address in every lane: one request
                    acc0 += a[0] * xa[q] + a[1] * xb[q] + a[2] * xc[q];
                    acc1 += a[3] * xa[q] + a[4] * xb[q] + a[5] * xc[q];
                    acc2 += a[6] * xa[q] + a[7] * xb[q] + a[8] * xc[q];
                }
            }
        }
        for (; u < cnt; ++u) {
            const int j = __builtin_amdgcn_readlane(colreg, u);
            const float* p = xbase + (int64_t)j * ldx3;
            const f4 x0 = *reinterpret_cast<const f4*>(p);
            if (KIND == 1) {
                const float m = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mreg), u));
                acc0 += m * x0;
                if (RS == 1) {
                    acc1 += m * *reinterpret_cast<const f4*>(p + ldx);
                    acc2 += m * *reinterpret_cast<const f4*>(p + 2 * ldx);
                }
            } else if (RS == 3) {
                const float* a = sv + u * 9 + r;
                acc0 += a[0] * x0;
                acc1 += a[3] * x0;
                acc2 += a[6] * x0;
            } else {
                const float* a = vals + (int64_t)(kc + u) * 9;
                const f4 x1 = *reinterpret_cast<const f4*>(p + ldx);
                const f4 x2 = *reinterpret_cast<const f4*>(p + 2 * ldx);
                acc0 += a[0] * x0 + a[1] * x1 + a[2] * x2;
                acc1 += a[3] * x0 + a[4] * x1 + a[5] * x2;
                acc2 += a[6] * x0 + a[7] * x1 + a[8] * x2;
            }
        }
        if (KIND == 0 && RS == 3 && kc + WN_CHUNK < ke) {  // the slab is rewritten by the next pass
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (RS == 3 && KIND == 0) {
        // merge the three row groups: lanes [0,lpn) += lanes [lpn,2lpn) + lanes [2lpn,3lpn)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            acc0[v] += __shfl(acc0[v], lane + lpn) + __shfl(acc0[v], lane + 2 * lpn);
            acc1[v] += __shfl(acc1[v], lane + lpn) + __shfl(acc1[v], lane + 2 * lpn);
            acc2[v] += __shfl(acc2[v], lane + lpn) + __shfl(acc2[v], lane + 2 * lpn);
        }
    }
    float* yp = Y + (node * 3) * ldy + c0;
    if (EPI == 1) {
        if (active && r_raw == 0) {
            const float* rp = epi.r0 + (node * 3) * epi.ldr + c0;
            const float* xp = X + (node * 3) * ldx + c0;
            const float* d = epi.dinv + node * 9;
            const f4 q0 = *reinterpret_cast<const f4*>(rp) - acc0;
            const f4 q1 = *reinterpret_cast<const f4*>(rp + epi.ldr) - acc1;
            const f4 q2 = *reinterpret_cast<const f4*>(rp + 2 * epi.ldr) - acc2;
            const f4 t0 = d[0] * q0 + d[1] * q1 + d[2] * q2;
            const f4 t1 = d[3] * q0 + d[4] * q1 + d[5] * q2;
            const f4 t2 = d[6] * q0 + d[7] * q1 + d[8] * q2;
            const f4 w0 = *reinterpret_cast<const f4*>(xp);
            const f4 w1 = *reinterpret_cast<const f4*>(xp + ldx);
            const f4 w2 = *reinterpret_cast<const f4*>(xp + 2 * ldx);
            f4 o0 = w0 + epi.c2 * t0, o1 = w1 + epi.c2 * t1, o2 = w2 + epi.c2 * t2;
            if (!epi.first) {
                o0 += epi.c1 * (w0 - *reinterpret_cast<const f4*>(yp));
                o1 += epi.c1 * (w1 - *reinterpret_cast<const f4*>(yp + ldy));
                o2 += epi.c1 * (w2 - *reinterpret_cast<const f4*>(yp + 2 * ldy));
            } else {  // W_prev = 0 (not read)
                o0 += epi.c1 * w0;
                o1 += epi.c1 * w1;
                o2 += epi.c1 * w2;
            }
            *reinterpret_cast<f4*>(yp) = o0;
            *reinterpret_cast<f4*>(yp + ldy) = o1;
            *reinterpret_cast<f4*>(yp + 2 * ldy) = o2;
        }
    } else if (EPI == 2) {  // Y = R0 - K X (block residual feeding the coarse-level restriction)
        if (active && r_raw == 0) {
            const float* rp = epi.r0 + (node * 3) * epi.ldr + c0;
            *reinterpret_cast<f4*>(yp) = *reinterpret_cast<const f4*>(rp) - acc0;
            *reinterpret_cast<f4*>(yp + ldy) = *reinterpret_cast<const f4*>(rp + epi.ldr) - acc1;
            *reinterpret_cast<f4*>(yp + 2 * ldy) = *reinterpret_cast<const f4*>(rp + 2 * epi.ldr) - acc2;
        }
    } else if (RS == 3 && KIND == 1) {
        if (active) *reinterpret_cast<f4*>(yp + (int64_t)r * ldy) = acc0;
    } else if (active && r_raw == 0) {
        *reinterpret_cast<f4*>(yp) = acc0;
        *reinterpret_cast<f4*>(yp + ldy) = acc1;
        *reinterpret_cast<f4*>(yp + 2 * ldy) = acc2;
    }
}

template <int KIND, int RS, int LPN_CT, int EPI = 0>
int launch_wn(const int32_t* rowptr, const int32_t* colidx, const void* vals, const void* vals_t, int64_t nv,
              const void* X, int64_t ldx, void* Y, int64_t ldy, int lpn, hipStream_t st,
              ChebEpilogue epi = ChebEpilogue{nullptr, 0, nullptr, 0.f, 0.f, 0}) {
    const int64_t nblk = ds::ceil_div(nv, 4);
    spmm_wave_node_kernel<KIND, RS, LPN_CT, EPI><<<(unsigned)nblk, 256, 0, st>>>(
        rowptr, colidx, static_cast<const float*>(vals), static_cast<const float*>(vals_t), nv,
        static_cast<const float*>(X), ldx, static_cast<float*>(Y), ldy, lpn, (unsigned)nblk, epi,
        (3 * nv * ldx * 4 >= ((int64_t)1 << 32)) ? 1 : 0);
    DS_LAUNCH_CHECK("spmm_wave_node_kernel");
    return DS_OK;
}

#include "spmm_union.inc"
#include "spmm_mfma.inc"
#include "spmm_mfma32.inc"
#include "spmm_narrow.inc"
#include "spmm_f64_union.inc"

template <int KIND>
int launch_fast(const int32_t* rowptr, const int32_t* colidx, const void* vals, const void* vals_t, int64_t nv,
                const void* X, int64_t ldx, void* Y, int64_t ldy, int ncols, hipStream_t st) {
    const int lpn = ncols / 4;
    if (lpn <= 21) {
        switch (lpn) {  // the solver's block widths get compile-time lane splits and their own kernel names
            case 18: return launch_wn<KIND, 3, 18>(rowptr, colidx, vals, vals_t, nv, X, ldx, Y, ldy, lpn, st);
            case 20: return launch_wn<KIND, 3, 20>(rowptr, colidx, vals, vals_t, nv, X, ldx, Y, ldy, lpn, st);
            default: return launch_wn<KIND, 3, 0>(rowptr, colidx, vals, vals_t, nv, X, ldx, Y, ldy, lpn, st);
        }
    }
    switch (lpn) {
        case 54: return launch_wn<KIND, 1, 54>(rowptr, colidx, vals, vals_t, nv, X, ldx, Y, ldy, lpn, st);
        case 60: return launch_wn<KIND, 1, 60>(rowptr, colidx, vals, vals_t, nv, X, ldx, Y, ldy, lpn, st);
        default: return launch_wn<KIND, 1, 0>(rowptr, colidx, vals, vals_t, nv, X, ldx, Y, ldy, lpn, st);
    }
}

// ------------------------------------------------------------------------------------------------
// fp64-value products of the read-out (X^T K_lambda X, X^T K_mu X, X^T M X on the converged block: kinds 2 / 3,
// fp64 values and result, fp32 X) for <= 84 columns: ONE WAVEFRONT PER NODE in the lane layout of the fp32
// kernels above - lane (g, cl) loads 16 bytes of row g of every neighbour panel and, for 3x3 blocks, column g of
// the block (3 doubles), accumulates its share of all three output rows in fp64 and the three lane groups are
// merged at the end.  The generic kernel above gives a lane 2 columns of one node and makes it load all 9 values
// of every block itself: 12 dependent-address loads per block and lane, 1.17 ms per 64-column product on the
// benchmark mesh; this one issues 4.
// (round 4: the row's fp64 coefficients are staged per chunk in LDS with coalesced loads and the neighbour ids come back with
// v_readlane, as in spmm_f64_polish_kernel below - three dependent-address global loads per block and lane fewer; same sums in
// the same order)
constexpr int F64_CHUNK = 32;  // blocks of a row staged per pass
constexpr int F64_WIN = 8;     // neighbour panels in flight per wave
template <int KIND, typename TX>
__global__ void __launch_bounds__(256)
    spmm_f64_node_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colidx,
                         const double* __restrict__ vals, int64_t nv, const TX* __restrict__ X, int64_t ldx,
                         double* __restrict__ Y, int64_t ldy, int lpn, unsigned nblk) {
    using d2 = __attribute__((ext_vector_type(2))) double;
    using xv4 = __attribute__((ext_vector_type(4))) TX;  // 16 bytes of an f32 row, 32 bytes of an f64 row
    __shared__ double s_v[4][F64_CHUNK * (KIND == 0 ? 9 : 1)];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t node = (int64_t)ds::xcd_remap(blockIdx.x, nblk) * 4 + wave;
    if (node >= nv) return;  // wave-uniform
    const int g = lane / lpn, cl = lane - g * lpn;
    const bool active = g < 3;
    const int ga = active ? g : 0;
    const int kb = __builtin_amdgcn_readfirstlane(rowptr[node]), ke = __builtin_amdgcn_readfirstlane(rowptr[node + 1]);
    const TX* xb = X + (int64_t)ga * ldx + cl * 4;
    double* sv = s_v[wave];
    double acc[3][4];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int v = 0; v < 4; ++v) acc[r][v] = 0.0;
    if (KIND == 1) {  // node scalars: index and value of a block are wave-uniform loads (the scalar path): nothing to stage
#pragma unroll 4
        for (int k = kb; k < ke; ++k) {
            const int64_t col = colidx[k];
            const xv4 x = *reinterpret_cast<const xv4*>(xb + col * 3 * ldx);
            const double m = vals[k];
#pragma unroll
            for (int v = 0; v < 4; ++v) acc[0][v] = fma(m, (double)x[v], acc[0][v]);  // m I3: row g feeds row g only
        }
    } else {
        for (int kc = kb; kc < ke; kc += F64_CHUNK) {
            const int cnt = min(F64_CHUNK, ke - kc);  // wave-uniform
            const int colreg = colidx[kc + min(lane, cnt - 1)];
            const double* pv = vals + (int64_t)kc * 9;
            // (all the chunk's coefficient loads in flight at once, clamped instead of predicated: a loop of one load per
            // trip met every load with a full wait)
            constexpr int NSL = (F64_CHUNK * 9 + 63) / 64;
            double stg[NSL];
#pragma unroll
            for (int i = 0; i < NSL; ++i) stg[i] = __builtin_nontemporal_load(pv + min(lane + 64 * i, cnt * 9 - 1));  // (read once: spmm_union.inc)
#pragma unroll
            for (int i = 0; i < NSL; ++i)
                if (lane + 64 * i < cnt * 9) sv[lane + 64 * i] = stg[i];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // The gathers of F64_WIN neighbour panels are in flight before the first of them is used (the loop used to
            // issue one and wait for it: every block of a row paid a full L2 round trip).  Slots past the chunk's last
            // block re-read it and are skipped; the sums and their order are unchanged.
            for (int u0 = 0; u0 < cnt; u0 += F64_WIN) {
                xv4 xs[F64_WIN];
#pragma unroll
                for (int j = 0; j < F64_WIN; ++j) {
                    const int64_t col = __builtin_amdgcn_readlane(colreg, min(u0 + j, cnt - 1));
                    xs[j] = *reinterpret_cast<const xv4*>(xb + col * 3 * ldx);
                }
#pragma unroll
                for (int j = 0; j < F64_WIN; ++j) {
                    if (u0 + j < cnt) {  // wave-uniform
                        const double* a = sv + (u0 + j) * 9 + ga;  // column g of the block
                        const double a0 = a[0], a1 = a[3], a2 = a[6];
#pragma unroll
                        for (int v = 0; v < 4; ++v) {
                            const double xv = (double)xs[j][v];
                            acc[0][v] = fma(a0, xv, acc[0][v]);
                            acc[1][v] = fma(a1, xv, acc[1][v]);
                            acc[2][v] = fma(a2, xv, acc[2][v]);
                        }
                    }
                }
            }
            if (kc + F64_CHUNK < ke) {  // the slab is rewritten by the next pass
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
    double out[4];
    if (KIND == 0) {
        // lane (g, cl) ends with output row g: its own share plus the shares of the two other groups, which send
        // their row (g_src + s) mod 3 to the group s steps ahead
#pragma unroll
        for (int v = 0; v < 4; ++v) out[v] = ga == 0 ? acc[0][v] : (ga == 1 ? acc[1][v] : acc[2][v]);
#pragma unroll
        for (int s_ = 1; s_ <= 2; ++s_) {
            const int to_row = (ga + s_) % 3;
            int src_g = ga - s_;
            if (src_g < 0) src_g += 3;
            const int src_lane = active ? src_g * lpn + cl : lane;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const double send = to_row == 0 ? acc[0][v] : (to_row == 1 ? acc[1][v] : acc[2][v]);
                out[v] += __shfl(send, src_lane, 64);
            }
        }
    } else {
#pragma unroll
        for (int v = 0; v < 4; ++v) out[v] = acc[0][v];
    }
    if (active) {
        double* yp = Y + (node * 3 + g) * ldy + cl * 4;
        __builtin_nontemporal_store(d2{out[0], out[1]}, reinterpret_cast<d2*>(yp));  // (written once)
        __builtin_nontemporal_store(d2{out[2], out[3]}, reinterpret_cast<d2*>(yp + 2));
    }
}

// The read-out's three fp64-value products in ONE walk of the pattern: Ya = A X, Yb = B X (3x3 fp64 blocks: K_lambda and
// K_mu) and Ym = (m (x) I3) X (the node-scalar mass values) share every gathered panel of the fp32 block X.  Three
// launches of spmm_f64_node_kernel are bound by exactly those gathers (3.2 GB out of L2 per 64-column product at the
// benchmark size: 0.63 + 0.63 + 0.22 ms); per output the arithmetic and its order are theirs, so the results are
// bit-identical.
// Round 4: the row's fp64 coefficients (9 + 9 + 1 doubles per block) no longer come through seven dependent-address global
// loads per block and lane (the kernel was bound by vector-memory ISSUE: 33 M wave-loads for 4.1 M blocks) but are staged per
// chunk of the row in LDS with coalesced 8-byte loads and read back as broadcasts; the neighbour ids of the chunk sit in one
// register (a coalesced load) and come back with v_readlane, so a panel's address is scalar arithmetic.  The sums and their
// order are unchanged: results bit-identical to the three separate launches, as before.
constexpr int PL_CHUNK = 32;  // blocks of a row staged per pass
constexpr int PL_WIN = 4;     // panels per gather window (two windows: 8 in flight)
// TY: the type the three results are STORED in - double, or float (round 5: the sums are formed in fp64 either way and rounded
// once; the fp32 blocks halve the 0.86 GB the kernel writes and the Gram launch behind it reads, and that Gram then runs on the
// fp32 matrix-core path; a result entry carries 6e-8 of relative rounding, the Gram entries ~1e-10 - see polish_products)
template <typename TY>
__global__ void __launch_bounds__(256)
    spmm_f64_polish_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colidx,
                           const double* __restrict__ va, const double* __restrict__ vb, const double* __restrict__ vm,
                           int64_t nv, const float* __restrict__ X, int64_t ldx, TY* __restrict__ Ya,
                           TY* __restrict__ Yb, TY* __restrict__ Ym, int64_t ldy, int lpn, unsigned nblk) {
    using d2 = __attribute__((ext_vector_type(2))) double;
    using xv4 = __attribute__((ext_vector_type(4))) float;
    __shared__ double s_a[4][PL_CHUNK * 9], s_b[4][PL_CHUNK * 9], s_m[4][PL_CHUNK];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t node = (int64_t)ds::xcd_remap(blockIdx.x, nblk) * 4 + wave;
    if (node >= nv) return;  // wave-uniform
    const int g = lane / lpn, cl = lane - g * lpn;
    const bool active = g < 3;
    const int ga = active ? g : 0;
    const int kb = __builtin_amdgcn_readfirstlane(rowptr[node]), ke = __builtin_amdgcn_readfirstlane(rowptr[node + 1]);
    const float* xb = X + (int64_t)ga * ldx + cl * 4;
    double* sa = s_a[wave];
    double* sb = s_b[wave];
    double* sm = s_m[wave];
    double aa[3][4], ab[3][4], am[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        am[v] = 0.0;
#pragma unroll
        for (int r = 0; r < 3; ++r) aa[r][v] = ab[r][v] = 0.0;
    }
    for (int kc = kb; kc < ke; kc += PL_CHUNK) {
        const int cnt = min(PL_CHUNK, ke - kc);  // wave-uniform
        const int colreg = colidx[kc + min(lane, cnt - 1)];
        const double* pa = va + (int64_t)kc * 9;
        const double* pb = vb + (int64_t)kc * 9;
        {  // all the chunk's coefficient loads in flight at once (clamped, not predicated)
            constexpr int NSL = (PL_CHUNK * 9 + 63) / 64;
            double ra[NSL], rb[NSL];
#pragma unroll
            for (int i = 0; i < NSL; ++i) {
                const int t = min(lane + 64 * i, cnt * 9 - 1);
                ra[i] = __builtin_nontemporal_load(pa + t), rb[i] = __builtin_nontemporal_load(pb + t);  // (read once)
            }
            const double rm = vm[kc + min(lane, cnt - 1)];
#pragma unroll
            for (int i = 0; i < NSL; ++i)
                if (lane + 64 * i < cnt * 9) sa[lane + 64 * i] = ra[i], sb[lane + 64 * i] = rb[i];
            if (lane < cnt) sm[lane] = rm;
        }
        // same-wave LDS write -> read (rows differ in length: no workgroup barrier)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // Two windows of PL_WIN panel gathers: the next window is requested before the current one is used (see
        // spmm_f64_node_kernel); same sums, same order
        auto issue = [&](xv4 (&xs)[PL_WIN], int u0) {
#pragma unroll
            for (int j = 0; j < PL_WIN; ++j) {
                const int64_t col = __builtin_amdgcn_readlane(colreg, min(u0 + j, cnt - 1));
                xs[j] = *reinterpret_cast<const xv4*>(xb + col * 3 * ldx);
            }
        };
        auto consume = [&](const xv4 (&xs)[PL_WIN], int u0) {
#pragma unroll
            for (int j = 0; j < PL_WIN; ++j) {
                if (u0 + j < cnt) {  // wave-uniform
                    const int u = u0 + j;
                    const double* ca = sa + u * 9 + ga;  // column g of the blocks
                    const double* cb = sb + u * 9 + ga;
                    const double a0 = ca[0], a1 = ca[3], a2 = ca[6], b0 = cb[0], b1 = cb[3], b2 = cb[6], m = sm[u];
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const double xv = (double)xs[j][v];
                        aa[0][v] = fma(a0, xv, aa[0][v]);
                        aa[1][v] = fma(a1, xv, aa[1][v]);
                        aa[2][v] = fma(a2, xv, aa[2][v]);
                        ab[0][v] = fma(b0, xv, ab[0][v]);
                        ab[1][v] = fma(b1, xv, ab[1][v]);
                        ab[2][v] = fma(b2, xv, ab[2][v]);
                        am[v] = fma(m, xv, am[v]);
                    }
                }
            }
        };
        xv4 x0[PL_WIN], x1[PL_WIN];
        issue(x0, 0);
        for (int u0 = 0; u0 < cnt; u0 += 2 * PL_WIN) {
            issue(x1, u0 + PL_WIN);  // (a window past the chunk re-reads its last panel and is skipped)
            consume(x0, u0);
            issue(x0, u0 + 2 * PL_WIN);
            consume(x1, u0 + PL_WIN);
        }
        if (kc + PL_CHUNK < ke) {  // the slabs are rewritten by the next pass
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
    // lane (g, cl) ends with output row g: its own share plus the shares of the two other groups (as the node kernel)
    double oa[4], ob[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        oa[v] = ga == 0 ? aa[0][v] : (ga == 1 ? aa[1][v] : aa[2][v]);
        ob[v] = ga == 0 ? ab[0][v] : (ga == 1 ? ab[1][v] : ab[2][v]);
    }
#pragma unroll
    for (int s_ = 1; s_ <= 2; ++s_) {
        const int to_row = (ga + s_) % 3;
        int src_g = ga - s_;
        if (src_g < 0) src_g += 3;
        const int src_lane = active ? src_g * lpn + cl : lane;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const double sa = to_row == 0 ? aa[0][v] : (to_row == 1 ? aa[1][v] : aa[2][v]);
            const double sb = to_row == 0 ? ab[0][v] : (to_row == 1 ? ab[1][v] : ab[2][v]);
            oa[v] += __shfl(sa, src_lane, 64);
            ob[v] += __shfl(sb, src_lane, 64);
        }
    }
    if (active) {
        const int64_t o = (node * 3 + g) * ldy + cl * 4;
        if constexpr (sizeof(TY) == 8) {
            __builtin_nontemporal_store(d2{oa[0], oa[1]}, reinterpret_cast<d2*>(Ya + o));  // (written once)
            __builtin_nontemporal_store(d2{oa[2], oa[3]}, reinterpret_cast<d2*>(Ya + o + 2));
            __builtin_nontemporal_store(d2{ob[0], ob[1]}, reinterpret_cast<d2*>(Yb + o));
            __builtin_nontemporal_store(d2{ob[2], ob[3]}, reinterpret_cast<d2*>(Yb + o + 2));
            __builtin_nontemporal_store(d2{am[0], am[1]}, reinterpret_cast<d2*>(Ym + o));
            __builtin_nontemporal_store(d2{am[2], am[3]}, reinterpret_cast<d2*>(Ym + o + 2));
        } else {
            *reinterpret_cast<xv4*>(Ya + o) = xv4{(float)oa[0], (float)oa[1], (float)oa[2], (float)oa[3]};
            *reinterpret_cast<xv4*>(Yb + o) = xv4{(float)ob[0], (float)ob[1], (float)ob[2], (float)ob[3]};
            *reinterpret_cast<xv4*>(Ym + o) = xv4{(float)am[0], (float)am[1], (float)am[2], (float)am[3]};
        }
    }
}

}  // namespace

extern "C" int ds_spmm_bsr3(int kind, const int32_t* rowptr, const int32_t* colidx, const void* vals,
                            const void* vals_t, int64_t nv, const void* X, int64_t ldx, void* Y, int64_t ldy,
                            int ncols, ds_stream_t stream) {
    DS_REQUIRE(rowptr && colidx && vals && X && Y, "ds_spmm_bsr3: null pointer");
    DS_REQUIRE(nv > 0 && ncols > 0, "ds_spmm_bsr3: empty problem");
    DS_REQUIRE(kind >= 0 && kind <= 5, "ds_spmm_bsr3: unknown kind %d", kind);
    DS_REQUIRE(ldx >= ncols && ldy >= ncols, "ds_spmm_bsr3: leading dimension smaller than ncols");
    hipStream_t st = ds::as_stream(stream);
    const bool f64out = kind >= 2, f64in = kind >= 4;
    const int xalign = (int)(reinterpret_cast<uintptr_t>(X) | (uintptr_t)(ldx * (f64in ? 8 : 4)));
    const int yalign = (int)(reinterpret_cast<uintptr_t>(Y) | (uintptr_t)(ldy * (f64out ? 8 : 4)));
    if (!f64out) {
        // float4 path needs 16-byte aligned rows; float2 path 8-byte
        if (ncols % 4 == 0 && ncols <= 256 && (xalign & 15) == 0 && (yalign & 15) == 0) {
            // (the fast path addresses X through one buffer descriptor: 32-bit byte offsets)
            return kind == 0 ? launch_fast<0>(rowptr, colidx, vals, vals_t, nv, X, ldx, Y, ldy, ncols, st)
                             : launch_fast<1>(rowptr, colidx, vals, vals_t, nv, X, ldx, Y, ldy, ncols, st);
        }
        DS_REQUIRE(ncols % 2 == 0 && ncols <= 128 && (xalign & 7) == 0 && (yalign & 7) == 0,
                   "ds_spmm_bsr3: f32 blocks need an even column count <= 256 (multiple of 4 above 128) and "
                   "8/16-byte aligned rows (ncols=%d ldx=%lld ldy=%lld)",
                   ncols, (long long)ldx, (long long)ldy);
        return kind == 0 ? launch<0, float, float, float, 2>(rowptr, colidx, vals, nv, X, ldx, Y, ldy, ncols, st)
                         : launch<1, float, float, float, 2>(rowptr, colidx, vals, nv, X, ldx, Y, ldy, ncols, st);
    }
    const unsigned nblk = (unsigned)ds::ceil_div(nv, 4);
    const double* dv = static_cast<const double*>(vals);
    double* dy = static_cast<double*>(Y);
    if (f64in) {  // fp64 iterates of the refinement phase: the wave-per-node kernel only
        DS_REQUIRE(ncols % 4 == 0 && ncols <= 84 && (xalign & 31) == 0 && (yalign & 15) == 0,
                   "ds_spmm_bsr3: f64-input blocks need a column count that is a multiple of 4, <= 84, and 32-byte "
                   "aligned rows (ncols=%d)", ncols);
        const double* dx = static_cast<const double*>(X);
        if (kind == 4)
            spmm_f64_node_kernel<0, double><<<nblk, 256, 0, st>>>(rowptr, colidx, dv, nv, dx, ldx, dy, ldy, ncols / 4, nblk);
        else
            spmm_f64_node_kernel<1, double><<<nblk, 256, 0, st>>>(rowptr, colidx, dv, nv, dx, ldx, dy, ldy, ncols / 4, nblk);
        DS_LAUNCH_CHECK("spmm_f64_node_kernel<double>");
        return DS_OK;
    }
    DS_REQUIRE(ncols % 2 == 0 && ncols <= 128 && (xalign & 7) == 0 && (yalign & 15) == 0,
               "ds_spmm_bsr3: f64-output blocks need an even column count <= 128 and aligned rows (ncols=%d)", ncols);
    if (ncols % 4 == 0 && ncols <= 84 && (xalign & 15) == 0) {
        const float* fx = static_cast<const float*>(X);
        if (kind == 2)
            spmm_f64_node_kernel<0, float><<<nblk, 256, 0, st>>>(rowptr, colidx, dv, nv, fx, ldx, dy, ldy, ncols / 4, nblk);
        else
            spmm_f64_node_kernel<1, float><<<nblk, 256, 0, st>>>(rowptr, colidx, dv, nv, fx, ldx, dy, ldy, ncols / 4, nblk);
        DS_LAUNCH_CHECK("spmm_f64_node_kernel");
        return DS_OK;
    }
    return kind == 2 ? launch<0, double, float, double, 2>(rowptr, colidx, vals, nv, X, ldx, Y, ldy, ncols, st)
                     : launch<1, double, float, double, 2>(rowptr, colidx, vals, nv, X, ldx, Y, ldy, ncols, st);
}

// ------------------------------------------------------------------------------------------------
// Pack the (transposed) blocks into the entry-major / node-minor order of the neighbour-union tables.
__global__ void pack_groups_kernel(const float* __restrict__ vals_t, const int32_t* __restrict__ kperm, int64_t nnzb,
                                   float* __restrict__ kgrp) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nnzb * 9) return;
    const int64_t p = i / 9;
    kgrp[i] = vals_t[(int64_t)kperm[p] * 9 + (i - p * 9)];
}

extern "C" int ds_pack_groups(const float* vals_t, const int32_t* kperm, int64_t nnzb, float* kgrp,
                              ds_stream_t stream) {
    DS_REQUIRE(vals_t && kperm && kgrp && nnzb > 0, "ds_pack_groups: bad argument");
    pack_groups_kernel<<<(unsigned)ds::ceil_div(nnzb * 9, 256), 256, 0, ds::as_stream(stream)>>>(vals_t, kperm, nnzb,
                                                                                                kgrp);
    DS_LAUNCH_CHECK("pack_groups_kernel");
    return DS_OK;
}

extern "C" int ds_cheb_spmm(const int32_t* rowptr, const int32_t* colidx, const float* vals, int64_t nv, const float* W,
                            int64_t ldw, float* Wprev, int64_t ldp, const float* R0, int64_t ldr, const float* dinv,
                            int ncols, float c1, float c2, int first, ds_stream_t stream) {
    DS_REQUIRE(rowptr && colidx && vals && W && Wprev && R0 && dinv, "ds_cheb_spmm: null pointer");
    DS_REQUIRE(nv > 0 && ncols > 0 && ncols % 4 == 0 && ncols <= 84, "ds_cheb_spmm: ncols must be a multiple of 4 <= 84");
    DS_REQUIRE(ldw >= ncols && ldp >= ncols && ldr >= ncols, "ds_cheb_spmm: leading dimension smaller than ncols");
    const uintptr_t al = reinterpret_cast<uintptr_t>(W) | reinterpret_cast<uintptr_t>(Wprev) |
                         reinterpret_cast<uintptr_t>(R0) | (uintptr_t)(ldw * 4) | (uintptr_t)(ldp * 4) |
                         (uintptr_t)(ldr * 4);
    DS_REQUIRE((al & 15) == 0, "ds_cheb_spmm: rows must be 16-byte aligned");
    DS_REQUIRE(W != Wprev, "ds_cheb_spmm: W and Wprev must be different buffers");
    hipStream_t st = ds::as_stream(stream);
    const ChebEpilogue epi{R0, ldr, dinv, c1, c2, first};
    const int lpn = ncols / 4;
    switch (lpn) {
        case 18: return launch_wn<0, 3, 18, 1>(rowptr, colidx, vals, nullptr, nv, W, ldw, Wprev, ldp, lpn, st, epi);
        case 20: return launch_wn<0, 3, 20, 1>(rowptr, colidx, vals, nullptr, nv, W, ldw, Wprev, ldp, lpn, st, epi);
        default: return launch_wn<0, 3, 0, 1>(rowptr, colidx, vals, nullptr, nv, W, ldw, Wprev, ldp, lpn, st, epi);
    }
}

// Y = R0 - K X on a block of <= 84 columns: the residual the two-level preconditioner restricts to the
// corner-node (P1) level, produced without writing K X to HBM.
extern "C" int ds_spmm_residual(const int32_t* rowptr, const int32_t* colidx, const float* vals, int64_t nv,
                                const float* X, int64_t ldx, const float* R0, int64_t ldr, float* Y, int64_t ldy,
                                int ncols, ds_stream_t stream) {
    DS_REQUIRE(rowptr && colidx && vals && X && R0 && Y, "ds_spmm_residual: null pointer");
    DS_REQUIRE(nv > 0 && ncols > 0 && ncols % 4 == 0 && ncols <= 84, "ds_spmm_residual: ncols must be a multiple of 4 <= 84");
    DS_REQUIRE(ldx >= ncols && ldy >= ncols && ldr >= ncols, "ds_spmm_residual: leading dimension smaller than ncols");
    const uintptr_t al = reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(Y) |
                         reinterpret_cast<uintptr_t>(R0) | (uintptr_t)(ldx * 4) | (uintptr_t)(ldy * 4) |
                         (uintptr_t)(ldr * 4);
    DS_REQUIRE((al & 15) == 0, "ds_spmm_residual: rows must be 16-byte aligned");
    DS_REQUIRE(X != Y, "ds_spmm_residual: X and Y must be different buffers");
    hipStream_t st = ds::as_stream(stream);
    const ChebEpilogue epi{R0, ldr, nullptr, 0.f, 0.f, 0};
    const int lpn = ncols / 4;
    switch (lpn) {
        case 20: return launch_wn<0, 3, 20, 2>(rowptr, colidx, vals, nullptr, nv, X, ldx, Y, ldy, lpn, st, epi);
        default: return launch_wn<0, 3, 0, 2>(rowptr, colidx, vals, nullptr, nv, X, ldx, Y, ldy, lpn, st, epi);
    }
}


// ------------------------------------------------------------------------------------------------
// Launch timing hook (bench.py's live roofline figures): while a stream is registered with ds_profile_stream, every launch
// of an ENABLED kind (ds_profile_kinds; default: the fused Chebyshev term only) on it is bracketed by HIP events;
// ds_profile_collect returns the durations with the shape of each launch.  One registered stream at a time; a relaxed flag
// keeps the cost off every other launch.
namespace {
struct TermRecord {
    hipEvent_t e0, e1;
    int64_t nv, nnzb;
    int ncols, first;  // first | element bytes of the vector blocks << 8 | kind << 16
};
std::atomic<void*> g_prof_stream{nullptr};
std::atomic<unsigned> g_prof_kinds{1u};
std::mutex g_prof_mutex;
std::vector<TermRecord> g_prof_records;
size_t g_prof_cap = 0;
}  // namespace

ds::ProfScope::ProfScope(ds_stream_t stream, int kind, int64_t a, int64_t b, int c, int d) {
    if (stream == nullptr || g_prof_stream.load(std::memory_order_relaxed) != stream) return;
    if (!((g_prof_kinds.load(std::memory_order_relaxed) >> kind) & 1u)) return;
    std::lock_guard<std::mutex> lock(g_prof_mutex);
    if (g_prof_records.size() >= g_prof_cap) return;
    auto* r = new TermRecord{nullptr, nullptr, a, b, c, (d & 0xffff) | (kind << 16)};
    if (hipEventCreate(&r->e0) != hipSuccess) {
        delete r;
        return;
    }
    if (hipEventCreate(&r->e1) != hipSuccess) {
        (void)hipEventDestroy(r->e0);
        delete r;
        return;
    }
    st_ = ds::as_stream(stream);
    (void)hipEventRecord(r->e0, st_);
    rec_ = r;
}

ds::ProfScope::~ProfScope() {
    if (!rec_) return;
    auto* r = static_cast<TermRecord*>(rec_);
    (void)hipEventRecord(r->e1, st_);
    std::lock_guard<std::mutex> lock(g_prof_mutex);
    g_prof_records.push_back(*r);
    delete r;
}

extern "C" int ds_profile_kinds(unsigned mask) {
    g_prof_kinds.store(mask ? mask : 1u);
    return DS_OK;
}

extern "C" int ds_profile_stream(ds_stream_t stream, int64_t capacity) {
    std::lock_guard<std::mutex> lock(g_prof_mutex);
    for (auto& r : g_prof_records) {
        (void)hipEventDestroy(r.e0);
        (void)hipEventDestroy(r.e1);
    }
    g_prof_records.clear();
    g_prof_cap = capacity > 0 ? (size_t)capacity : 0;
    g_prof_stream.store(capacity > 0 ? stream : nullptr);
    return DS_OK;
}

extern "C" int64_t ds_profile_collect(float* ms, int64_t* nv, int64_t* nnzb, int32_t* ncols, int32_t* first, int64_t cap) {
    std::lock_guard<std::mutex> lock(g_prof_mutex);
    g_prof_stream.store(nullptr);
    int64_t n = 0;
    for (auto& r : g_prof_records) {
        if (n < cap && hipEventSynchronize(r.e1) == hipSuccess) {
            float t = 0.f;
            if (hipEventElapsedTime(&t, r.e0, r.e1) == hipSuccess) {
                ms[n] = t, nv[n] = r.nv, nnzb[n] = r.nnzb, ncols[n] = r.ncols, first[n] = r.first;
                ++n;
            }
        }
        (void)hipEventDestroy(r.e0);
        (void)hipEventDestroy(r.e1);
    }
    g_prof_records.clear();
    return n;
}

extern "C" int ds_spmm_union(int epilogue, int level_tag, const int32_t* utab, const int32_t* ctab, int64_t ngroups, int cap_blocks,
                             const int32_t* gent,
                             const float* kgrp, int64_t nnzb, int64_t nv, const float* X, int64_t ldx, float* Y,
                             int64_t ldy, const float* R0, int64_t ldr, const float* dinv, int ncols, float c1, float c2,
                             int first, const float* Wprev, int64_t ldp, ds_stream_t stream) {
    DS_REQUIRE(ctab && gent && kgrp && X && Y, "ds_spmm_union: null pointer");
    DS_REQUIRE(epilogue >= 0 && epilogue <= 3, "ds_spmm_union: bad epilogue %d", epilogue);
    DS_REQUIRE(level_tag == 0 || level_tag == 1, "ds_spmm_union: level_tag must be 0 (fine) or 1 (corner-node level)");
    DS_REQUIRE(epilogue == 0 || epilogue == 3 || R0, "ds_spmm_union: the epilogue needs R0");
    DS_REQUIRE(epilogue != 1 || dinv, "ds_spmm_union: the Chebyshev epilogue needs dinv");
    DS_REQUIRE(nv > 0 && ngroups == (nv + 3) / 4 && nnzb > 0 && ncols > 0 && ncols % 4 == 0 && ncols <= 84,
               "ds_spmm_union: ncols must be a multiple of 4 <= 84 and ngroups = ceil(nv / 4)");
    DS_REQUIRE(cap_blocks > 0 && cap_blocks <= UN_CAPB, "ds_spmm_union: a chunk of %d blocks exceeds the LDS image (DS_UNION_CAP = 140)", cap_blocks);
    DS_REQUIRE(ldx >= ncols && ldy >= ncols && (epilogue == 0 || epilogue == 3 || ldr >= ncols),
               "ds_spmm_union: leading dimension smaller than ncols");
    DS_REQUIRE(X != Y, "ds_spmm_union: X and Y must be different buffers");
    DS_REQUIRE(ldx * 12 < (int64_t)PIPE_OOB && ldy * 12 < (int64_t)PIPE_OOB && ldr * 12 < (int64_t)PIPE_OOB &&
                   nv * 36 < (int64_t)PIPE_OOB,
               "ds_spmm_union: a 3-row panel of %lld bytes exceeds the descriptor range", (long long)(ldx * 12));
    DS_REQUIRE(nnzb * 36 < ((int64_t)1 << 32), "ds_spmm_union: value array exceeds the descriptor range");
    uintptr_t al = reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(Y) | (uintptr_t)(ldx * 4) |
                   (uintptr_t)(ldy * 4) | reinterpret_cast<uintptr_t>(kgrp) | reinterpret_cast<uintptr_t>(ctab);
    if (epilogue == 1 || epilogue == 2) al |= reinterpret_cast<uintptr_t>(R0) | (uintptr_t)(ldr * 4);
    DS_REQUIRE((al & 15) == 0, "ds_spmm_union: rows, kgrp and ctab must be 16-byte aligned");
    hipStream_t st = ds::as_stream(stream);
    ChebEpilogue epi{R0, ldr, dinv, c1, c2, first};
    if (Wprev && epilogue == 1) {
        DS_REQUIRE(ldp >= ncols && ldp * 12 < (int64_t)PIPE_OOB &&
                       ((reinterpret_cast<uintptr_t>(Wprev) | (uintptr_t)(ldp * 4)) & 15) == 0 && Wprev != X,
                   "ds_spmm_union: bad W_prev block");
        epi.wprev = Wprev, epi.ldp = ldp;
    }
    const int lpn = ncols / 4;
    // (kinds of the timing hook: the fused term, Y = K X, Y = M_s X; the residual epilogue 2 is not recorded)
    const int pkind = epilogue == 1 ? DS_PROF_TERM : (epilogue == 0 ? DS_PROF_KX : (epilogue == 3 ? DS_PROF_MX : -1));
    ds::ProfScope prof(pkind < 0 ? nullptr : stream, pkind < 0 ? 0 : pkind, nv, nnzb, ncols, (first ? 1 : 0) | (4 << 8));
#define DS_U(L, E) return launch_union<L, E>(utab, ctab, ngroups, cap_blocks, gent, kgrp, nnzb, nv, X, ldx, Y, ldy, lpn, st, epi)
#define DS_UL(L, E) do { if (level_tag == 1) return launch_union<L, E, false, true, 1>(utab, ctab, ngroups, cap_blocks, gent, kgrp, nnzb, nv, X, ldx, Y, ldy, lpn, st, epi); \
                         return launch_union<L, E, false, true, 0>(utab, ctab, ngroups, cap_blocks, gent, kgrp, nnzb, nv, X, ldx, Y, ldy, lpn, st, epi); } while (0)
    if (lpn == 20) {
        if (epilogue == 1) DS_U(20, 1);
        if (epilogue == 2) DS_U(20, 2);
        if (epilogue == 3) DS_UL(20, 3);
        DS_UL(20, 0);
    }
    if (epilogue == 1) DS_U(0, 1);
    if (epilogue == 2) DS_U(0, 2);
    if (epilogue == 3) DS_UL(0, 3);
    DS_UL(0, 0);
#undef DS_UL
#undef DS_U
}


// Narrow blocks (<= 16 columns): the lanes dealt over the union's entries (spmm_narrow.inc)
extern "C" int ds_spmm_union_narrow(int kind, int level_tag, const int32_t* utab, const int32_t* ctab, int64_t ngroups,
                                    const int32_t* gent, const float* vals, int64_t nnzb, int64_t nv, const float* X, int64_t ldx,
                                    float* Y, int64_t ldy, int ncols, ds_stream_t stream) {
    DS_REQUIRE(ctab && gent && vals && X && Y, "ds_spmm_union_narrow: null pointer");
    DS_REQUIRE(kind == 0 || kind == 3, "ds_spmm_union_narrow: kind must be 0 (3x3 blocks) or 3 (node-scalar values)");
    DS_REQUIRE(level_tag == 0 || level_tag == 1, "ds_spmm_union_narrow: level_tag must be 0 (fine) or 1 (corner-node level)");
    DS_REQUIRE(nv > 0 && ngroups == (nv + 3) / 4 && nnzb > 0 && ncols > 0 && ncols % 4 == 0 && ncols <= 16,
               "ds_spmm_union_narrow: ncols must be a multiple of 4 <= 16 and ngroups = ceil(nv / 4)");
    DS_REQUIRE(ldx >= ncols && ldy >= ncols, "ds_spmm_union_narrow: leading dimension smaller than ncols");
    DS_REQUIRE(X != Y, "ds_spmm_union_narrow: X and Y must be different buffers");
    const uintptr_t al = reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(Y) | (uintptr_t)(ldx * 4) | (uintptr_t)(ldy * 4) |
                         reinterpret_cast<uintptr_t>(ctab);
    DS_REQUIRE((al & 15) == 0, "ds_spmm_union_narrow: rows and ctab must be 16-byte aligned");
    hipStream_t st = ds::as_stream(stream);
    const unsigned nwg = (unsigned)ds::ceil_div(ngroups, 4);
    ds::ProfScope prof(stream, kind == 0 ? DS_PROF_KX : DS_PROF_MX, nv, nnzb, ncols, 4 << 8);
#define DS_UN(S, V) spmm_union_narrow_kernel<S, V><<<nwg, 256, 0, st>>>(reinterpret_cast<const int2*>(utab), reinterpret_cast<const int4*>(ctab), \
                                                                       (unsigned)ngroups, gent, vals, nv, X, ldx, Y, ldy, ncols, nwg)
    if (kind == 0) {
        if (level_tag == 1) DS_UN(false, 1);
        else DS_UN(false, 0);
    } else {
        if (level_tag == 1) DS_UN(true, 1);
        else DS_UN(true, 0);
    }
#undef DS_UN
    DS_LAUNCH_CHECK("spmm_union_narrow_kernel");
    return DS_OK;
}

// The eigensolver's residual in ONE walk of the unions (spmm_union.inc, epilogue 4): R = K X - (M_s (x) I3) X diag(lam) and
// the column norms ||R_j||^2, ||X_j||^2, with neither K X nor M X written to memory.
extern "C" int64_t ds_union_residual_workspace_bytes(int64_t ngroups, int ncols) {
    return ngroups * 2 * (int64_t)ncols * 4 + (int64_t)UN_NORM_BLOCKS * 2 * ncols * 8;
}

extern "C" int ds_union_residual(int level_tag, const int32_t* utab, const int32_t* ctab, int64_t ngroups, int cap_blocks,
                                 const int32_t* gent, const float* kgrp, const float* mgrp, int64_t nnzb, int64_t nv,
                                 const float* X, int64_t ldx, const double* lam, float* R, int64_t ldr, int ncols, void* work,
                                 int64_t work_bytes, double* rn2, double* xn2, ds_stream_t stream) {
    DS_REQUIRE(ctab && gent && kgrp && mgrp && X && lam && R && work && rn2 && xn2, "ds_union_residual: null pointer");
    DS_REQUIRE(level_tag == 0 || level_tag == 1, "ds_union_residual: level_tag must be 0 (fine) or 1 (corner-node level)");
    DS_REQUIRE(nv > 0 && ngroups == (nv + 3) / 4 && nnzb > 0 && ncols > 0 && ncols % 4 == 0 && ncols <= 84,
               "ds_union_residual: ncols must be a multiple of 4 <= 84 and ngroups = ceil(nv / 4)");
    DS_REQUIRE(cap_blocks > 0 && cap_blocks <= UN_CAPB, "ds_union_residual: a chunk of %d blocks exceeds the LDS image", cap_blocks);
    DS_REQUIRE(ldx >= ncols && ldr >= ncols, "ds_union_residual: leading dimension smaller than ncols");
    DS_REQUIRE(X != R, "ds_union_residual: X and R must be different buffers");
    DS_REQUIRE(ldx * 12 < (int64_t)PIPE_OOB && ldr * 12 < (int64_t)PIPE_OOB && nnzb * 36 < ((int64_t)1 << 32),
               "ds_union_residual: operand beyond the descriptor range");
    const uintptr_t al = reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(R) | (uintptr_t)(ldx * 4) | (uintptr_t)(ldr * 4) |
                         reinterpret_cast<uintptr_t>(kgrp) | reinterpret_cast<uintptr_t>(ctab) | reinterpret_cast<uintptr_t>(work);
    DS_REQUIRE((al & 15) == 0, "ds_union_residual: rows, kgrp, ctab and the workspace must be 16-byte aligned");
    DS_REQUIRE(work_bytes >= ds_union_residual_workspace_bytes(ngroups, ncols), "ds_union_residual: workspace of %lld bytes needed",
               (long long)ds_union_residual_workspace_bytes(ngroups, ncols));
    hipStream_t st = ds::as_stream(stream);
    ChebEpilogue epi{nullptr, 0, nullptr, 0.f, 0.f, 0};
    epi.mvals = mgrp, epi.lam = lam, epi.nwork = static_cast<float*>(work);
    const int lpn = ncols / 4;
    int rc;
    {
    ds::ProfScope prof(stream, DS_PROF_RESID, nv, nnzb, ncols, 4 << 8);  // (the walk of the unions; the two norm reductions follow)
#define DS_UR(L, V) rc = launch_union<L, 4, false, true, V>(utab, ctab, ngroups, cap_blocks, gent, kgrp, nnzb, nv, X, ldx, R, ldr, lpn, st, epi)
    if (lpn == 20) {
        if (level_tag == 1) DS_UR(20, 1);
        else DS_UR(20, 0);
    } else {
        if (level_tag == 1) DS_UR(0, 1);
        else DS_UR(0, 0);
    }
#undef DS_UR
    }
    if (rc != DS_OK) return rc;
    double* partial = reinterpret_cast<double*>(static_cast<char*>(work) + ngroups * 2 * (int64_t)ncols * 4);
    union_norm_partial_kernel<<<UN_NORM_BLOCKS, 256, 0, st>>>(static_cast<const float*>(work), (unsigned)ngroups, 2 * ncols, partial);
    DS_LAUNCH_CHECK("union_norm_partial_kernel");
    union_norm_final_kernel<<<(unsigned)(2 * ncols), 256, 0, st>>>(partial, ncols, rn2, xn2);
    DS_LAUNCH_CHECK("union_norm_final_kernel");
    return DS_OK;
}

// Y = K X and Y2 = (M_s (x) I3) X of ONE block in one walk of the unions (spmm_union.inc, epilogue 5): the eigensolver's
// products of the raw preconditioned residuals W (round 5: Rayleigh-Ritz on the raw basis, csrc/lobpcg.cpp).
extern "C" int ds_spmm_union_km(int level_tag, const int32_t* utab, const int32_t* ctab, int64_t ngroups, int cap_blocks,
                                const int32_t* gent, const float* kgrp, const float* mgrp, int64_t nnzb, int64_t nv,
                                const float* X, int64_t ldx, float* KX, int64_t ldk, float* MX, int64_t ldm, int ncols,
                                ds_stream_t stream) {
    DS_REQUIRE(ctab && gent && kgrp && mgrp && X && KX && MX, "ds_spmm_union_km: null pointer");
    DS_REQUIRE(level_tag == 0 || level_tag == 1, "ds_spmm_union_km: level_tag must be 0 (fine) or 1 (corner-node level)");
    DS_REQUIRE(nv > 0 && ngroups == (nv + 3) / 4 && nnzb > 0 && ncols > 0 && ncols % 4 == 0 && ncols <= 84,
               "ds_spmm_union_km: ncols must be a multiple of 4 <= 84 and ngroups = ceil(nv / 4)");
    DS_REQUIRE(cap_blocks > 0 && cap_blocks <= UN_CAPB, "ds_spmm_union_km: a chunk of %d blocks exceeds the LDS image", cap_blocks);
    DS_REQUIRE(ldx >= ncols && ldk >= ncols && ldm >= ncols, "ds_spmm_union_km: leading dimension smaller than ncols");
    DS_REQUIRE(X != KX && X != MX && KX != MX, "ds_spmm_union_km: X, KX and MX must be different buffers");
    DS_REQUIRE(ldx * 12 < (int64_t)PIPE_OOB && nnzb * 36 < ((int64_t)1 << 32), "ds_spmm_union_km: operand beyond the descriptor range");
    const uintptr_t al = reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(KX) | reinterpret_cast<uintptr_t>(MX) |
                         (uintptr_t)(ldx * 4) | (uintptr_t)(ldk * 4) | (uintptr_t)(ldm * 4) | reinterpret_cast<uintptr_t>(kgrp) |
                         reinterpret_cast<uintptr_t>(ctab);
    DS_REQUIRE((al & 15) == 0, "ds_spmm_union_km: rows, kgrp and ctab must be 16-byte aligned");
    hipStream_t st = ds::as_stream(stream);
    ChebEpilogue epi{nullptr, ldm, nullptr, 0.f, 0.f, 0};
    epi.mvals = mgrp, epi.nwork = MX;
    const int lpn = ncols / 4;
    ds::ProfScope prof(stream, DS_PROF_KM, nv, nnzb, ncols, 4 << 8);
#define DS_UKM(L, V) return launch_union<L, 5, false, true, V>(utab, ctab, ngroups, cap_blocks, gent, kgrp, nnzb, nv, X, ldx, KX, ldk, lpn, st, epi)
    if (lpn == 20) {
        if (level_tag == 1) DS_UKM(20, 1);
        DS_UKM(20, 0);
    }
    if (level_tag == 1) DS_UKM(0, 1);
    DS_UKM(0, 0);
#undef DS_UKM
}

// bf16-block form of the two preconditioner epilogues (the V-cycle's iterates are stored in bf16, arithmetic in fp32):
// X, R0, W_prev bf16; Y bf16, or fp32 when y_f32 (the last term of a cycle, written into the solver's basis buffer).
extern "C" int ds_spmm_union16(int epilogue, const int32_t* utab, const int32_t* ctab, int64_t ngroups, int cap_blocks,
                               const int32_t* gent, const float* kgrp, int64_t nnzb, int64_t nv, const void* X,
                               int64_t ldx, void* Y, int64_t ldy, int y_f32, const void* R0, int64_t ldr,
                               const float* dinv, int ncols, float c1, float c2, int first, const void* Wprev,
                               int64_t ldp, ds_stream_t stream) {
    DS_REQUIRE(ctab && gent && kgrp && X && Y && R0, "ds_spmm_union16: null pointer");
    DS_REQUIRE(epilogue == 1 || epilogue == 2, "ds_spmm_union16: epilogue must be 1 (Chebyshev term) or 2 (residual)");
    DS_REQUIRE(epilogue != 1 || dinv, "ds_spmm_union16: the Chebyshev epilogue needs dinv");
    DS_REQUIRE(nv > 0 && ngroups == (nv + 3) / 4 && nnzb > 0 && ncols > 0 && ncols % 4 == 0 && ncols <= 84,
               "ds_spmm_union16: ncols must be a multiple of 4 <= 84 and ngroups = ceil(nv / 4)");
    DS_REQUIRE(cap_blocks > 0 && cap_blocks <= UN_CAPB, "ds_spmm_union16: a chunk of %d blocks exceeds the LDS image", cap_blocks);
    DS_REQUIRE(ldx >= ncols && ldy >= ncols && ldr >= ncols, "ds_spmm_union16: leading dimension smaller than ncols");
    DS_REQUIRE(X != Y, "ds_spmm_union16: X and Y must be different buffers");
    // (the bf16 inputs go through buffer descriptors; the result is stored through plain 64-bit addresses)
    DS_REQUIRE(3 * nv * std::max(std::max(ldx, ldr), ldp) * 2 < (int64_t)PIPE_OOB && nv * 36 < (int64_t)PIPE_OOB,
               "ds_spmm_union16: a bf16 operand block of %lld bytes exceeds the descriptor range",
               (long long)(3 * nv * std::max(std::max(ldx, ldr), ldp) * 2));
    DS_REQUIRE(nnzb * 36 < ((int64_t)1 << 32), "ds_spmm_union16: value array exceeds the descriptor range");
    uintptr_t al = reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(R0) | (uintptr_t)(ldx * 2) | (uintptr_t)(ldr * 2);
    al |= reinterpret_cast<uintptr_t>(Y) | (uintptr_t)(ldy * (y_f32 ? 4 : 2));
    DS_REQUIRE((al & 7) == 0 && (!y_f32 || ((reinterpret_cast<uintptr_t>(Y) | (uintptr_t)(ldy * 4)) & 15) == 0) &&
                   ((reinterpret_cast<uintptr_t>(kgrp) | reinterpret_cast<uintptr_t>(ctab)) & 15) == 0,
               "ds_spmm_union16: bf16 rows must be 8-byte aligned (fp32 output rows, kgrp and ctab 16-byte)");
    hipStream_t st = ds::as_stream(stream);
    ChebEpilogue epi{static_cast<const float*>(R0), ldr, dinv, c1, c2, first};
    if (Wprev && epilogue == 1) {
        DS_REQUIRE(ldp >= ncols && ((reinterpret_cast<uintptr_t>(Wprev) | (uintptr_t)(ldp * 2)) & 7) == 0 && Wprev != X,
                   "ds_spmm_union16: bad W_prev block");
        epi.wprev = static_cast<const float*>(Wprev), epi.ldp = ldp;
    }
    DS_REQUIRE(!y_f32 || epilogue == 1, "ds_spmm_union16: an fp32 result is the Chebyshev term's only");
    const float* Xf = static_cast<const float*>(X);
    float* Yf = static_cast<float*>(Y);
    const int lpn = ncols / 4;
    // (the bf16-in / bf16-out term: the launch the benchmark's roofline figure is about)
    ds::ProfScope prof((epilogue == 1 && !y_f32) ? stream : nullptr, DS_PROF_TERM, nv, nnzb, ncols, (first ? 1 : 0) | (2 << 8));
#define DS_U16(L, E, O) return launch_union<L, E, true, O>(utab, ctab, ngroups, cap_blocks, gent, kgrp, nnzb, nv, Xf, ldx, Yf, ldy, lpn, st, epi)
    if (lpn == 20) {
        if (epilogue == 2) DS_U16(20, 2, false);
        if (y_f32) DS_U16(20, 1, true);
        DS_U16(20, 1, false);
    }
    if (epilogue == 2) DS_U16(0, 2, false);
    if (y_f32) DS_U16(0, 1, true);
    DS_U16(0, 1, false);
#undef DS_U16
}

// ------------------------------------------------------------------------------------------------
// MFMA form of the bf16 preconditioner terms (spmm_mfma.inc)
extern "C" int ds_pack_kc(const float* k32, const int32_t* kperm, int64_t nnzb, void* kc, ds_stream_t stream) {
    DS_REQUIRE(k32 && kperm && kc && nnzb > 0, "ds_pack_kc: bad argument");
    DS_REQUIRE((reinterpret_cast<uintptr_t>(kc) & 7) == 0, "ds_pack_kc: kc must be 8-byte aligned");
    pack_kc_kernel<<<(unsigned)ds::ceil_div(nnzb * 3, 256), 256, 0, ds::as_stream(stream)>>>(k32, kperm, nnzb,
                                                                                            static_cast<i2s*>(kc));
    DS_LAUNCH_CHECK("pack_kc_kernel");
    return DS_OK;
}

template <int G, int NT, int BATCH, int LVL, int TAIL>
static int launch_mfma(int epilogue, int y_f32, const int32_t* gptr, const int32_t* gcol, const int32_t* gmeta,
                       const int32_t* gbase, const int32_t* ghead, const void* kc, int64_t nnzb, int64_t ngroups, int64_t nv, const float* X,
                       int64_t ldx, float* Y, int64_t ldy, int lpn, int acap, hipStream_t st, const ChebEpilogue& epi) {
    const char* kcp = static_cast<const char*>(kc);
    const size_t lds = (size_t)mf_panel_bytes(lpn * 4, G, BATCH, TAIL) + (size_t)acap + 32;
    if (epilogue == 2)
        spmm_union_mfma_kernel<G, NT, 2, false, BATCH, LVL, TAIL><<<(unsigned)ngroups, 64, lds, st>>>(gptr, gcol, gmeta, gbase, ghead, kcp, nnzb, (unsigned)ngroups, nv, X, ldx, Y, ldy, lpn, acap, epi);
    else if (y_f32)
        spmm_union_mfma_kernel<G, NT, 1, true, BATCH, LVL, TAIL><<<(unsigned)ngroups, 64, lds, st>>>(gptr, gcol, gmeta, gbase, ghead, kcp, nnzb, (unsigned)ngroups, nv, X, ldx, Y, ldy, lpn, acap, epi);
    else
        spmm_union_mfma_kernel<G, NT, 1, false, BATCH, LVL, TAIL><<<(unsigned)ngroups, 64, lds, st>>>(gptr, gcol, gmeta, gbase, ghead, kcp, nnzb, (unsigned)ngroups, nv, X, ldx, Y, ldy, lpn, acap, epi);
    DS_LAUNCH_CHECK("spmm_union_mfma_kernel");
    return DS_OK;
}

// entries per batch on the corner-node level (level_tag 1): DS_MF_BATCH, or twice that (one wave's chain of dependent round
// trips is what a level with about as many groups as the device has wave slots is made of; measured at C3's corner level,
// 2 461 groups: see profiles/r04_corner_batch_ab.txt)
#ifndef DS_MF_CORNER_BATCH
#define DS_MF_CORNER_BATCH DS_MF_BATCH
#endif

extern "C" int ds_spmm_union16m(int epilogue, int group_nodes, int level_tag, const int32_t* gptr, const int32_t* gcol,
                                const int32_t* gmeta, const int32_t* gbase, const int32_t* ghead, const void* kc, int64_t nnzb,
                                int64_t ngroups, int max_entries, int max_batch_blocks, int64_t nv, const void* X,
                                int64_t ldx, void* Y,
                                int64_t ldy, int y_f32, const void* R0, int64_t ldr, const float* dinv, int ncols,
                                float c1, float c2, int first, const void* Wprev, int64_t ldp, ds_stream_t stream) {
    DS_REQUIRE(gptr && gcol && gmeta && gbase && ghead && kc && X && Y && R0, "ds_spmm_union16m: null pointer");
    DS_REQUIRE((reinterpret_cast<uintptr_t>(ghead) & 3) == 0 && ngroups * 512 < (int64_t)1 << 40, "ds_spmm_union16m: bad ghead");
    DS_REQUIRE(epilogue == 1 || epilogue == 2, "ds_spmm_union16m: epilogue must be 1 (Chebyshev term) or 2 (residual)");
    DS_REQUIRE(epilogue != 1 || dinv, "ds_spmm_union16m: the Chebyshev epilogue needs dinv");
    // (4-node groups were slower on both levels; only the 8-node kernel is built)
    DS_REQUIRE(group_nodes == 8, "ds_spmm_union16m: groups of 8 nodes");
    DS_REQUIRE(level_tag == 0 || level_tag == 1, "ds_spmm_union16m: level_tag must be 0 (fine) or 1 (corner-node level)");
    DS_REQUIRE(nv > 0 && ngroups == (nv + group_nodes - 1) / group_nodes && ncols > 0 && ncols % 4 == 0 && ncols <= 84,
               "ds_spmm_union16m: ncols must be a multiple of 4 <= 84 and ngroups = ceil(nv / group_nodes)");
    DS_REQUIRE(max_entries > 0 && max_entries <= 256, "ds_spmm_union16m: a group with %d union entries exceeds 256", max_entries);
    DS_REQUIRE(max_batch_blocks > 0 && max_batch_blocks <= DS_MF_BATCH * group_nodes,
               "ds_spmm_union16m: max_batch_blocks must be in (0, DS_MF_BATCH x group_nodes]");
    DS_REQUIRE(nnzb > 0 && nnzb * 24 < (int64_t)PIPE_OOB, "ds_spmm_union16m: the block array exceeds the descriptor range");
    DS_REQUIRE(ldx >= ncols && ldy >= ncols && ldr >= ncols, "ds_spmm_union16m: leading dimension smaller than ncols");
    DS_REQUIRE(X != Y, "ds_spmm_union16m: X and Y must be different buffers");
    DS_REQUIRE(3 * nv * std::max(std::max(ldx, ldr), ldp) * 2 < (int64_t)PIPE_OOB && nv * 36 < (int64_t)PIPE_OOB,
               "ds_spmm_union16m: a bf16 operand block exceeds the descriptor range");
    uintptr_t al = reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(R0) | (uintptr_t)(ldx * 2) | (uintptr_t)(ldr * 2);
    al |= reinterpret_cast<uintptr_t>(Y) | (uintptr_t)(ldy * (y_f32 ? 4 : 2)) | reinterpret_cast<uintptr_t>(kc);
    DS_REQUIRE((al & 7) == 0 && (!y_f32 || ((reinterpret_cast<uintptr_t>(Y) | (uintptr_t)(ldy * 4)) & 15) == 0),
               "ds_spmm_union16m: bf16 rows and kc must be 8-byte aligned (fp32 output rows 16-byte)");
    DS_REQUIRE(!y_f32 || epilogue == 1, "ds_spmm_union16m: an fp32 result is the Chebyshev term's only");
    hipStream_t st = ds::as_stream(stream);
    ChebEpilogue epi{static_cast<const float*>(R0), ldr, dinv, c1, c2, first};
    if (Wprev && epilogue == 1) {
        DS_REQUIRE(ldp >= ncols && ((reinterpret_cast<uintptr_t>(Wprev) | (uintptr_t)(ldp * 2)) & 7) == 0 && Wprev != X,
                   "ds_spmm_union16m: bad W_prev block");
        epi.wprev = static_cast<const float*>(Wprev), epi.ldp = ldp;
    }
    const float* Xf = static_cast<const float*>(X);
    float* Yf = static_cast<float*>(Y);
    const int lpn = ncols / 4;
    const int nt = (ncols + 15) / 16;
    // two batches of DS_MF_BATCH entries never hold more blocks than 2 x max_batch_blocks
    constexpr int CB = DS_MF_CORNER_BATCH;
    const int acap = ((max_batch_blocks * 24 + 1023) / 1024) * 1024;  // whole 1 KiB staging pieces
    const int acapc = (((CB / DS_MF_BATCH) * max_batch_blocks * 24 + 1023) / 1024) * 1024;
    // groups of at most 128 entries: the form whose last batch takes DS_MF_TAIL entries more (see spmm_mfma.inc)
    const bool tail = max_entries <= 128 && max_batch_blocks * 24 <= 2048;
    constexpr int CT = CB == DS_MF_BATCH ? DS_MF_TAIL : 0;
#define DS_MF_ARGS(ACAP) epilogue, y_f32, gptr, gcol, gmeta, gbase, ghead, kc, nnzb, ngroups, nv, Xf, ldx, Yf, ldy, lpn, ACAP, st, epi
    // (blocks of >= 49 columns only: with fewer accumulator tiles the compiler spills the tail form at three waves per SIMD)
#define DS_MF_GO(N) return tail ? launch_mfma<8, N, DS_MF_BATCH, 0, (N >= 4 ? DS_MF_TAIL : 0)>(DS_MF_ARGS(acap)) : launch_mfma<8, N, DS_MF_BATCH, 0, 0>(DS_MF_ARGS(acap))
#define DS_MF_GOC(N) return tail ? launch_mfma<8, N, CB, 1, (N >= 4 ? CT : 0)>(DS_MF_ARGS(acapc)) : launch_mfma<8, N, CB, 1, 0>(DS_MF_ARGS(acapc))
    auto go = [&]() -> int {
        if (level_tag == 1) {
            switch (nt) {
                case 1: DS_MF_GOC(1); case 2: DS_MF_GOC(2); case 3: DS_MF_GOC(3);
                case 4: DS_MF_GOC(4); case 5: DS_MF_GOC(5); default: DS_MF_GOC(6);
            }
        }
        switch (nt) {
            case 1: DS_MF_GO(1); case 2: DS_MF_GO(2); case 3: DS_MF_GO(3);
            case 4: DS_MF_GO(4); case 5: DS_MF_GO(5); default: DS_MF_GO(6);
        }
    };
#undef DS_MF_GO
#undef DS_MF_GOC
#undef DS_MF_ARGS
    ds::ProfScope prof((epilogue == 1 && !y_f32) ? stream : nullptr, DS_PROF_TERM, nv, nnzb, ncols, (first ? 1 : 0) | (2 << 8));
    return go();
}

// ------------------------------------------------------------------------------------------------
// fp32 matrix-core form of the eigensolver's own products (spmm_mfma32.inc)
#ifndef DS_MF32_BATCH
#define DS_MF32_BATCH 8
#endif
template <int NT, int EPI, int LVL>
static int launch_mfma32(const int32_t* gptr, const int32_t* gcol, const int32_t* gmeta, const int32_t* gbase, const float* vals,
                         unsigned vals_bytes, int64_t ngroups, int64_t nv, const float* X, int64_t ldx, float* Y, int64_t ldy,
                         int lpn, hipStream_t st) {
    constexpr int EB = DS_MF32_BATCH;
    constexpr int acap = ((EB * MF32_G * (EPI == 3 ? 4 : 36) + 1023) / 1024) * 1024;  // whole 1 KiB staging pieces
    const size_t lds = (size_t)mf32_panel_bytes(NT, EB) + acap + 48;
    spmm_union_mfma32_kernel<NT, EPI, EB, LVL><<<(unsigned)ngroups, 64, lds, st>>>(gptr, gcol, gmeta, gbase, vals, vals_bytes,
                                                                                  (unsigned)ngroups, nv, X, ldx, Y, ldy, lpn, acap);
    DS_LAUNCH_CHECK("spmm_union_mfma32_kernel");
    return DS_OK;
}

extern "C" int ds_spmm_union32m(int epilogue, int level_tag, const int32_t* gptr, const int32_t* gcol, const int32_t* gmeta,
                                const int32_t* gbase, const float* vals, int64_t vals_bytes, int64_t nblocks, int64_t ngroups,
                                int max_entries, int max_batch_blocks, int64_t nv, const float* X, int64_t ldx, float* Y,
                                int64_t ldy, int ncols, ds_stream_t stream) {
    DS_REQUIRE(gptr && gcol && gmeta && gbase && vals && X && Y, "ds_spmm_union32m: null pointer");
    DS_REQUIRE(epilogue == 0 || epilogue == 3, "ds_spmm_union32m: epilogue must be 0 (3x3 blocks) or 3 (node-scalar values)");
    DS_REQUIRE(level_tag == 0 || level_tag == 1, "ds_spmm_union32m: level_tag must be 0 (fine) or 1 (corner-node level)");
    DS_REQUIRE(nv > 0 && ngroups == (nv + MF32_G - 1) / MF32_G && ncols > 0 && ncols % 4 == 0 && ncols <= 84,
               "ds_spmm_union32m: ncols must be a multiple of 4 <= 84 and ngroups = ceil(nv / 4)");
    DS_REQUIRE(max_entries > 0 && max_entries <= 256, "ds_spmm_union32m: a group with %d union entries exceeds 256", max_entries);
    DS_REQUIRE(max_batch_blocks > 0 && max_batch_blocks <= DS_MF32_BATCH * MF32_G,
               "ds_spmm_union32m: max_batch_blocks must be in (0, DS_MF32_BATCH x 4]");
    const int64_t vb = epilogue == 3 ? 4 : 36;
    // (the value stream is read in 16-byte pieces: the array carries 16 bytes of slack behind its last block)
    DS_REQUIRE(nblocks > 0 && vals_bytes >= nblocks * vb + 16 && vals_bytes < (int64_t)PIPE_OOB,
               "ds_spmm_union32m: the value array must hold nblocks x %d bytes + 16 of slack, under the descriptor range", (int)vb);
    DS_REQUIRE(ldx >= ncols && ldy >= ncols, "ds_spmm_union32m: leading dimension smaller than ncols");
    DS_REQUIRE(X != Y, "ds_spmm_union32m: X and Y must be different buffers");
    DS_REQUIRE(3 * nv * ldx * 4 < (int64_t)PIPE_OOB, "ds_spmm_union32m: the operand block exceeds the descriptor range");
    const uintptr_t al = reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(Y) | (uintptr_t)(ldx * 4) | (uintptr_t)(ldy * 4);
    DS_REQUIRE((al & 15) == 0 && (reinterpret_cast<uintptr_t>(vals) & 3) == 0, "ds_spmm_union32m: rows must be 16-byte aligned");
    hipStream_t st = ds::as_stream(stream);
    const int lpn = ncols / 4, nt = (ncols + 15) / 16;
#define DS_M32_GO(N, E, L) return launch_mfma32<N, E, L>(gptr, gcol, gmeta, gbase, vals, (unsigned)vals_bytes, ngroups, nv, X, ldx, Y, ldy, lpn, st)
#define DS_M32_NT(E, L)                                                                     \
    switch (nt) {                                                                           \
        case 1: DS_M32_GO(1, E, L); case 2: DS_M32_GO(2, E, L); case 3: DS_M32_GO(3, E, L); \
        case 4: DS_M32_GO(4, E, L); case 5: DS_M32_GO(5, E, L); default: DS_M32_GO(6, E, L); \
    }
    if (epilogue == 0) {
        if (level_tag == 0) DS_M32_NT(0, 0)
        DS_M32_NT(0, 1)
    }
    if (level_tag == 0) DS_M32_NT(3, 0)
    DS_M32_NT(3, 1)
#undef DS_M32_NT
#undef DS_M32_GO
}

#ifdef DS_DIAG
extern "C" int ds_m32_diag(unsigned long long* out, int nwaves) {  // diagnostic build only: out[nwaves][8]
    return ds::check_hip(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_m32_dbg), (size_t)nwaves * 8 * sizeof(unsigned long long)), "ds_m32_diag read");
}
extern "C" int ds_mf_diag(unsigned long long* out, int nwaves) {  // the bf16 term kernel's records
    return ds::check_hip(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mf_dbg), (size_t)nwaves * 8 * sizeof(unsigned long long)), "ds_mf_diag read");
}
#endif

// fp64 values (and fp64 or fp32 vectors) on the neighbour-union tables (spmm_f64_union.inc): the fp64 refinement's K W / M W
extern "C" int ds_spmm_f64_union(int kind, int x_f64, const int32_t* utab, const int32_t* ctab, int64_t ngroups, int cap_blocks,
                                 const int32_t* gent, const double* vals, int64_t nnzb, int64_t nv, const void* X, int64_t ldx,
                                 double* Y, int64_t ldy, int ncols, ds_stream_t stream) {
    DS_REQUIRE(ctab && gent && vals && X && Y, "ds_spmm_f64_union: null pointer");
    DS_REQUIRE(kind == 0 || kind == 1, "ds_spmm_f64_union: kind must be 0 (3x3 blocks in group order, transposed) or 1 (node scalars)");
    DS_REQUIRE(nv > 0 && ngroups == (nv + 3) / 4 && nnzb > 0 && ncols > 0 && ncols % 4 == 0 && ncols <= 84,
               "ds_spmm_f64_union: ncols must be a multiple of 4 <= 84 and ngroups = ceil(nv / 4)");
    DS_REQUIRE(cap_blocks > 0 && cap_blocks <= DS_UNION_CAP, "ds_spmm_f64_union: a chunk of %d blocks exceeds the LDS image", cap_blocks);
    DS_REQUIRE(ldx >= ncols && ldy >= ncols, "ds_spmm_f64_union: leading dimension smaller than ncols");
    DS_REQUIRE(static_cast<const void*>(Y) != X, "ds_spmm_f64_union: X and Y must be different buffers");
    const int xb = x_f64 ? 8 : 4;
    DS_REQUIRE(((reinterpret_cast<uintptr_t>(X) | (uintptr_t)(ldx * xb) | reinterpret_cast<uintptr_t>(Y) | (uintptr_t)(ldy * 8) |
                 reinterpret_cast<uintptr_t>(ctab)) & 15) == 0, "ds_spmm_f64_union: rows and ctab must be 16-byte aligned");
    hipStream_t st = ds::as_stream(stream);
    const unsigned nwg = (unsigned)ds::ceil_div(ngroups, 4);
    const int lpn = ncols / 4;
    const int2* ut = reinterpret_cast<const int2*>(utab);
    const int4* ct = reinterpret_cast<const int4*>(ctab);
#define DS_F64U(K, T) spmm_f64_union_kernel<K, T><<<nwg, 256, 0, st>>>(ut, ct, (unsigned)ngroups, gent, vals, nv, static_cast<const T*>(X), ldx, Y, ldy, lpn, nwg)
    if (kind == 0) {
        if (x_f64) DS_F64U(0, double);
        else DS_F64U(0, float);
    } else {
        if (x_f64) DS_F64U(1, double);
        else DS_F64U(1, float);
    }
#undef DS_F64U
    DS_LAUNCH_CHECK("spmm_f64_union_kernel");
    return DS_OK;
}

// Ya = A X, Yb = B X (fp64 3x3 block values), Ym = (m (x) I3) X (fp64 node scalars) for one fp32 block X in one walk.
extern "C" int ds_spmm_f64_polish(const int32_t* rowptr, const int32_t* colidx, const double* a, const double* b,
                                  const double* m, int64_t nv, const float* X, int64_t ldx, double* Ya, double* Yb,
                                  double* Ym, int64_t ldy, int ncols, ds_stream_t stream) {
    DS_REQUIRE(rowptr && colidx && a && b && m && X && Ya && Yb && Ym, "ds_spmm_f64_polish: null pointer");
    DS_REQUIRE(nv > 0 && ncols > 0 && ncols % 4 == 0 && ncols <= 84, "ds_spmm_f64_polish: ncols must be a multiple of 4 <= 84");
    DS_REQUIRE(ldx >= ncols && ldy >= ncols, "ds_spmm_f64_polish: leading dimension smaller than ncols");
    const uintptr_t ya = reinterpret_cast<uintptr_t>(Ya) | reinterpret_cast<uintptr_t>(Yb) | reinterpret_cast<uintptr_t>(Ym) |
                         (uintptr_t)(ldy * 8);
    DS_REQUIRE(((reinterpret_cast<uintptr_t>(X) | (uintptr_t)(ldx * 4) | ya) & 15) == 0,
               "ds_spmm_f64_polish: rows must be 16-byte aligned");
    DS_REQUIRE(Ya != Yb && Ya != Ym && Yb != Ym, "ds_spmm_f64_polish: the three results must be different buffers");
    const unsigned nblk = (unsigned)ds::ceil_div(nv, 4);
    spmm_f64_polish_kernel<double><<<nblk, 256, 0, ds::as_stream(stream)>>>(rowptr, colidx, a, b, m, nv, X, ldx, Ya, Yb, Ym, ldy,
                                                                            ncols / 4, nblk);
    DS_LAUNCH_CHECK("spmm_f64_polish_kernel");
    return DS_OK;
}

// the same products, stored as fp32 blocks (sums in fp64, one rounding)
extern "C" int ds_spmm_f64_polish_f32out(const int32_t* rowptr, const int32_t* colidx, const double* a, const double* b,
                                         const double* m, int64_t nv, const float* X, int64_t ldx, float* Ya, float* Yb,
                                         float* Ym, int64_t ldy, int ncols, ds_stream_t stream) {
    DS_REQUIRE(rowptr && colidx && a && b && m && X && Ya && Yb && Ym, "ds_spmm_f64_polish_f32out: null pointer");
    DS_REQUIRE(nv > 0 && ncols > 0 && ncols % 4 == 0 && ncols <= 84, "ds_spmm_f64_polish_f32out: ncols must be a multiple of 4 <= 84");
    DS_REQUIRE(ldx >= ncols && ldy >= ncols, "ds_spmm_f64_polish_f32out: leading dimension smaller than ncols");
    const uintptr_t ya = reinterpret_cast<uintptr_t>(Ya) | reinterpret_cast<uintptr_t>(Yb) | reinterpret_cast<uintptr_t>(Ym) |
                         (uintptr_t)(ldy * 4);
    DS_REQUIRE(((reinterpret_cast<uintptr_t>(X) | (uintptr_t)(ldx * 4) | ya) & 15) == 0,
               "ds_spmm_f64_polish_f32out: rows must be 16-byte aligned");
    DS_REQUIRE(Ya != Yb && Ya != Ym && Yb != Ym && Ya != X && Yb != X && Ym != X,
               "ds_spmm_f64_polish_f32out: X and the three results must be different buffers");
    const unsigned nblk = (unsigned)ds::ceil_div(nv, 4);
    spmm_f64_polish_kernel<float><<<nblk, 256, 0, ds::as_stream(stream)>>>(rowptr, colidx, a, b, m, nv, X, ldx, Ya, Yb, Ym, ldy,
                                                                           ncols / 4, nblk);
    DS_LAUNCH_CHECK("spmm_f64_polish_kernel<float>");
    return DS_OK;
}
