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
#include "ds_common.h"

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

}  // namespace

extern "C" int ds_spmm_bsr3(int kind, const int32_t* rowptr, const int32_t* colidx, const void* vals, int64_t nv,
                            const void* X, int64_t ldx, void* Y, int64_t ldy, int ncols, ds_stream_t stream) {
    DS_REQUIRE(rowptr && colidx && vals && X && Y, "ds_spmm_bsr3: null pointer");
    DS_REQUIRE(nv > 0 && ncols > 0, "ds_spmm_bsr3: empty problem");
    DS_REQUIRE(kind >= 0 && kind <= 3, "ds_spmm_bsr3: unknown kind %d", kind);
    DS_REQUIRE(ldx >= ncols && ldy >= ncols, "ds_spmm_bsr3: leading dimension smaller than ncols");
    hipStream_t st = ds::as_stream(stream);
    const bool f64out = kind >= 2;
    const int xalign = (int)(reinterpret_cast<uintptr_t>(X) | (uintptr_t)(ldx * 4));
    const int yalign = (int)(reinterpret_cast<uintptr_t>(Y) | (uintptr_t)(ldy * (f64out ? 8 : 4)));
    if (!f64out) {
        // float4 path needs 16-byte aligned rows; float2 path 8-byte
        if (ncols % 4 == 0 && ncols <= 256 && (xalign & 15) == 0 && (yalign & 15) == 0) {
            return kind == 0 ? launch<0, float, float, float, 4>(rowptr, colidx, vals, nv, X, ldx, Y, ldy, ncols, st)
                             : launch<1, float, float, float, 4>(rowptr, colidx, vals, nv, X, ldx, Y, ldy, ncols, st);
        }
        DS_REQUIRE(ncols % 2 == 0 && ncols <= 128 && (xalign & 7) == 0 && (yalign & 7) == 0,
                   "ds_spmm_bsr3: f32 blocks need an even column count <= 256 (multiple of 4 above 128) and "
                   "8/16-byte aligned rows (ncols=%d ldx=%lld ldy=%lld)",
                   ncols, (long long)ldx, (long long)ldy);
        return kind == 0 ? launch<0, float, float, float, 2>(rowptr, colidx, vals, nv, X, ldx, Y, ldy, ncols, st)
                         : launch<1, float, float, float, 2>(rowptr, colidx, vals, nv, X, ldx, Y, ldy, ncols, st);
    }
    DS_REQUIRE(ncols % 2 == 0 && ncols <= 128 && (xalign & 7) == 0 && (yalign & 15) == 0,
               "ds_spmm_bsr3: f64-output blocks need an even column count <= 128 and aligned rows (ncols=%d)", ncols);
    return kind == 2 ? launch<0, double, float, double, 2>(rowptr, colidx, vals, nv, X, ldx, Y, ldy, ncols, st)
                     : launch<1, double, float, double, 2>(rowptr, colidx, vals, nv, X, ldx, Y, ldy, ncols, st);
}
