// bf16-storage pieces of the two-level preconditioner - gfx950.  The V-cycle (csrc/vcycle.cpp) keeps its iterates,
// residuals and corner-level blocks in bf16 (fp32 arithmetic in registers): the outer iteration count of the
// eigensolver does not notice an 8-bit mantissa in its PRECONDITIONER, and every term of the cycle is bound by the
// bytes of its vector streams.  The fused SpMM terms are ds_spmm_union16 (spmm.hip); here are the two streaming steps:
//   ds_cheb_init16        W1 = c T R (block-Jacobi, first Chebyshev iterate) written as bf16 - R fp32 (the solver's
//                         residual block: a bf16 copy of it is written alongside, every later term reads that) or bf16
//                         (corner level);
//   ds_scalar_csr_spmm16  the level transfer Y_i = beta Y_i + sum_k w_k X_col_k on bf16 node panels.
// (reference: the preconditioner of the reference's LOBPCG is an opaque callable, src/lobpcg/_lobpcg.py:441.)
#include "ds_common.h"

namespace {

using f4 = __attribute__((ext_vector_type(4))) float;
using i2 = __attribute__((ext_vector_type(2))) int;

__device__ __forceinline__ f4 unpack4(i2 w) {
    f4 r;
    r.x = __builtin_bit_cast(float, w.x << 16);
    r.y = __builtin_bit_cast(float, w.x & (int)0xffff0000);
    r.z = __builtin_bit_cast(float, w.y << 16);
    r.w = __builtin_bit_cast(float, w.y & (int)0xffff0000);
    return r;
}
__device__ __forceinline__ i2 pack4(f4 v) {  // round to nearest even
    using bf2 = __attribute__((ext_vector_type(2))) __bf16;
    using f2 = __attribute__((ext_vector_type(2))) float;
    const bf2 lo = __builtin_convertvector(f2{v.x, v.y}, bf2), hi = __builtin_convertvector(f2{v.z, v.w}, bf2);
    return i2{__builtin_bit_cast(int, lo), __builtin_bit_cast(int, hi)};
}
template <bool IN32>
__device__ __forceinline__ f4 load_piece(const void* base, int64_t row, int64_t ld, int c0) {
    if (IN32) return *reinterpret_cast<const f4*>(static_cast<const float*>(base) + row * ld + c0);
    return unpack4(*reinterpret_cast<const i2*>(static_cast<const uint16_t*>(base) + row * ld + c0));
}
__device__ __forceinline__ void store_piece(uint16_t* base, int64_t row, int64_t ld, int c0, f4 v) {
    *reinterpret_cast<i2*>(base + row * ld + c0) = pack4(v);
}

template <bool IN32>
__global__ void __launch_bounds__(256)
    cheb_init16_kernel(const void* __restrict__ R, int64_t ldr, uint16_t* __restrict__ W, int64_t ldw,
                       uint16_t* __restrict__ Rcopy, int64_t ldc, const float* __restrict__ dinv, int64_t nv,
                       int cgroups, float c) {
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t node = tid / cgroups;
    if (node >= nv) return;
    const int c0 = (int)(tid - node * cgroups) * 4;
    const float* d = dinv + node * 9;
    const int64_t r = node * 3;
    const f4 r0 = load_piece<IN32>(R, r, ldr, c0), r1 = load_piece<IN32>(R, r + 1, ldr, c0),
             r2 = load_piece<IN32>(R, r + 2, ldr, c0);
    store_piece(W, r, ldw, c0, c * (d[0] * r0 + d[1] * r1 + d[2] * r2));
    store_piece(W, r + 1, ldw, c0, c * (d[3] * r0 + d[4] * r1 + d[5] * r2));
    store_piece(W, r + 2, ldw, c0, c * (d[6] * r0 + d[7] * r1 + d[8] * r2));
    if (Rcopy) {
        store_piece(Rcopy, r, ldc, c0, r0);
        store_piece(Rcopy, r + 1, ldc, c0, r1);
        store_piece(Rcopy, r + 2, ldc, c0, r2);
    }
}

__global__ void __launch_bounds__(256)
    scalar_csr16_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colidx,
                        const float* __restrict__ w, int64_t nrows, const uint16_t* __restrict__ X, int64_t ldx,
                        uint16_t* __restrict__ Y, int64_t ldy, int lpn, float beta) {
    const int cpn = 3 * lpn;  // 8-byte pieces per node panel
    // (each XCD a contiguous range of rows: neighbouring rows gather overlapping panels, which then meet in one L2)
    const int64_t gid = (int64_t)ds::xcd_remap(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x;
    const int64_t node = gid / cpn;
    if (node >= nrows) return;
    const int c = (int)(gid - node * cpn);
    const int r = c / lpn;
    const int c0 = (c - r * lpn) * 4;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    const int kb = rowptr[node], ke = rowptr[node + 1];
    // four independent gathers in flight (a restriction row has 30-60 entries and the plain loop met every gather - an
    // index load and the panel load that depends on it - with a full wait); the sum keeps its order
    int k = kb;
    for (; k + 4 <= ke; k += 4) {
        const int64_t j0 = colidx[k], j1 = colidx[k + 1], j2 = colidx[k + 2], j3 = colidx[k + 3];
        const float w0 = w[k], w1 = w[k + 1], w2 = w[k + 2], w3 = w[k + 3];
        const f4 x0 = load_piece<false>(X, 3 * j0 + r, ldx, c0), x1 = load_piece<false>(X, 3 * j1 + r, ldx, c0);
        const f4 x2 = load_piece<false>(X, 3 * j2 + r, ldx, c0), x3 = load_piece<false>(X, 3 * j3 + r, ldx, c0);
        acc += w0 * x0;
        acc += w1 * x1;
        acc += w2 * x2;
        acc += w3 * x3;
    }
    for (; k < ke; ++k) acc += w[k] * load_piece<false>(X, 3 * (int64_t)colidx[k] + r, ldx, c0);
    if (beta != 0.f) acc += beta * load_piece<false>(Y, 3 * node + r, ldy, c0);
    store_piece(Y, 3 * node + r, ldy, c0, acc);
}

}  // namespace

extern "C" int ds_cheb_init16(const void* R, int r_f32, int64_t ldr, void* W, int64_t ldw, void* Rcopy, int64_t ldc,
                              const float* dinv, int64_t nv, int ncols, float c, ds_stream_t stream) {
    DS_REQUIRE(R && W && dinv, "ds_cheb_init16: null pointer");
    DS_REQUIRE(nv > 0 && ncols > 0 && ncols % 4 == 0, "ds_cheb_init16: ncols must be a positive multiple of 4");
    DS_REQUIRE(ldr >= ncols && ldw >= ncols && (!Rcopy || ldc >= ncols), "ds_cheb_init16: leading dimension smaller than ncols");
    uintptr_t al = reinterpret_cast<uintptr_t>(W) | (uintptr_t)(ldw * 2) | reinterpret_cast<uintptr_t>(Rcopy) | (uintptr_t)(ldc * 2);
    al |= r_f32 ? 0 : (reinterpret_cast<uintptr_t>(R) | (uintptr_t)(ldr * 2));
    DS_REQUIRE((al & 7) == 0 && (!r_f32 || ((reinterpret_cast<uintptr_t>(R) | (uintptr_t)(ldr * 4)) & 15) == 0),
               "ds_cheb_init16: bf16 rows must be 8-byte aligned, fp32 rows 16-byte");
    const int cgroups = ncols / 4;
    const unsigned grid = (unsigned)ds::ceil_div(nv * cgroups, (int64_t)256);
    hipStream_t st = ds::as_stream(stream);
    if (r_f32)
        cheb_init16_kernel<true><<<grid, 256, 0, st>>>(R, ldr, static_cast<uint16_t*>(W), ldw, static_cast<uint16_t*>(Rcopy),
                                                       ldc, dinv, nv, cgroups, c);
    else
        cheb_init16_kernel<false><<<grid, 256, 0, st>>>(R, ldr, static_cast<uint16_t*>(W), ldw, static_cast<uint16_t*>(Rcopy),
                                                        ldc, dinv, nv, cgroups, c);
    DS_LAUNCH_CHECK("cheb_init16_kernel");
    return DS_OK;
}

extern "C" int ds_scalar_csr_spmm16(const int32_t* rowptr, const int32_t* colidx, const float* w, int64_t nrows,
                                    const void* X, int64_t ldx, void* Y, int64_t ldy, int ncols, float beta,
                                    ds_stream_t stream) {
    DS_REQUIRE(rowptr && colidx && w && X && Y, "ds_scalar_csr_spmm16: null pointer");
    DS_REQUIRE(nrows > 0 && ncols > 0 && ncols % 4 == 0, "ds_scalar_csr_spmm16: ncols must be a positive multiple of 4");
    DS_REQUIRE(ldx >= ncols && ldy >= ncols, "ds_scalar_csr_spmm16: leading dimension smaller than ncols");
    const uintptr_t al = reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(Y) | (uintptr_t)(ldx * 2) | (uintptr_t)(ldy * 2);
    DS_REQUIRE((al & 7) == 0, "ds_scalar_csr_spmm16: rows must be 8-byte aligned");
    DS_REQUIRE(X != Y, "ds_scalar_csr_spmm16: X and Y must be different buffers");
    const int lpn = ncols / 4;
    const int64_t blocks = ds::ceil_div(nrows * 3 * lpn, (int64_t)256);
    DS_REQUIRE(blocks < ((int64_t)1 << 31), "ds_scalar_csr_spmm16: grid too large");
    scalar_csr16_kernel<<<(unsigned)blocks, 256, 0, ds::as_stream(stream)>>>(
        rowptr, colidx, w, nrows, static_cast<const uint16_t*>(X), ldx, static_cast<uint16_t*>(Y), ldy, lpn, beta);
    DS_LAUNCH_CHECK("scalar_csr16_kernel");
    return DS_OK;
}
