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


// ---- GROUP-block Jacobi (round 6).  T = the inverse of the 24 x 24 diagonal block K_gg of every group of 8 consecutive nodes (the
// groups of the matrix-core term kernel's tiles) instead of the 3 x 3 node blocks: Chebyshev(14, ratio 150) in T_g K then does on
// the corner-node level what Chebyshev(22, ratio 350) does with the node blocks (profiles/r06_group_block_jacobi*.txt).  The TERM
// kernel does not change: it is handed the blocks of T_g K - dense over (node of the group) x (entry of the group's union), every
// presence bit set - with an identity in the place of dinv, and a right-hand side that went through T_g first:
//   W' = W + c1 (W - W_prev) + c2 (T R - (T K) W).
constexpr int GJ_G = 8, GJ_N = 3 * GJ_G;

// T_g = K_gg^-1: one workgroup per group; K_gg gathered from the BSR rows of the group's nodes into LDS (fp64), Gauss-Jordan
// without pivoting (a principal submatrix of the stiffness matrix on a strict subset of the nodes is positive definite),
// nodes behind the last one: identity rows.
__global__ void __launch_bounds__(64)
    group_inverse_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colidx,
                         const float* __restrict__ k32, int64_t nv, float* __restrict__ T) {
    __shared__ double A[GJ_N][2 * GJ_N + 1];
    __shared__ double prow[2 * GJ_N];
    __shared__ double fcol[GJ_N];
    __shared__ double Dg[GJ_G][9];  // the nodes' own 3 x 3 blocks: what a group whose elimination breaks down falls back to
    __shared__ int bad;
    const int64_t g = blockIdx.x;
    const int tid = threadIdx.x;
    if (tid == 0) bad = 0;
    for (int i = tid; i < GJ_N * 2 * GJ_N; i += 64) {
        const int r = i / (2 * GJ_N), c = i - r * 2 * GJ_N;
        A[r][c] = (c == GJ_N + r) ? 1.0 : 0.0;
    }
    for (int i = tid; i < GJ_G * 9; i += 64) Dg[i / 9][i % 9] = (i % 9) % 4 == 0 ? 1.0 : 0.0;
    __syncthreads();
    {   // 8 lanes per node of the group walk that node's row
        const int s = tid >> 3, l = tid & 7;
        const int64_t row = g * GJ_G + s;
        if (row < nv) {
            for (int j = rowptr[row] + l; j < rowptr[row + 1]; j += 8) {
                const int64_t c = colidx[j];
                if (c / GJ_G != g) continue;
                const int cs = (int)(c - g * GJ_G);
                const float* b = k32 + (int64_t)j * 9;
                for (int i = 0; i < 3; ++i)
                    for (int k = 0; k < 3; ++k) {
                        A[3 * s + i][3 * cs + k] = (double)b[3 * i + k];
                        if (cs == s) Dg[s][3 * i + k] = (double)b[3 * i + k];
                    }
            }
        } else if (l < 3) {
            A[3 * s + l][3 * s + l] = 1.0;
        }
    }
    __syncthreads();
    for (int p = 0; p < GJ_N; ++p) {
        const double app = A[p][p];
        if (tid == 0 && !(app > 0.0)) bad = 1;  // (not positive, or not a number: K_gg is not positive definite in fp64)
        const double piv = 1.0 / app;
        for (int c = tid; c < 2 * GJ_N; c += 64) prow[c] = A[p][c] * piv;
        if (tid < GJ_N) fcol[tid] = A[tid][p];
        __syncthreads();
        for (int i = tid; i < GJ_N * 2 * GJ_N; i += 64) {
            const int r = i / (2 * GJ_N), c = i - r * 2 * GJ_N;
            A[r][c] = (r == p) ? prow[c] : A[r][c] - fcol[r] * prow[c];
        }
        __syncthreads();
    }
    if (bad) {  // the node blocks' inverses for this group (the polynomial's interval bound holds for them as well)
        for (int i = tid; i < GJ_N * GJ_N; i += 64) T[g * GJ_N * GJ_N + i] = 0.f;
        __syncthreads();
        if (tid < GJ_G) {
            const double* d = Dg[tid];
            const double c00 = d[4] * d[8] - d[5] * d[7], c01 = d[5] * d[6] - d[3] * d[8], c02 = d[3] * d[7] - d[4] * d[6];
            const double det = d[0] * c00 + d[1] * c01 + d[2] * c02, id = 1.0 / det;
            const double inv[9] = {c00 * id, (d[2] * d[7] - d[1] * d[8]) * id, (d[1] * d[5] - d[2] * d[4]) * id,
                                   c01 * id, (d[0] * d[8] - d[2] * d[6]) * id, (d[2] * d[3] - d[0] * d[5]) * id,
                                   c02 * id, (d[1] * d[6] - d[0] * d[7]) * id, (d[0] * d[4] - d[1] * d[3]) * id};
            for (int i = 0; i < 3; ++i)
                for (int k = 0; k < 3; ++k)
                    T[g * GJ_N * GJ_N + (3 * tid + i) * GJ_N + 3 * tid + k] = (float)(0.5 * (inv[3 * i + k] + inv[3 * k + i]));
        }
        return;
    }
    for (int i = tid; i < GJ_N * GJ_N; i += 64) {
        const int r = i / GJ_N, c = i - r * GJ_N;
        T[g * GJ_N * GJ_N + i] = (float)(0.5 * (A[r][GJ_N + c] + A[c][GJ_N + r]));  // (symmetric to the last bit)
    }
}

// The blocks of T_g K in the term kernel's order, DENSE: position (first entry of the group + e) * 8 + s' holds
// sum_s T_g[s', s] K(s, e) for every node s' of the group and every entry e of its union (bf16 rows padded to 8 bytes, as ds_pack_kc).
// K(s, e) is found through the COMPACT tables of the same topology (presence mask | first block of the entry, kperm).
__global__ void __launch_bounds__(64)
    group_pack_kernel(const float* __restrict__ k32, const float* __restrict__ T, const int32_t* __restrict__ gptr,
                      const int32_t* __restrict__ gmeta, const int32_t* __restrict__ gbase, const int32_t* __restrict__ kperm,
                      int64_t nv, uint16_t* __restrict__ kc) {
    __shared__ float Ts[GJ_N][GJ_N + 1];
    const int64_t g = blockIdx.x;
    const int tid = threadIdx.x;
    for (int i = tid; i < GJ_N * GJ_N; i += 64) Ts[i / GJ_N][i % GJ_N] = T[g * GJ_N * GJ_N + i];
    __syncthreads();
    const int e0 = gptr[g], ne = gptr[g + 1] - e0;
    const int base = gbase[g];
    for (int w = tid; w < ne * GJ_G; w += 64) {
        const int e = w / GJ_G, sp = w - e * GJ_G;
        const int meta = gmeta[e0 + e];
        const int mask = meta & 0xff, first = (int)((unsigned)meta >> 8);
        float acc[3][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
        if (g * GJ_G + sp < nv) {
            int rank = 0;
            for (int s = 0; s < GJ_G; ++s) {
                if (!((mask >> s) & 1)) continue;
                const float* b = k32 + (int64_t)kperm[base + first + rank] * 9;
                ++rank;
                for (int i = 0; i < 3; ++i)
                    for (int k = 0; k < 3; ++k) {
                        const float t = Ts[3 * sp + i][3 * s + k];
                        acc[i][0] += t * b[3 * k + 0];
                        acc[i][1] += t * b[3 * k + 1];
                        acc[i][2] += t * b[3 * k + 2];
                    }
            }
        }
        uint16_t* out = kc + ((int64_t)(e0 + e) * GJ_G + sp) * 12;
        for (int i = 0; i < 3; ++i) {
            const i2 pk = pack4(f4{acc[i][0], acc[i][1], acc[i][2], 0.f});
            *reinterpret_cast<i2*>(out + 4 * i) = pk;
        }
    }
}

// Y = T_g X per group (Y bf16, or fp32 for the power iteration's block): the group's 24 x ncols panel is staged in LDS by all
// lanes, then every lane forms (row, four columns) pieces of the product from LDS - T_g by broadcast reads, the panel by
// consecutive 16-byte reads.  Y may be X (every load of the workgroup completes before its first store).
template <bool IN32, bool OUT32>
__global__ void __launch_bounds__(256)
    group_apply_kernel(const float* __restrict__ T, const void* __restrict__ X, int64_t ldx, void* __restrict__ Y, int64_t ldy,
                       int64_t nv, int cgroups) {
    extern __shared__ float gj_lds[];
    float(*Ts)[GJ_N + 1] = reinterpret_cast<float(*)[GJ_N + 1]>(gj_lds);
    f4* Xs = reinterpret_cast<f4*>(gj_lds + GJ_N * (GJ_N + 1) + 4);  // (604 floats in: a 16-byte boundary)
    const int64_t g = blockIdx.x;
    const int tid = threadIdx.x;
    for (int i = tid; i < GJ_N * GJ_N; i += 256) Ts[i / GJ_N][i % GJ_N] = T[g * GJ_N * GJ_N + i];
    const int64_t r0 = g * GJ_N;
    const int nr = (int)(((3 * nv - r0) < GJ_N) ? (3 * nv - r0) : GJ_N);
    const int np = GJ_N * cgroups;
    for (int i = tid; i < np; i += 256) {
        const int r = i / cgroups, cg = i - r * cgroups;
        Xs[i] = r < nr ? load_piece<IN32>(X, r0 + r, ldx, cg * 4) : f4{0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();
    for (int i = tid; i < np; i += 256) {
        const int r = i / cgroups, cg = i - r * cgroups;
        if (r >= nr) continue;
        f4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < GJ_N; ++k) acc += Ts[r][k] * Xs[k * cgroups + cg];
        if (OUT32)
            *reinterpret_cast<f4*>(static_cast<float*>(Y) + (r0 + r) * ldy + cg * 4) = acc;
        else
            store_piece(static_cast<uint16_t*>(Y), r0 + r, ldy, cg * 4, acc);
    }
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

extern "C" int ds_group_inverse(const int32_t* rowptr, const int32_t* colidx, const float* k32, int64_t nv, int group_nodes,
                                float* T, ds_stream_t stream) {
    DS_REQUIRE(rowptr && colidx && k32 && T, "ds_group_inverse: null pointer");
    DS_REQUIRE(nv > 0 && group_nodes == GJ_G, "ds_group_inverse: groups of 8 nodes");
    const int64_t ng = ds::ceil_div(nv, (int64_t)GJ_G);
    DS_REQUIRE(ng < ((int64_t)1 << 31), "ds_group_inverse: grid too large");
    group_inverse_kernel<<<(unsigned)ng, 64, 0, ds::as_stream(stream)>>>(rowptr, colidx, k32, nv, T);
    DS_LAUNCH_CHECK("group_inverse_kernel");
    return DS_OK;
}

extern "C" int ds_group_pack_kc(const float* k32, const float* T, const int32_t* gptr, const int32_t* gmeta,
                                const int32_t* gbase, const int32_t* kperm, int group_nodes, int64_t nv, void* kc,
                                ds_stream_t stream) {
    DS_REQUIRE(k32 && T && gptr && gmeta && gbase && kperm && kc, "ds_group_pack_kc: null pointer");
    DS_REQUIRE(nv > 0 && group_nodes == GJ_G, "ds_group_pack_kc: groups of 8 nodes");
    DS_REQUIRE((reinterpret_cast<uintptr_t>(kc) & 7) == 0, "ds_group_pack_kc: kc must be 8-byte aligned");
    const int64_t ng = ds::ceil_div(nv, (int64_t)GJ_G);
    DS_REQUIRE(ng < ((int64_t)1 << 31), "ds_group_pack_kc: grid too large");
    group_pack_kernel<<<(unsigned)ng, 64, 0, ds::as_stream(stream)>>>(k32, T, gptr, gmeta, gbase, kperm, nv,
                                                                      static_cast<uint16_t*>(kc));
    DS_LAUNCH_CHECK("group_pack_kernel");
    return DS_OK;
}

extern "C" int ds_group_apply16(const float* T, int group_nodes, const void* X, int x_f32, int64_t ldx, void* Y, int y_f32,
                                int64_t ldy, int64_t nv, int ncols, ds_stream_t stream) {
    DS_REQUIRE(T && X && Y, "ds_group_apply16: null pointer");
    DS_REQUIRE(nv > 0 && group_nodes == GJ_G, "ds_group_apply16: groups of 8 nodes");
    DS_REQUIRE(ncols > 0 && ncols % 4 == 0 && ncols <= 256 && ldx >= ncols && ldy >= ncols,
               "ds_group_apply16: ncols must be a positive multiple of 4, <= 256 and <= ld");
    uintptr_t al16 = 0, al8 = 0;
    (x_f32 ? al16 : al8) |= reinterpret_cast<uintptr_t>(X) | (uintptr_t)(ldx * (x_f32 ? 4 : 2));
    (y_f32 ? al16 : al8) |= reinterpret_cast<uintptr_t>(Y) | (uintptr_t)(ldy * (y_f32 ? 4 : 2));
    DS_REQUIRE((al8 & 7) == 0 && (al16 & 15) == 0, "ds_group_apply16: bf16 rows must be 8-byte aligned, fp32 rows 16-byte");
    DS_REQUIRE(X != Y || x_f32 == y_f32, "ds_group_apply16: in place only with one element type");
    const int64_t ng = ds::ceil_div(nv, (int64_t)GJ_G);
    DS_REQUIRE(ng < ((int64_t)1 << 31), "ds_group_apply16: grid too large");
    hipStream_t st = ds::as_stream(stream);
    const int cg = ncols / 4;
    // LDS: T_g (24 x 25 floats, padded to a 16-byte boundary: 604 floats) + the panel (24 x ncols floats)
    const size_t lds = (size_t)(GJ_N * (GJ_N + 1) + 4) * 4 + (size_t)GJ_N * ncols * 4;
#define DS_GJ(I, O) group_apply_kernel<I, O><<<(unsigned)ng, 256, lds, st>>>(T, X, ldx, Y, ldy, nv, cg)
    if (x_f32 && y_f32) DS_GJ(true, true);
    else if (x_f32) DS_GJ(true, false);
    else if (y_f32) DS_GJ(false, true);
    else DS_GJ(false, false);
#undef DS_GJ
    DS_LAUNCH_CHECK("group_apply_kernel");
    return DS_OK;
}
