// Numeric assembly of K_lambda, K_mu (BSR-3) and M_s (node-scalar CSR) - gfx950.
//
// Restates update_stiff_matrix / update_mass_matrix of the reference
// (src/diffelastic/diff_model.py:184-312).  Because the element map is affine (the reference's
// transform matrix uses the four corners only, src/diffelastic/mesh.py:58-99) the Gauss sum
//   Ke[(a,i),(b,j)] = sum_g w_g |detA| [ mu (d_ij grad N_a . grad N_b + grad N_a[j] grad N_b[i])
//                                       + lam grad N_a[i] grad N_b[j] ]
// collapses to  H_ab = sum_{k,l} D[a,k,b,l] grad L_k (x) grad L_l  with the constant table
// D[a,k,b,l] = sum_g w_g dN_a/dL_k dN_b/dL_l built by the host from the reference's own rule:
//   K_lambda block = |detA| H_ab ,  K_mu block = |detA| (tr(H_ab) I + H_ab^T).
//
// Design: no atomics, no COO.  Kernel 1 computes the 13 geometry scalars per element; kernel 2
// gives ONE THREAD PER BLOCK SLOT of the pattern and sums that slot's contribution list (built by
// the symbolic phase, ascending ids) in registers -> bitwise reproducible.  HBM traffic is the
// contribution list (4 B each), the L2-resident geometry gather and the 19 fp64 outputs per slot.
#include "ds_common.h"

namespace {

__global__ void tet_geometry_kernel(const float* __restrict__ verts, const int32_t* __restrict__ tets, int64_t T,
                                    int N, double* __restrict__ geo) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    const int32_t* tt = tets + t * N;
    // corner nodes in the reference's local order: ord-1 (0,1,2,3), ord-2 (0,2,4,9)  (mesh.py:75-84)
    const int c0 = 0, c1 = N == 4 ? 1 : 2, c2 = N == 4 ? 2 : 4, c3 = N == 4 ? 3 : 9;
    double p[4][3];
    const int cs[4] = {c0, c1, c2, c3};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float* v = verts + (int64_t)tt[cs[k]] * 3;
        p[k][0] = (double)v[0];
        p[k][1] = (double)v[1];
        p[k][2] = (double)v[2];
    }
    // A = [p0-p3, p1-p3, p2-p3] as columns ; rows of A^-1 are grad L_1..3 ; grad L_4 = -(sum)
    double a[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) a[r][c] = p[c][r] - p[3][r];
    const double c00 = a[1][1] * a[2][2] - a[1][2] * a[2][1];
    const double c01 = a[1][2] * a[2][0] - a[1][0] * a[2][2];
    const double c02 = a[1][0] * a[2][1] - a[1][1] * a[2][0];
    const double det = a[0][0] * c00 + a[0][1] * c01 + a[0][2] * c02;
    const double id = 1.0 / det;
    double g[4][3];
    g[0][0] = c00 * id;
    g[0][1] = (a[0][2] * a[2][1] - a[0][1] * a[2][2]) * id;
    g[0][2] = (a[0][1] * a[1][2] - a[0][2] * a[1][1]) * id;
    g[1][0] = c01 * id;
    g[1][1] = (a[0][0] * a[2][2] - a[0][2] * a[2][0]) * id;
    g[1][2] = (a[0][2] * a[1][0] - a[0][0] * a[1][2]) * id;
    g[2][0] = c02 * id;
    g[2][1] = (a[0][1] * a[2][0] - a[0][0] * a[2][1]) * id;
    g[2][2] = (a[0][0] * a[1][1] - a[0][1] * a[1][0]) * id;
#pragma unroll
    for (int c = 0; c < 3; ++c) g[3][c] = -(g[0][c] + g[1][c] + g[2][c]);
    double* o = geo + t * 13;
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int c = 0; c < 3; ++c) o[k * 3 + c] = g[k][c];
    o[12] = fabs(det);  // the reference integrates with |det| (deform.py:143-144, diff_model.py:289)
}

template <int N>
__global__ void __launch_bounds__(256) assemble_blocks_kernel(const int32_t* __restrict__ cptr,
                                                              const int32_t* __restrict__ clist, int64_t nnzb,
                                                              const double* __restrict__ geo,
                                                              const double* __restrict__ dtab,
                                                              const double* __restrict__ mtab,
                                                              double* __restrict__ klam, double* __restrict__ kmu,
                                                              double* __restrict__ ms) {
    __shared__ double sD[N * 4 * N * 4];
    __shared__ double sM[N * N];
    for (int i = threadIdx.x; i < N * 4 * N * 4; i += blockDim.x) sD[i] = dtab[i];
    for (int i = threadIdx.x; i < N * N; i += blockDim.x) sM[i] = mtab[i];
    __syncthreads();
    const int64_t s0 = (int64_t)blockIdx.x * blockDim.x;
    const int64_t s = s0 + threadIdx.x;
    double H[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    double msum = 0.0;
    const int c_end = s < nnzb ? cptr[s + 1] : 0;
    for (int c = s < nnzb ? cptr[s] : 0; c < c_end; ++c) {
        const int cid = clist[c];
        const int t = cid / (N * N);
        const int ab = cid - t * (N * N);
        const int a = ab / N, b = ab - a * N;
        const double* g = geo + (int64_t)t * 13;
        double G[4][3];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int i = 0; i < 3; ++i) G[k][i] = g[k * 3 + i];
        const double J = g[12];
        const double* D = sD + (a * 4) * (N * 4) + b * 4;  // D[a][k][b][l] at D[k*(N*4) + l]
        // u[l][i] = sum_k D[a,k,b,l] G[k][i] ;  h[i][j] = sum_l u[l][i] G[l][j]
        double h[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
#pragma unroll
        for (int l = 0; l < 4; ++l) {
            double u0 = 0, u1 = 0, u2 = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const double d = D[k * (N * 4) + l];
                u0 = fma(d, G[k][0], u0);
                u1 = fma(d, G[k][1], u1);
                u2 = fma(d, G[k][2], u2);
            }
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                h[0][j] = fma(u0, G[l][j], h[0][j]);
                h[1][j] = fma(u1, G[l][j], h[1][j]);
                h[2][j] = fma(u2, G[l][j], h[2][j]);
            }
        }
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) H[i][j] = fma(J, h[i][j], H[i][j]);
        msum = fma(J, sM[a * N + b], msum);
    }
    // The 9 doubles of a slot are contiguous in memory, the slots of neighbouring lanes 72 bytes apart: written
    // straight from the registers every store instruction covered 8 of every 72 bytes (1.9 ms for 0.63 GB on the
    // benchmark mesh).  The workgroup's 256 x 9 values go through LDS instead and leave as whole rows.
    __shared__ double sOut[256 * 9];
    const double tr = H[0][0] + H[1][1] + H[2][2];
    const int64_t nslot = min<int64_t>(256, nnzb - s0);
    const int nval = (int)nslot * 9;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        __syncthreads();  // (pass 0: the tables are no longer read; pass 1: sOut has been drained)
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j)
                sOut[threadIdx.x * 9 + i * 3 + j] = pass == 0 ? H[i][j] : H[j][i] + (i == j ? tr : 0.0);
        __syncthreads();
        double* dst = (pass == 0 ? klam : kmu) + s0 * 9;
        for (int e = threadIdx.x; e < nval; e += 256) dst[e] = sOut[e];
    }
    if (s < nnzb) ms[s] = msum;
}

__global__ void combine_values_kernel(const double* __restrict__ klam, const double* __restrict__ kmu,
                                      const double* __restrict__ ms, int64_t nnzb, double lam, double mu,
                                      float* __restrict__ k32, float* __restrict__ k32t, float* __restrict__ ms32) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nnzb * 9) {
        const float v = (float)(lam * klam[i] + mu * kmu[i]);
        k32[i] = v;
        if (k32t) {
            const int64_t blk = i / 9;
            const int e = (int)(i - blk * 9), row = e / 3, col = e - row * 3;
            k32t[blk * 9 + col * 3 + row] = v;
        }
    }
    if (i < nnzb) ms32[i] = (float)ms[i];
}

__global__ void diag_inverse_kernel(const double* __restrict__ klam, const double* __restrict__ kmu,
                                    const int32_t* __restrict__ diagidx, int64_t nv, double lam, double mu,
                                    float* __restrict__ dinv) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nv) return;
    const int d = diagidx[i];
    float* o = dinv + i * 9;
    if (d < 0) {  // node not referenced by any element: identity keeps the preconditioner defined
        for (int k = 0; k < 9; ++k) o[k] = (k % 4 == 0) ? 1.f : 0.f;
        return;
    }
    double a[9];
    for (int k = 0; k < 9; ++k) a[k] = lam * klam[(int64_t)d * 9 + k] + mu * kmu[(int64_t)d * 9 + k];
    const double c0 = a[4] * a[8] - a[5] * a[7];
    const double c1 = a[5] * a[6] - a[3] * a[8];
    const double c2 = a[3] * a[7] - a[4] * a[6];
    const double id = 1.0 / (a[0] * c0 + a[1] * c1 + a[2] * c2);
    o[0] = (float)(c0 * id);
    o[1] = (float)((a[2] * a[7] - a[1] * a[8]) * id);
    o[2] = (float)((a[1] * a[5] - a[2] * a[4]) * id);
    o[3] = (float)(c1 * id);
    o[4] = (float)((a[0] * a[8] - a[2] * a[6]) * id);
    o[5] = (float)((a[2] * a[3] - a[0] * a[5]) * id);
    o[6] = (float)(c2 * id);
    o[7] = (float)((a[1] * a[6] - a[0] * a[7]) * id);
    o[8] = (float)((a[0] * a[4] - a[1] * a[3]) * id);
}

}  // namespace

extern "C" int ds_assemble_kml(const float* verts, const int32_t* tets, int64_t T, int N, int64_t nv,
                               const int32_t* cptr, const int32_t* clist, int64_t nnzb, const double* dtab,
                               const double* mtab, double* tetgeo, double* klam, double* kmu, double* ms,
                               ds_stream_t stream) {
    DS_REQUIRE(N == 4 || N == 10, "ds_assemble_kml: N must be 4 or 10 (got %d)", N);
    DS_REQUIRE(verts && tets && cptr && clist && dtab && mtab && tetgeo && klam && kmu && ms,
               "ds_assemble_kml: null pointer");
    DS_REQUIRE(T > 0 && nv > 0 && nnzb > 0, "ds_assemble_kml: empty problem");
    hipStream_t st = ds::as_stream(stream);
    tet_geometry_kernel<<<(unsigned)ds::ceil_div(T, 256), 256, 0, st>>>(verts, tets, T, N, tetgeo);
    DS_LAUNCH_CHECK("tet_geometry_kernel");
    const unsigned grid = (unsigned)ds::ceil_div(nnzb, 256);
    if (N == 4)
        assemble_blocks_kernel<4><<<grid, 256, 0, st>>>(cptr, clist, nnzb, tetgeo, dtab, mtab, klam, kmu, ms);
    else
        assemble_blocks_kernel<10><<<grid, 256, 0, st>>>(cptr, clist, nnzb, tetgeo, dtab, mtab, klam, kmu, ms);
    DS_LAUNCH_CHECK("assemble_blocks_kernel");
    return DS_OK;
}

extern "C" int ds_combine_material(const double* klam, const double* kmu, const double* ms, int64_t nnzb,
                                   const int32_t* diagidx, int64_t nv, double lam, double mu, float* k32,
                                   float* k32t, float* ms32, float* dinv32, ds_stream_t stream) {
    DS_REQUIRE(klam && kmu && ms && diagidx && k32 && ms32 && dinv32, "ds_combine_material: null pointer");
    DS_REQUIRE(nnzb > 0 && nv > 0, "ds_combine_material: empty problem");
    hipStream_t st = ds::as_stream(stream);
    combine_values_kernel<<<(unsigned)ds::ceil_div(nnzb * 9, 256), 256, 0, st>>>(klam, kmu, ms, nnzb, lam, mu, k32,
                                                                                 k32t, ms32);
    DS_LAUNCH_CHECK("combine_values_kernel");
    diag_inverse_kernel<<<(unsigned)ds::ceil_div(nv, 256), 256, 0, st>>>(klam, kmu, diagidx, nv, lam, mu, dinv32);
    DS_LAUNCH_CHECK("diag_inverse_kernel");
    return DS_OK;
}
