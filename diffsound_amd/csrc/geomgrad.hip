// Geometry backward of the modal read-out - gfx950.
//
//   s(x) = sum_i g_i [ u_i^T K(x) u_i - lambda_i u_i^T M(x) u_i ]          (u_i, lambda_i constants)
//   out  = ds/dx   for the node coordinates x
//
// This is the gradient the reference obtains by autograd through get_vals -> sparse K, M -> coalesce ->
// torch.inverse / torch.det of the per-element transform matrices (reference
// src/diffelastic/diff_model.py:390-399, deform.py:35-68,136-147, mesh.py:58-99), which
// retains every per-Gauss-point intermediate (58 GB at 10k ord-2 tets).  Here nothing is retained: one
// wavefront per element, one lane per mode.  With the affine element map, F = sum_k c_k (x) grad L_k with
// c_k(g) = sum_a dN_a/dL_k(g) u_a, the element energy is  J sum_g w_g [ mu (F:F + F:F^T) + lam tr(F)^2 ],
// so d/d(grad L_k) = J sum_g w_g (2 P(F))^T c_k  and  d/dJ = energy/J - lambda_i rho u^T (M^ (x) I) u.
// The lanes' 14 partial numbers are wave-reduced, lane 0 applies  d(A^-1) = -A^-1 dA A^-1,
// d|det A| = |det A| tr(A^-1 dA)  and scatters 12 fp64 atomics to the four corner nodes
// (mid-edge nodes do not enter the element map).
#include "ds_common.h"

namespace {

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

template <int N>
__global__ void __launch_bounds__(256)
    geometry_grad_kernel(const int32_t* __restrict__ tets, int64_t T, const double* __restrict__ tetgeo,
                         const float* __restrict__ U, int64_t ldu, int m, const double* __restrict__ gk,
                         const double* __restrict__ gm, double lam, double mu, const double* __restrict__ gtab,
                         const double* __restrict__ gw, int ng, const double* __restrict__ mtab,
                         double* __restrict__ grad) {
    __shared__ double s_gt[4 * N * 4];  // up to 4 quadrature points
    __shared__ double s_m[N * N];
    for (int i = threadIdx.x; i < ng * N * 4; i += blockDim.x) s_gt[i] = gtab[i];
    for (int i = threadIdx.x; i < N * N; i += blockDim.x) s_m[i] = mtab[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int64_t e = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (e >= T) return;  // wave-uniform
    const int32_t* tt = tets + e * N;
    const double* geo = tetgeo + e * 13;
    double G[4][3];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int j = 0; j < 3; ++j) G[k][j] = geo[k * 3 + j];
    const double J = geo[12];

    double dG[4][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    double sW = 0.0, sM = 0.0;
    for (int i = lane; i < m; i += 64) {
        double u[N][3];
#pragma unroll
        for (int a = 0; a < N; ++a) {
            const int64_t row = (int64_t)tt[a] * 3;
#pragma unroll
            for (int d = 0; d < 3; ++d) u[a][d] = (double)U[(row + d) * ldu + i];
        }
        const double gi = gk[i];
        double mass = 0.0;
#pragma unroll
        for (int a = 0; a < N; ++a)
#pragma unroll
            for (int b = 0; b < N; ++b)
                mass += s_m[a * N + b] * (u[a][0] * u[b][0] + u[a][1] * u[b][1] + u[a][2] * u[b][2]);
        sM += gm[i] * mass;
        for (int g = 0; g < ng; ++g) {
            const double* gt = s_gt + g * N * 4;
            double c[4][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
#pragma unroll
            for (int a = 0; a < N; ++a)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const double w = gt[a * 4 + k];
                    c[k][0] = fma(w, u[a][0], c[k][0]);
                    c[k][1] = fma(w, u[a][1], c[k][1]);
                    c[k][2] = fma(w, u[a][2], c[k][2]);
                }
            double F[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int j = 0; j < 3; ++j) F[r][j] = fma(c[k][r], G[k][j], F[r][j]);
            const double tr = F[0][0] + F[1][1] + F[2][2];
            double W = lam * tr * tr;
            double P2[3][3];  // dW/dF = 2 mu (F + F^T) + 2 lam tr I
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    W += mu * (F[r][j] * F[r][j] + F[r][j] * F[j][r]);
                    P2[r][j] = 2.0 * mu * (F[r][j] + F[j][r]) + (r == j ? 2.0 * lam * tr : 0.0);
                }
            const double wg = gw[g] * gi;
            sW = fma(wg, W, sW);
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    dG[k][j] += wg * (P2[0][j] * c[k][0] + P2[1][j] * c[k][1] + P2[2][j] * c[k][2]);
        }
    }
    sW = wave_sum(sW);
    sM = wave_sum(sM);
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int j = 0; j < 3; ++j) dG[k][j] = wave_sum(dG[k][j]);
    if (lane != 0) return;
    // s_e = J * sW(G) - J * sM ;  G_3 = -(G_0 + G_1 + G_2) ;  rows of A^-1 are G_0..2
    double B[3][3];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int j = 0; j < 3; ++j) B[k][j] = J * (dG[k][j] - dG[3][j]);
    // dS/dA = -Ainv^T B Ainv^T + (sW - sM) J Ainv^T      (Ainv[k][j] = G[k][j])
    double Tm[3][3];  // Tm = B Ainv^T : Tm[k][r] = sum_j B[k][j] Ainv[r][j]
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int r = 0; r < 3; ++r) Tm[k][r] = B[k][0] * G[r][0] + B[k][1] * G[r][1] + B[k][2] * G[r][2];
    const double sj = (sW - sM) * J;
    double dA[3][3];  // dA[r][c] : derivative w.r.t. A[r][c] = p_c[r] - p_3[r]
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            // (Ainv^T Tm)[r][c] = sum_k Ainv[k][r] Tm[k][c]
            const double t = G[0][r] * Tm[0][c] + G[1][r] * Tm[1][c] + G[2][r] * Tm[2][c];
            dA[r][c] = -t + sj * G[c][r];
        }
    const int cs[4] = {0, N == 4 ? 1 : 2, N == 4 ? 2 : 4, N == 4 ? 3 : 9};
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        double s3 = 0.0;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            atomicAdd(&grad[(int64_t)tt[cs[c]] * 3 + r], dA[r][c]);
            s3 += dA[r][c];
        }
        atomicAdd(&grad[(int64_t)tt[cs[3]] * 3 + r], -s3);
    }
}

}  // namespace

extern "C" int ds_geometry_grad(const int32_t* tets, int64_t T, int N, int64_t nv, const double* tetgeo, const float* U,
                                int64_t ldu, int m, const double* gk, const double* gm, double lam, double mu,
                                const double* gtab, const double* gw, int ng, const double* mtab, double* grad,
                                ds_stream_t stream) {
    DS_REQUIRE(tets && tetgeo && U && gk && gm && gtab && gw && mtab && grad, "ds_geometry_grad: null pointer");
    DS_REQUIRE(N == 4 || N == 10, "ds_geometry_grad: N must be 4 or 10 (got %d)", N);
    DS_REQUIRE(T > 0 && nv > 0 && m > 0 && ldu >= m, "ds_geometry_grad: bad sizes");
    DS_REQUIRE(ng >= 1 && ng <= 4, "ds_geometry_grad: 1..4 quadrature points supported (got %d)", ng);
    hipStream_t st = ds::as_stream(stream);
    const unsigned grid = (unsigned)ds::ceil_div(T, 4);
    if (N == 4)
        geometry_grad_kernel<4><<<grid, 256, 0, st>>>(tets, T, tetgeo, U, ldu, m, gk, gm, lam, mu, gtab, gw, ng, mtab, grad);
    else
        geometry_grad_kernel<10><<<grid, 256, 0, st>>>(tets, T, tetgeo, U, ldu, m, gk, gm, lam, mu, gtab, gw, ng, mtab, grad);
    DS_LAUNCH_CHECK("geometry_grad_kernel");
    return DS_OK;
}
