// The differentiable read-out + render + loss + backward of a pass in ONE call - gfx950.
//
// Steps 3-6 of a pass (diffsound_amd/pipeline.py), i.e. the body the reference's training loop runs every epoch between
// eigendecompositions (experiments/material_sync_train.py:135-167 with EIGEN_DECOMPOSE_CYCLE = 15):
//   pred_i = ev_i + (lam a_i + mu b_i) - ev_i m_i           get_undamped_freqs, src/diffelastic/diff_model.py:371-388
//   f_i    = float(sqrt(pred_i) / 2 / pi)                    (the reference hands fp32 frequencies to the oscillator)
//   w0sq_i = (2 pi f_i)^2,  d_i = (alpha + beta w0sq_i) / 2,  w_i = sqrt(w0sq_i - d_i^2)       src/ddsp/oscillator.py:282-310
//   y      = oscillator bank (ds_osc_bank_fwd),  loss = mean((y - target)^2)  (target NULL: mean(y^2))
//   backward: gy = 2 (y - target) / S -> ds_osc_bank_bwd -> (gd_i, gw_i) -> g_pred_i -> (dloss/dE, dloss/dnu) through
//             lam(E, nu), mu(E, nu), whose four partial derivatives the caller passes in.
// Written as torch operations this is ~45 element-wise launches on 64-element vectors plus the autograd engine: 0.83 ms of
// host time per pass for 0.07 ms of kernels (profiles/r03_cached_pass_profile.txt).  Here: three tiny kernels around the two
// oscillator launches and ONE 24-byte result (loss, dloss/dE, dloss/dnu) for the caller to copy back.
#include <cmath>

#include "ds_common.h"

namespace {

constexpr double TWO_PI = 6.283185307179586476925286766559;

// one thread per mode: frequencies, decay rates, damped angular frequencies and what the chain rule needs later
__global__ void readout_pre_kernel(const double* __restrict__ ev, const double* __restrict__ a, const double* __restrict__ b,
                                   const double* __restrict__ md, int m, double lam, double mu, double alpha, double beta,
                                   float* __restrict__ freqs, double* __restrict__ work) {
#pragma clang fp contract(off)  // the torch formulation rounds every product: no fused multiply-adds here (bit-identical d, w)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const double pred = ev[i] + (lam * a[i] + mu * b[i]) - ev[i] * md[i];
    const double sp = sqrt(pred);
    const float f32 = (float)(sp / 2.0 / M_PI);
    freqs[i] = f32;
    const double fd = (double)f32;
    const double x = fd * TWO_PI;
    const double w0sq = x * x;
    const double d = 0.5 * (alpha + beta * w0sq);
    const double w = sqrt(w0sq - d * d);
    work[i] = d;
    work[m + i] = w;
    work[2 * m + i] = fd;
    work[3 * m + i] = sp;
}

// loss = mean(diff^2) and its gradient gy = 2 diff / S; one workgroup (S is a clip: thousands of samples)
__global__ void __launch_bounds__(1024)
    readout_loss_kernel(const float* __restrict__ y, const float* __restrict__ target, int S, float* __restrict__ gy,
                        double* __restrict__ out) {
    __shared__ double part[16];
    double acc = 0.0;
    const float scale = 2.0f / (float)S;
    for (int t = threadIdx.x; t < S; t += blockDim.x) {
        const float diff = target ? y[t] - target[t] : y[t];
        gy[t] = diff * scale;
        acc += (double)diff * (double)diff;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int k = 0; k < (int)(blockDim.x >> 6); ++k) s += part[k];
        out[0] = s / (double)S;
    }
}

// (gd, gw) per mode -> dloss/dpred per mode -> dloss/dlam = sum_i g_i a_i, dloss/dmu = sum_i g_i b_i -> (E, nu); one wave
__global__ void __launch_bounds__(64)
    readout_post_kernel(const double* __restrict__ a, const double* __restrict__ b, int m, double beta, double dlam_dE,
                        double dlam_dnu, double dmu_dE, double dmu_dnu, const double* __restrict__ work,
                        double* __restrict__ out) {
    double gl = 0.0, gm = 0.0;
    for (int i = threadIdx.x; i < m; i += 64) {
        const double d = work[i], w = work[m + i], fd = work[2 * m + i], sp = work[3 * m + i];
        const double gd = work[4 * m + i], gw = work[5 * m + i];
        // w = sqrt(w0sq - d^2), d = (alpha + beta w0sq) / 2
        const double gu = gw / (2.0 * w);
        const double gw0 = gu + (gd - 2.0 * d * gu) * (0.5 * beta);
        // w0sq = (2 pi f)^2, f = sqrt(pred) / 2 / pi (the fp32 rounding of f passes gradients through unchanged)
        const double gf = gw0 * 2.0 * TWO_PI * TWO_PI * fd;
        const double gp = gf / (2.0 * M_PI) * (0.5 / sp);
        gl += gp * a[i];
        gm += gp * b[i];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        gl += __shfl_xor(gl, o);
        gm += __shfl_xor(gm, o);
    }
    if (threadIdx.x == 0) {
        out[1] = gl * dlam_dE + gm * dmu_dE;
        out[2] = gl * dlam_dnu + gm * dmu_dnu;
    }
}

}  // namespace

extern "C" int ds_readout_pass(const double* ev, const double* a_lam, const double* b_mu, const double* m_diag, int m,
                               double lam, double mu, double dlam_dE, double dlam_dnu, double dmu_dE, double dmu_dnu,
                               double alpha, double beta, const float* force, int F, int S, double sr, const float* target,
                               int backward, float* audio, float* freqs, double* work, float* fwork, double* out,
                               ds_stream_t stream) {
    DS_REQUIRE(ev && a_lam && b_mu && m_diag && force && audio && freqs && work && fwork && out, "ds_readout_pass: null pointer");
    DS_REQUIRE(m > 0 && S > 0 && sr > 0, "ds_readout_pass: empty problem");
    hipStream_t st = ds::as_stream(stream);
    readout_pre_kernel<<<(unsigned)ds::ceil_div(m, 64), 64, 0, st>>>(ev, a_lam, b_mu, m_diag, m, lam, mu, alpha, beta, freqs, work);
    DS_LAUNCH_CHECK("readout_pre_kernel");
    int rc = ds_osc_bank_fwd(work, work + m, nullptr, force, 1, m, F, S, sr, audio, stream);
    if (rc != DS_OK) return rc;
    float* gy = fwork;
    readout_loss_kernel<<<1, 1024, 0, st>>>(audio, target, S, gy, out);
    DS_LAUNCH_CHECK("readout_loss_kernel");
    if (!backward) return DS_OK;
    rc = ds_osc_bank_bwd(gy, work, work + m, nullptr, force, 1, m, F, S, sr, fwork + S, work + 4 * (int64_t)m,
                         work + 5 * (int64_t)m, nullptr, stream);
    if (rc != DS_OK) return rc;
    readout_post_kernel<<<1, 64, 0, st>>>(a_lam, b_mu, m, beta, dlam_dE, dlam_dnu, dmu_dE, dmu_dnu, work, out);
    DS_LAUNCH_CHECK("readout_post_kernel");
    return DS_OK;
}
