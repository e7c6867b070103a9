// Damped-oscillator bank, forward and backward - gfx950.
//
// Reference: TraditionalDampedOscillator.forward / DampedOscillator.forward
// (src/ddsp/oscillator.py:282-310, 113-141) materialise (A, m, S) tensors five times (repeat, two
// cumsums, exp, sin), sum over modes and call a grouped conv1d.  Here each mode is time-stepped by
// ONE WAVEFRONT: lane c owns a contiguous run of samples, seeds the complex phasor
// z = exp((-d + i w) tau) once in fp64 (closed form, so the phase never accumulates fp32 cumsum
// error) and advances it with one complex multiply per sample; the mode sum stays in registers,
// the 4 waves of a workgroup merge through LDS, and the causal FIR with the force runs out of the
// same LDS tile.  Nothing of size (A, m, S) ever touches HBM.
#include <algorithm>

#include "ds_common.h"

namespace {

constexpr int TILE = 1024;    // output samples per workgroup
constexpr int MAXF = 512;     // max force taps
constexpr int MAXCH = 24;     // max samples per lane per tile: ceil((TILE + MAXF - 1) / 64)

__global__ void __launch_bounds__(256)
    osc_fwd_kernel(const double* __restrict__ dd, const double* __restrict__ ww, const float* __restrict__ amp,
                   const float* __restrict__ force, int m, int F, int S, double inv_sr, float* __restrict__ y) {
    __shared__ float s_sig[4][TILE + MAXF];
    __shared__ float s_force[MAXF];
    const int a = blockIdx.y;
    const int t0 = blockIdx.x * TILE;
    const int H = F - 1;
    const int L = min(TILE, S - t0) + H;  // samples t0-H .. t0+tile-1 (negative times are zero)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int CH = (L + 63) / 64;
    for (int i = threadIdx.x; i < F; i += blockDim.x) s_force[i] = force[(int64_t)a * F + i];

    double accum[MAXCH];
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) accum[i] = 0.0;
    const int l0 = lane * CH;             // first local sample of this lane
    const int tg0 = t0 - H + l0;          // its global sample index
    for (int mm = wave; mm < m; mm += 4) {
        const double d = dd[mm], w = ww[mm];
        const double am = amp ? (double)amp[(int64_t)a * m + mm] : 1.0;
        // seed at tau = (tg0 + 1)/sr ; step multiplier q = exp((-d + i w)/sr)
        const double tau = (double)(tg0 + 1) * inv_sr;
        double sn, cs;
        sincos(w * tau, &sn, &cs);
        const double e = exp(-d * tau) * am;
        double zr = e * cs, zi = e * sn;
        double qs, qc;
        sincos(w * inv_sr, &qs, &qc);
        const double qe = exp(-d * inv_sr);
        const double qr = qe * qc, qi = qe * qs;
#pragma unroll
        for (int i = 0; i < MAXCH; ++i) {
            if (i < CH) {
                accum[i] += zi;
                const double nr = zr * qr - zi * qi;
                zi = zr * qi + zi * qr;
                zr = nr;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {
        const int l = l0 + i;
        if (i < CH && l < L) s_sig[wave][l] = (tg0 + i >= 0) ? (float)accum[i] : 0.f;
    }
    __syncthreads();
    for (int l = threadIdx.x; l < L; l += blockDim.x) s_sig[0][l] = (s_sig[0][l] + s_sig[1][l]) + (s_sig[2][l] + s_sig[3][l]);
    __syncthreads();
    const int nout = min(TILE, S - t0);
    for (int j = threadIdx.x; j < nout; j += blockDim.x) {
        float acc = 0.f;
        const float* sp = &s_sig[0][H + j];
        for (int f = 0; f < F; ++f) acc = fmaf(s_force[f], sp[-f], acc);
        y[(int64_t)a * S + t0 + j] = acc;
    }
}

// gs[a,t] = sum_f force[a,f] gy[a,t+f]   (adjoint of the causal FIR + crop)
__global__ void osc_bwd_corr_kernel(const float* __restrict__ gy, const float* __restrict__ force, int F, int S,
                                    float* __restrict__ gs) {
    __shared__ float s_force[MAXF];
    const int a = blockIdx.y;
    for (int i = threadIdx.x; i < F; i += blockDim.x) s_force[i] = force[(int64_t)a * F + i];
    __syncthreads();
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= S) return;
    const float* g = gy + (int64_t)a * S;
    float acc = 0.f;
    const int fmax = min(F, S - t);
    for (int f = 0; f < fmax; ++f) acc = fmaf(s_force[f], g[t + f], acc);
    gs[(int64_t)a * S + t] = acc;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// one wavefront per mode: gd = dL/dd_m, gw = dL/dw_m, gamp[a,m] = dL/damp[a,m]
__global__ void __launch_bounds__(64)
    osc_bwd_mode_kernel(const float* __restrict__ gs, const double* __restrict__ dd, const double* __restrict__ ww,
                        const float* __restrict__ amp, int A, int m, int S, double inv_sr, double* __restrict__ gd,
                        double* __restrict__ gw, float* __restrict__ gamp) {
    const int mm = blockIdx.x;
    const int lane = threadIdx.x;
    const double d = dd[mm], w = ww[mm];
    const int CH = (S + 63) / 64;
    const int t_begin = lane * CH, t_end = min(S, t_begin + CH);
    double qs, qc;
    sincos(w * inv_sr, &qs, &qc);
    const double qe = exp(-d * inv_sr);
    const double qr = qe * qc, qi = qe * qs;
    double gd_acc = 0.0, gw_acc = 0.0;
    for (int a = 0; a < A; ++a) {
        const double am = amp ? (double)amp[(int64_t)a * m + mm] : 1.0;
        const float* g = gs + (int64_t)a * S;
        double tau = (double)(t_begin + 1) * inv_sr;
        double sn, cs;
        sincos(w * tau, &sn, &cs);
        const double e = exp(-d * tau);
        double zr = e * cs, zi = e * sn;
        double p = 0.0, qd = 0.0, qw = 0.0;
        for (int t = t_begin; t < t_end; ++t) {
            const double gv = (double)g[t];
            p += gv * zi;               // sum g e sin
            qd -= gv * tau * zi;        // d/dd : -tau e sin
            qw += gv * tau * zr;        // d/dw :  tau e cos
            const double nr = zr * qr - zi * qi;
            zi = zr * qi + zi * qr;
            zr = nr;
            tau += inv_sr;
        }
        p = wave_sum(p);
        gd_acc += am * qd;
        gw_acc += am * qw;
        if (gamp && lane == 0) gamp[(int64_t)a * m + mm] = (float)p;
    }
    gd_acc = wave_sum(gd_acc);
    gw_acc = wave_sum(gw_acc);
    if (lane == 0) {
        gd[mm] = gd_acc;
        gw[mm] = gw_acc;
    }
}

}  // namespace

extern "C" int ds_osc_bank_fwd(const double* d, const double* w, const float* amp, const float* force, int A, int m,
                               int F, int S, double sr, float* y, ds_stream_t stream) {
    DS_REQUIRE(d && w && force && y, "ds_osc_bank_fwd: null pointer");
    DS_REQUIRE(A > 0 && m > 0 && S > 0 && sr > 0, "ds_osc_bank_fwd: empty problem");
    DS_REQUIRE(F >= 1 && F <= MAXF, "ds_osc_bank_fwd: force length %d not in 1..%d", F, MAXF);
    dim3 grid((unsigned)ds::ceil_div(S, TILE), (unsigned)A);
    osc_fwd_kernel<<<grid, 256, 0, ds::as_stream(stream)>>>(d, w, amp, force, m, F, S, 1.0 / sr, y);
    DS_LAUNCH_CHECK("osc_fwd_kernel");
    return DS_OK;
}

extern "C" int ds_osc_bank_bwd(const float* gy, const double* d, const double* w, const float* amp,
                               const float* force, int A, int m, int F, int S, double sr, float* gs, double* gd,
                               double* gw, float* gamp, ds_stream_t stream) {
    DS_REQUIRE(gy && d && w && force && gs && gd && gw, "ds_osc_bank_bwd: null pointer");
    DS_REQUIRE(A > 0 && m > 0 && S > 0 && sr > 0, "ds_osc_bank_bwd: empty problem");
    DS_REQUIRE(F >= 1 && F <= MAXF, "ds_osc_bank_bwd: force length %d not in 1..%d", F, MAXF);
    hipStream_t st = ds::as_stream(stream);
    dim3 grid((unsigned)ds::ceil_div(S, 256), (unsigned)A);
    osc_bwd_corr_kernel<<<grid, 256, 0, st>>>(gy, force, F, S, gs);
    DS_LAUNCH_CHECK("osc_bwd_corr_kernel");
    osc_bwd_mode_kernel<<<(unsigned)m, 64, 0, st>>>(gs, d, w, amp, A, m, S, 1.0 / sr, gd, gw, gamp);
    DS_LAUNCH_CHECK("osc_bwd_mode_kernel");
    return DS_OK;
}
