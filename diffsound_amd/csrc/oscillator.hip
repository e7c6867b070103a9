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

// ---------------------------------------------------------------------------------------------------------------
// Time-varying bank (reference GTDampedOscillator.forward with non_linear_rate != 0, src/ddsp/oscillator.py:217-243):
// every (clip, mode) has its own damping rate dmp[a,m,t] and damped frequency frq[a,m,t] per sample; the reference
// takes two cumsums over time of (A, m, S) tensors, exp, sin, the mode sum and a grouped conv1d.  Here a workgroup
// owns 16 modes of one clip (4 waves x 4 modes, carries of the two running sums in registers, fp64) and walks the
// clip in chunks of 64 samples: coalesced loads, wave-level inclusive scans, exp / sin, the 16 modes summed in a fixed
// order (registers, then LDS) into one partial signal per workgroup; a second kernel adds the partials in order and
// applies the force FIR.  Deterministic; nothing of size (A, m, S) is written.
constexpr int TV_MPW = 4;            // modes per wave
constexpr int TV_MPG = 4 * TV_MPW;   // modes per workgroup
constexpr double TWO_PI = 6.283185307179586476925286766559;

__device__ __forceinline__ double wave_scan_incl(double v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const double u = __shfl_up(v, o, 64);
        if (lane >= o) v += u;
    }
    return v;
}

__global__ void __launch_bounds__(256)
    osc_tv_modes_kernel(const float* __restrict__ dmp, const float* __restrict__ frq, const float* __restrict__ amp,
                        int m, int S, double inv_sr, float* __restrict__ part) {
    __shared__ double s_p[4][64];
    const int a = blockIdx.y, g = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m0 = g * TV_MPG + wave * TV_MPW;
    double cd[TV_MPW], cf[TV_MPW], am[TV_MPW];
#pragma unroll
    for (int j = 0; j < TV_MPW; ++j) {
        cd[j] = cf[j] = 0.0;
        am[j] = (m0 + j < m) ? (amp ? (double)amp[(int64_t)a * m + m0 + j] : 1.0) : 0.0;
    }
    float* out = part + ((int64_t)a * gridDim.x + g) * S;
    for (int t0 = 0; t0 < S; t0 += 64) {
        const int t = t0 + lane;
        const bool in = t < S;
        double acc = 0.0;
#pragma unroll
        for (int j = 0; j < TV_MPW; ++j) {
            if (m0 + j < m) {  // wave-uniform
                const int64_t o = ((int64_t)a * m + m0 + j) * S + t;
                const double dv = in ? (double)dmp[o] * inv_sr : 0.0;
                const double fv = in ? (double)frq[o] * inv_sr : 0.0;
                const double D = cd[j] + wave_scan_incl(dv, lane);
                const double P = cf[j] + wave_scan_incl(fv, lane);
                cd[j] = __shfl(D, 63, 64);
                cf[j] = __shfl(P, 63, 64);
                cf[j] -= floor(cf[j]);  // whole turns do not change the sine: keep the running phase small
                acc += am[j] * exp(-D) * sin(TWO_PI * (P - floor(P)));
            }
        }
        s_p[wave][lane] = acc;
        __syncthreads();
        if (wave == 0 && in) out[t] = (float)((s_p[0][lane] + s_p[1][lane]) + (s_p[2][lane] + s_p[3][lane]));
        __syncthreads();
    }
}

// y[a,t] = sum_f force[a,f] s[a,t-f],  s = sum of the ng partial signals in order
__global__ void __launch_bounds__(256)
    osc_tv_fir_kernel(const float* __restrict__ part, int ng, const float* __restrict__ force, int F, int S,
                      float* __restrict__ y) {
    __shared__ float s_sig[TILE + MAXF];
    __shared__ float s_force[MAXF];
    const int a = blockIdx.y;
    const int t0 = blockIdx.x * TILE;
    const int H = F - 1;
    const int L = min(TILE, S - t0) + H;
    for (int i = threadIdx.x; i < F; i += blockDim.x) s_force[i] = force[(int64_t)a * F + i];
    for (int l = threadIdx.x; l < L; l += blockDim.x) {
        const int t = t0 - H + l;
        float v = 0.f;
        if (t >= 0)
            for (int g = 0; g < ng; ++g) v += part[((int64_t)a * ng + g) * S + t];
        s_sig[l] = v;
    }
    __syncthreads();
    const int nout = min(TILE, S - t0);
    for (int j = threadIdx.x; j < nout; j += blockDim.x) {
        float acc = 0.f;
        const float* sp = &s_sig[H + j];
        for (int f = 0; f < F; ++f) acc = fmaf(s_force[f], sp[-f], acc);
        y[(int64_t)a * S + t0 + j] = acc;
    }
}

// one wavefront per (clip, mode):  with  u_t = gs_t amp e^{-D_t},  gD_t = -u_t sin(2 pi P_t),  gP_t = 2 pi u_t cos(2 pi P_t)
//   g_dmp[t] = (1/sr) sum_{t' >= t} gD_t' ,  g_frq[t] = (1/sr) sum_{t' >= t} gP_t' ,  gamp = sum_t gs_t e^{-D_t} sin(2 pi P_t)
// (D, P are inclusive running sums, so sample t feeds every later one).  Two sweeps: totals first, then
// suffix = total - exclusive prefix.
__global__ void __launch_bounds__(64)
    osc_tv_bwd_kernel(const float* __restrict__ gs, const float* __restrict__ dmp, const float* __restrict__ frq,
                      const float* __restrict__ amp, int m, int S, double inv_sr, float* __restrict__ g_dmp,
                      float* __restrict__ g_frq, float* __restrict__ gamp) {
    const int mm = blockIdx.x, a = blockIdx.y, lane = threadIdx.x;
    const double am = amp ? (double)amp[(int64_t)a * m + mm] : 1.0;
    const int64_t base = ((int64_t)a * m + mm) * S;
    const float* g = gs + (int64_t)a * S;
    double totD = 0.0, totP = 0.0, ga = 0.0;
    for (int sweep = 0; sweep < 2; ++sweep) {
        double cd = 0.0, cf = 0.0, pD = 0.0, pP = 0.0;  // carries of the forward sums and of the gradient prefixes
        for (int t0 = 0; t0 < S; t0 += 64) {
            const int t = t0 + lane;
            const bool in = t < S;
            const double dv = in ? (double)dmp[base + t] * inv_sr : 0.0;
            const double fv = in ? (double)frq[base + t] * inv_sr : 0.0;
            const double D = cd + wave_scan_incl(dv, lane);
            const double P = cf + wave_scan_incl(fv, lane);
            cd = __shfl(D, 63, 64);
            cf = __shfl(P, 63, 64);
            cf -= floor(cf);
            double sn, cs;
            sincos(TWO_PI * (P - floor(P)), &sn, &cs);
            const double e = exp(-D);
            const double gv = in ? (double)g[t] : 0.0;
            const double u = gv * am * e;
            const double gD = -u * sn, gP = TWO_PI * u * cs;
            if (sweep == 0) {
                totD += gD;
                totP += gP;
                ga += gv * e * sn;
            } else {
                const double iD = wave_scan_incl(gD, lane), iP = wave_scan_incl(gP, lane);
                if (in) {
                    g_dmp[base + t] = (float)((totD - (pD + iD - gD)) * inv_sr);
                    g_frq[base + t] = (float)((totP - (pP + iP - gP)) * inv_sr);
                }
                pD += __shfl(iD, 63, 64);
                pP += __shfl(iP, 63, 64);
            }
        }
        if (sweep == 0) {
            totD = wave_sum(totD);
            totP = wave_sum(totP);
            ga = wave_sum(ga);
            if (gamp && lane == 0) gamp[(int64_t)a * m + mm] = (float)ga;
        }
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

extern "C" int64_t ds_osc_tv_workspace_floats(int A, int m, int S) {
    if (A <= 0 || m <= 0 || S <= 0) return 0;
    return (int64_t)A * ds::ceil_div(m, TV_MPG) * S;
}

extern "C" int ds_osc_tv_fwd(const float* dmp, const float* frq, const float* amp, const float* force, int A, int m,
                             int F, int S, double sr, float* work, float* y, ds_stream_t stream) {
    DS_REQUIRE(dmp && frq && force && work && y, "ds_osc_tv_fwd: null pointer");
    DS_REQUIRE(A > 0 && A < 65536 && m > 0 && S > 0 && sr > 0, "ds_osc_tv_fwd: empty problem");
    DS_REQUIRE(F >= 1 && F <= MAXF, "ds_osc_tv_fwd: force length %d not in 1..%d", F, MAXF);
    hipStream_t st = ds::as_stream(stream);
    const int ng = (int)ds::ceil_div(m, TV_MPG);
    osc_tv_modes_kernel<<<dim3((unsigned)ng, (unsigned)A), 256, 0, st>>>(dmp, frq, amp, m, S, 1.0 / sr, work);
    DS_LAUNCH_CHECK("osc_tv_modes_kernel");
    osc_tv_fir_kernel<<<dim3((unsigned)ds::ceil_div(S, TILE), (unsigned)A), 256, 0, st>>>(work, ng, force, F, S, y);
    DS_LAUNCH_CHECK("osc_tv_fir_kernel");
    return DS_OK;
}

extern "C" int ds_osc_tv_bwd(const float* gy, const float* dmp, const float* frq, const float* amp, const float* force,
                             int A, int m, int F, int S, double sr, float* gs, float* g_dmp, float* g_frq, float* gamp,
                             ds_stream_t stream) {
    DS_REQUIRE(gy && dmp && frq && force && gs && g_dmp && g_frq, "ds_osc_tv_bwd: null pointer");
    DS_REQUIRE(A > 0 && A < 65536 && m > 0 && S > 0 && sr > 0, "ds_osc_tv_bwd: empty problem");
    DS_REQUIRE(F >= 1 && F <= MAXF, "ds_osc_tv_bwd: force length %d not in 1..%d", F, MAXF);
    hipStream_t st = ds::as_stream(stream);
    osc_bwd_corr_kernel<<<dim3((unsigned)ds::ceil_div(S, 256), (unsigned)A), 256, 0, st>>>(gy, force, F, S, gs);
    DS_LAUNCH_CHECK("osc_bwd_corr_kernel");
    osc_tv_bwd_kernel<<<dim3((unsigned)m, (unsigned)A), 64, 0, st>>>(gs, dmp, frq, amp, m, S, 1.0 / sr, g_dmp, g_frq, gamp);
    DS_LAUNCH_CHECK("osc_tv_bwd_kernel");
    return DS_OK;
}
