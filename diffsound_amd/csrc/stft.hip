// Multi-scale spectral loss head - gfx950: Hann-windowed STFT power spectrogram (forward + backward) and the
// weighted-L1 / log-RMSE reductions of the reference's SSSLoss (src/ddsp/mss_loss.py:50-62, 70-122), consuming the
// oscillator bank's audio where it lies in HBM.
//
// The reference computes  S = torchaudio.transforms.Spectrogram(n_fft, hop_length = n_fft / 4)(x)  - i.e.
// torch.stft(center=True, pad_mode="reflect", window = periodic Hann, onesided) followed by |.|^2 (power = 2,
// not normalised) - for n_fft = 32 ... 2048 on clips of S = 8000 samples, once for the prediction and once for the
// target, per scale and per training step, and then takes element-wise logs and weighted means over the
// (batch, n_fft / 2 + 1, 1 + S / hop) arrays: ~12 full-size temporaries per scale.  Here:
//   ds_stft_power      one workgroup per (frame, clip): the reflect-padded, windowed frame and a twiddle table go
//                      to LDS, every thread evaluates its bins by a direct DFT (n_fft <= 2048: at most 2.1 M
//                      multiply-adds per frame, fp64 accumulation so the low-level bins that the log loss looks at
//                      carry no summation noise) and writes P = re^2 + im^2 (and re, im when a backward will follow);
//   ds_spec_loss       |w_t (log2(P_p + eps) - log2(P_t + eps))|, |w_t (P_p - P_t)| or the squared log difference,
//                      summed per workgroup in a fixed order (deterministic), plus d loss / d P_p;
//   ds_stft_power_bwd  d loss / d frame = window * sum_k g_k (2 re_k cos - 2 im_k sin) (the transposed DFT, same
//                      structure), then overlap-add and the fold of the reflect padding as a GATHER per sample
//                      (no atomics, deterministic).
// Frames: T = 1 + S / hop, frame t covers padded samples [t hop, t hop + n_fft), padded sample p = x[reflect(p - n_fft/2)].
#include <algorithm>
#include <cmath>

#include "ds_common.h"

namespace {

constexpr int STFT_MAXN = 2048;

__device__ __forceinline__ int reflect_index(int i, int S) {  // i in [-S+1, 2S-2]
    if (i < 0) i = -i;
    if (i >= S) i = 2 * (S - 1) - i;
    return i;
}

__global__ void __launch_bounds__(256)
    stft_power_kernel(const float* __restrict__ x, int S, int N, int hop, int T, float* __restrict__ P,
                      float* __restrict__ re_out, float* __restrict__ im_out) {
    __shared__ float s_x[STFT_MAXN];
    __shared__ float s_c[STFT_MAXN];
    __shared__ float s_s[STFT_MAXN];
    const int t = blockIdx.x, b = blockIdx.y;
    const int F = N / 2 + 1, pad = N / 2;
    const float* xb = x + (int64_t)b * S;
    for (int n = threadIdx.x; n < N; n += 256) {
        double sn, cs;
        sincospi(2.0 * n / N, &sn, &cs);
        s_c[n] = (float)cs;
        s_s[n] = (float)sn;
        const float w = (float)(0.5 - 0.5 * cs);  // periodic Hann
        s_x[n] = w * xb[reflect_index(t * hop + n - pad, S)];
    }
    __syncthreads();
    const int mask = N - 1;  // N is a power of two
    for (int k = threadIdx.x; k < F; k += 256) {
        double re = 0.0, im = 0.0;
        int ph = 0;
        for (int n = 0; n < N; ++n) {
            const double xv = (double)s_x[n];
            re = fma(xv, (double)s_c[ph], re);
            im = fma(-xv, (double)s_s[ph], im);
            ph = (ph + k) & mask;
        }
        const int64_t o = ((int64_t)b * F + k) * T + t;
        P[o] = (float)(re * re + im * im);
        if (re_out) {
            re_out[o] = (float)re;
            im_out[o] = (float)im;
        }
    }
}

// gframe[b, t, n] = w[n] sum_k gP[b, k, t] (2 re cos(2 pi k n / N) - 2 im sin(2 pi k n / N))
__global__ void __launch_bounds__(256)
    stft_bwd_frames_kernel(const float* __restrict__ gP, const float* __restrict__ re, const float* __restrict__ im,
                           int N, int T, float gscale, float* __restrict__ gframes) {
    __shared__ float s_gr[STFT_MAXN / 2 + 1];
    __shared__ float s_gi[STFT_MAXN / 2 + 1];
    __shared__ float s_c[STFT_MAXN];
    __shared__ float s_s[STFT_MAXN];
    const int t = blockIdx.x, b = blockIdx.y;
    const int F = N / 2 + 1;
    for (int n = threadIdx.x; n < N; n += 256) {
        double sn, cs;
        sincospi(2.0 * n / N, &sn, &cs);
        s_c[n] = (float)cs;
        s_s[n] = (float)sn;
    }
    for (int k = threadIdx.x; k < F; k += 256) {
        const int64_t o = ((int64_t)b * F + k) * T + t;
        const float g = 2.f * gscale * gP[o];
        s_gr[k] = g * re[o];
        s_gi[k] = g * im[o];
    }
    __syncthreads();
    const int mask = N - 1;
    for (int n = threadIdx.x; n < N; n += 256) {
        double acc = 0.0;
        int ph = 0;
        for (int k = 0; k < F; ++k) {
            acc = fma((double)s_gr[k], (double)s_c[ph], acc);
            acc = fma(-(double)s_gi[k], (double)s_s[ph], acc);
            ph = (ph + n) & mask;
        }
        const float w = 0.5f - 0.5f * s_c[n];
        gframes[((int64_t)b * T + t) * N + n] = w * (float)acc;
    }
}

// gx[b, s] = sum over the padded positions p that read x[s] (the direct one and up to two reflections) of the
// frames covering p
__global__ void __launch_bounds__(256)
    stft_bwd_fold_kernel(const float* __restrict__ gframes, int S, int N, int hop, int T, float* __restrict__ gx) {
    const int s = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (s >= S) return;
    const int pad = N / 2;
    const float* gf = gframes + (int64_t)b * T * N;
    float acc = 0.f;
    int pos[3];
    int np = 0;
    pos[np++] = pad + s;
    if (s >= 1 && s <= pad) pos[np++] = pad - s;
    if (s <= S - 2 && s >= S - 1 - pad) pos[np++] = pad + 2 * (S - 1) - s;
    for (int q = 0; q < np; ++q) {
        const int p = pos[q];
        int t0 = p - N + 1;
        t0 = t0 <= 0 ? 0 : (t0 + hop - 1) / hop;
        const int t1 = min(T - 1, p / hop);
        for (int t = t0; t <= t1; ++t) acc += gf[(int64_t)t * N + (p - t * hop)];
    }
    gx[(int64_t)b * S + s] = acc;
}

// kind 0 (weighted L1, reference weighted_l1_loss on log2 and linear power, DC bin dropped):
//   sums[blk][0] = sum |w_t (log2(Pp + eps) - log2(Pt + eps))|, sums[blk][1] = sum |w_t (Pp - Pt)| over f >= 1
//   gP = w_t (alpha sign(dlog) / ((Pp + eps) ln 2) + sign(dlin)) / count,  count = B (F - 1) T
// kind 1 (log RMSE on the first fclip bins, DC included): sums[blk][0] = sum (log2(Pp + eps) - log2(Pt + eps))^2
//   gP = dlog / ((Pp + eps) ln 2 count)          (the caller divides by the RMSE), count = B fclip T
// One workgroup per (b, f) row; time weights w_t = (1 - linspace(1, 0.9, T)) T / sum.
__global__ void __launch_bounds__(256)
    spec_loss_kernel(int kind, const float* __restrict__ Pp, const float* __restrict__ Pt, int F, int T, float alpha,
                     float eps, int fclip, double inv_count, double* __restrict__ sums, float* __restrict__ gP) {
    const int f = blockIdx.x, b = blockIdx.y;
    const int64_t row = ((int64_t)b * F + f) * T;
    const bool in = kind == 0 ? (f >= 1) : (f < fclip);
    double a0 = 0.0, a1 = 0.0;
    // weights: 1 - (1 - 0.1 t / (T - 1)) = 0.1 t / (T - 1); sum = 0.05 T  ->  w_t = 2 t / (T - 1)   (T = 1: 0/0 in the reference)
    const double wnorm = T > 1 ? 2.0 / (T - 1) : 0.0;
    const float il2 = 1.4426950408889634f;  // 1 / ln 2
    for (int t = threadIdx.x; t < T; t += 256) {
        float g = 0.f;
        if (in) {
            const float pp = Pp[row + t], pt = Pt[row + t];
            const float dlog = log2f(pp + eps) - log2f(pt + eps);
            if (kind == 0) {
                const float w = (float)(wnorm * t);
                const float dlin = pp - pt;
                a0 += fabs((double)(w * dlog));
                a1 += fabs((double)(w * dlin));
                const float sl = dlog > 0.f ? 1.f : (dlog < 0.f ? -1.f : 0.f);
                const float sn = dlin > 0.f ? 1.f : (dlin < 0.f ? -1.f : 0.f);
                g = w * (alpha * sl * il2 / (pp + eps) + sn) * (float)inv_count;
            } else {
                a0 += (double)dlog * (double)dlog;
                g = dlog * il2 / (pp + eps) * (float)inv_count;
            }
        }
        if (gP) gP[row + t] = g;
    }
    // fixed-order reduction: wave shuffles, then the four waves through LDS
    for (int o = 32; o > 0; o >>= 1) {
        a0 += __shfl_down(a0, o, 64);
        a1 += __shfl_down(a1, o, 64);
    }
    __shared__ double part[4][2];
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        part[wave][0] = a0;
        part[wave][1] = a1;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double* o = sums + ((int64_t)b * F + f) * 2;
        o[0] = (part[0][0] + part[1][0]) + (part[2][0] + part[3][0]);
        o[1] = (part[0][1] + part[1][1]) + (part[2][1] + part[3][1]);
    }
}

bool stft_shape_ok(int B, int S, int n_fft, int hop) {
    return B > 0 && S > 1 && n_fft >= 8 && n_fft <= STFT_MAXN && (n_fft & (n_fft - 1)) == 0 && hop > 0 && hop <= n_fft &&
           n_fft / 2 < S && B < 65536;
}

}  // namespace

extern "C" int ds_stft_power(const float* x, int B, int S, int n_fft, int hop, float* P, float* re, float* im,
                             ds_stream_t stream) {
    DS_REQUIRE(x && P, "ds_stft_power: null pointer");
    DS_REQUIRE((re == nullptr) == (im == nullptr), "ds_stft_power: re and im go together");
    DS_REQUIRE(stft_shape_ok(B, S, n_fft, hop),
               "ds_stft_power: need n_fft a power of two in [8, 2048], 0 < hop <= n_fft, n_fft / 2 < S (reflect padding)");
    const int T = 1 + S / hop;
    stft_power_kernel<<<dim3((unsigned)T, (unsigned)B), 256, 0, ds::as_stream(stream)>>>(x, S, n_fft, hop, T, P, re, im);
    DS_LAUNCH_CHECK("stft_power_kernel");
    return DS_OK;
}

extern "C" int ds_stft_power_bwd(const float* gP, const float* re, const float* im, int B, int S, int n_fft, int hop,
                                 float gscale, float* gframes, float* gx, ds_stream_t stream) {
    DS_REQUIRE(gP && re && im && gframes && gx, "ds_stft_power_bwd: null pointer");
    DS_REQUIRE(stft_shape_ok(B, S, n_fft, hop), "ds_stft_power_bwd: bad shape (see ds_stft_power)");
    const int T = 1 + S / hop;
    hipStream_t st = ds::as_stream(stream);
    stft_bwd_frames_kernel<<<dim3((unsigned)T, (unsigned)B), 256, 0, st>>>(gP, re, im, n_fft, T, gscale, gframes);
    DS_LAUNCH_CHECK("stft_bwd_frames_kernel");
    stft_bwd_fold_kernel<<<dim3((unsigned)ds::ceil_div(S, 256), (unsigned)B), 256, 0, st>>>(gframes, S, n_fft, hop, T, gx);
    DS_LAUNCH_CHECK("stft_bwd_fold_kernel");
    return DS_OK;
}

extern "C" int ds_spec_loss(int kind, const float* Pp, const float* Pt, int B, int F, int T, float alpha, float eps,
                            int fclip, double* sums, float* gP, ds_stream_t stream) {
    DS_REQUIRE(Pp && Pt && sums, "ds_spec_loss: null pointer");
    DS_REQUIRE(kind == 0 || kind == 1, "ds_spec_loss: kind must be 0 (weighted L1) or 1 (log RMSE)");
    DS_REQUIRE(B > 0 && B < 65536 && F > 1 && T > 0 && fclip >= 0 && fclip <= F, "ds_spec_loss: bad shape");
    const double count = kind == 0 ? (double)B * (F - 1) * T : (double)B * fclip * T;
    DS_REQUIRE(count > 0, "ds_spec_loss: empty reduction");
    spec_loss_kernel<<<dim3((unsigned)F, (unsigned)B), 256, 0, ds::as_stream(stream)>>>(kind, Pp, Pt, F, T, alpha, eps,
                                                                                        fclip, 1.0 / count, sums, gP);
    DS_LAUNCH_CHECK("spec_loss_kernel");
    return DS_OK;
}
