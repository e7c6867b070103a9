// What rate does v_mfma_f32_16x16x4_f32 sustain on this device, and at what shader clock?  Register-only kernel: every wave issues
// ITERS x 8 independent MFMAs (8 accumulators, so no MFMA waits for its predecessor); the clock comes from s_memtime (shader
// cycles) against s_memrealtime (100 MHz).  mix_lds_kernel / gram32_partial_kernel issue these MFMAs at 88-102 TF/s when
// everything else is knocked out (profiles/r03_mix_knockout.txt): is that the instruction's rate or the kernels'?
// hipcc --offload-arch=gfx950 -O3 tools/mfma_rate_probe.hip -o tools/mfma_rate_probe.bin && tools/mfma_rate_probe.bin
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

using f4 = __attribute__((ext_vector_type(4))) float;

template <int WHAT>
__global__ void __launch_bounds__(256) spin(int iters, float* out, unsigned long long* clk) {
    f4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
    float a = (float)threadIdx.x * 1e-3f, b = 1.0f + (float)blockIdx.x * 1e-6f;
    const unsigned long long c0 = __builtin_readcyclecounter();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < (WHAT >= 2 ? 0 : iters); ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (WHAT == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
            else {  // plain VALU FMAs: 4 per accumulator
#pragma unroll
                for (int k = 0; k < 4; ++k) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i][k]) : "v"(a), "v"(b));
            }
        }
    }
    if (WHAT >= 2) {  // the MFMA pattern of mix_lds_kernel<5>: 2 x 5 accumulators, operands from 2 + 5 four-component registers,
                      // 40 MFMAs per k-step; WHAT == 3: with the 8 v_cndmask per k-step the kernel has between them
        f4 a2[2] = {f4{a, b, a, b}, f4{b, a, b, a}}, b5[5], acc2[2][5];
        for (int j = 0; j < 5; ++j) b5[j] = f4{a + j, b, a, b + j};
        for (int t = 0; t < 2; ++t)
            for (int j = 0; j < 5; ++j) acc2[t][j] = f4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters / 5; ++it) {
            if (WHAT == 3) {
                const bool kv = it + (int)threadIdx.x >= 0;
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int k = 0; k < 4; ++k) asm volatile("v_cndmask_b32 %0, 0, %0, %1" : "+v"(a2[t][k]) : "s"(__builtin_amdgcn_ballot_w64(kv)));
            }
#pragma unroll
            for (int s_ = 0; s_ < 4; ++s_)
#pragma unroll
                for (int j = 0; j < 5; ++j)
#pragma unroll
                    for (int t = 0; t < 2; ++t)
                        acc2[t][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[t][s_], b5[j][s_], acc2[t][j], 0, 0, 0);
        }
        for (int t = 0; t < 2; ++t)
            for (int j = 0; j < 5; ++j) acc[0] += acc2[t][j];
    }
    const unsigned long long c1 = __builtin_readcyclecounter();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678f) out[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = c1 - c0, clk[1] = r1 - r0;
}

int main() {
    float* out;
    unsigned long long *clk, h[2];
    hipMalloc(&out, 4);
    hipMalloc(&clk, 16);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    for (int wpe = 1; wpe <= 4; wpe *= 2) {      // waves per SIMD
        for (int what = 0; what < 4; ++what) {
            const int iters = what == 1 ? 40000 : 20000;
            const int blocks = cus * wpe;        // 256 threads = 4 waves = one per SIMD
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0, 0);
                if (what == 0) spin<0><<<blocks, 256>>>(iters, out, clk);
                else if (what == 1) spin<1><<<blocks, 256>>>(iters, out, clk);
                else if (what == 2) spin<2><<<blocks, 256>>>(iters, out, clk);
                else spin<3><<<blocks, 256>>>(iters, out, clk);
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
            }
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
            const double waves = (double)blocks * 4;
            const double flops = what == 1 ? waves * iters * 8.0 * 4 * 64 * 2.0 : (what == 0 ? waves * iters * 8.0 * 2048.0 : waves * (iters / 5) * 40.0 * 2048.0);
            printf("%s, %d wave(s) per SIMD: %.3f ms, %.1f TF/s; shader clock %.2f GHz (%llu cycles in %.1f us)\n",
                   what == 0 ? "v_mfma_f32_16x16x4_f32" : what == 1 ? "v_fma_f32             " : what == 2 ? "mfma, mix pattern 2x5 " : "mfma, mix pattern+cnd ", wpe, ms, flops / ms / 1e9,
                   (double)h[0] / ((double)h[1] * 10.0) , h[0], (double)h[1] / 100.0);
        }
    }
    return 0;
}
