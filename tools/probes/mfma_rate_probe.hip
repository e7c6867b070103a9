// Probe (round 6): issue rate of the matrix instructions the Rayleigh-Ritz kernels could run on, one wave per SIMD,
// register operands only (no memory): v_mfma_f32_16x16x4_f32 (what gram32_partial_kernel / mix_lds_kernel issue today),
// v_mfma_f32_16x16x16_bf16 (the only bf16 form this library allows itself: profiles/r03_mfma_interference_matrix.txt) and, for
// reference, v_mfma_f32_16x16x32_bf16.  A split-precision Gram product (fp32 = three bf16 pieces, six cross products) pays six bf16
// instructions of K = 16 for every four fp32 instructions of K = 4 it replaces: it wins on the matrix pipe iff
// 6 x cycles(16x16x16_bf16) < 4 x cycles(16x16x4_f32).
//   hipcc -O3 --offload-arch=gfx950 tools/probes/mfma_rate_probe.hip -o tools/probes/mfma_rate_probe.bin && tools/probes/mfma_rate_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>

typedef __attribute__((ext_vector_type(4))) float f4;
typedef __attribute__((ext_vector_type(4))) short s4;
typedef __attribute__((ext_vector_type(8))) __bf16 b8;

template <int FORM, int NACC>
__global__ void __launch_bounds__(256) rate_kernel(float* out, unsigned long long* cyc, int iters) {
    f4 acc[NACC];
    for (int j = 0; j < NACC; ++j) acc[j] = {0.f, 0.f, 0.f, 0.f};
    const float x = 1.0f + 1e-3f * (threadIdx.x & 63), y = 0.5f - 1e-3f * (threadIdx.x & 31);
    s4 a16 = {(short)(0x3f80 + (threadIdx.x & 7)), 0x3f00, 0x3e80, 0x3f81}, b16 = {0x3f01, (short)(0x3e00 + (threadIdx.x & 3)), 0x3f40, 0x3ec0};
    b8 a32, b32;
    for (int k = 0; k < 8; ++k) a32[k] = (__bf16)(1.0f + 0.01f * k), b32[k] = (__bf16)(0.5f - 0.01f * k);
    __builtin_amdgcn_s_barrier();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < NACC; ++j) {
            if (FORM == 4) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, acc[j], 0, 0, 0);
            if (FORM == 16) acc[j] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a16, b16, acc[j], 0, 0, 0);
            if (FORM == 32) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a32, b32, acc[j], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int j = 0; j < NACC; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int FORM, int NACC>
void run(const char* name, double flops_per_mfma) {
    const int blocks = 256 * 4, iters = 20000;  // 4 workgroups of 4 waves per CU resident; one wave per SIMD issues at a time per slot
    float* out;
    unsigned long long* cyc;
    hipMalloc(&out, blocks * 256 * sizeof(float));
    hipMalloc(&cyc, blocks * sizeof(unsigned long long));
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    for (int grid : {256, 1024}) {
        rate_kernel<FORM, NACC><<<grid, 256>>>(out, cyc, 200);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        rate_kernel<FORM, NACC><<<grid, 256>>>(out, cyc, iters);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        unsigned long long c0 = 0;
        hipMemcpy(&c0, cyc, sizeof(c0), hipMemcpyDeviceToHost);
        const double n_mfma_wave = (double)iters * NACC;
        const double tf = flops_per_mfma * n_mfma_wave * grid * 4 / (ms * 1e-3) / 1e12;
        printf("%-28s acc %2d  grid %4d (%d waves per SIMD): %7.2f timer ticks per MFMA and wave, %8.1f TFLOP/s chip-wide, %.3f ms\n", name, NACC, grid,
               grid / 256, (double)c0 / n_mfma_wave, tf, ms);
    }
    hipFree(out), hipFree(cyc);
}

int main() {
    run<4, 4>("v_mfma_f32_16x16x4_f32", 2.0 * 16 * 16 * 4);
    run<16, 4>("v_mfma_f32_16x16x16_bf16", 2.0 * 16 * 16 * 16);
    run<16, 1>("v_mfma_f32_16x16x16_bf16", 2.0 * 16 * 16 * 16);
    run<32, 4>("v_mfma_f32_16x16x32_bf16", 2.0 * 16 * 16 * 32);
    return 0;
}
