// Bare v_mfma_f64_16x16x4_f64 issue-rate probe (operands in registers): prints TFLOP/s chip-wide.
#include <hip/hip_runtime.h>
#include <cstdio>
using d4 = __attribute__((ext_vector_type(4))) double;
template <int NACC>
__global__ void __launch_bounds__(256) k(double* out, int iters) {
    d4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = threadIdx.x * 2e-3;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
void run(int wgs_per_cu) {
    const int blocks = 256 * wgs_per_cu, iters = 2000;
    double* out; hipMalloc(&out, blocks * 256 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC><<<blocks, 256>>>(out, 10);
    hipEventRecord(e0); k<NACC><<<blocks, 256>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double fl = (double)blocks * 4 * iters * NACC * 2048.0;
    printf("NACC=%d wg/CU=%d: %.3f ms  %.1f TF/s f64\n", NACC, wgs_per_cu, ms, fl / ms / 1e9);
    hipFree(out);
}
int main() { run<9>(1); run<9>(2); run<9>(3); run<9>(4); run<4>(4); run<4>(8); return 0; }
