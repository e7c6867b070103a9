// Minimal, memory-free probe for the gfx950 finding of DESIGN.md 5b: do MFMA instructions of ONE wave change the results
// of packed-FMA instructions of OTHER waves that share the compute unit?
//   aggressor  mfma_spin<FORM>: a register-only loop of matrix-core instructions (no loads, no stores inside the loop, no
//              LDS): FORM 32 = v_mfma_f32_16x16x32_bf16 (gfx950's double-rate form), 16 = v_mfma_f32_16x16x16_bf16,
//              4 = v_mfma_f32_16x16x4_f32, 0 = the same loop with plain v_fma_f32 instead (control);
//   victim     fma_chain<PACKED>: a register-only chain of v_pk_fma_f32 (the three operand-selection forms the
//              neighbour-union SpMM issues) or of v_fma_f32 on lane-dependent data; the result of a launch is a pure
//              function of its arguments, so any launch that differs from the solo launch bit for bit is a corruption.
// Built by `make -C diffsound_amd/csrc probe` into tests/probes/libmfma_probe.so; drivers: tools/mfma_interference.py (the
// whole matrix) and tests/test_hip_kernels.py::test_no_packed_fp32_and_immunity_to_foreign_mfma.
#include <hip/hip_runtime.h>
#include <cstdint>

using f4 = __attribute__((ext_vector_type(4))) float;
using f2 = __attribute__((ext_vector_type(2))) float;
using bf8v = __attribute__((ext_vector_type(8))) __bf16;
using sh4v = __attribute__((ext_vector_type(4))) short;

template <int FORM>
__global__ void __launch_bounds__(64) mfma_spin(int iters, float* out) {
    const int lane = threadIdx.x;
    f4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = f4{0.f, 0.f, 0.f, 0.f};
    // small exactly representable operands: the accumulators stay finite for any iteration count (products sum to ~0)
    const float av = ((lane * 7 + 3) % 13 - 6) * 0.0078125f, bv = ((lane * 5 + 1) % 11 - 5) * 0.0078125f;
    if constexpr (FORM == 32) {
        bf8v a, b;
#pragma unroll
        for (int k = 0; k < 8; ++k) a[k] = (__bf16)(av * (k & 1 ? 1.f : -1.f)), b[k] = (__bf16)(bv * (k & 2 ? 1.f : -1.f));
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[j], 0, 0, 0);
        }
    } else if constexpr (FORM == 16) {
        sh4v a, b;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            a[k] = __builtin_bit_cast(short, (__bf16)(av * (k & 1 ? 1.f : -1.f)));
            b[k] = __builtin_bit_cast(short, (__bf16)(bv * (k & 2 ? 1.f : -1.f)));
        }
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {  // two per step: the same multiply-accumulate count as one 16x16x32
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(b, a, acc[j], 0, 0, 0);
            }
        }
    } else if constexpr (FORM == 4) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[j], 0, 0, 0);
        }
    } else {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int c = 0; c < 4; ++c) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[j][c]) : "v"(av), "v"(bv));
        }
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    out[(size_t)blockIdx.x * 64 + lane] = s;
}

// acc <- acc * d + x c in a long dependent chain, six accumulator pairs per lane as in the SpMM's inner loop; |d| < 1
// keeps the values bounded, every lane and iteration uses different operands
template <bool PACKED>
__global__ void __launch_bounds__(256) fma_chain(int iters, const float* __restrict__ seed, float* __restrict__ out) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    f2 acc[6];
    const float s0 = seed[t & 4095], s1 = seed[(t * 7 + 1) & 4095];
#pragma unroll
    for (int j = 0; j < 6; ++j) acc[j] = f2{s0 + 0.125f * j, s1 - 0.0625f * j};
    f2 xl = {s0 * 0.5f + 0.25f, s1 * 0.5f - 0.25f}, xh = {s1 * 0.375f, s0 * 0.625f}, c01 = {0.4375f, -0.3125f}, c2p = {0.28125f, 0.f};
    for (int i = 0; i < iters; ++i) {
        if constexpr (PACKED) {
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc[0]) : "v"(xl), "v"(c01));
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc[1]) : "v"(xh), "v"(c01));
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(acc[2]) : "v"(xl), "v"(c01));
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(acc[3]) : "v"(xh), "v"(c01));
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc[4]) : "v"(xl), "v"(c2p));
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc[5]) : "v"(xh), "v"(c2p));
            // damp and rotate the operands (packed as well): the chain never repeats and never grows
            asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(acc[i % 6 == 0 ? 0 : 1]) : "v"(c01));
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(xl) : "v"(acc[2]), "v"(c2p));
            asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(xl) : "v"(c01));
        } else {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[0][h]) : "v"(xl[h]), "v"(c01[0]));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[1][h]) : "v"(xh[h]), "v"(c01[0]));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[2][h]) : "v"(xl[h]), "v"(c01[1]));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[3][h]) : "v"(xh[h]), "v"(c01[1]));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[4][h]) : "v"(xl[h]), "v"(c2p[0]));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[5][h]) : "v"(xh[h]), "v"(c2p[0]));
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                asm volatile("v_mul_f32 %0, %0, %1" : "+v"(acc[i % 6 == 0 ? 0 : 1][h]) : "v"(c01[h]));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(xl[h]) : "v"(acc[2][h]), "v"(c2p[h]));
                asm volatile("v_mul_f32 %0, %0, %1" : "+v"(xl[h]) : "v"(c01[h]));
            }
        }
    }
    float* o = out + (size_t)t * 12;
#pragma unroll
    for (int j = 0; j < 6; ++j) o[2 * j] = acc[j][0], o[2 * j + 1] = acc[j][1];
}

extern "C" int probe_mfma_spin(int form, int iters, int nwaves, float* out, void* stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (form) {
        case 32: mfma_spin<32><<<nwaves, 64, 0, st>>>(iters, out); break;
        case 16: mfma_spin<16><<<nwaves, 64, 0, st>>>(iters, out); break;
        case 4: mfma_spin<4><<<nwaves, 64, 0, st>>>(iters, out); break;
        default: mfma_spin<0><<<nwaves, 64, 0, st>>>(iters, out); break;
    }
    return (int)hipGetLastError();
}

extern "C" int probe_fma_chain(int packed, int iters, int nblocks, const float* seed, float* out, void* stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (packed)
        fma_chain<true><<<nblocks, 256, 0, st>>>(iters, seed, out);
    else
        fma_chain<false><<<nblocks, 256, 0, st>>>(iters, seed, out);
    return (int)hipGetLastError();
}
