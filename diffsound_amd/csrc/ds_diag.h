// Wave-level time stamps for DIAGNOSTIC builds of the SpMM kernels (make EXTRA=-DDS_DIAG; read by tools/m32_diag.py).
// In the shipped library every macro below is empty: the kernels carry the instrumentation POINTS, no instrumentation.
// A record per wave (plain stores - contended atomics cost more than the kernels): [0] start, [1] end (s_memtime, shader
// cycles; the counter is per XCD), [2] the 100 MHz s_memrealtime ticks between them (-> the in-kernel clock), [3..5] three
// accumulators of phase time, [6] start -> mark 0 (head), [7] mark 3 -> end (epilogue).
#pragma once
#ifdef DS_DIAG
#define DS_DIAG_DECL(sym, waves) __device__ unsigned long long sym[(waves) * 8];
#define DS_DIAG_BEGIN()                                                                                     \
    const unsigned long long dg_t0 = __builtin_amdgcn_s_memtime(), dg_r0 = __builtin_amdgcn_s_memrealtime(); \
    unsigned long long dg_acc[3] = {0, 0, 0}, dg_mark[4] = {0, 0, 0, 0}
#define DS_DIAG_MARK(i) do { asm volatile("s_nop 0" ::: "memory"); dg_mark[i] = __builtin_amdgcn_s_memtime(); } while (0)
#define DS_DIAG_ADD(acc, from, to) dg_acc[acc] += dg_mark[to] - dg_mark[from]
#define DS_DIAG_END(sym, waves, lane)                                                                       \
    do {                                                                                                    \
        const unsigned long long t1_ = __builtin_amdgcn_s_memtime(), r1_ = __builtin_amdgcn_s_memrealtime();  \
        if ((lane) == 0 && blockIdx.x < (unsigned)(waves)) {                                                \
            unsigned long long* r_ = sym + (size_t)blockIdx.x * 8;                                          \
            r_[0] = dg_t0, r_[1] = t1_, r_[2] = r1_ - dg_r0, r_[3] = dg_acc[0], r_[4] = dg_acc[1], r_[5] = dg_acc[2];  \
            r_[6] = dg_mark[0] - dg_t0, r_[7] = t1_ - dg_mark[3];                                           \
        }                                                                                                   \
    } while (0)
#else
#define DS_DIAG_DECL(sym, waves)
#define DS_DIAG_BEGIN() do { } while (0)
#define DS_DIAG_MARK(i) do { } while (0)
#define DS_DIAG_ADD(acc, from, to) do { } while (0)
#define DS_DIAG_END(sym, waves, lane) do { } while (0)
#endif
