// STREAM probe (tuning aid for ds_stream_triad): triad a = b + s c and copy a = b with 16-byte accesses, over grid
// sizes, loads in flight and store policy.  hipcc --offload-arch=gfx950 -O3 tools/stream_probe.hip -o /tmp/stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using f4 = __attribute__((ext_vector_type(4))) float;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int U, bool NT, bool COPY>
__global__ void __launch_bounds__(256) k(f4* __restrict__ a, const f4* __restrict__ b, const f4* __restrict__ c, long n4, float s) {
    const long stride = (long)gridDim.x * 256;
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n4; i += U * stride) {
        f4 bv[U], cv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            bv[u] = NT ? __builtin_nontemporal_load(b + i + u * stride) : b[i + u * stride];
            if (!COPY) cv[u] = NT ? __builtin_nontemporal_load(c + i + u * stride) : c[i + u * stride];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const f4 r = COPY ? bv[u] : bv[u] + s * cv[u];
            if (NT) __builtin_nontemporal_store(r, a + i + u * stride); else a[i + u * stride] = r;
        }
    }
    for (; i < n4; i += stride) a[i] = COPY ? b[i] : b[i] + s * c[i];
}

// contiguous-per-block form: every workgroup owns a contiguous chunk (no grid stride)
template <bool NT, bool COPY>
__global__ void __launch_bounds__(256) kc(f4* __restrict__ a, const f4* __restrict__ b, const f4* __restrict__ c, long n4, float s) {
    const long i0 = ((long)blockIdx.x * 256 + threadIdx.x);
    if (i0 < n4) {
        const f4 bv = b[i0];
        const f4 r = COPY ? bv : bv + s * c[i0];
        if (NT) __builtin_nontemporal_store(r, a + i0); else a[i0] = r;
    }
}

int main() {
    const long n = 1L << 28;  // 1 GiB per array
    f4 *a, *b, *c;
    CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMalloc(&c, n * 4));
    CK(hipMemset(a, 0, n * 4)); CK(hipMemset(b, 0, n * 4)); CK(hipMemset(c, 0, n * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const long n4 = n / 4;
    auto run = [&](const char* name, auto launch, double bytes) {
        for (int i = 0; i < 3; ++i) launch();
        hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-44s %8.1f GB/s\n", name, bytes / (ms / 20 * 1e-3) / 1e9);
        return 0;
    };
    char nm[128];
    for (int wgcu : {2, 4, 8, 16, 32}) {
        const unsigned g = 256 * wgcu;
#define V(U, NT, COPY) snprintf(nm, sizeof nm, "%s U=%d nt=%d wg/cu=%d", COPY ? "copy " : "triad", U, NT, wgcu); \
        run(nm, [&] { k<U, NT, COPY><<<g, 256>>>(a, b, c, n4, 0.5f); }, (COPY ? 2.0 : 3.0) * n * 4);
        V(1, false, false) V(2, false, false) V(4, false, false) V(8, false, false)
        V(4, true, false) V(8, true, false)
        V(4, false, true) V(8, false, true) V(4, true, true)
    }
    run("triad one-f4-per-thread", [&] { kc<false, false><<<(unsigned)((n4 + 255) / 256), 256>>>(a, b, c, n4, 0.5f); }, 3.0 * n * 4);
    run("triad one-f4-per-thread nt", [&] { kc<true, false><<<(unsigned)((n4 + 255) / 256), 256>>>(a, b, c, n4, 0.5f); }, 3.0 * n * 4);
    run("copy  one-f4-per-thread", [&] { kc<false, true><<<(unsigned)((n4 + 255) / 256), 256>>>(a, b, c, n4, 0.5f); }, 2.0 * n * 4);
    run("copy  one-f4-per-thread nt", [&] { kc<true, true><<<(unsigned)((n4 + 255) / 256), 256>>>(a, b, c, n4, 0.5f); }, 2.0 * n * 4);
    // hipMemcpy D2D as a cross-check
    run("hipMemcpyAsync D2D", [&] { hipMemcpyAsync(a, b, n * 4, hipMemcpyDeviceToDevice, 0); }, 2.0 * n * 4);
    return 0;
}
