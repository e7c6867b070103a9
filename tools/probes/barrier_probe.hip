// Probe (round 5): what does a device-wide barrier cost on MI355X for a grid shaped like the corner-node level's polynomial - 2 461
// single-wave workgroups, all co-resident - ?  Hierarchical, monotone counters: a wave adds 1 to the counter of its part
// (blockIdx & 7: workgroups are dealt round-robin over the 8 XCDs, so a part's counter stays in one L2); the last arriver of a part
// adds 1 to the global counter; the last of those publishes the generation; everybody polls the generation word.
//   hipcc -O3 --offload-arch=gfx950 tools/probes/barrier_probe.hip -o tools/probes/barrier_probe.bin && tools/probes/barrier_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

struct Bar {
    unsigned part[8 * 32];  // one counter per part, 128 bytes apart
    unsigned global_count[32];
    unsigned gen[32];
    unsigned timeout[32];
};

__device__ __forceinline__ bool grid_barrier(Bar* b, unsigned k, unsigned nparts_expected, const unsigned* expect) {
    // k: index of this barrier (0, 1, ...); counters are monotone: no reset, no ABA
    const unsigned p = blockIdx.x & 7u;
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        const unsigned old = __hip_atomic_fetch_add(&b->part[p * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1u == expect[p] * (k + 1u)) {
            const unsigned g = __hip_atomic_fetch_add(&b->global_count[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (g + 1u == nparts_expected * (k + 1u)) __hip_atomic_store(&b->gen[0], k + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const unsigned long long t0 = __builtin_readcyclecounter();
        while (__hip_atomic_load(&b->gen[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < k + 1u) {
            __builtin_amdgcn_s_sleep(2);
            if (__builtin_readcyclecounter() - t0 > 400000000ull) {  // ~0.2 s: give up instead of hanging the device
                __hip_atomic_store(&b->timeout[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return false;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __builtin_amdgcn_wave_barrier();
    return true;
}

// Variant 2: the fences are issued ONCE PER PART - the last arriver of a part writes the part's L2 back (release) before it
// reports to the global counter; after the generation flips every wave invalidates (acquire; mode 2) or only polls (mode 3: the
// floor: atomics and polling alone).  Correct only if a part's waves share one L2, i.e. if part = the wave's XCD.
__device__ __forceinline__ bool grid_barrier2(Bar* b, unsigned k, unsigned nparts_expected, const unsigned* expect, int mode) {
    const unsigned p = blockIdx.x & 7u;
    if (threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned old = __hip_atomic_fetch_add(&b->part[p * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1u == expect[p] * (k + 1u)) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            const unsigned g = __hip_atomic_fetch_add(&b->global_count[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (g + 1u == nparts_expected * (k + 1u)) __hip_atomic_store(&b->gen[0], k + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const unsigned long long t0 = __builtin_readcyclecounter();
        while (__hip_atomic_load(&b->gen[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < k + 1u) {
            __builtin_amdgcn_s_sleep(1);
            if (__builtin_readcyclecounter() - t0 > 400000000ull) {
                __hip_atomic_store(&b->timeout[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return false;
            }
        }
        if (mode == 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __builtin_amdgcn_wave_barrier();
    return true;
}

__global__ void __launch_bounds__(64) probe_kernel2(Bar* b, int nbar, const unsigned* expect, unsigned nparts, float* data, int work, int mode) {
    float v = (float)threadIdx.x;
    for (int k = 0; k < nbar; ++k) {
        for (int w = 0; w < work; ++w) {
            float* p = data + ((size_t)blockIdx.x * 64 + threadIdx.x + (size_t)w * 4096) % (1u << 22);
            v += *p;
            *p = v * 0.5f;
        }
        if (!grid_barrier2(b, (unsigned)k, nparts, expect, mode)) return;
    }
    if (v == -1.f) data[0] = v;
}

__global__ void __launch_bounds__(64) probe_kernel(Bar* b, int nbar, const unsigned* expect, unsigned nparts, float* data, int work) {
    float v = (float)threadIdx.x;
    for (int k = 0; k < nbar; ++k) {
        // a little "work" between barriers: a dependent global read-modify-write per lane (work = 0: barriers only)
        for (int w = 0; w < work; ++w) {
            float* p = data + ((size_t)blockIdx.x * 64 + threadIdx.x + (size_t)w * 4096) % (1u << 22);
            v += *p;
            *p = v * 0.5f;
        }
        if (!grid_barrier(b, (unsigned)k, nparts, expect)) return;
    }
    if (v == -1.f) data[0] = v;
}

int main(int argc, char** argv) {
    const int grid = argc > 1 ? atoi(argv[1]) : 2461;
    Bar* b;
    unsigned* expect;
    float* data;
    hipMalloc(&b, sizeof(Bar));
    hipMalloc(&expect, 8 * sizeof(unsigned));
    hipMalloc(&data, (1u << 22) * sizeof(float));
    hipMemset(data, 0, (1u << 22) * sizeof(float));
    std::vector<unsigned> ex(8, 0);
    for (int i = 0; i < grid; ++i) ex[i & 7]++;
    unsigned nparts = 0;
    for (unsigned e : ex) nparts += e > 0;
    hipMemcpy(expect, ex.data(), 8 * sizeof(unsigned), hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int work : {0, 1}) {
        for (int nbar : {1, 21, 101}) {
            float best = 1e30f;
            for (int rep = 0; rep < 5; ++rep) {
                hipMemset(b, 0, sizeof(Bar));
                hipEventRecord(e0, 0);
                probe_kernel<<<grid, 64, 0, 0>>>(b, nbar, expect, nparts, data, work);
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                float ms = 0;
                hipEventElapsedTime(&ms, e0, e1);
                best = ms < best ? ms : best;
            }
            Bar h;
            hipMemcpy(&h, b, sizeof(Bar), hipMemcpyDeviceToHost);
            printf("grid %d, work %d, %3d barriers: %.1f us per launch (best of 5)%s\n", grid, work, nbar, best * 1e3, h.timeout[0] ? "  TIMEOUT" : "");
        }
    }
    printf("=> per barrier: (t(101) - t(1)) / 100\n");
    for (int mode : {2, 3}) {
        float t1 = 0, t101 = 0;
        for (int nbar : {1, 101}) {
            float best = 1e30f;
            for (int rep = 0; rep < 5; ++rep) {
                hipMemset(b, 0, sizeof(Bar));
                hipEventRecord(e0, 0);
                probe_kernel2<<<grid, 64, 0, 0>>>(b, nbar, expect, nparts, data, 1, mode);
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                float ms = 0;
                hipEventElapsedTime(&ms, e0, e1);
                best = ms < best ? ms : best;
            }
            (nbar == 1 ? t1 : t101) = best;
        }
        Bar h;
        hipMemcpy(&h, b, sizeof(Bar), hipMemcpyDeviceToHost);
        printf("grid %d, fences once per part, mode %d (%s): %.2f us per barrier%s\n", grid, mode,
               mode == 2 ? "last arriver releases, every wave acquires" : "last arriver releases, nobody acquires: the floor",
               (t101 - t1) * 10.0f, h.timeout[0] ? "  TIMEOUT" : "");
    }
    return 0;
}
