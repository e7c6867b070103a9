// Internal helpers shared by the translation units of libdiffsound_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

#include "diffsound_hip.h"

namespace ds {

void set_error(const char* fmt, ...);

inline int check_hip(hipError_t e, const char* what) {
    if (e == hipSuccess) return DS_OK;
    set_error("%s: %s", what, hipGetErrorString(e));
    return DS_ERR_HIP;
}

inline hipStream_t as_stream(ds_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Workgroups are dealt round-robin over the 8 XCDs (blocks b and b+8 share one L2).  Give each
// XCD a contiguous slice of the logical work list so neighbouring rows share an L2 (bijective for
// any grid size).  Speed only, never correctness.
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nblk) {
    const unsigned q = nblk >> 3, r = nblk & 7u;
    const unsigned xcd = bid & 7u, idx = bid >> 3;
    const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

// Launch timing hook (include/diffsound_hip.h: ds_profile_stream / ds_profile_kinds / ds_profile_collect).  A scope brackets
// the launches issued inside it with HIP events when `stream` is the registered one and `kind` is enabled - else it does
// nothing (one relaxed load).  Kinds: DS_PROF_* of the public header; (a, b, c, d) = the launch's shape as that header lists.
struct ProfScope {
    ProfScope(ds_stream_t stream, int kind, int64_t a, int64_t b, int c, int d);
    ~ProfScope();
    ProfScope(const ProfScope&) = delete;
    ProfScope& operator=(const ProfScope&) = delete;
    void* rec_ = nullptr;
    hipStream_t st_ = nullptr;
};

}  // namespace ds

#define DS_REQUIRE(cond, ...)            \
    do {                                 \
        if (!(cond)) {                   \
            ds::set_error(__VA_ARGS__);  \
            return DS_ERR_ARG;           \
        }                                \
    } while (0)

#define DS_LAUNCH_CHECK(name)                                         \
    do {                                                              \
        int _rc = ds::check_hip(hipGetLastError(), name " launch");   \
        if (_rc != DS_OK) return _rc;                                 \
    } while (0)
