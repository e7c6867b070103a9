// STREAM triad  a = b + s c  on f32 arrays - gfx950.  Not part of the modal path: the measured HBM bandwidth the
// benchmark quotes the SpMM against (BASELINE.json north_star: "LOBPCG SpMM at >= 40 % of STREAM HBM bandwidth"),
// taken on the same device in the same run.  One 16-byte piece per thread, non-temporal stores: the fastest of the
// forms tried on MI355X (tools/stream_probe.hip: 6.13 TB/s; grid-stride loops over an occupancy-sized grid with
// 1-8 loads in flight per lane reach 4.5-5.3 TB/s; a float4 copy in the same form 6.37 TB/s).
#include <algorithm>

#include "ds_common.h"

namespace {

using f4 = __attribute__((ext_vector_type(4))) float;

__global__ void __launch_bounds__(256)
    stream_triad_kernel(f4* __restrict__ a, const f4* __restrict__ b, const f4* __restrict__ c, int64_t n4, float s) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) __builtin_nontemporal_store(b[i] + s * c[i], a + i);
}

}  // namespace

extern "C" int ds_stream_triad(float* a, const float* b, const float* c, int64_t n, float s, ds_stream_t stream) {
    DS_REQUIRE(a && b && c, "ds_stream_triad: null pointer");
    DS_REQUIRE(n > 0 && n % 4 == 0, "ds_stream_triad: n must be a positive multiple of 4");
    const uintptr_t al = reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c);
    DS_REQUIRE((al & 15) == 0, "ds_stream_triad: arrays must be 16-byte aligned");
    const int64_t n4 = n / 4;
    const int64_t blocks = ds::ceil_div(n4, (int64_t)256);
    DS_REQUIRE(blocks < ((int64_t)1 << 31), "ds_stream_triad: array too large");
    stream_triad_kernel<<<(unsigned)blocks, 256, 0, ds::as_stream(stream)>>>(
        reinterpret_cast<f4*>(a), reinterpret_cast<const f4*>(b), reinterpret_cast<const f4*>(c), n4, s);
    DS_LAUNCH_CHECK("stream_triad_kernel");
    return DS_OK;
}
