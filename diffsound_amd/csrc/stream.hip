// STREAM triad  a = b + s c  on f32 arrays - gfx950.  Not part of the modal path: the measured HBM bandwidth the
// benchmark quotes the SpMM against (BASELINE.json north_star: "LOBPCG SpMM at >= 40 % of STREAM HBM bandwidth"),
// taken on the same device in the same run.  16 bytes per lane per access, grid-stride over an occupancy-sized grid
// (8 workgroups of 256 threads per CU), 4 independent 16-byte loads per array in flight per lane.
#include <algorithm>

#include "ds_common.h"

namespace {

using f4 = __attribute__((ext_vector_type(4))) float;

__global__ void __launch_bounds__(256)
    stream_triad_kernel(f4* __restrict__ a, const f4* __restrict__ b, const f4* __restrict__ c, int64_t n4, float s) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {
        const f4 b0 = b[i], b1 = b[i + stride], b2 = b[i + 2 * stride], b3 = b[i + 3 * stride];
        const f4 c0 = c[i], c1 = c[i + stride], c2 = c[i + 2 * stride], c3 = c[i + 3 * stride];
        a[i] = b0 + s * c0;
        a[i + stride] = b1 + s * c1;
        a[i + 2 * stride] = b2 + s * c2;
        a[i + 3 * stride] = b3 + s * c3;
    }
    for (; i < n4; i += stride) a[i] = b[i] + s * c[i];
}

}  // namespace

extern "C" int ds_stream_triad(float* a, const float* b, const float* c, int64_t n, float s, ds_stream_t stream) {
    DS_REQUIRE(a && b && c, "ds_stream_triad: null pointer");
    DS_REQUIRE(n > 0 && n % 4 == 0, "ds_stream_triad: n must be a positive multiple of 4");
    const uintptr_t al = reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c);
    DS_REQUIRE((al & 15) == 0, "ds_stream_triad: arrays must be 16-byte aligned");
    const int64_t n4 = n / 4;
    const int64_t blocks = std::min<int64_t>(ds::ceil_div(n4, (int64_t)256), 256 * 8);
    stream_triad_kernel<<<(unsigned)blocks, 256, 0, ds::as_stream(stream)>>>(
        reinterpret_cast<f4*>(a), reinterpret_cast<const f4*>(b), reinterpret_cast<const f4*>(c), n4, s);
    DS_LAUNCH_CHECK("stream_triad_kernel");
    return DS_OK;
}
