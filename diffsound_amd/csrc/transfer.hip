// Level transfer of the two-level (P2 -> P1) preconditioner - gfx950.
//
//   Y_i = beta Y_i + sum_k w_k X_{col_k}          (3 x ncols node panels, scalar weights)
//
// is both directions of the transfer between the quadratic mesh and its corner-node (linear) sub-mesh:
// prolongation (rows = fine nodes, two entries of weight 1/2 per row: a corner copies its coarse value,
// a mid-edge node averages the edge's end points - the P1 function written in the P2 nodal basis,
// reference shape_func.py:14-24 / mesh.py:139-154 give the node layout) and restriction = its transpose
// (rows = coarse nodes, entries = the corner itself with weight 1 and the ~14 mid-edge nodes of its edges
// with weight 1/2).  Pure gather + stream: one thread per 16-byte piece of the output panel.
#include "ds_common.h"

namespace {

__global__ void __launch_bounds__(256)
    scalar_csr_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colidx,
                      const float* __restrict__ w, int64_t nrows, const float* __restrict__ X, int64_t ldx,
                      float* __restrict__ Y, int64_t ldy, int lpn, float beta) {
    using f4 = __attribute__((ext_vector_type(4))) float;
    const int cpn = 3 * lpn;  // 16-byte pieces per node panel
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t node = gid / cpn;
    if (node >= nrows) return;
    const int c = (int)(gid - node * cpn);
    const int r = c / lpn;
    const int c0 = (c - r * lpn) * 4;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    const int kb = rowptr[node], ke = rowptr[node + 1];
    // four independent gathers in flight (see scalar_csr16_kernel); the sum keeps its order
    int k = kb;
    for (; k + 4 <= ke; k += 4) {
        const int64_t j0 = colidx[k], j1 = colidx[k + 1], j2 = colidx[k + 2], j3 = colidx[k + 3];
        const float w0 = w[k], w1 = w[k + 1], w2 = w[k + 2], w3 = w[k + 3];
        const f4 x0 = *reinterpret_cast<const f4*>(X + (3 * j0 + r) * ldx + c0), x1 = *reinterpret_cast<const f4*>(X + (3 * j1 + r) * ldx + c0);
        const f4 x2 = *reinterpret_cast<const f4*>(X + (3 * j2 + r) * ldx + c0), x3 = *reinterpret_cast<const f4*>(X + (3 * j3 + r) * ldx + c0);
        acc += w0 * x0;
        acc += w1 * x1;
        acc += w2 * x2;
        acc += w3 * x3;
    }
    for (; k < ke; ++k) {
        const int64_t j = colidx[k];
        acc += w[k] * *reinterpret_cast<const f4*>(X + (3 * j + r) * ldx + c0);
    }
    float* yp = Y + (3 * node + r) * ldy + c0;
    if (beta != 0.f) acc += beta * *reinterpret_cast<const f4*>(yp);
    *reinterpret_cast<f4*>(yp) = acc;
}

}  // namespace

extern "C" int ds_scalar_csr_spmm(const int32_t* rowptr, const int32_t* colidx, const float* w, int64_t nrows,
                                  const float* X, int64_t ldx, float* Y, int64_t ldy, int ncols, float beta,
                                  ds_stream_t stream) {
    DS_REQUIRE(rowptr && colidx && w && X && Y, "ds_scalar_csr_spmm: null pointer");
    DS_REQUIRE(nrows > 0 && ncols > 0 && ncols % 4 == 0, "ds_scalar_csr_spmm: ncols must be a positive multiple of 4");
    DS_REQUIRE(ldx >= ncols && ldy >= ncols, "ds_scalar_csr_spmm: leading dimension smaller than ncols");
    const uintptr_t al = reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(Y) | (uintptr_t)(ldx * 4) |
                         (uintptr_t)(ldy * 4);
    DS_REQUIRE((al & 15) == 0, "ds_scalar_csr_spmm: rows must be 16-byte aligned");
    DS_REQUIRE(X != Y, "ds_scalar_csr_spmm: X and Y must be different buffers");
    const int lpn = ncols / 4;
    const int64_t threads = nrows * 3 * lpn;
    const int64_t blocks = ds::ceil_div(threads, (int64_t)256);
    DS_REQUIRE(blocks < ((int64_t)1 << 31), "ds_scalar_csr_spmm: grid too large");
    scalar_csr_kernel<<<(unsigned)blocks, 256, 0, ds::as_stream(stream)>>>(rowptr, colidx, w, nrows, X, ldx, Y, ldy, lpn, beta);
    DS_LAUNCH_CHECK("scalar_csr_kernel");
    return DS_OK;
}
