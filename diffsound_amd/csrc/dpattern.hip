// Symbolic phase ON THE DEVICE - gfx950: BSR-3 pattern, per-slot contribution lists and the neighbour-union tables
// straight from the element connectivity in HBM.  Same outputs, entry for entry, as the host routines ds_pattern_build
// / ds_groups_build (pattern.cpp) and the chunk table of the Python binding; tests/test_hip_kernels.py compares them.
//
// The reference rebuilds its sparsity pattern implicitly on every assembly (COO triplets + coalesce(),
// src/diffelastic/diff_model.py:214-220, 299-312), and the geometry experiments build a NEW mesh every iteration
// (src/dmtet/geometry/dmtet_thickness.py:251), so the symbolic phase is on their critical path: 0.3-0.5 s per
// topology on the host for the 100k-tet ord-2 mesh, against ~40 ms for a whole modal pass.  Here:
//   1. key = row * nv + col for each of the T N^2 (element, a, b) contributions, value = the contribution id;
//      a stable radix sort by key (rocPRIM) groups the contributions of a block slot together in ascending id order;
//   2. run-length encoding of the sorted keys = the blocks (colidx, cptr); rowptr / diagidx by binary search;
//   3. blocks re-keyed by (group of 4 rows, col, row-in-group) and sorted again: runs of equal (group, col) are
//      the union entries (gent = col | presence mask << 28), the sorted order is kperm, run starts are goff;
//   4. chunk table: one chunk (e0, e1, b0, b1) per group when the group fits the SpMM's LDS image (the rule - rare
//      exceptions are reported through `single` so the caller can take the host path for the table alone).
// Sorting and scanning are rocPRIM device primitives; the rest are the small kernels below.
#include <algorithm>
#include <cstring>
#include <new>
#include <vector>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_run_length_encode.hpp>
#include <rocprim/device/device_scan.hpp>

#include "ds_common.h"

struct ds_dpattern {
    int64_t nv = 0, T = 0, nnzb = 0, ncontrib = 0, ne = 0, ngroups = 0;
    int N = 0, cap = 0, single = 0;
    // device arrays (owned)
    int32_t *rowptr = nullptr, *colidx = nullptr, *diagidx = nullptr, *cptr = nullptr, *clist = nullptr;
    int32_t *gptr = nullptr, *gent = nullptr, *goff = nullptr, *kperm = nullptr, *ctab = nullptr;
};

namespace {

#define DP_HIP(x)                                       \
    do {                                                \
        int rc_ = ds::check_hip((x), #x);               \
        if (rc_ != DS_OK) return rc_;                   \
    } while (0)

struct Scratch {  // frees on scope exit
    std::vector<void*> ptrs;
    hipStream_t st;
    explicit Scratch(hipStream_t s) : st(s) {}
    ~Scratch() {
        for (void* p : ptrs) (void)hipFree(p);
    }
    template <typename T>
    int alloc(T** p, int64_t n) {
        void* q = nullptr;
        int rc = ds::check_hip(hipMalloc(&q, (size_t)std::max<int64_t>(n, 1) * sizeof(T)), "hipMalloc(symbolic scratch)");
        if (rc != DS_OK) return rc;
        ptrs.push_back(q);
        *p = static_cast<T*>(q);
        return DS_OK;
    }
};

__global__ void contrib_keys_kernel(const int32_t* __restrict__ tets, int64_t total, int N, int64_t nv,
                                    uint64_t* __restrict__ keys, int32_t* __restrict__ vals) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int nn = N * N;
    const int64_t t = i / nn;
    const int ab = (int)(i - t * nn);
    const int a = ab / N, b = ab - a * N;
    keys[i] = (uint64_t)tets[t * N + a] * (uint64_t)nv + (uint64_t)tets[t * N + b];
    vals[i] = (int32_t)i;
}

__global__ void block_meta_kernel(const uint64_t* __restrict__ ukeys, int64_t nnzb, int64_t nv,
                                  int32_t* __restrict__ colidx, int32_t* __restrict__ diagidx,
                                  uint64_t* __restrict__ gkeys, int32_t* __restrict__ gvals) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nnzb) return;
    const uint64_t key = ukeys[k];
    const uint64_t row = key / (uint64_t)nv, col = key - row * (uint64_t)nv;
    colidx[k] = (int32_t)col;
    if (row == col) diagidx[row] = (int32_t)k;
    gkeys[k] = (((row >> 2) * (uint64_t)nv + col) << 2) | (row & 3u);
    gvals[k] = (int32_t)k;
}

// out[r] = first index i with keys[i] >= r * stride   (r = 0 .. nrows inclusive)
__global__ void lower_bound_rows_kernel(const uint64_t* __restrict__ keys, int64_t n, int64_t nrows, uint64_t stride,
                                        int shift, int32_t* __restrict__ out) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r > nrows) return;
    const uint64_t target = (uint64_t)r * stride;
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if ((keys[mid] >> shift) < target) lo = mid + 1; else hi = mid;
    }
    out[r] = (int32_t)lo;
}

__global__ void fill_kernel(int32_t* __restrict__ p, int64_t n, int32_t v) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

__global__ void shift_keys_kernel(const uint64_t* __restrict__ in, int64_t n, uint64_t* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i] >> 2;
}

// entry e: run [goff[e], goff[e+1]) of the sorted group keys; gent = col | (OR of 1 << row-in-group) << 28
__global__ void entries_kernel(const uint64_t* __restrict__ gkeys_sorted, const int32_t* __restrict__ goff, int64_t ne,
                               int64_t nv, int32_t* __restrict__ gent) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= ne) return;
    const int b0 = goff[e], b1 = goff[e + 1];
    const uint64_t ek = gkeys_sorted[b0] >> 2;
    const uint64_t col = ek % (uint64_t)nv;
    unsigned mask = 0;
    for (int b = b0; b < b1; ++b) mask |= 1u << (unsigned)(gkeys_sorted[b] & 3u);
    gent[e] = (int32_t)((uint32_t)col | (mask << 28));
}

__global__ void chunks_kernel(const int32_t* __restrict__ gptr, const int32_t* __restrict__ goff, int64_t ngroups,
                              int cap, int32_t* __restrict__ ctab, int32_t* __restrict__ not_single) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= ngroups) return;
    const int e0 = gptr[g], e1 = gptr[g + 1];
    const int b0 = goff[e0], b1 = goff[e1];
    ctab[4 * g + 0] = e0;
    ctab[4 * g + 1] = e1;
    ctab[4 * g + 2] = b0;
    ctab[4 * g + 3] = b1;
    if (e1 - e0 > cap || b1 - b0 > cap) atomicOr(not_single, 1);
}

int bits_for(uint64_t max_value) {
    int b = 1;
    while (b < 64 && (max_value >> b) != 0) ++b;
    return b;
}

template <typename T>
int dev_alloc(T** p, int64_t n) {
    void* q = nullptr;
    int rc = ds::check_hip(hipMalloc(&q, (size_t)std::max<int64_t>(n, 1) * sizeof(T)), "hipMalloc(symbolic tables)");
    if (rc == DS_OK) *p = static_cast<T*>(q);
    return rc;
}

int sort_pairs(Scratch& sc, uint64_t* kin, uint64_t* kout, int32_t* vin, int32_t* vout, int64_t n, int bits,
               hipStream_t st) {
    size_t bytes = 0;
    DP_HIP(rocprim::radix_sort_pairs(nullptr, bytes, kin, kout, vin, vout, (size_t)n, 0u, (unsigned)bits, st));
    char* tmp;
    int rc = sc.alloc(&tmp, (int64_t)bytes);
    if (rc != DS_OK) return rc;
    DP_HIP(rocprim::radix_sort_pairs(tmp, bytes, kin, kout, vin, vout, (size_t)n, 0u, (unsigned)bits, st));
    return DS_OK;
}

// unique keys, run lengths -> exclusive offsets (n_runs + 1 entries); *nruns on the host (synchronises the stream)
int runs(Scratch& sc, const uint64_t* keys_sorted, int64_t n, uint64_t* ukeys, int32_t* offsets, int64_t* nruns,
         hipStream_t st) {
    int32_t* counts;
    int64_t* dcount;
    int rc = sc.alloc(&counts, n + 1);
    if (rc != DS_OK) return rc;
    rc = sc.alloc(&dcount, 1);
    if (rc != DS_OK) return rc;
    size_t bytes = 0;
    DP_HIP(rocprim::run_length_encode(nullptr, bytes, keys_sorted, (unsigned int)n, ukeys, counts, dcount, st));
    char* tmp;
    rc = sc.alloc(&tmp, (int64_t)bytes);
    if (rc != DS_OK) return rc;
    DP_HIP(rocprim::run_length_encode(tmp, bytes, keys_sorted, (unsigned int)n, ukeys, counts, dcount, st));
    DP_HIP(hipMemcpyAsync(nruns, dcount, sizeof(int64_t), hipMemcpyDeviceToHost, st));
    DP_HIP(hipStreamSynchronize(st));
    // offsets[0..nruns] = exclusive scan of counts (counts[nruns] is read as the (nruns+1)-th input: zero it first)
    DP_HIP(hipMemsetAsync(counts + *nruns, 0, sizeof(int32_t), st));
    bytes = 0;
    DP_HIP(rocprim::exclusive_scan(nullptr, bytes, counts, offsets, 0, (size_t)(*nruns + 1), rocprim::plus<int32_t>(), st));
    char* tmp2;
    rc = sc.alloc(&tmp2, (int64_t)bytes);
    if (rc != DS_OK) return rc;
    DP_HIP(rocprim::exclusive_scan(tmp2, bytes, counts, offsets, 0, (size_t)(*nruns + 1), rocprim::plus<int32_t>(), st));
    return DS_OK;
}

inline unsigned blocks_for(int64_t n) { return (unsigned)ds::ceil_div(std::max<int64_t>(n, 1), (int64_t)256); }

int build(ds_dpattern* p, const int32_t* tets, hipStream_t st) {
    const int64_t nv = p->nv, T = p->T;
    const int N = p->N;
    Scratch sc(st);
    // ---- 1. contributions sorted by block key
    const int64_t nc = T * N * N;
    p->ncontrib = nc;
    uint64_t *k0, *k1;
    int32_t* v0;
    int rc;
    if ((rc = sc.alloc(&k0, nc)) != DS_OK || (rc = sc.alloc(&k1, nc)) != DS_OK || (rc = sc.alloc(&v0, nc)) != DS_OK) return rc;
    if ((rc = dev_alloc(&p->clist, nc)) != DS_OK) return rc;
    contrib_keys_kernel<<<blocks_for(nc), 256, 0, st>>>(tets, nc, N, nv, k0, v0);
    DS_LAUNCH_CHECK("contrib_keys_kernel");
    const int kbits = bits_for((uint64_t)nv * (uint64_t)nv);
    if ((rc = sort_pairs(sc, k0, k1, v0, p->clist, nc, kbits, st)) != DS_OK) return rc;
    // ---- 2. blocks = runs of equal keys
    uint64_t* ukeys = k0;  // (the unsorted keys are no longer needed)
    int32_t* cptr_tmp;
    if ((rc = sc.alloc(&cptr_tmp, nc + 1)) != DS_OK) return rc;
    if ((rc = runs(sc, k1, nc, ukeys, cptr_tmp, &p->nnzb, st)) != DS_OK) return rc;
    const int64_t nnzb = p->nnzb;
    if ((rc = dev_alloc(&p->cptr, nnzb + 1)) != DS_OK || (rc = dev_alloc(&p->colidx, nnzb)) != DS_OK ||
        (rc = dev_alloc(&p->rowptr, nv + 1)) != DS_OK || (rc = dev_alloc(&p->diagidx, nv)) != DS_OK ||
        (rc = dev_alloc(&p->kperm, nnzb)) != DS_OK)
        return rc;
    DP_HIP(hipMemcpyAsync(p->cptr, cptr_tmp, sizeof(int32_t) * (nnzb + 1), hipMemcpyDeviceToDevice, st));
    fill_kernel<<<blocks_for(nv), 256, 0, st>>>(p->diagidx, nv, -1);
    DS_LAUNCH_CHECK("fill_kernel");
    uint64_t *g0, *g1, *ek;
    int32_t* gv0;
    if ((rc = sc.alloc(&g0, nnzb)) != DS_OK || (rc = sc.alloc(&g1, nnzb)) != DS_OK || (rc = sc.alloc(&ek, nnzb)) != DS_OK ||
        (rc = sc.alloc(&gv0, nnzb)) != DS_OK)
        return rc;
    block_meta_kernel<<<blocks_for(nnzb), 256, 0, st>>>(ukeys, nnzb, nv, p->colidx, p->diagidx, g0, gv0);
    DS_LAUNCH_CHECK("block_meta_kernel");
    lower_bound_rows_kernel<<<blocks_for(nv + 1), 256, 0, st>>>(ukeys, nnzb, nv, (uint64_t)nv, 0, p->rowptr);
    DS_LAUNCH_CHECK("lower_bound_rows_kernel");
    // ---- 3. union entries of the groups of 4 rows
    p->ngroups = (nv + 3) / 4;
    if ((rc = sort_pairs(sc, g0, g1, gv0, p->kperm, nnzb, std::min(64, kbits + 1), st)) != DS_OK) return rc;
    shift_keys_kernel<<<blocks_for(nnzb), 256, 0, st>>>(g1, nnzb, ek);
    DS_LAUNCH_CHECK("shift_keys_kernel");
    uint64_t* uek = g0;
    int32_t* goff_tmp;
    if ((rc = sc.alloc(&goff_tmp, nnzb + 1)) != DS_OK) return rc;
    if ((rc = runs(sc, ek, nnzb, uek, goff_tmp, &p->ne, st)) != DS_OK) return rc;
    const int64_t ne = p->ne;
    if ((rc = dev_alloc(&p->goff, ne + 1)) != DS_OK || (rc = dev_alloc(&p->gent, ne)) != DS_OK ||
        (rc = dev_alloc(&p->gptr, p->ngroups + 1)) != DS_OK || (rc = dev_alloc(&p->ctab, 4 * p->ngroups)) != DS_OK)
        return rc;
    DP_HIP(hipMemcpyAsync(p->goff, goff_tmp, sizeof(int32_t) * (ne + 1), hipMemcpyDeviceToDevice, st));
    entries_kernel<<<blocks_for(ne), 256, 0, st>>>(g1, p->goff, ne, nv, p->gent);
    DS_LAUNCH_CHECK("entries_kernel");
    lower_bound_rows_kernel<<<blocks_for(p->ngroups + 1), 256, 0, st>>>(uek, ne, p->ngroups, (uint64_t)nv, 0, p->gptr);
    DS_LAUNCH_CHECK("lower_bound_rows_kernel(groups)");
    // ---- 4. chunk table
    int32_t* flag;
    if ((rc = sc.alloc(&flag, 1)) != DS_OK) return rc;
    DP_HIP(hipMemsetAsync(flag, 0, sizeof(int32_t), st));
    chunks_kernel<<<blocks_for(p->ngroups), 256, 0, st>>>(p->gptr, p->goff, p->ngroups, p->cap, p->ctab, flag);
    DS_LAUNCH_CHECK("chunks_kernel");
    int32_t hflag = 0;
    DP_HIP(hipMemcpyAsync(&hflag, flag, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    DP_HIP(hipStreamSynchronize(st));
    p->single = hflag ? 0 : 1;
    return DS_OK;
}

}  // namespace

extern "C" void ds_dpattern_free(ds_dpattern_t* p) {
    if (!p) return;
    for (int32_t* q : {p->rowptr, p->colidx, p->diagidx, p->cptr, p->clist, p->gptr, p->gent, p->goff, p->kperm, p->ctab})
        if (q) (void)hipFree(q);
    delete p;
}

extern "C" int ds_dpattern_build(const int32_t* tets, int64_t T, int N, int64_t nv, int cap_blocks, ds_stream_t stream,
                                 ds_dpattern_t** out) {
    DS_REQUIRE(tets && out, "ds_dpattern_build: null argument");
    DS_REQUIRE(T > 0 && (N == 4 || N == 10), "ds_dpattern_build: need T > 0 and N in {4, 10}");
    DS_REQUIRE(nv > 0 && nv < ((int64_t)1 << 28), "ds_dpattern_build: nv must be in (0, 2^28)");
    DS_REQUIRE(T * N * N < ((int64_t)1 << 31), "ds_dpattern_build: more than 2^31 contributions");
    DS_REQUIRE(cap_blocks > 0, "ds_dpattern_build: cap_blocks must be positive");
    auto* p = new (std::nothrow) ds_dpattern;
    if (!p) {
        ds::set_error("ds_dpattern_build: out of memory");
        return DS_ERR_NOMEM;
    }
    p->nv = nv, p->T = T, p->N = N, p->cap = cap_blocks;
    const int rc = build(p, tets, ds::as_stream(stream));
    if (rc != DS_OK) {
        ds_dpattern_free(p);
        return rc;
    }
    *out = p;
    return DS_OK;
}

extern "C" int ds_dpattern_sizes(const ds_dpattern_t* p, int64_t* nnzb, int64_t* ncontrib, int64_t* ne,
                                 int64_t* ngroups, int* single) {
    DS_REQUIRE(p != nullptr, "ds_dpattern_sizes: null handle");
    if (nnzb) *nnzb = p->nnzb;
    if (ncontrib) *ncontrib = p->ncontrib;
    if (ne) *ne = p->ne;
    if (ngroups) *ngroups = p->ngroups;
    if (single) *single = p->single;
    return DS_OK;
}

extern "C" int ds_dpattern_export(const ds_dpattern_t* p, int32_t* rowptr, int32_t* colidx, int32_t* diagidx,
                                  int32_t* cptr, int32_t* clist, int32_t* gptr, int32_t* gent, int32_t* goff,
                                  int32_t* kperm, int32_t* ctab, ds_stream_t stream) {
    DS_REQUIRE(p != nullptr, "ds_dpattern_export: null handle");
    hipStream_t st = ds::as_stream(stream);
    struct {
        int32_t* dst;
        const int32_t* src;
        int64_t n;
    } jobs[] = {{rowptr, p->rowptr, p->nv + 1}, {colidx, p->colidx, p->nnzb},   {diagidx, p->diagidx, p->nv},
                {cptr, p->cptr, p->nnzb + 1},   {clist, p->clist, p->ncontrib}, {gptr, p->gptr, p->ngroups + 1},
                {gent, p->gent, p->ne},         {goff, p->goff, p->ne + 1},     {kperm, p->kperm, p->nnzb},
                {ctab, p->ctab, 4 * p->ngroups}};
    for (auto& j : jobs)
        if (j.dst) DP_HIP(hipMemcpyAsync(j.dst, j.src, sizeof(int32_t) * (size_t)j.n, hipMemcpyDeviceToDevice, st));
    return DS_OK;
}
