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
//   4. chunk table: each group cut greedily into chunks (e0, e1, b0, b1) of whole entries that fit the SpMM's LDS
//      images (almost always one chunk per group); utab = chunk range of every group.
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
    int64_t nv = 0, T = 0, nnzb = 0, ncontrib = 0, ne = 0, ngroups = 0, nchunks = 0;
    int N = 0, cap = 0, single = 0;
    // device arrays (owned)
    int32_t *rowptr = nullptr, *colidx = nullptr, *diagidx = nullptr, *cptr = nullptr, *clist = nullptr;
    int32_t *gptr = nullptr, *gent = nullptr, *goff = nullptr, *kperm = nullptr, *utab = nullptr, *ctab = nullptr;
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

// Chunks of whole entries with at most cap entries and cap blocks (the LDS images of ds_spmm_union): the greedy cut of
// the host rule (union_chunks in _hip.py) - furthest entry end within both limits, at least one entry.
__device__ __forceinline__ int next_cut(const int32_t* __restrict__ goff, int e, int e1, int cap) {
    const int lim = min(e1, e + cap);
    const int bound = goff[e] + cap;
    int j = e + 1;
    while (j < lim && goff[j + 1] <= bound) ++j;
    return j;
}

__global__ void chunk_count_kernel(const int32_t* __restrict__ gptr, const int32_t* __restrict__ goff, int64_t ngroups,
                                   int cap, int32_t* __restrict__ counts) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= ngroups) return;
    const int e1 = gptr[g + 1];
    int c = 0;
    for (int e = gptr[g]; e < e1; e = next_cut(goff, e, e1, cap)) ++c;
    counts[g] = c > 0 ? c : 1;  // a group without entries keeps one (empty) chunk, as on the host
}

__global__ void chunk_write_kernel(const int32_t* __restrict__ gptr, const int32_t* __restrict__ goff, int64_t ngroups,
                                   int cap, const int32_t* __restrict__ start, int32_t* __restrict__ utab,
                                   int32_t* __restrict__ ctab) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= ngroups) return;
    const int e0 = gptr[g], e1 = gptr[g + 1];
    int c = start[g];
    utab[2 * g + 0] = c;
    utab[2 * g + 1] = start[g + 1];
    if (e0 >= e1) {
        ctab[4 * c + 0] = e0, ctab[4 * c + 1] = e1, ctab[4 * c + 2] = goff[e0], ctab[4 * c + 3] = goff[e1];
        return;
    }
    for (int e = e0; e < e1; ++c) {
        const int nxt = next_cut(goff, e, e1, cap);
        ctab[4 * c + 0] = e, ctab[4 * c + 1] = nxt, ctab[4 * c + 2] = goff[e], ctab[4 * c + 3] = goff[nxt];
        e = nxt;
    }
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
        (rc = dev_alloc(&p->gptr, p->ngroups + 1)) != DS_OK || (rc = dev_alloc(&p->utab, 2 * p->ngroups)) != DS_OK)
        return rc;
    DP_HIP(hipMemcpyAsync(p->goff, goff_tmp, sizeof(int32_t) * (ne + 1), hipMemcpyDeviceToDevice, st));
    entries_kernel<<<blocks_for(ne), 256, 0, st>>>(g1, p->goff, ne, nv, p->gent);
    DS_LAUNCH_CHECK("entries_kernel");
    lower_bound_rows_kernel<<<blocks_for(p->ngroups + 1), 256, 0, st>>>(uek, ne, p->ngroups, (uint64_t)nv, 0, p->gptr);
    DS_LAUNCH_CHECK("lower_bound_rows_kernel(groups)");
    // ---- 4. chunk table: count per group, exclusive scan, write
    int32_t *counts, *start;
    if ((rc = sc.alloc(&counts, p->ngroups + 1)) != DS_OK || (rc = sc.alloc(&start, p->ngroups + 1)) != DS_OK) return rc;
    DP_HIP(hipMemsetAsync(counts + p->ngroups, 0, sizeof(int32_t), st));
    chunk_count_kernel<<<blocks_for(p->ngroups), 256, 0, st>>>(p->gptr, p->goff, p->ngroups, p->cap, counts);
    DS_LAUNCH_CHECK("chunk_count_kernel");
    size_t bytes = 0;
    DP_HIP(rocprim::exclusive_scan(nullptr, bytes, counts, start, 0, (size_t)(p->ngroups + 1), rocprim::plus<int32_t>(), st));
    char* tmp;
    if ((rc = sc.alloc(&tmp, (int64_t)bytes)) != DS_OK) return rc;
    DP_HIP(rocprim::exclusive_scan(tmp, bytes, counts, start, 0, (size_t)(p->ngroups + 1), rocprim::plus<int32_t>(), st));
    int32_t nchunks = 0;
    DP_HIP(hipMemcpyAsync(&nchunks, start + p->ngroups, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    DP_HIP(hipStreamSynchronize(st));
    p->nchunks = nchunks;
    if ((rc = dev_alloc(&p->ctab, 4 * p->nchunks)) != DS_OK) return rc;
    chunk_write_kernel<<<blocks_for(p->ngroups), 256, 0, st>>>(p->gptr, p->goff, p->ngroups, p->cap, start, p->utab, p->ctab);
    DS_LAUNCH_CHECK("chunk_write_kernel");
    DP_HIP(hipStreamSynchronize(st));  // `start` is scratch
    p->single = p->nchunks == p->ngroups ? 1 : 0;
    return DS_OK;
}


// ---- mesh front end (lifting / duplicate merge) -------------------------------------------------
__global__ void edge_keys_kernel(const int64_t* __restrict__ tets, int64_t T, int64_t nv, uint64_t* __restrict__ keys,
                                 int32_t* __restrict__ vals, int32_t* __restrict__ bad) {
    // local edge order of the reference's lifting: (0,1) (1,2) (0,2) (0,3) (1,3) (2,3)  (src/diffelastic/mesh.py:112-135)
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 6 * T) return;
    const int64_t t = i / 6;
    const int e = (int)(i - 6 * t);
    const int ca = (e == 0 || e == 2 || e == 3) ? 0 : (e == 1 || e == 4) ? 1 : 2;
    const int cb = (e == 0) ? 1 : (e == 1 || e == 2) ? 2 : 3;
    const int64_t a = tets[4 * t + ca], b = tets[4 * t + cb];
    if (a < 0 || b < 0 || a >= nv || b >= nv) {
        *bad = 1;
        keys[i] = 0, vals[i] = (int32_t)i;
        return;
    }
    const int64_t lo = a < b ? a : b, hi = a < b ? b : a;
    keys[i] = (uint64_t)lo * (uint64_t)nv + (uint64_t)hi;
    vals[i] = (int32_t)i;
}

__global__ void edge_scatter_kernel(const uint64_t* __restrict__ ukeys, const int32_t* __restrict__ offsets,
                                    const int32_t* __restrict__ vals_sorted, int64_t ne, int64_t nv,
                                    int64_t* __restrict__ ea, int64_t* __restrict__ eb, int64_t* __restrict__ tet_edge) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= ne) return;
    const uint64_t k = ukeys[r];
    ea[r] = (int64_t)(k / (uint64_t)nv);
    eb[r] = (int64_t)(k % (uint64_t)nv);
    for (int32_t j = offsets[r]; j < offsets[r + 1]; ++j) tet_edge[vals_sorted[j]] = r;
}

// order-preserving map fp32 -> u32 (-0.0 and +0.0 share a key: torch.unique merges them as equal values)
__device__ __forceinline__ uint32_t float_key(float f) {
    uint32_t b = __float_as_uint(f);
    if ((b << 1) == 0u) b = 0u;
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

__global__ void coord_keys_kernel(const float* __restrict__ xyz, const uint32_t* __restrict__ perm, int64_t n, int c,
                                  uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t src = perm ? perm[i] : (uint32_t)i;
    keys[i] = float_key(xyz[3 * (int64_t)src + c]);
    vals[i] = src;
}

__global__ void row_heads_kernel(const float* __restrict__ xyz, const uint32_t* __restrict__ perm, int64_t n,
                                 int32_t* __restrict__ head) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int h = 1;
    if (i > 0) {
        const float* a = xyz + 3 * (int64_t)perm[i];
        const float* b = xyz + 3 * (int64_t)perm[i - 1];
        h = (float_key(a[0]) != float_key(b[0])) || (float_key(a[1]) != float_key(b[1])) ||
            (float_key(a[2]) != float_key(b[2]));
    }
    head[i] = h;
}

__global__ void row_ids_kernel(const uint32_t* __restrict__ perm, const int32_t* __restrict__ head,
                               const int32_t* __restrict__ rank, int64_t n, int64_t* __restrict__ inv,
                               int64_t* __restrict__ first) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t id = (int64_t)rank[i] - 1;  // inclusive scan of the head flags
    inv[perm[i]] = id;
    if (head[i]) first[id] = perm[i];  // stable sort: the head of a run is its lowest original index
}

int sort_pairs32(Scratch& sc, uint32_t* kin, uint32_t* kout, uint32_t* vin, uint32_t* vout, int64_t n, hipStream_t st) {
    size_t bytes = 0;
    DP_HIP(rocprim::radix_sort_pairs(nullptr, bytes, kin, kout, vin, vout, (size_t)n, 0u, 32u, st));
    char* tmp;
    int rc = sc.alloc(&tmp, (int64_t)bytes);
    if (rc != DS_OK) return rc;
    DP_HIP(rocprim::radix_sort_pairs(tmp, bytes, kin, kout, vin, vout, (size_t)n, 0u, 32u, st));
    return DS_OK;
}

}  // namespace

extern "C" void ds_dpattern_free(ds_dpattern_t* p) {
    if (!p) return;
    for (int32_t* q : {p->rowptr, p->colidx, p->diagidx, p->cptr, p->clist, p->gptr, p->gent, p->goff, p->kperm, p->utab,
                       p->ctab})
        if (q) (void)hipFree(q);
    delete p;
}

extern "C" int ds_dpattern_build(const int32_t* tets, int64_t T, int N, int64_t nv, int cap_blocks, ds_stream_t stream,
                                 ds_dpattern_t** out) {
    DS_REQUIRE(tets && out, "ds_dpattern_build: null argument");
    DS_REQUIRE(T > 0 && (N == 4 || N == 10), "ds_dpattern_build: need T > 0 and N in {4, 10}");
    DS_REQUIRE(nv > 0 && nv < ((int64_t)1 << 28), "ds_dpattern_build: nv must be in (0, 2^28)");
    DS_REQUIRE(T * N * N < ((int64_t)1 << 31), "ds_dpattern_build: more than 2^31 contributions");
    DS_REQUIRE(cap_blocks >= 4, "ds_dpattern_build: cap_blocks must be at least 4 (one entry holds up to 4 blocks)");
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
                                 int64_t* ngroups, int64_t* nchunks) {
    DS_REQUIRE(p != nullptr, "ds_dpattern_sizes: null handle");
    if (nnzb) *nnzb = p->nnzb;
    if (ncontrib) *ncontrib = p->ncontrib;
    if (ne) *ne = p->ne;
    if (ngroups) *ngroups = p->ngroups;
    if (nchunks) *nchunks = p->nchunks;
    return DS_OK;
}

extern "C" int ds_dpattern_export(const ds_dpattern_t* p, int32_t* rowptr, int32_t* colidx, int32_t* diagidx,
                                  int32_t* cptr, int32_t* clist, int32_t* gptr, int32_t* gent, int32_t* goff,
                                  int32_t* kperm, int32_t* utab, int32_t* ctab, ds_stream_t stream) {
    DS_REQUIRE(p != nullptr, "ds_dpattern_export: null handle");
    hipStream_t st = ds::as_stream(stream);
    struct {
        int32_t* dst;
        const int32_t* src;
        int64_t n;
    } jobs[] = {{rowptr, p->rowptr, p->nv + 1}, {colidx, p->colidx, p->nnzb},   {diagidx, p->diagidx, p->nv},
                {cptr, p->cptr, p->nnzb + 1},   {clist, p->clist, p->ncontrib}, {gptr, p->gptr, p->ngroups + 1},
                {gent, p->gent, p->ne},         {goff, p->goff, p->ne + 1},     {kperm, p->kperm, p->nnzb},
                {utab, p->utab, 2 * p->ngroups}, {ctab, p->ctab, 4 * p->nchunks}};
    for (auto& j : jobs)
        if (j.dst) DP_HIP(hipMemcpyAsync(j.dst, j.src, sizeof(int32_t) * (size_t)j.n, hipMemcpyDeviceToDevice, st));
    return DS_OK;
}


// Distinct undirected edges of an ord-1 tet mesh and the edge ids of every element (the "edge hash" of the ord-2
// lifting: one midpoint per edge instead of one per (element, edge) followed by a float duplicate merge).
extern "C" int ds_edge_table(const int64_t* tets, int64_t T, int64_t nv, int64_t* ea, int64_t* eb, int64_t* tet_edge,
                             int64_t* n_edges, ds_stream_t stream) {
    DS_REQUIRE(tets && ea && eb && tet_edge && n_edges, "ds_edge_table: null argument");
    DS_REQUIRE(T > 0 && 6 * T < ((int64_t)1 << 31), "ds_edge_table: need 0 < 6 T < 2^31");
    DS_REQUIRE(nv > 0 && nv < ((int64_t)1 << 31), "ds_edge_table: nv must be in (0, 2^31)");
    hipStream_t st = ds::as_stream(stream);
    Scratch sc(st);
    const int64_t n = 6 * T;
    uint64_t *k0, *k1, *uk;
    int32_t *v0, *v1, *off, *bad;
    int rc;
    if ((rc = sc.alloc(&k0, n)) != DS_OK || (rc = sc.alloc(&k1, n)) != DS_OK || (rc = sc.alloc(&uk, n)) != DS_OK ||
        (rc = sc.alloc(&v0, n)) != DS_OK || (rc = sc.alloc(&v1, n)) != DS_OK || (rc = sc.alloc(&off, n + 1)) != DS_OK ||
        (rc = sc.alloc(&bad, 1)) != DS_OK)
        return rc;
    DP_HIP(hipMemsetAsync(bad, 0, sizeof(int32_t), st));
    edge_keys_kernel<<<blocks_for(n), 256, 0, st>>>(tets, T, nv, k0, v0, bad);
    DP_HIP(hipGetLastError());
    int32_t hbad = 0;
    DP_HIP(hipMemcpyAsync(&hbad, bad, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    DP_HIP(hipStreamSynchronize(st));
    DS_REQUIRE(hbad == 0, "ds_edge_table: node index outside [0, nv)");
    rc = sort_pairs(sc, k0, k1, v0, v1, n, bits_for((uint64_t)nv * (uint64_t)nv), st);
    if (rc != DS_OK) return rc;
    int64_t ne = 0;
    rc = runs(sc, k1, n, uk, off, &ne, st);
    if (rc != DS_OK) return rc;
    edge_scatter_kernel<<<blocks_for(ne), 256, 0, st>>>(uk, off, v1, ne, nv, ea, eb, tet_edge);
    DP_HIP(hipGetLastError());
    DP_HIP(hipStreamSynchronize(st));  // scratch is freed on return
    *n_edges = ne;
    return DS_OK;
}

// Unique rows of an (n, 3) fp32 array in lexicographic order: torch.unique(x, dim=0, return_inverse=True) of the
// reference's remove_duplicate_vertices (src/diffelastic/mesh.py:162-179) as three stable radix-sort passes.
extern "C" int ds_unique_rows3(const float* xyz, int64_t n, int64_t* inv, int64_t* first, int64_t* n_unique,
                               ds_stream_t stream) {
    DS_REQUIRE(xyz && inv && first && n_unique, "ds_unique_rows3: null argument");
    DS_REQUIRE(n > 0 && n < ((int64_t)1 << 31), "ds_unique_rows3: need 0 < n < 2^31");
    hipStream_t st = ds::as_stream(stream);
    Scratch sc(st);
    uint32_t *k0, *k1, *p0, *p1;
    int32_t *head, *rank;
    int rc;
    if ((rc = sc.alloc(&k0, n)) != DS_OK || (rc = sc.alloc(&k1, n)) != DS_OK || (rc = sc.alloc(&p0, n)) != DS_OK ||
        (rc = sc.alloc(&p1, n)) != DS_OK || (rc = sc.alloc(&head, n)) != DS_OK || (rc = sc.alloc(&rank, n)) != DS_OK)
        return rc;
    for (int c = 2; c >= 0; --c) {  // least significant coordinate first; each pass is stable
        coord_keys_kernel<<<blocks_for(n), 256, 0, st>>>(xyz, c == 2 ? nullptr : p1, n, c, k0, p0);
        DP_HIP(hipGetLastError());
        rc = sort_pairs32(sc, k0, k1, p0, p1, n, st);
        if (rc != DS_OK) return rc;
    }
    row_heads_kernel<<<blocks_for(n), 256, 0, st>>>(xyz, p1, n, head);
    DP_HIP(hipGetLastError());
    size_t bytes = 0;
    DP_HIP(rocprim::inclusive_scan(nullptr, bytes, head, rank, (size_t)n, rocprim::plus<int32_t>(), st));
    char* tmp;
    rc = sc.alloc(&tmp, (int64_t)bytes);
    if (rc != DS_OK) return rc;
    DP_HIP(rocprim::inclusive_scan(tmp, bytes, head, rank, (size_t)n, rocprim::plus<int32_t>(), st));
    row_ids_kernel<<<blocks_for(n), 256, 0, st>>>(p1, head, rank, n, inv, first);
    DP_HIP(hipGetLastError());
    int32_t last = 0;
    DP_HIP(hipMemcpyAsync(&last, rank + (n - 1), sizeof(int32_t), hipMemcpyDeviceToHost, st));
    DP_HIP(hipStreamSynchronize(st));
    *n_unique = last;
    return DS_OK;
}
