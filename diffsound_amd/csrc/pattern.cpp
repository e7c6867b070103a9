// Symbolic phase (host): node-adjacency BSR pattern and per-block contribution lists.
//
// What the reference does instead: it emits (3N)^2 COO triplets per Gauss point / element and lets
// torch.sparse coalesce() sort and merge them on every call (reference
// src/diffelastic/diff_model.py:214-220, 299-312).  Here the merge is done once per topology:
// for every node-pair block (i,j) of the pattern we record which (element, a, b) pairs add into
// it, in ascending contribution-id order, so the numeric kernel can sum them without atomics and
// bit-reproducibly.
#include <algorithm>
#include <climits>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#include "ds_common.h"

namespace ds {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace ds

extern "C" const char* ds_last_error(void) { return ds::g_err; }
extern "C" int ds_abi_version(void) { return DS_ABI_VERSION; }

struct ds_pattern {
    int64_t nv = 0, nnzb = 0, ncontrib = 0;
    std::vector<int32_t> rowptr, colidx, diagidx, cptr, clist;
};

namespace {
struct RowScratch {
    std::vector<uint64_t> keys;  // (col << 32) | contribution id
};
}  // namespace

extern "C" int ds_pattern_build(const int32_t* tets, int64_t T, int N, int64_t nv, int nthreads,
                                ds_pattern_t** out) {
    DS_REQUIRE(out != nullptr && tets != nullptr, "ds_pattern_build: null argument");
    DS_REQUIRE(N == 4 || N == 10, "ds_pattern_build: N must be 4 or 10 (got %d)", N);
    DS_REQUIRE(T > 0 && nv > 0, "ds_pattern_build: empty mesh (T=%lld nv=%lld)", (long long)T, (long long)nv);
    DS_REQUIRE(T * (int64_t)N * N < (int64_t)1 << 31, "ds_pattern_build: T*N*N overflows int32");
    for (int64_t i = 0; i < T * N; ++i)
        DS_REQUIRE(tets[i] >= 0 && tets[i] < nv, "ds_pattern_build: node id %d out of range at %lld", tets[i],
                   (long long)i);
    if (nthreads <= 0) nthreads = (int)std::max(1u, std::thread::hardware_concurrency());
    nthreads = (int)std::min<int64_t>(nthreads, std::max<int64_t>(1, nv / 1024));

    // node -> (element, local index) incidence by counting sort
    std::vector<int64_t> nptr(nv + 1, 0);
    for (int64_t i = 0; i < T * N; ++i) nptr[tets[i] + 1]++;
    for (int64_t i = 0; i < nv; ++i) nptr[i + 1] += nptr[i];
    std::vector<int32_t> ninc(T * N);
    {
        std::vector<int64_t> fill(nptr.begin(), nptr.end() - 1);
        for (int64_t i = 0; i < T * N; ++i) ninc[fill[tets[i]]++] = (int32_t)i;  // i = t*N + a
    }

    auto* p = new (std::nothrow) ds_pattern;
    if (!p) {
        ds::set_error("ds_pattern_build: out of memory");
        return DS_ERR_NOMEM;
    }
    p->nv = nv;
    p->ncontrib = T * (int64_t)N * N;
    p->rowptr.assign(nv + 1, 0);
    p->diagidx.assign(nv, -1);

    // pass 1 (parallel over row ranges): per row, sorted (col, contribution) keys
    std::vector<std::vector<uint64_t>> chunk_keys(nthreads);
    std::vector<int64_t> bounds(nthreads + 1);
    for (int c = 0; c <= nthreads; ++c) bounds[c] = nv * c / nthreads;
    std::vector<int32_t> rowcnt(nv, 0);
    auto work = [&](int c) {
        auto& keys = chunk_keys[c];
        std::vector<uint64_t> row;
        for (int64_t i = bounds[c]; i < bounds[c + 1]; ++i) {
            row.clear();
            for (int64_t e = nptr[i]; e < nptr[i + 1]; ++e) {
                const int64_t ta = ninc[e];
                const int64_t t = ta / N, a = ta % N;
                const int32_t* tt = tets + t * N;
                for (int b = 0; b < N; ++b)
                    row.push_back(((uint64_t)(uint32_t)tt[b] << 32) | (uint32_t)(t * N * N + a * N + b));
            }
            std::sort(row.begin(), row.end());
            int32_t nb = 0;
            uint32_t prev = 0xffffffffu;
            for (uint64_t k : row) {
                const uint32_t col = (uint32_t)(k >> 32);
                if (col != prev) {
                    ++nb;
                    prev = col;
                }
            }
            rowcnt[i] = nb;
            keys.insert(keys.end(), row.begin(), row.end());
        }
    };
    {
        std::vector<std::thread> th;
        for (int c = 0; c < nthreads; ++c) th.emplace_back(work, c);
        for (auto& t : th) t.join();
    }
    for (int64_t i = 0; i < nv; ++i) {
        const int64_t nxt = (int64_t)p->rowptr[i] + rowcnt[i];
        if (nxt >= ((int64_t)1 << 31)) {
            delete p;
            ds::set_error("ds_pattern_build: nnzb overflows int32");
            return DS_ERR_ARG;
        }
        p->rowptr[i + 1] = (int32_t)nxt;
    }
    p->nnzb = p->rowptr[nv];
    p->colidx.resize(p->nnzb);
    p->cptr.assign(p->nnzb + 1, 0);
    p->clist.resize(p->ncontrib);

    // pass 2: scatter the per-chunk key streams into colidx / cptr / clist
    std::vector<int64_t> chunk_c0(nthreads + 1, 0);  // first contribution index of each chunk
    for (int c = 0; c < nthreads; ++c) chunk_c0[c + 1] = chunk_c0[c] + (int64_t)chunk_keys[c].size();
    auto fill = [&](int c) {
        const auto& keys = chunk_keys[c];
        int64_t slot = p->rowptr[bounds[c]] - 1;
        int64_t ci = chunk_c0[c];
        size_t pos = 0;
        for (int64_t i = bounds[c]; i < bounds[c + 1]; ++i) {
            int64_t cnt = 0;
            for (int64_t e = nptr[i]; e < nptr[i + 1]; ++e) cnt += N;
            uint32_t prev = 0xffffffffu;
            for (int64_t q = 0; q < cnt; ++q, ++pos, ++ci) {
                const uint32_t col = (uint32_t)(keys[pos] >> 32);
                if (col != prev) {
                    ++slot;
                    p->colidx[slot] = (int32_t)col;
                    p->cptr[slot] = (int32_t)ci;
                    if ((int64_t)col == i) p->diagidx[i] = (int32_t)slot;
                    prev = col;
                }
                p->clist[ci] = (int32_t)(uint32_t)keys[pos];
            }
        }
    };
    {
        std::vector<std::thread> th;
        for (int c = 0; c < nthreads; ++c) th.emplace_back(fill, c);
        for (auto& t : th) t.join();
    }
    p->cptr[p->nnzb] = (int32_t)p->ncontrib;
    *out = p;
    return DS_OK;
}

extern "C" int ds_pattern_sizes(const ds_pattern_t* p, int64_t* nv, int64_t* nnzb, int64_t* ncontrib) {
    DS_REQUIRE(p != nullptr, "ds_pattern_sizes: null pattern");
    if (nv) *nv = p->nv;
    if (nnzb) *nnzb = p->nnzb;
    if (ncontrib) *ncontrib = p->ncontrib;
    return DS_OK;
}

extern "C" int ds_pattern_export(const ds_pattern_t* p, int32_t* rowptr, int32_t* colidx, int32_t* diagidx,
                                 int32_t* cptr, int32_t* clist) {
    DS_REQUIRE(p != nullptr, "ds_pattern_export: null pattern");
    if (rowptr) std::memcpy(rowptr, p->rowptr.data(), sizeof(int32_t) * (p->nv + 1));
    if (colidx) std::memcpy(colidx, p->colidx.data(), sizeof(int32_t) * p->nnzb);
    if (diagidx) std::memcpy(diagidx, p->diagidx.data(), sizeof(int32_t) * p->nv);
    if (cptr) std::memcpy(cptr, p->cptr.data(), sizeof(int32_t) * (p->nnzb + 1));
    if (clist) std::memcpy(clist, p->clist.data(), sizeof(int32_t) * p->ncontrib);
    return DS_OK;
}

extern "C" void ds_pattern_free(ds_pattern_t* p) { delete p; }

// ------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------
// Node groups for the register-blocked SpMM: G = 4 consecutive (Morton-ordered) nodes share one wave.
// Their neighbour sets overlap (~1.7x on P2 meshes), so the wave walks the UNION of the four column
// lists once: one X-panel load serves up to four blocks.  Per union entry: column id + 4-bit presence
// mask (packed in one int32), block offset into a group-ordered copy of the values (goff), and kperm
// maps that copy back to the BSR slots so it can be refreshed whenever the material changes.
struct ds_groups {
    int64_t ngroups = 0, ne = 0, nnzb = 0;
    std::vector<int32_t> gptr, gent, goff, kperm;
};

extern "C" int ds_groups_build(const int32_t* rowptr, const int32_t* colidx, int64_t nv, ds_groups_t** out) {
    DS_REQUIRE(rowptr && colidx && out, "ds_groups_build: null argument");
    DS_REQUIRE(nv > 0 && nv < ((int64_t)1 << 28), "ds_groups_build: nv must be in (0, 2^28)");
    constexpr int G = 4;
    auto* g = new (std::nothrow) ds_groups;
    if (!g) {
        ds::set_error("ds_groups_build: out of memory");
        return DS_ERR_NOMEM;
    }
    g->nnzb = rowptr[nv];
    g->ngroups = (nv + G - 1) / G;
    g->gptr.reserve(g->ngroups + 1);
    g->gptr.push_back(0);
    g->kperm.reserve(g->nnzb);
    g->goff.push_back(0);
    for (int64_t gi = 0; gi < g->ngroups; ++gi) {
        const int64_t n0 = gi * G, n1 = std::min<int64_t>(nv, n0 + G);
        int64_t pos[G];
        for (int q = 0; q < G; ++q) pos[q] = (n0 + q < n1) ? rowptr[n0 + q] : 0;
        for (;;) {  // G-way merge of the sorted rows
            int32_t cmin = INT32_MAX;
            for (int q = 0; q < n1 - n0; ++q)
                if (pos[q] < rowptr[n0 + q + 1]) cmin = std::min(cmin, colidx[pos[q]]);
            if (cmin == INT32_MAX) break;
            uint32_t mask = 0;
            for (int q = 0; q < n1 - n0; ++q)
                if (pos[q] < rowptr[n0 + q + 1] && colidx[pos[q]] == cmin) {
                    mask |= 1u << q;
                    g->kperm.push_back((int32_t)pos[q]);
                    ++pos[q];
                }
            g->gent.push_back((int32_t)((uint32_t)cmin | (mask << 28)));
            g->goff.push_back((int32_t)g->kperm.size());
        }
        g->gptr.push_back((int32_t)g->gent.size());
    }
    g->ne = (int64_t)g->gent.size();
    *out = g;
    return DS_OK;
}

extern "C" int ds_groups_sizes(const ds_groups_t* g, int64_t* ngroups, int64_t* ne) {
    DS_REQUIRE(g != nullptr, "ds_groups_sizes: null handle");
    if (ngroups) *ngroups = g->ngroups;
    if (ne) *ne = g->ne;
    return DS_OK;
}

extern "C" int ds_groups_export(const ds_groups_t* g, int32_t* gptr, int32_t* gent, int32_t* goff, int32_t* kperm) {
    DS_REQUIRE(g != nullptr, "ds_groups_export: null handle");
    if (gptr) std::memcpy(gptr, g->gptr.data(), sizeof(int32_t) * (g->ngroups + 1));
    if (gent) std::memcpy(gent, g->gent.data(), sizeof(int32_t) * g->ne);
    if (goff) std::memcpy(goff, g->goff.data(), sizeof(int32_t) * (g->ne + 1));
    if (kperm) std::memcpy(kperm, g->kperm.data(), sizeof(int32_t) * g->nnzb);
    return DS_OK;
}

extern "C" void ds_groups_free(ds_groups_t* g) { delete g; }
