// Gather probe (tuning aid for the neighbour-union SpMM): how fast can the panels of the real union tables be
// pulled into registers, as a function of the bytes per lane and the panels per wave-instruction?
//   mode 0: bf16 panels (3 rows x ld x 2 B), 8 B per lane, ONE panel per instruction (the production bf16 layout)
//   mode 1: bf16 panels, 16 B per lane, TWO panels per instruction (lanes 0-29 / 30-59)
//   mode 2: fp32 panels (3 rows x ld x 4 B), 16 B per lane, one panel per instruction (the production fp32 layout)
//   mode 3: bf16 panels, 16 B per lane, two panels per instruction, ONE WORKGROUP-wide union of 16 nodes walked by the
//           4 waves together (each wave a quarter of the entries): the gather volume of a 16-node group
// hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/gather_probe.hip -o tools/gather_probe.so ; driven by
// tools/mb_gather_probe.py
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

using i2v = __attribute__((ext_vector_type(2))) int;
using i4v = __attribute__((ext_vector_type(4))) int;

template <int MODE>
__global__ void __launch_bounds__(256) probe(const int32_t* __restrict__ gptr, const int32_t* __restrict__ gent,
                                             unsigned ngroups, const char* __restrict__ X, unsigned xbytes, int panel_bytes,
                                             int row_bytes, int lanes_per_row, int* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const unsigned grp = blockIdx.x * 4 + wave;
    if (grp >= ngroups) return;
    const int e0 = __builtin_amdgcn_readfirstlane(gptr[grp]);
    const int ne = __builtin_amdgcn_readfirstlane(gptr[grp + 1]) - e0;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)X, 0, (int)xbytes, 0x00020000);  // loads at 0x7f000000 fall outside: zeros
    int acc = 0;
    if (MODE == 0 || MODE == 2) {
        const int g = lane / lanes_per_row, cl = lane - g * lanes_per_row;
        const bool active = g < 3;
        const int voff = active ? g * row_bytes + cl * (MODE == 0 ? 8 : 16) : 0x7f000000;
        for (int b = 0; b < ne; b += 64) {
            const int mine = (b + lane < ne) ? (gent[e0 + b + lane] & 0x0fffffff) : -1;
            const int cnt = min(64, ne - b);
            for (int q = 0; q < cnt; q += 8) {
                if (MODE == 0) {
                    i2v x[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int e = __builtin_amdgcn_readlane(mine, (q + u) & 63);
                        const int so = e < 0 ? 0 : e * panel_bytes;
                        x[u] = __builtin_bit_cast(i2v, __builtin_amdgcn_raw_buffer_load_b64(rs, e < 0 ? 0x7f000000 : voff, so, 0));
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) acc ^= x[u].x ^ x[u].y;
                } else {
                    i4v x[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int e = __builtin_amdgcn_readlane(mine, (q + u) & 63);
                        const int so = e < 0 ? 0 : e * panel_bytes;
                        x[u] = __builtin_bit_cast(i4v, __builtin_amdgcn_raw_buffer_load_b128(rs, e < 0 ? 0x7f000000 : voff, so, 0));
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) acc ^= x[u].x ^ x[u].y ^ x[u].z ^ x[u].w;
                }
            }
        }
    } else if (MODE == 6) {
        // outer-product layout with the columns split over TWO waves (blockIdx.y = column half of 40): per entry three
        // dword loads of 40 lanes (160 bytes each)
        const int c0 = blockIdx.y * 40;
        for (int b = 0; b < ne; b += 64) {
            const int mine = (b + lane < ne) ? (gent[e0 + b + lane] & 0x0fffffff) : -1;
            const int cnt = min(64, ne - b);
            for (int q = 0; q < cnt; q += 8) {
                int x[24];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int e = __builtin_amdgcn_readlane(mine, (q + u) & 63);
                    const int so = e < 0 ? 0 : e * panel_bytes;
                    const int bad = (e < 0 || lane >= 40) ? 0x7f000000 : 0;
#pragma unroll
                    for (int g = 0; g < 3; ++g)
                        x[3 * u + g] = __builtin_amdgcn_raw_buffer_load_b32(rs, (bad ? bad : (c0 + lane) * 4 + g * row_bytes), so, 0);
                }
#pragma unroll
                for (int u = 0; u < 24; ++u) acc ^= x[u];
            }
        }
    } else if (MODE == 5) {
        // fp32 panels in the OUTER-PRODUCT layout (one column per lane: the B operand of v_mfma_f32_4x4x1_16B_f32): per
        // entry three dword loads (row g, columns 0..63) and one for the tails (lane l: row l >> 4, column 64 + (l & 15))
        const int tvoff = lane < 48 ? (lane >> 4) * row_bytes + (64 + (lane & 15)) * 4 : 0x7f000000;
        for (int b = 0; b < ne; b += 64) {
            const int mine = (b + lane < ne) ? (gent[e0 + b + lane] & 0x0fffffff) : -1;
            const int cnt = min(64, ne - b);
            for (int q = 0; q < cnt; q += 4) {
                int x[16];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int e = __builtin_amdgcn_readlane(mine, (q + u) & 63);
                    const int so = e < 0 ? 0 : e * panel_bytes;
                    const int bad = e < 0 ? 0x7f000000 : 0;
#pragma unroll
                    for (int g = 0; g < 3; ++g)
                        x[4 * u + g] = __builtin_amdgcn_raw_buffer_load_b32(rs, (bad ? bad : lane * 4 + g * row_bytes), so, 0);
                    x[4 * u + 3] = __builtin_amdgcn_raw_buffer_load_b32(rs, bad ? bad : tvoff, so, 0);
                }
#pragma unroll
                for (int u = 0; u < 16; ++u) acc ^= x[u];
            }
        }
    } else {  // MODE 1: 30 lanes x 16 B = one 480-byte bf16 panel; two panels per instruction
        const int half = lane >= 30 ? 1 : 0;
        const int l30 = lane - 30 * half;
        const bool active = lane < 60;
        const int voff = active ? l30 * 16 : 0x7f000000;
        for (int b = 0; b < ne; b += 64) {
            const int mine = (b + lane < ne) ? (gent[e0 + b + lane] & 0x0fffffff) : -1;
            const int cnt = min(64, ne - b);
            for (int q = 0; q < cnt; q += 16) {
                i4v x[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    // lane-dependent entry: the even one for the first 30 lanes, the odd one for the next 30
                    const int ea = __builtin_amdgcn_readlane(mine, (q + 2 * u) & 63);
                    const int eb = __builtin_amdgcn_readlane(mine, (q + 2 * u + 1) & 63);
                    const int e = half ? eb : ea;
                    const int vo = (e < 0) ? 0x7f000000 : voff + e * panel_bytes;
                    x[u] = __builtin_bit_cast(i4v, __builtin_amdgcn_raw_buffer_load_b128(rs, vo, 0, 0));
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) acc ^= x[u].x ^ x[u].y ^ x[u].z ^ x[u].w;
            }
        }
    }
    if (acc == 0x12345678) out[grp] = acc;  // keeps the loads alive, (almost) never writes
}

// mode 7: the production fp32 load (mode 2) with knobs: DEPTH loads in flight per wave, the workgroup -> group mapping of the
// library (each XCD a contiguous range of groups) or the plain one, WPG groups walked by every wave one after the other, and
// `colmask` != 0 folds every panel index into a table of colmask + 1 panels (the ceiling of the load shape out of the L2)
__device__ __forceinline__ unsigned probe_xcd_remap(unsigned bid, unsigned nblk) {
    const unsigned q = nblk >> 3, r = nblk & 7, x = bid & 7, k = bid >> 3;
    return x < r ? x * (q + 1) + k : r * (q + 1) + (x - r) * q + k;
}
template <int DEPTH>
__global__ void __launch_bounds__(256) probe7(const int32_t* __restrict__ gptr, const int32_t* __restrict__ gent, unsigned ngroups,
                                              const char* __restrict__ X, unsigned xbytes, int panel_bytes, int row_bytes,
                                              int lanes_per_row, int remap, int colmask, unsigned nwg, int* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const unsigned grp = (remap ? probe_xcd_remap(blockIdx.x, nwg) : blockIdx.x) * 4 + wave;
    if (grp >= ngroups) return;
    const int e0 = __builtin_amdgcn_readfirstlane(gptr[grp]);
    const int ne = __builtin_amdgcn_readfirstlane(gptr[grp + 1]) - e0;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)X, 0, (int)xbytes, 0x00020000);
    int acc = 0;
    const int g = lane / lanes_per_row, cl = lane - g * lanes_per_row;
    const int voff = g < 3 ? g * row_bytes + cl * 16 : 0x7f000000;
    for (int b = 0; b < ne; b += 64) {
        int mine = (b + lane < ne) ? (gent[e0 + b + lane] & 0x0fffffff) : -1;
        if (colmask && mine >= 0) mine &= colmask;
        const int cnt = min(64, ne - b);
        for (int q = 0; q < cnt; q += DEPTH) {
            i4v x[DEPTH];
#pragma unroll
            for (int u = 0; u < DEPTH; ++u) {
                const int e = __builtin_amdgcn_readlane(mine, (q + u) & 63);
                const int so = e < 0 ? 0 : e * panel_bytes;
                x[u] = __builtin_bit_cast(i4v, __builtin_amdgcn_raw_buffer_load_b128(rs, e < 0 ? 0x7f000000 : voff, so, 0));
            }
#pragma unroll
            for (int u = 0; u < DEPTH; ++u) acc ^= x[u].x ^ x[u].y ^ x[u].z ^ x[u].w;
        }
    }
    if (acc == 0x12345678) out[grp] = acc;
}

extern "C" float gather_probe7(int depth, int remap, int colmask, const int32_t* gptr, const int32_t* gent, unsigned ngroups,
                               const void* X, unsigned xbytes, int ld_elems, int* out, int reps) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    const unsigned nwg = (ngroups + 3) / 4;
    const int row_bytes = ld_elems * 4, panel_bytes = 3 * row_bytes;
    float best = 1e30f;
    for (int r = 0; r < reps + 1; ++r) {
        hipEventRecord(a, 0);
        if (depth == 4) probe7<4><<<nwg, 256>>>(gptr, gent, ngroups, (const char*)X, xbytes, panel_bytes, row_bytes, ld_elems / 4, remap, colmask, nwg, out);
        else if (depth == 8) probe7<8><<<nwg, 256>>>(gptr, gent, ngroups, (const char*)X, xbytes, panel_bytes, row_bytes, ld_elems / 4, remap, colmask, nwg, out);
        else probe7<16><<<nwg, 256>>>(gptr, gent, ngroups, (const char*)X, xbytes, panel_bytes, row_bytes, ld_elems / 4, remap, colmask, nwg, out);
        hipEventRecord(b, 0);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        if (r > 0 && ms < best) best = ms;
    }
    hipEventDestroy(a);
    hipEventDestroy(b);
    if (hipGetLastError() != hipSuccess) return -1.f;
    return best;
}

// mode 3: gptr / gent describe 16-node groups; the 4 waves of a workgroup split the entries of ONE group
__global__ void __launch_bounds__(256) probe16(const int32_t* __restrict__ gptr, const int32_t* __restrict__ gent,
                                               unsigned ngroups, const char* __restrict__ X, unsigned xbytes, int panel_bytes,
                                               int* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const unsigned grp = blockIdx.x;
    if (grp >= ngroups) return;
    const int e0 = __builtin_amdgcn_readfirstlane(gptr[grp]);
    const int ne = __builtin_amdgcn_readfirstlane(gptr[grp + 1]) - e0;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)X, 0, (int)xbytes, 0x00020000);  // loads at 0x7f000000 fall outside: zeros
    const int half = lane >= 30 ? 1 : 0;
    const int l30 = lane - 30 * half;
    const int voff = lane < 60 ? l30 * 16 : 0x7f000000;
    int acc = 0;
    for (int b = wave * 16; b < ne; b += 64) {  // 16 entries per wave and round
        const int mine = (b + (lane & 15) < ne) ? (gent[e0 + b + (lane & 15)] & 0x0fffffff) : -1;
        i4v x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int ea = __builtin_amdgcn_readlane(mine, 2 * u);
            const int eb = __builtin_amdgcn_readlane(mine, 2 * u + 1);
            const int e = half ? eb : ea;
            const int vo = (e < 0) ? 0x7f000000 : voff + e * panel_bytes;
            x[u] = __builtin_bit_cast(i4v, __builtin_amdgcn_raw_buffer_load_b128(rs, vo, 0, 0));
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc ^= x[u].x ^ x[u].y ^ x[u].z ^ x[u].w;
    }
    if (acc == 0x12345678) out[grp] = acc;
}

extern "C" float gather_probe(int mode, const int32_t* gptr, const int32_t* gent, unsigned ngroups, const void* X,
                              unsigned xbytes, int ld_elems, int* out, int reps) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    const unsigned nwg = (ngroups + 3) / 4;
    float best = 1e30f;
    for (int r = 0; r < reps + 1; ++r) {
        hipEventRecord(a, 0);
        const int eb = (mode == 2 || mode == 5 || mode == 6) ? 4 : 2;
        const int row_bytes = ld_elems * eb, panel_bytes = 3 * row_bytes;
        if (mode == 0) probe<0><<<nwg, 256>>>(gptr, gent, ngroups, (const char*)X, xbytes, panel_bytes, row_bytes, ld_elems / 4, out);
        else if (mode == 1) probe<1><<<nwg, 256>>>(gptr, gent, ngroups, (const char*)X, xbytes, panel_bytes, row_bytes, 0, out);
        else if (mode == 2) probe<2><<<nwg, 256>>>(gptr, gent, ngroups, (const char*)X, xbytes, panel_bytes, row_bytes, ld_elems / 4, out);
        else if (mode == 5) probe<5><<<nwg, 256>>>(gptr, gent, ngroups, (const char*)X, xbytes, panel_bytes, row_bytes, ld_elems / 4, out);
        else if (mode == 6) probe<6><<<dim3(nwg, 2), 256>>>(gptr, gent, ngroups, (const char*)X, xbytes, panel_bytes, row_bytes, ld_elems / 4, out);
        else probe16<<<ngroups, 256>>>(gptr, gent, ngroups, (const char*)X, xbytes, panel_bytes, out);
        hipEventRecord(b, 0);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        if (r > 0 && ms < best) best = ms;
    }
    hipEventDestroy(a);
    hipEventDestroy(b);
    if (hipGetLastError() != hipSuccess) return -1.f;
    return best;
}
