/* diffsound_hip.h - C ABI of libdiffsound_hip.so (MI355X / gfx950 only).
 *
 * Drop-in boundary for DiffSound's modal-sound hot path.  The reference's only native-op
 * precedent on this path is
 *     void assemble_mass_matrix(const Tensor& vertices, const Tensor& tets, Tensor& values,
 *                               Tensor& rows, Tensor& cols, Tensor& element_mm,
 *                               const double density, const int order)
 *     (reference src/cuda/massMatrixDouble.h:14-15, bound in src/cuda/bind.cu:9-12),
 * whose conventions are kept: the CALLER allocates every input and output buffer, the op fills
 * outputs in place and owns nothing; errors surface to Python as RuntimeError.  Differences, on
 * purpose: plain pointers + sizes instead of torch types, an explicit hipStream_t instead of the
 * legacy default stream (reference src/include/macro.h:146-152), an int status + ds_last_error()
 * instead of C++ exceptions, and no global mutable state besides the opaque host-side pattern
 * handle.  Every function is asynchronous on `stream` unless stated otherwise; none allocates or
 * synchronises, so all of them can be captured into a hipGraph.
 *
 * Layout conventions
 *   - DOF ordering: global DOF 3*node + c (reference src/diffelastic/deform.py:118-125).
 *   - K, K_lambda, K_mu: BSR with 3x3 blocks on the node-adjacency pattern (rowptr[nv+1],
 *     colidx[nnzb], ascending per row), values [nnzb][3][3] row-major.
 *   - M = M_s (x) I3 is stored as the node-level scalar CSR M_s on the same pattern: values [nnzb].
 *   - dense blocks X are row-major (n x ncols) with an explicit leading dimension `ld` (elements)
 *     so that column slices of a wider buffer can be passed without copies.
 *   - dtype codes: DS_F32 = 0, DS_F64 = 1.
 */
#ifndef DIFFSOUND_HIP_H
#define DIFFSOUND_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* ds_stream_t; /* hipStream_t */

enum { DS_F32 = 0, DS_F64 = 1 };
enum { DS_OK = 0, DS_ERR_ARG = 1, DS_ERR_HIP = 2, DS_ERR_NOMEM = 3 };

/* Thread-local text of the last error returned on this thread ("" if none). */
const char* ds_last_error(void);
/* Library ABI version (bumped on any signature change); ds_abi_version() returns the value the library was built with. */
#define DS_ABI_VERSION 33
int ds_abi_version(void);

/* ------------------------------------------------------------------------------------------------
 * Symbolic phase (HOST, CPU only): node-adjacency BSR pattern + per-block contribution lists.
 * Replaces the reference's COO triplet emission + coalesce() (src/diffelastic/diff_model.py:
 * 214-220, 299-312; src/cuda/massMatrixDouble.cu:70-77).  Once per mesh topology.
 *   tets   : (T x N) int32 host array, N = 4 (ord-1) or 10 (ord-2), reference local node order
 *            (src/diffelastic/mesh.py:139-154).
 * Contribution id of element t, local pair (a,b):  t*N*N + a*N + b.
 * ---------------------------------------------------------------------------------------------- */
typedef struct ds_pattern ds_pattern_t;
int ds_pattern_build(const int32_t* tets, int64_t T, int N, int64_t nv, int nthreads, ds_pattern_t** out);
int ds_pattern_sizes(const ds_pattern_t* p, int64_t* nv, int64_t* nnzb, int64_t* ncontrib);
/* Copies into caller-allocated HOST arrays: rowptr[nv+1], colidx[nnzb], diagidx[nv] (slot of block
 * (i,i)), cptr[nnzb+1], clist[ncontrib] (contribution ids grouped by block slot, ascending). */
int ds_pattern_export(const ds_pattern_t* p, int32_t* rowptr, int32_t* colidx, int32_t* diagidx,
                      int32_t* cptr, int32_t* clist);
void ds_pattern_free(ds_pattern_t* p);

/* Node groups of the neighbour-union SpMM (HOST): 4 consecutive nodes share one wave and walk the union of
 * their column lists.  Exports gptr[ngroups+1], gent[ne] (column id | presence mask << 28), goff[ne+1] (block
 * offsets into the group-ordered value copy) and kperm[nnzb] (group-ordered position -> BSR slot). */
typedef struct ds_groups ds_groups_t;
int ds_groups_build(const int32_t* rowptr, const int32_t* colidx, int64_t nv, ds_groups_t** out);
int ds_groups_sizes(const ds_groups_t* g, int64_t* ngroups, int64_t* ne);
int ds_groups_export(const ds_groups_t* g, int32_t* gptr, int32_t* gent, int32_t* goff, int32_t* kperm);
void ds_groups_free(ds_groups_t* g);

/* The same symbolic phase ON THE DEVICE (csrc/dpattern.hip): pattern, contribution lists, neighbour-union tables and
 * the chunk tables (every group of 4 rows cut into chunks of whole entries with at most cap_blocks entries / blocks;
 * almost always one chunk per group) from device connectivity, entry for entry what ds_pattern_build +
 * ds_groups_build and the host cutting rule produce.  The handle owns device arrays; ds_dpattern_build synchronises
 * `stream` (the table sizes come back to the host); ds_dpattern_export copies into caller-allocated DEVICE arrays (any
 * pointer may be NULL): rowptr[nv+1], colidx[nnzb], diagidx[nv], cptr[nnzb+1], clist[ncontrib], gptr[ngroups+1],
 * gent[ne], goff[ne+1], kperm[nnzb], utab[ngroups x 2] (chunk range of each group), ctab[nchunks x 4] (e0, e1, b0, b1).
 * (reference: COO triplets + coalesce() on every assembly, src/diffelastic/diff_model.py:214-220, 299-312, for a mesh
 * that the geometry experiments rebuild every iteration, src/dmtet/geometry/dmtet_thickness.py:251) */
typedef struct ds_dpattern ds_dpattern_t;
int ds_dpattern_build(const int32_t* tets, int64_t T, int N, int64_t nv, int cap_blocks, ds_stream_t stream,
                      ds_dpattern_t** out);
int ds_dpattern_sizes(const ds_dpattern_t* p, int64_t* nnzb, int64_t* ncontrib, int64_t* ne, int64_t* ngroups,
                      int64_t* nchunks);
int ds_dpattern_export(const ds_dpattern_t* p, int32_t* rowptr, int32_t* colidx, int32_t* diagidx, int32_t* cptr,
                       int32_t* clist, int32_t* gptr, int32_t* gent, int32_t* goff, int32_t* kperm, int32_t* utab,
                       int32_t* ctab, ds_stream_t stream);
void ds_dpattern_free(ds_dpattern_t* p);

/* Mesh front end ON THE DEVICE (csrc/dpattern.hip), replacing the float-`unique` of the reference's ord-2 lifting and
 * duplicate merge (src/diffelastic/mesh.py:101-179), which the geometry experiments run on a new mesh every iteration
 * (src/dmtet/geometry/dmtet_thickness.py:251-285).  Both synchronise `stream` (a count comes back to the host).
 *  ds_edge_table  : tets (T x 4) int64 DEVICE (torch long) -> the distinct undirected edges, ea[i] < eb[i], sorted by
 *                   (ea, eb) (arrays of capacity 6 T), and tet_edge (T x 6): the edge id of each local edge in the
 *                   reference's order (0,1) (1,2) (0,2) (0,3) (1,3) (2,3); *n_edges on the HOST.
 *  ds_unique_rows3: xyz (n x 3) fp32 DEVICE -> inv[n] (id of every row among the distinct rows in lexicographic
 *                   (x, y, z) order; -0.0 == +0.0 as in torch.unique) and first[id] = lowest row index with that
 *                   coordinate (capacity n); *n_unique on the HOST. */
int ds_edge_table(const int64_t* tets, int64_t T, int64_t nv, int64_t* ea, int64_t* eb, int64_t* tet_edge,
                  int64_t* n_edges, ds_stream_t stream);
int ds_unique_rows3(const float* xyz, int64_t n, int64_t* inv, int64_t* first, int64_t* n_unique, ds_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Numeric assembly (DEVICE).  K_lambda, K_mu (geometry-only parts of K = lam*K_lambda + mu*K_mu,
 * SURVEY.md 0.6) and M_s in one pass, fp64, deterministic (no atomics): one thread per block slot
 * sums its contribution list in fixed order.  Restates update_stiff_matrix / update_mass_matrix
 * (reference src/diffelastic/diff_model.py:184-312) with the element integrals in closed form over
 * the reference's own Gauss rule (tables built by the host from src/diffelastic/gauss.py:17-38).
 *   verts  : (nv x 3) f32 device          tets : (T x N) i32 device
 *   dtab   : (N x 4 x N x 4) f64 device,  dtab[a][k][b][l] = sum_g w_g dN_a/dL_k dN_b/dL_l
 *   mtab   : (N x N) f64 device,          mtab[a][b]       = sum_g w_g N_a N_b   (already * density
 *            in fp32 where the reference does so, diff_model.py:301-303)
 *   tetgeo : workspace (T x 13) f64 device (barycentric gradients 4x3 + |det|)
 *   out    : klam, kmu (nnzb x 9) f64 ; ms (nnzb) f64
 * ---------------------------------------------------------------------------------------------- */
int ds_assemble_kml(const float* verts, const int32_t* tets, int64_t T, int N, int64_t nv,
                    const int32_t* cptr, const int32_t* clist, int64_t nnzb,
                    const double* dtab, const double* mtab, double* tetgeo,
                    double* klam, double* kmu, double* ms, ds_stream_t stream);

/* K32 = (float)(lam*K_lambda + mu*K_mu) (k32t: the same with every 3x3 block transposed, may be NULL),
 * Ms32 = (float)M_s, and the fp32 inverse of the 3x3 diagonal blocks of K (block-Jacobi preconditioner).
 * Per material hypothesis. */
int ds_combine_material(const double* klam, const double* kmu, const double* ms, int64_t nnzb,
                        const int32_t* diagidx, int64_t nv, double lam, double mu,
                        float* k32, float* k32t, float* ms32, float* dinv32, ds_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Block SpMM  Y = A X  on the BSR-3 pattern (the HBM-roofline kernel).  Replaces torch.sparse.mm
 * in the reference's LOBPCG (src/lobpcg/_linalg_utils.py:36-37) and in get_vals
 * (src/diffelastic/diff_model.py:395-397).
 *   kind 0: A = K,   vals (nnzb x 9) f32           X, Y f32
 *   kind 1: A = M,   vals (nnzb)     f32 (M_s)     X, Y f32
 *   kind 2: A = K_*, vals (nnzb x 9) f64           X f32, Y f64   (polish / read-out)
 *   kind 3: A = M,   vals (nnzb)     f64 (M_s)     X f32, Y f64
 *   kind 4: A = K_*, vals (nnzb x 9) f64           X f64, Y f64   (fp64 refinement; ncols % 4 == 0, <= 84,
 *   kind 5: A = M,   vals (nnzb)     f64 (M_s)     X f64, Y f64    32-byte aligned rows)
 * vals_t (kind 0 only, may be NULL): the same blocks stored transposed, vals_t[k][r][i] = K_k[i][r]
 *   (ds_combine_material writes it); enables the one-load-per-block path for ncols <= 84.
 * X: (3nv x ncols) ld = ldx ; Y: (3nv x ncols) ld = ldy ; X and Y must not overlap.
 * ---------------------------------------------------------------------------------------------- */
int ds_spmm_bsr3(int kind, const int32_t* rowptr, const int32_t* colidx, const void* vals,
                 const void* vals_t, int64_t nv, const void* X, int64_t ldx, void* Y, int64_t ldy, int ncols,
                 ds_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Tall-skinny Gram  G = A^T B  (p x q, fp64, row-major, ld = q) with MFMA.
 * Replaces the dense (k x n)(n x k) products of Rayleigh-Ritz / svqb / ortho in the reference
 * (src/lobpcg/_linalg_utils.py:64-73 via torch.matmul).
 *   A: (n x p) f32 or f64 (a_dtype; an f64 A needs an f64 B), lda ;  B: (n x q) f32 or f64 (b_dtype), ldb
 *   flags: DS_GRAM_SYMMETRIC (needs p == q): the caller asserts G is symmetric (e.g. S^T (K S)); only the
 *     block-upper part is computed and mirrored.
 *     DS_GRAM_EXACT: exact products, fp64 accumulation throughout (fp64 MFMA).  Without it an f32 x f32 product
 *     runs on the fp32 MFMA with the accumulators folded into fp64 every 48 rows (error ~1e-9 |A_i||B_j|, far
 *     below the operands' own fp32 rounding) at twice the rate; an f64 B always takes the exact path.
 *   work: device scratch of ds_gram_workspace_bytes(n, p, q) bytes.
 * ---------------------------------------------------------------------------------------------- */
int64_t ds_gram_workspace_bytes(int64_t n, int p, int q);
#define DS_GRAM_SYMMETRIC 1
#define DS_GRAM_EXACT 2
int ds_gram(const void* A, int a_dtype, int64_t lda, int p, const void* B, int b_dtype, int64_t ldb, int q,
            int64_t n, int flags, double* G, void* work, int64_t work_bytes, ds_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Fused block-vector updates of the eigensolver (all (n x ncols) f32 with leading dimensions).
 * ---------------------------------------------------------------------------------------------- */
/* R <- KX - MX * diag(lam) ;  rn2[j] = ||R_j||^2, xn2[j] = ||X_j||^2 (f64, zeroed by this call).  KX may be R (in place).
 * (reference update_residual + the norms of update_converged_count, _lobpcg.py:301-333) */
int ds_residual(const float* KX, int64_t ldk, float* R, int64_t ldr, const float* MX, int64_t ldm, const float* X,
                int64_t ldx, const double* lam, int64_t n, int ncols, double* rn2, double* xn2, ds_stream_t stream);
/* D <- c * T R ; W <- D      (T = block-Jacobi, dinv (nv x 9) f32) */
int ds_cheb_init(const float* R, int64_t ldr, float* D, int64_t ldd, float* W, int64_t ldw,
                 const float* dinv, int64_t nv, int ncols, float c, ds_stream_t stream);
/* R <- R - AD ; D <- c1 D + c2 T R ; W <- W + D */
int ds_cheb_step(const float* AD, int64_t lda, float* R, int64_t ldr, float* D, int64_t ldd,
                 float* W, int64_t ldw, const float* dinv, int64_t nv, int ncols, float c1, float c2,
                 ds_stream_t stream);
/* One fused term of the Chebyshev block-Jacobi polynomial preconditioner (three-term form):
 *   W_next <- W + c1 (W - W_prev) + c2 T (R0 - K W),   written over W_prev ;  first != 0: W_prev = 0, not read.
 * K: (rowptr, colidx, vals f32 (nnzb x 9)); ncols a multiple of 4, <= 84; W and W_prev distinct buffers. */
int ds_cheb_spmm(const int32_t* rowptr, const int32_t* colidx, const float* vals, int64_t nv,
                 const float* W, int64_t ldw, float* Wprev, int64_t ldp, const float* R0, int64_t ldr,
                 const float* dinv, int ncols, float c1, float c2, int first, ds_stream_t stream);
/* Y <- R0 - K X on a block of <= 84 columns (X, Y distinct): the fine-level residual that the two-level
 * preconditioner restricts to the corner-node level; K X itself is never written. */
int ds_spmm_residual(const int32_t* rowptr, const int32_t* colidx, const float* vals, int64_t nv,
                     const float* X, int64_t ldx, const float* R0, int64_t ldr, float* Y, int64_t ldy,
                     int ncols, ds_stream_t stream);
/* Y_i <- beta Y_i + sum_k w[k] X_{colidx[k]}, k in [rowptr[i], rowptr[i+1]), on 3 x ncols node panels
 * (nrows output nodes; X may have a different node count): prolongation / restriction between an ord-2
 * mesh and its corner-node ord-1 sub-mesh (the P1 space written in the P2 nodal basis; node layout of
 * reference src/diffelastic/mesh.py:139-154).  ncols a multiple of 4, 16-byte aligned rows. */
int ds_scalar_csr_spmm(const int32_t* rowptr, const int32_t* colidx, const float* w, int64_t nrows,
                       const float* X, int64_t ldx, float* Y, int64_t ldy, int ncols, float beta,
                       ds_stream_t stream);
/* kgrp: (nnzb x 9) f32 = the TRANSPOSED blocks in group order, kgrp[p] = vals_t[kperm[p]] (tables of ds_groups_build). */
/* The three fp64-value products of the read-out in one walk of the pattern (the panels of X are gathered once instead of
 * three times): Ya = A X, Yb = B X with (nnzb x 9) fp64 blocks (K_lambda, K_mu), Ym = (m (x) I3) X with nnzb fp64 node
 * scalars (M_s); X fp32, results fp64, ncols a multiple of 4 <= 84, 16-byte aligned rows.  Bit-identical to ds_spmm_bsr3
 * kinds 2, 2, 3.  (reference: the autograd read-out of get_undamped_freqs, src/diffelastic/diff_model.py:371-388) */
int ds_spmm_f64_polish(const int32_t* rowptr, const int32_t* colidx, const double* a, const double* b, const double* m,
                       int64_t nv, const float* X, int64_t ldx, double* Ya, double* Yb, double* Ym, int64_t ldy, int ncols,
                       ds_stream_t stream);
/* fp64 values with fp64 (x_f64 != 0) or fp32 vectors on the neighbour-union tables of ds_spmm_union (ABI 28): Y (fp64) = A X for a
 * block of <= 84 columns, one wave per group of 4 nodes walking the union of their neighbours - the fp64 refinement's K W / M W
 * (ds_spmm_bsr3 kinds 4 / 5 gather every neighbour's panel once per ROW: 2.2 x the panels).  vals: kind 0 = the 3x3 blocks in
 * the tables' group order, TRANSPOSED (vals[p][g][i] = A_block[i][g], 9 doubles per block, i.e. ds_pack_groups' layout in fp64);
 * kind 1 = node scalars in that order (A = a (x) I3).  Equal to ds_spmm_bsr3's results to fp64 rounding (another summation order).
 * (reference: torch.sparse.mm of src/lobpcg/_linalg_utils.py:36-37 in fp64) */
int ds_spmm_f64_union(int kind, int x_f64, const int32_t* utab, const int32_t* ctab, int64_t ngroups, int cap_blocks,
                      const int32_t* gent, const double* vals, int64_t nnzb, int64_t nv, const void* X, int64_t ldx, double* Y,
                      int64_t ldy, int ncols, ds_stream_t stream);
/* The same three products stored as fp32 blocks (ABI 28): the sums are formed in fp64 and rounded once.  Halves what the
 * kernel writes and the Gram product behind it reads; an OPTION of the eigensolver's polish (off by default: the polish then
 * carries ~3e-8 of relative noise instead of being accurate to second order in the iteration error). */
int ds_spmm_f64_polish_f32out(const int32_t* rowptr, const int32_t* colidx, const double* a, const double* b,
                              const double* m, int64_t nv, const float* X, int64_t ldx, float* Ya, float* Yb, float* Ym,
                              int64_t ldy, int ncols, ds_stream_t stream);
int ds_pack_groups(const float* vals_t, const int32_t* kperm, int64_t nnzb, float* kgrp, ds_stream_t stream);
/* Neighbour-union form of ds_spmm_bsr3 / ds_cheb_spmm / ds_spmm_residual for ncols <= 84 (the eigensolver's
 * b-column products and every preconditioner term): one wavefront per group of 4 consecutive nodes walks the UNION
 * of their neighbours, so a neighbour panel shared inside the group is gathered once (Morton order: 0.58 x the
 * panel loads of one wavefront per node).  Tables from ds_groups_build - gent (union entries col | mask << 28),
 * kgrp (TRANSPOSED 3x3 blocks in group order, ds_pack_groups) - cut into chunks of whole entries with at most
 * cap_blocks (<= DS_UNION_CAP = 140) blocks: ctab (nchunks x 4) = (e0, e1, b0, b1), utab (ngroups x 2) = chunk range of each
 * group, ngroups = ceil(nv / 4); utab may be NULL when every group is exactly one chunk (ctab has ngroups rows).
 * epilogue 0: Y <- A X ; 1: Y (= W_prev) <- X + c1 (X - Y) + c2 T (R0 - A X) (first != 0: Y not read) ;
 * (Wprev != NULL, epilogue 1 only: W_prev is read from there and Y is only written - out-of-place form) ;
 * 2: Y <- R0 - A X ; 3: Y <- (A_s (x) I3) X with kgrp = the node-SCALAR values in group order (nnzb floats: the mass
 * matrix).  X and Y distinct, 16-byte aligned rows; operand blocks of 2 GB and more (3 nv ld 4 >= 0x7f000000 bytes)
 * take a variant that builds one buffer descriptor per panel load.
 * level_tag (0 fine, 1 corner-node level) selects instantiations of epilogues 0 and 3 whose kernel symbols differ - nothing else.
 * (reference: torch.sparse.mm in src/lobpcg/_linalg_utils.py:36-37 and the iK callable of _lobpcg.py:441) */
#define DS_UNION_CAP 140
int ds_spmm_union(int epilogue, int level_tag, const int32_t* utab, const int32_t* ctab, int64_t ngroups, int cap_blocks,
                  const int32_t* gent, const float* kgrp, int64_t nnzb, int64_t nv, const float* X, int64_t ldx,
                  float* Y, int64_t ldy, const float* R0, int64_t ldr, const float* dinv, int ncols, float c1, float c2,
                  int first, const float* Wprev, int64_t ldp, ds_stream_t stream);
/* The eigensolver's residual in ONE walk of the neighbour unions (ABI 26): R <- K X - (M_s (x) I3) X diag(lam) on a block of
 * <= 84 columns, and rn2[j] = ||R_j||^2, xn2[j] = ||X_j||^2 (fp64) - what ds_spmm_union (epilogue 0), ds_spmm_union
 * (epilogue 3) and ds_residual compute in three launches and five passes over (n x ncols) blocks (reference:
 * update_residual / update_converged_count, src/lobpcg/_lobpcg.py:301-333).  kgrp / mgrp: the transposed 3x3 blocks and the
 * node-scalar mass values in group order, as for ds_spmm_union; lam: ncols Ritz values on the device (fp64, rounded to fp32 as
 * ds_residual does).  R equals the three-launch result bit for bit; the norms are summed over the groups in a fixed order (no
 * atomics: reproducible, which ds_residual's are not).  work: ds_union_residual_workspace_bytes(ngroups, ncols) bytes. */
int64_t ds_union_residual_workspace_bytes(int64_t ngroups, int ncols);
int ds_union_residual(int level_tag, const int32_t* utab, const int32_t* ctab, int64_t ngroups, int cap_blocks,
                      const int32_t* gent, const float* kgrp, const float* mgrp, int64_t nnzb, int64_t nv, const float* X,
                      int64_t ldx, const double* lam, float* R, int64_t ldr, int ncols, void* work, int64_t work_bytes,
                      double* rn2, double* xn2, ds_stream_t stream);
/* Narrow blocks (ABI 28): Y = K X (kind 0, vals = kgrp) or Y = (M_s (x) I3) X (kind 3, vals = mgrp) on <= 16 columns with the
 * lanes of a wave dealt over the union's ENTRIES instead of over the columns (ds_spmm_union keeps 6 of 64 lanes busy on an
 * 8-column block and takes as long as on 80 columns): the block power iteration of the Chebyshev interval and the operator-norm
 * estimates of the eigensolver.  Same tables as ds_spmm_union; sums in another order: equal to it to fp32 rounding, not bit for
 * bit.  (reference: the operator-norm estimates of src/lobpcg/_lobpcg.py:280-285) */
int ds_spmm_union_narrow(int kind, int level_tag, const int32_t* utab, const int32_t* ctab, int64_t ngroups, const int32_t* gent,
                         const float* vals, int64_t nnzb, int64_t nv, const float* X, int64_t ldx, float* Y, int64_t ldy,
                         int ncols, ds_stream_t stream);
/* Y = K X and Y2 = (M_s (x) I3) X of ONE block (<= 84 columns) in one walk of the neighbour unions (ABI 28; epilogue 5 of the
 * kernel): X is gathered once instead of twice; each product is formed exactly as ds_spmm_union's epilogues 0 and 3 form it
 * (bit-identical results).  The eigensolver's K W and M W of the raw preconditioned residuals (ds_lobpcg_t.raw_rr).
 * (reference: the two torch.sparse.mm of _lobpcg.py:441-459 on the same block) */
int ds_spmm_union_km(int level_tag, const int32_t* utab, const int32_t* ctab, int64_t ngroups, int cap_blocks,
                     const int32_t* gent, const float* kgrp, const float* mgrp, int64_t nnzb, int64_t nv, const float* X,
                     int64_t ldx, float* KX, int64_t ldk, float* MX, int64_t ldm, int ncols, ds_stream_t stream);
/* ------------------------------------------------------------------------------------------------
 * Two-level V-cycle preconditioner in one call (host-side driver, csrc/vcycle.cpp): issues on `stream` the launch
 * sequence  W1 = S R ; W2 = W1 + P C P^T (R - K W1) ; W = W2 + S (R - K W2)  out of ds_cheb_init, ds_spmm_union and
 * ds_scalar_csr_spmm - S: degree-`fine.degree` Chebyshev block-Jacobi smoother for the interval [lmin, lmax] of T K
 * on the fine level, C: the same on the corner-node level, P / P^T the transfer operators.  The preconditioner of
 * the reference's LOBPCG is an opaque callable (src/lobpcg/_lobpcg.py:441); this is the one diffsound_amd supplies.
 * All blocks (rows x ncols) f32 with leading dimensions, 16-byte aligned rows, ncols a multiple of 4 <= 84; R is only
 * read; Wc, D, AD (one common leading dimension), Rr (fine) and Rc, Ec, Dc, ADc (corner-node level, common leading
 * dimension ldc) are scratch.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    const int32_t* utab;   /* neighbour-union tables and values of the level, as for ds_spmm_union */
    const int32_t* ctab;
    int64_t ngroups;
    int32_t cap_blocks;
    int32_t degree;        /* polynomial degree of the level's Chebyshev block-Jacobi operator */
    const int32_t* gent;
    const float* kgrp;
    int64_t nnzb;
    int64_t nv;
    const float* dinv;     /* (nv x 9) inverse diagonal blocks */
    double lmax, lmin;     /* the polynomial targets [lmin, lmax] of T K */
    /* bf16 cycles only: mf_group_nodes = 8 sends the level's bf16 terms to ds_spmm_union16m (tables and the packed
     * blocks as described there); 0: ds_spmm_union16 */
    int32_t mf_group_nodes;
    int32_t mf_max_entries;
    int32_t mf_max_batch_blocks;
    int32_t level_tag;     /* 0: fine level, 1: corner-node level - selects kernel instantiations whose SYMBOLS differ, so
                              that a rocprofv3 kernel table separates the two levels' launches (same code otherwise) */
    const int32_t *mf_gptr, *mf_gcol, *mf_gmeta, *mf_gbase, *mf_ghead;
    const void* mf_kc;
    /* fp32 matrix-core form of the level's own products K X / M X (ds_spmm_union32m): m32_gptr != NULL enables it in
     * ds_lobpcg_iterate; tables for groups of 4 nodes, m32_k / m32_m the values in the tables' order (with 16 bytes of
     * slack behind the last block), m32_m may be NULL (mass matrix not node-scalar) */
    int32_t m32_max_entries;
    int32_t m32_max_batch_blocks;
    const int32_t *m32_gptr, *m32_gcol, *m32_gmeta, *m32_gbase;
    const float* m32_k;
    const float* m32_m;
    /* ABI 33, bf16 cycles on the matrix-core tables only: GROUP-block Jacobi.  tgrp != NULL: (ceil(nv / 8) x 24 x 24) f32, the
     * inverses T_g of the 24 x 24 diagonal blocks of the level's groups of 8 nodes (ds_group_inverse).  The level's polynomial is
     * then p(T_g K) T_g: its right-hand side goes through ds_group_apply16 first, mf_kc holds the blocks of T_g K (ds_group_pack_kc:
     * DENSE over node x entry, mf_gmeta / mf_gbase / mf_ghead with every presence bit set, mf_nblocks = 8 x the number of union
     * entries of the level) and dinv an identity per node.  NULL / 0: the node blocks (dinv) as before. */
    const float* tgrp;
    int64_t mf_nblocks;
} ds_level_t;
typedef struct {
    ds_level_t fine, coarse;
    const int32_t *rptr, *rcol;  /* restriction (corner node <- fine nodes) */
    const float* rw;
    const int32_t *pptr, *pcol;  /* prolongation (fine node <- corner nodes) */
    const float* pw;
    const float* R;
    int64_t ldr;
    float* W;
    int64_t ldw;
    float* D;
    int64_t ldd;
    float* AD;
    int64_t lda;
    float* Rr;
    int64_t ldrr;
    float *Rc, *Ec, *Dc, *ADc;
    int64_t ldc;
    int32_t ncols;
    float* Wc;       /* fine scratch: the cycle's iterate (compact), W itself is written once at the end */
    int64_t ldwc;    /* = ldd = lda */
    int32_t storage; /* 0: every scratch block fp32 ; 1: every scratch block bf16 (same fields, 2-byte elements; R and W
                        stay fp32): the preconditioner's iterates need no more mantissa, and its terms are bound by
                        the bytes of their vector streams */
    void* R16;       /* storage 1: fine bf16 scratch for the copy of R (rows x ncols, leading dimension ldr16) */
    int64_t ldr16;
} ds_twolevel_t;
int ds_twolevel_apply(const ds_twolevel_t* p, ds_stream_t stream);
/* One-level form: W <- p(T K) T R with the level's degree / [lmin, lmax] (Chebyshev block-Jacobi polynomial, every term
 * one fused ds_spmm_union launch); a, b: compact scratch blocks (rows x ncols, leading dimension lds). */
int ds_chebyshev_apply(const ds_level_t* level, const float* R, int64_t ldr, float* W, int64_t ldw, float* a, float* b,
                       int64_t lds, int ncols, ds_stream_t stream);
/* ... with bf16 scratch blocks a, b, r16 (rows x ncols bf16, leading dimension lds); degree >= 2. */
int ds_chebyshev_apply16(const ds_level_t* level, const float* R, int64_t ldr, float* W, int64_t ldw, void* a, void* b,
                         void* r16, int64_t lds, int ncols, ds_stream_t stream);
/* bf16-block forms used by the bf16 V-cycle: the fused terms (X, R0, W_prev bf16; Y bf16, or fp32 when y_f32), the
 * first Chebyshev iterate W1 = c T R (R fp32 - then Rcopy, if not NULL, receives its bf16 copy - or bf16) and the level
 * transfer.  Leading dimensions in elements; bf16 rows 8-byte aligned. */
int ds_spmm_union16(int epilogue, const int32_t* utab, const int32_t* ctab, int64_t ngroups, int cap_blocks,
                    const int32_t* gent, const float* kgrp, int64_t nnzb, int64_t nv, const void* X, int64_t ldx,
                    void* Y, int64_t ldy, int y_f32, const void* R0, int64_t ldr, const float* dinv, int ncols,
                    float c1, float c2, int first, const void* Wprev, int64_t ldp, ds_stream_t stream);
/* MFMA form of ds_spmm_union16 (csrc/spmm_mfma.inc): the same two epilogues on the same bf16 blocks, the block
 * products on the matrix cores (v_mfma_f32_16x16x16_bf16; the 3x3 blocks rounded to bf16, fp32 accumulation), one
 * wavefront per group of group_nodes = 8 consecutive nodes.  Topology tables (device): gptr (ngroups + 1) / gcol =
 * the sorted union of the column ids of each group's rows; gmeta per entry = presence mask of the group's nodes |
 * (index of the entry's first block inside the group) << 8; gbase (ngroups) = first block of each group; kperm = the
 * BSR block of every position of the group / entry / node order.  ds_pack_kc writes kc (nnzb x 3 x 4 bf16: the rows of
 * the blocks in that order, padded to 8 bytes) from the BSR values k32 - once per material.  max_entries = the largest
 * group's entry count (<= 256); max_batch_blocks = the largest number of blocks in DS_MF_BATCH = 16 consecutive entries
 * of a group (batches counted from the group's first entry; it sizes the wavefront's LDS) - when max_entries <= 128 the LAST
 * batch of a group also takes up to DS_MF_TAIL entries beyond the 16 (a group of 65 entries is four batches, not five), and
 * max_batch_blocks counts the blocks of the batches so formed.  ghead (ngroups x 128): per group
 * a record of FIXED stride with the first 64 entries of gcol (words 0..63) and of gmeta (words 64..127), zero behind the
 * group's last entry - a wavefront asks for it before it knows where the group's entries start (ABI 29). */
#define DS_MF_BATCH 16
#define DS_MF_TAIL 2
int ds_pack_kc(const float* k32, const int32_t* kperm, int64_t nnzb, void* kc, ds_stream_t stream);
/* ABI 33: the group-block Jacobi of the bf16 polynomial (ds_level_t.tgrp; reference: the preconditioner is the caller's opaque
 * callable, src/lobpcg/_lobpcg.py:441).  Groups of group_nodes = 8 consecutive nodes, ng = ceil(nv / 8).
 * ds_group_inverse: T (ng x 24 x 24 f32) = the inverses of the diagonal blocks K_gg of the BSR-3 matrix (rowptr, colidx, k32:
 *   nnzb x 9 f32), nodes behind the last one as identity rows; once per material.
 * ds_group_pack_kc: kc (8 x entries x 3 x 4 bf16) = the blocks of T_g K, dense: position (gptr[g] + e) * 8 + s' = sum_s
 *   T_g[s', s] K(s, e); gptr / gmeta / gbase / kperm are the COMPACT tables of ds_spmm_union16m on the same topology.
 * ds_group_apply16: Y = T_g X on (3 nv x ncols) blocks, ncols <= 256, X and Y each f32 (x_f32 / y_f32) or bf16; Y may be X when
 *   both have one element type. */
int ds_group_inverse(const int32_t* rowptr, const int32_t* colidx, const float* k32, int64_t nv, int group_nodes, float* T,
                     ds_stream_t stream);
int ds_group_pack_kc(const float* k32, const float* T, const int32_t* gptr, const int32_t* gmeta, const int32_t* gbase,
                     const int32_t* kperm, int group_nodes, int64_t nv, void* kc, ds_stream_t stream);
int ds_group_apply16(const float* T, int group_nodes, const void* X, int x_f32, int64_t ldx, void* Y, int y_f32, int64_t ldy,
                     int64_t nv, int ncols, ds_stream_t stream);
int ds_spmm_union16m(int epilogue, int group_nodes, int level_tag, const int32_t* gptr, const int32_t* gcol,
                     const int32_t* gmeta, const int32_t* gbase, const int32_t* ghead, const void* kc, int64_t nnzb, int64_t ngroups, int max_entries,
                     int max_batch_blocks, int64_t nv, const void* X, int64_t ldx, void* Y, int64_t ldy, int y_f32,
                     const void* R0, int64_t ldr,
                     const float* dinv, int ncols, float c1, float c2, int first, const void* Wprev, int64_t ldp,
                     ds_stream_t stream);
/* fp32 matrix-core form of ds_spmm_union epilogues 0 and 3 (csrc/spmm_mfma32.inc) - the eigensolver's own products
 * K W, M W, M X on fp32 blocks of <= 84 columns (reference: torch.sparse.mm in src/lobpcg/_linalg_utils.py:36-37):
 * one wavefront per group of 4 consecutive nodes, the gathered panels of a batch of DS_MF32_BATCH union entries staged
 * in LDS as the B operand of v_mfma_f32_16x16x4_f32 (exact fp32: one rounding per product), the 3x3 blocks (epilogue 0:
 * `vals` = nblocks x 9 floats, row-major blocks) or node-scalar values (epilogue 3: nblocks floats) of the batch staged
 * beside them and spread over the 16 x 4 A tile.  Tables as for ds_spmm_union16m but for groups of 4 nodes (gmeta = 4-bit
 * presence mask | first block inside the group << 8), values in the tables' (group, entry, node) order with at least 16
 * bytes of slack behind the last block (vals_bytes = the readable size); max_batch_blocks = the largest number of blocks
 * in DS_MF32_BATCH consecutive entries of a group (batches counted from the group's first entry).  Per output element
 * the products are summed in ONE chain in entry order - the same precision as ds_spmm_union, whose three chains (one per
 * panel row) are added at the end: results agree to fp32 rounding, not bit for bit.  level_tag as in ds_level_t. */
#define DS_MF32_BATCH 8
int ds_spmm_union32m(int epilogue, int level_tag, const int32_t* gptr, const int32_t* gcol, const int32_t* gmeta,
                     const int32_t* gbase, const float* vals, int64_t vals_bytes, int64_t nblocks, int64_t ngroups,
                     int max_entries, int max_batch_blocks, int64_t nv, const float* X, int64_t ldx, float* Y, int64_t ldy,
                     int ncols, ds_stream_t stream);
int ds_cheb_init16(const void* R, int r_f32, int64_t ldr, void* W, int64_t ldw, void* Rcopy, int64_t ldc,
                   const float* dinv, int64_t nv, int ncols, float c, ds_stream_t stream);
int ds_scalar_csr_spmm16(const int32_t* rowptr, const int32_t* colidx, const float* w, int64_t nrows, const void* X,
                         int64_t ldx, void* Y, int64_t ldy, int ncols, float beta, ds_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * The eigensolver's iteration as ONE native call (csrc/lobpcg.cpp): the loop of the reference's LOBPCG.run /
 * _update_ortho (src/lobpcg/_lobpcg.py:344-376, 433-477) in the re-designed form of lobpcg/modal_solver.py - residual
 * test, hard locking, preconditioner, single-sweep projected Cholesky-QR, Rayleigh-Ritz by recurrence, basis update -
 * issued on `stream`; the <= 3b x 3b dense steps run on the calling thread with the LAPACK / BLAS routines of
 * ds_lapack_t (Fortran calling convention, e.g. SciPy's cython_lapack / cython_blas function pointers).  Synchronises
 * the stream a few times per iteration (residual norms, Gram blocks).
 * On entry: S = [Y (ny rigid columns, 0 or a multiple of 4) | X (b Ritz vectors, M-orthonormal) | - | -], KS[:, :b] = K X,
 * lam = the b Ritz values, S2[:, :ny] = Y.  On exit: the basis buffer holding X (S2 if result_in_s2), lam, rerr (per
 * column: ||K x - lam M x|| / (||x|| (A_norm + |lam| B_norm))), iterations.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    void (*dsyevd)(char* jobz, char* uplo, int* n, double* a, int* lda, double* w, double* work, int* lwork, int* iwork,
                   int* liwork, int* info);
    void (*dgemm)(char* ta, char* tb, int* m, int* n, int* k, double* alpha, double* a, int* lda, double* b, int* ldb,
                  double* beta, double* c, int* ldc);
    /* ABI 31, optional (all three or none; NULL = the Ritz step calls dsyevd): the stages of dsyevd one by one, so that only the
     * wanted third of the eigenvectors of the 3 na x 3 na Ritz problem is back-transformed */
    void (*dsytrd)(char* uplo, int* n, double* a, int* lda, double* d, double* e, double* tau, double* work, int* lwork, int* info);
    void (*dstedc)(char* compz, int* n, double* d, double* e, double* z, int* ldz, double* work, int* lwork, int* iwork,
                   int* liwork, int* info);
    void (*dormtr)(char* side, char* uplo, char* trans, int* m, int* n, double* a, int* lda, double* tau, double* c, int* ldc,
                   double* work, int* lwork, int* info);
} ds_lapack_t;
typedef struct {
    int64_t n, nv;            /* n = 3 nv rows */
    int32_t b, k, ny;         /* block width (multiple of 4, <= 160: products of more than 84 columns run in column slices),
                                 wanted pairs, rigid columns */
    int32_t maxit, lock, ortho_passes, rr_refresh, gram_exact;
    int32_t kx_fresh;         /* != 0: K X' of the new Ritz block by ONE product K X' (b columns) instead of the update
                                 [K X' | K P'] = K [X P W] [Z1 Zp] (3b -> 2b columns) - K P is then never formed: the Gram
                                 blocks among X and P come from the small Ritz algebra, only the residual needs K X */
    int32_t raw_rr;           /* != 0 (needs kx_fresh and res_work): Rayleigh-Ritz on the RAW basis [Y X P W], W = the preconditioned
                                 residuals as they come - K W and M W in one walk (ds_spmm_union_km), ONE Gram launch
                                 [Y X P W]^T [K W | M W], the orthonormalisation of W against [Y X P] folded into the small dense
                                 algebra and into the coefficients of ONE update [X' P'] = [Y X P W] Z; an iteration whose W is
                                 too ill-conditioned for that (eps x amplification >= ortho_tol) takes the explicit route */
    double tol, ortho_tol, A_norm, B_norm;
    float *S, *S2;            /* (n x (ny + 3 b)), leading dimension lds */
    float *KS, *KS2;          /* (n x 3 b), leading dimension ldks */
    float *R, *MX, *MW;       /* (n x b), leading dimension ldr */
    int64_t lds, ldks, ldr;
    ds_level_t level;         /* K of this level: neighbour-union tables + block-Jacobi blocks; degree / lmin / lmax of
                                 the one-level polynomial when twolevel == NULL */
    const float* mgrp;        /* node-scalar mass values in group order (ds_spmm_union epilogue 3) */
    const int32_t *rowptr, *colidx;   /* BSR pattern and values for the wide products of the periodic full refresh */
    const float *k32, *k32t;
    const ds_twolevel_t* twolevel;    /* two-level preconditioner (scratch blocks with >= b columns), or NULL */
    float *pa, *pb;           /* one-level preconditioner scratch (n x b), leading dimension ldp */
    int64_t ldp;
    void* pr16;               /* not NULL: pa, pb, pr16 are bf16 blocks and the polynomial runs on bf16 iterates */
    double* gbuf;             /* device, (ny + 3 b) x 3 b doubles: Gram results */
    float* cbuf;              /* device, 8 x (ny + 3 b) x 2 b floats: update coefficients (a ring of 8 slots) */
    double* nrm;              /* device, 2 x 1024 doubles */
    double* lam_dev;          /* device, b doubles */
    void* res_work;           /* not NULL (and kx_fresh != 0): the residual of every iteration by ds_union_residual - K X' and
                                 M X' of the new Ritz block are then never written; ds_union_residual_workspace_bytes bytes */
    int64_t res_work_bytes;
    void* gram_work;
    int64_t gram_work_bytes;
    double* lam;              /* host, b: in / out */
    double* rerr;             /* host, b: out */
    double* history;          /* host, history_cap entries or NULL: max backward error of the wanted pairs per iteration */
    int32_t history_cap;
    int32_t iterations;       /* out */
    int32_t result_in_s2;     /* out */
    int32_t wait_mode;        /* ABI 31: how THIS solve's host thread waits for its stream - 0: hipStreamSynchronize, 1: a 20 us poll,
                                 then a sleep on a blocking event, -1: the process default (ds_host_wait_mode).  Per solve, so that
                                 the hypothesis lanes of one pipeline, another pipeline of the process and a single solve beside
                                 them each keep their own setting */
    double ritz_tol;          /* ABI 32: > 0: a pair counts as converged (and is locked) only when, besides rel < tol, its Ritz value
                                 moved by less than ritz_tol (relative) in the last step.  The backward error rel is relative to
                                 ||K|| + lambda ||M||: a SMOOTH vector passes a loose tol (the corner-level phase of a nested start:
                                 3e-3) whatever its Rayleigh quotient is - a start block that went through the preconditioner was
                                 locked with Ritz values 2 x off (profiles/r06_start_sweeps.txt).  0: the reference's test alone */
} ds_lobpcg_t;
int ds_lobpcg_iterate(ds_lobpcg_t* p, const ds_lapack_t* lapack, ds_stream_t stream);
/* ABI 31.  Self-check of the loop's host-side dense steps with the given LAPACK table - no device involved (the CPU test suite
 * calls it): on a seeded random symmetric positive definite n x n matrix, errs[0] / errs[1] = eigenvalue / residual error of the
 * lowest m pairs from the staged eigensolver (dsytrd + dstedc + dormtr on m columns) against dsyevd, errs[2] / errs[3] = the
 * Cholesky factor and its inverse, errs[4] = orthogonality of the twice-applied Cholesky-QR on n x m columns, errs[5] = 1 when the
 * staged path ran.  errs: 6 doubles. */
int ds_selftest_dense(const ds_lapack_t* lapack, int n, int m, unsigned seed, double* errs);
/* ABI 31.  The two small dense steps that frame the iteration, on the HOST with the caller's LAPACK table (no device; reference: the
 * first Rayleigh-Ritz step of _update_ortho, src/lobpcg/_lobpcg.py:443-448, and the read-out's quadratic forms, diff_model.py:371-399).
 * ds_host_start_block: G = [Y X0]^T [K X0 | M X0] ((ny + b) x 2 b doubles, row-major) of a start block X0 -> its projection against the
 *   M-orthonormal Y, its M-orthonormalisation and its first Ritz step in coefficients: lam (b), coef ((ny + b) x b: X = [Y X0] coef),
 *   cx (b x b: K X = (K X0) cx), *amp (the rounding amplification of the one-sweep orthonormalisation); *route = 1 (nothing else written)
 *   when eps x amp >= ortho_tol or the projected Gram matrix broke down: the caller orthonormalises explicitly.
 * ds_host_polish: GK = nterms (b x b) matrices X^T K_i X one after the other, coefs their weights, GM = X^T M X -> E (k lowest values of
 *   (sum c_i GK_i) z = e GM z), C (b x b, C^T GM C = I), qs ((nterms + 1) x k: c_j^T GK_i c_j, then c_j^T GM c_j). */
int ds_host_start_block(const ds_lapack_t* lapack, const double* G, int ny, int b, double ortho_tol, double eps, double* lam,
                        double* coef, double* cx, double* amp, int* route);
int ds_host_polish(const ds_lapack_t* lapack, int nterms, const double* GK, const double* coefs, const double* GM, int b, int k,
                   double* E, double* C, double* qs);
/* How the host thread of ds_lobpcg_iterate waits for its stream when its descriptor says wait_mode = -1 (ABI 30; process-wide default 0).  0: hipStreamSynchronize (the
 * runtime spins when the host has more cores than devices); 1: a 20 us poll, then a sleep on an event created with
 * hipEventBlockingSync - for callers that run several solves on several streams and threads at once (the hypothesis lanes of
 * diffsound_amd/pipeline.py): waiting lanes then leave their cores to the lanes that are computing. */
int ds_host_wait_mode(int mode);

/* Out <- alpha * A C + beta * Out,  A (n x p) f32, C (p x q) f32 row-major device, Out (n x q) f32.
 * Out may overlap A (e.g. be a column range of it) when q <= 160: every row tile is read completely before it is
 * written.  (reference: X <- S Z etc., _lobpcg.py:463-466) */
int ds_mix(const float* A, int64_t lda, int p, const float* C, int q, float* Out, int64_t ldo,
           int64_t n, float alpha, float beta, ds_stream_t stream);

/* Element-wise passes of the fp64 refinement fused (ABI 28, csrc/refine64.hip; reference: update_residual /
 * update_converged_count of src/lobpcg/_lobpcg.py:301-333 in fp64).  KX, MX, X: (n x b) fp64 blocks, b even <= 512, rows 16-byte
 * aligned; lam: b Ritz values on the device.
 *   ds_residual64_norms   rn2[j] = ||K x_j - lam_j M x_j||^2, xn2[j] = ||x_j||^2 in ONE pass over the three blocks (no residual
 *                         block is written); work: ds_residual64_workspace_doubles(b) doubles; fixed-order sums
 *   ds_residual64_scaled  R (n x nact fp32, nact a multiple of 4) = the residual columns cols[0 .. nact) (ascending or not),
 *                         each multiplied by scale[col] (e.g. 1 / its norm): the input of the fp32 preconditioner */
int64_t ds_residual64_workspace_doubles(int b);
int ds_residual64_norms(const double* KX, int64_t ldk, const double* MX, int64_t ldm, const double* X, int64_t ldx,
                        const double* lam, int64_t n, int b, double* work, int64_t work_doubles, double* rn2, double* xn2,
                        ds_stream_t stream);
int ds_residual64_scaled(const double* KX, int64_t ldk, const double* MX, int64_t ldm, const double* lam, const double* scale,
                         const int32_t* cols, int nact, float* R, int64_t ldr, int64_t n, ds_stream_t stream);

/* A basis held as a list of fp64 blocks (ds_gram64_blocks, ds_mix64). */
#define DS_MIX64_MAX_BLOCKS 4
typedef struct {
    const double* a;  /* device, n x p, row-major */
    int64_t lda;
    int32_t p;
    int32_t offset;   /* ds_mix64: first row of this block's coefficients in C; ds_gram64_blocks: the block's first row
                         (a block of A) or column (a block of B) of G */
} ds_block64_t;

/* G[oa_i + r][ob_j + c] = (A_i^T B_j)[r][c] for every pair of a block of A and a block of B (fp64 operands, fp64 MFMA,
 * fixed-order reduction over the row splits): the Gram blocks of a basis held as a LIST of arrays, [S]^T [K W | M W] of the
 * fp64 refinement in ONE pass over the rows instead of one ds_gram call per pair of blocks.  G is dense row-major,
 * P = sum p_i rows by Q = sum q_j columns: the blocks tile G in the order given (offset_k = the sum of the widths
 * before block k; checked).
 * flags: DS_GRAM_SYMMETRIC - G is symmetric (B = K A with symmetric K, same offsets on both sides): only the tiles on
 * and above the diagonal are computed, the rest mirrored.  At most DS_MIX64_MAX_BLOCKS blocks per side.
 * work: ds_gram_workspace_bytes(n, P, Q) bytes.  (reference: the Gram products of _lobpcg.py:516-525 on the concatenated
 * basis) */
int ds_gram64_blocks(int na, const ds_block64_t* A, int nb, const ds_block64_t* B, int64_t n, int flags, double* G,
                     void* work, int64_t work_bytes, ds_stream_t stream);

/* Out <- alpha * sum_b A_b C[offset_b : offset_b + p_b, :q] + beta * Out, all fp64: the dense n x b updates of the fp64
 * refinement over a basis given as a LIST of blocks (the rigid modes, X, P, W: separate arrays of different widths) and
 * one stacked coefficient matrix C (row-major device, leading dimension ldc) - every block is read once, the result
 * written once.  Out must not overlap any block.  (reference: X <- S Z, src/lobpcg/_lobpcg.py:457-477, there as
 * torch.matmul on the concatenated basis) */
int ds_mix64(int nblocks, const ds_block64_t* blocks, const double* C, int64_t ldc, int q, double* Out,
             int64_t ldo, int64_t n, double alpha, double beta, ds_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Geometry backward of the modal read-out (reference: autograd through get_vals -> K, M -> vertices,
 * src/diffelastic/diff_model.py:390-399 with deform.py:35-68,136-147, mesh.py:58-99):
 *   grad += d/dx sum_i gk[i] u_i^T K(x) u_i - gm[i] u_i^T M(x) u_i      (gm[i] = gk[i] * lambda_i)
 * tetgeo: the (T x 13) geometry workspace filled by ds_assemble_kml for the SAME coordinates;
 * U: (3nv x m) f32 modes; gtab (ng x N x 4) f64 = dN_a/dL_k at ng <= 4 quadrature points, gw (ng) weights
 * (a rule exact for degree 2(order-1)); mtab as in ds_assemble_kml; grad: (nv x 3) f64, ACCUMULATED with fp64
 * atomics (the caller zeroes it). */
int ds_geometry_grad(const int32_t* tets, int64_t T, int N, int64_t nv, const double* tetgeo, const float* U,
                     int64_t ldu, int m, const double* gk, const double* gm, double lam, double mu,
                     const double* gtab, const double* gw, int ng, const double* mtab, double* grad,
                     ds_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Damped-oscillator bank (reference src/ddsp/oscillator.py:113-141, 282-310):
 *   s[a,t] = sum_m amp[a,m] exp(-d_m tau_t) sin(w_m tau_t),  tau_t = (t+1)/sr
 *   y[a,t] = sum_{j<F} force[a,j] s[a,t-j]                   (causal FIR, cropped to S samples)
 * d, w: (m) f64 (decay rate and damped angular frequency) ; amp: (A x m) f32 or NULL (= 1) ;
 * force: (A x F) f32 ; y: (A x S) f32.
 * Backward: gy (A x S) f32 -> gd, gw (m) f64, gamp (A x m) f32 (may be NULL) ; gs is scratch (A x S) f32.
 * ---------------------------------------------------------------------------------------------- */
int ds_osc_bank_fwd(const double* d, const double* w, const float* amp, const float* force,
                    int A, int m, int F, int S, double sr, float* y, ds_stream_t stream);
int ds_osc_bank_bwd(const float* gy, const double* d, const double* w, const float* amp,
                    const float* force, int A, int m, int F, int S, double sr, float* gs,
                    double* gd, double* gw, float* gamp, ds_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Read-out + render + loss + backward of a pass in one call: steps 3-6 of the reference's training-loop body, the part it
 * runs EVERY epoch (the eigendecomposition only every EIGEN_DECOMPOSE_CYCLE-th: experiments/material_sync_train.py:135-167):
 *   pred_i = ev_i + (lam a_i + mu b_i) - ev_i m_i        get_undamped_freqs, src/diffelastic/diff_model.py:371-388
 *   f_i = float(sqrt(pred_i) / 2 / pi);  w0sq = (2 pi f)^2, d = (alpha + beta w0sq) / 2, w = sqrt(w0sq - d^2)
 *   audio = oscillator bank of (d, w), unit amplitudes, A = 1          TraditionalDampedOscillator, src/ddsp/oscillator.py:282-310
 *   loss = mean((audio - target)^2)   (target NULL: mean(audio^2))
 *   backward != 0: dloss/dE, dloss/dnu through the bank, the damping model and lam(E, nu), mu(E, nu), whose partial
 *   derivatives the caller passes (dlam_dE, dlam_dnu, dmu_dE, dmu_dnu).
 * ev, a_lam, b_mu, m_diag: (m) f64 - eigenvalues and the quadratic forms u^T K_lambda u, u^T K_mu u, u^T M u of the kept
 * eigenvectors; force (F) f32; target (S) f32 or NULL; audio (S) f32 out; freqs (m) f32 out; work: 6 m doubles, fwork: 2 S
 * floats (scratch); out: 3 doubles on the device (loss, dloss/dE, dloss/dnu; the last two only with backward).
 * ---------------------------------------------------------------------------------------------- */
int ds_readout_pass(const double* ev, const double* a_lam, const double* b_mu, const double* m_diag, int m,
                    double lam, double mu, double dlam_dE, double dlam_dnu, double dmu_dE, double dmu_dnu,
                    double alpha, double beta, const float* force, int F, int S, double sr, const float* target,
                    int backward, float* audio, float* freqs, double* work, float* fwork, double* out,
                    ds_stream_t stream);

/* Time-varying bank (reference GTDampedOscillator.forward with non_linear_rate != 0, oscillator.py:217-243):
 *   s[a,t] = sum_m amp[a,m] exp(-D[a,m,t]) sin(2 pi P[a,m,t]),  D = cumsum_t(dmp / sr),  P = cumsum_t(frq / sr)
 *   y = causal FIR of s with force, as above.
 * dmp, frq: (A x m x S) f32 per-sample damping rate [1/s] and damped frequency [Hz]; running sums in fp64.
 * work: scratch of ds_osc_tv_workspace_floats(A, m, S) floats.  Backward: g_dmp, g_frq (A x m x S) f32, gamp (A x m)
 * f32 or NULL; gs scratch (A x S) f32.  Deterministic. */
int64_t ds_osc_tv_workspace_floats(int A, int m, int S);
int ds_osc_tv_fwd(const float* dmp, const float* frq, const float* amp, const float* force, int A, int m, int F,
                  int S, double sr, float* work, float* y, ds_stream_t stream);
int ds_osc_tv_bwd(const float* gy, const float* dmp, const float* frq, const float* amp, const float* force, int A,
                  int m, int F, int S, double sr, float* gs, float* g_dmp, float* g_frq, float* gamp,
                  ds_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Multi-scale spectral loss head (reference src/ddsp/mss_loss.py:50-62, 70-122: SSSLoss types 'l1_loss' and
 * 'rmse_loss' on torchaudio.transforms.Spectrogram(n_fft, hop_length = n_fft / 4) = |torch.stft(center=True,
 * pad_mode="reflect", periodic Hann window, onesided)|^2).
 *   x: (B x S) f32 clips ; n_fft a power of two in [8, 2048], n_fft / 2 < S ; T = 1 + S / hop frames, F = n_fft/2 + 1 bins
 *   ds_stft_power      P (B x F x T) f32 power spectrogram ; re, im (B x F x T) f32 or both NULL (kept for the backward)
 *   ds_spec_loss       kind 0: sums (B x F x 2) f64 = per-row sums of |w_t dlog2| and |w_t dlin| over bins f >= 1
 *                        (loss = (alpha * sum0 + sum1) / (B (F-1) T), w_t = 2 t / (T - 1): the reference's time weights)
 *                      kind 1: sums[..][0] = per-row sums of (log2(Pp+eps) - log2(Pt+eps))^2 over bins f < fclip
 *                        (loss = sqrt(sum / (B fclip T)))
 *                      gP (B x F x T) f32 or NULL: d loss / d Pp (kind 1: to be divided by the loss value)
 *   ds_stft_power_bwd  gx (B x S) f32 = gscale * d/dx sum gP P(x), through re, im of the forward;
 *                      gframes: scratch (B x T x n_fft) f32.  Deterministic (gather form, no atomics).
 * ---------------------------------------------------------------------------------------------- */
int ds_stft_power(const float* x, int B, int S, int n_fft, int hop, float* P, float* re, float* im,
                  ds_stream_t stream);
int ds_spec_loss(int kind, const float* Pp, const float* Pt, int B, int F, int T, float alpha, float eps, int fclip,
                 double* sums, float* gP, ds_stream_t stream);
int ds_stft_power_bwd(const float* gP, const float* re, const float* im, int B, int S, int n_fft, int hop,
                      float gscale, float* gframes, float* gx, ds_stream_t stream);

/* Launch timing hook for the benchmark's live roofline figure: while `stream` is registered (capacity > 0; 0 or a
 * NULL stream un-registers), every fused Chebyshev-term launch (ds_spmm_union epilogue 1) issued on it - from Python or
 * from the native drivers - is bracketed by HIP events.  ds_profile_collect un-registers, waits for the events and
 * returns the number of records copied: duration [ms], nv, nnzb, ncols and `first` | element bytes of the vector
 * blocks << 8 (4: fp32 term, 2: bf16 term of ds_spmm_union16) of each launch - the algorithmic bytes follow from those.
 * One stream at a time. */
int ds_profile_stream(ds_stream_t stream, int64_t capacity);
/* Which launches the hook records (bit k = kind k; 0 restores the default, the fused term only).  The kind of a record
 * comes back in bits 16.. of ds_profile_collect's `first` word; its shape in (nv, nnzb, ncols):
 *   DS_PROF_TERM   fused Chebyshev term (ds_spmm_union / 16 / 16m epilogue 1)   nv, nnzb, ncols
 *   DS_PROF_KX     Y = K X   (ds_spmm_union epilogue 0)                        nv, nnzb, ncols
 *   DS_PROF_MX     Y = M_s X (ds_spmm_union epilogue 3)                        nv, nnzb, ncols
 *   DS_PROF_RESID  the walk of ds_union_residual (without its two reductions)  nv, nnzb, ncols
 *   DS_PROF_GRAM   ds_gram (partial products + reduction)                      p, n, q ; first = symmetric | fast << 1
 *   DS_PROF_MIX    ds_mix                                                      p, n, q */
#define DS_PROF_TERM 0
#define DS_PROF_KX 1
#define DS_PROF_MX 2
#define DS_PROF_RESID 3
#define DS_PROF_GRAM 4
#define DS_PROF_MIX 5
#define DS_PROF_KM 6 /* ds_spmm_union_km: nv, nnzb, ncols */
int ds_profile_kinds(unsigned mask);
int64_t ds_profile_collect(float* ms, int64_t* nv, int64_t* nnzb, int32_t* ncols, int32_t* first, int64_t cap);

/* ------------------------------------------------------------------------------------------------
 * STREAM triad  a = b + s c  (n f32 elements, n % 4 == 0, 16-byte aligned): the measured HBM bandwidth
 * (3 n 4 bytes per call) that bench.py quotes the SpMM against.  Not part of the modal path.
 * ---------------------------------------------------------------------------------------------- */
int ds_stream_triad(float* a, const float* b, const float* c, int64_t n, float s, ds_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DIFFSOUND_HIP_H */
