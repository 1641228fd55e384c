/*
 * clane_hip.h -- C ABI of libclane_hip.so: CLANE's iterative embedding loop on MI355X (gfx950).
 *
 * The reference (helloybz/CLANE @ v2) is pure Python on PyTorch-CPU and has no FFI.  This
 * header is the boundary a maintainer would bind (ctypes) underneath
 * the reference's own Python surface.  Each entry point names the reference code it
 * replaces (paths are relative to the reference checkout).
 *
 * Conventions
 *  - Every function returns 0 on success, <0 on error (CLANE_ERR_*); the text of the last
 *    error on the calling thread is clane_last_error().
 *  - All array arguments are DEVICE pointers to caller-owned memory.  The library never
 *    allocates, frees or retains them, keeps no global mutable state, and never
 *    synchronises: work is enqueued on `stream` (a hipStream_t, NULL = default stream).
 *  - Matrices are row-major with a leading dimension in ELEMENTS (ldz, ldx, ldo >= d).
 *    FAST PATH: when every matrix base pointer is 16-byte aligned and every leading
 *    dimension is a multiple of 16/sizeof(T), rows are moved with 16-byte accesses and the
 *    columns [d, roundup(d, 16/sizeof(T))) of each row are read AND written: they must be
 *    zero on entry (they stay zero).  Otherwise a 1-element-per-lane path is taken.
 *  - CSR: rowptr int64 [nrows+1], offsets into colidx / P; colidx int32, GLOBAL column ids
 *    (rows of the full Z), sorted and unique within a row (reference: graph.py:104-110,
 *    sparse_coo_tensor(...).coalesce(): row = source, col = destination).
 *  - `row0` is the global row id of local row 0 (row-partitioned multi-GPU runs hand each
 *    rank a contiguous block of rows; on one GPU row0 = 0).
 *  - Suffix _f32: T = float,  accumulate float,  P/scores float.
 *    Suffix _f64: T = double, accumulate double, P/scores double (C.npy may be float64 and
 *                 the reference keeps that dtype: graph.py:51).
 *    Suffix _bf16: T = bf16 storage for Z/X, accumulate float, P/scores float.
 *  - Reductions that feed control flow (the L1 delta) are deterministic: one partial per workgroup /
 *    listed row, summed in index order, no float atomics -- two launches give bitwise equal results.
 */
#ifndef CLANE_HIP_H_
#define CLANE_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CLANE_ABI_VERSION 4 /* 2: + clane_build_info, clane_xcc_ids, clane_check_csr, clane_spmm_update_class_*, clane_edge_score_class_*
                             * 3: clane_l1_distance_* takes `sq_a` (the rows' squared norms from the outer-delta pass: free),
                             *    clane_device_alloc_contiguous; every clane_spmm_update* took `sq_out` (the same norms out of K3)
                             * 4: `sq_out` is gone again: it cost every sweep 1.1-1.4 % to save one 0.34 ms pass per build_P, and a
                             *    propagate runs >= 11 sweeps per build_P (profiles/r04_fused_norms_ab.jsonl);
                             *    clane_spmm_update_class_* takes `flags` (CLANE_SPMM_TABLE_BEYOND_CACHE, also a flag of clane_spmm_update_*) */

#define CLANE_OK 0
#define CLANE_ERR_INVALID_ARGUMENT (-1)
#define CLANE_ERR_LAUNCH (-2)

/* score modes of clane_edge_score_* */
#define CLANE_SCORE_REFERENCE 0 /* dot / (||Z[src_all]||_F * ||Z[dst_all]||_F)  -- what similarity.py:37 computes */
#define CLANE_SCORE_PER_EDGE 1  /* dot / (||z_src|| * ||z_dst||)                 -- what its docstring describes  */
#define CLANE_SCORE_RAW_DOT 2   /* dot                                           -- stage test                     */

/* flags of clane_edge_score_* */
#define CLANE_SCORE_FUSE_SOFTMAX 1
/* clane_edge_score_class_* with CLANE_SCORE_FUSE_SOFTMAX: n (1..255) workgroups share the final rescale pass of each
 * listed row -- for graphs with rows of millions of edges; results do not depend on n; 0 means 1 */
#define CLANE_SCORE_ROW_PARTS(n) (((n) & 0xff) << 8)

/* flags of clane_spmm_update_* and clane_spmm_update_class_* */
#define CLANE_SPMM_SINKS_UNTOUCHED 1
/* A hint, never a change of results: the embedding table is far beyond the caches (the engine: more than twice the
 * 256 MiB Infinity Cache).  The instances with two fp32 rows per instruction (512-byte rows: column tiles / slices)
 * then keep 4 (row kernel) / 6 (class chunks) row loads in flight per wave instead of 8 and run 8 / 7 waves per SIMD
 * instead of 6 / 5 -- config 3 in two column tiles 3.79 -> 3.71 ms per sweep; a cache-resident table (config 2) loses
 * 18 % with it.  The class chunks with four bf16 rows per instruction keep 4 (config 4: 7.10 -> 6.99 ms).  With the
 * hint every finished row of z_new is also stored non-temporal (config 3 another 0.8 %; on a cache-resident table the
 * next sweep gathers what this one wrote and non-temporal stores cost 4 %). */
#define CLANE_SPMM_TABLE_BEYOND_CACHE 2

/* Optional further destinations of the rows a clane_spmm_update* call finishes: row r (relative to the call's
 * first row) is also stored at the places slot[row_ptr[r] .. row_ptr[r+1]); a place is (buffer << 28 | row) into
 * `bufs`, a DEVICE array of up to 8 matrix base addresses (leading dimension ld, element type of Z_new).
 * One buffer: the send buffer of the multi-GPU halo exchange gets packed by the kernel that produces a row
 * instead of by a separate gather pass.  Several: the other GPUs' tables mapped into this process (IPC) -- rows
 * go straight over xGMI.  aligned16: the caller vouches that every base is 16-byte aligned (the library cannot
 * look into device memory); 0 selects the scalar path.  NULL, or row_ptr == NULL: no mirror. */
typedef struct {
    const int64_t *row_ptr; /* [rows of the call + 1] */
    const int32_t *slot;
    void *const *bufs;
    int64_t ld;
    int32_t aligned16;
} clane_mirror_t;

int clane_abi_version(void);
const char *clane_last_error(void);
/* Compile-time tuning of this build as "KEY=value;..." (neighbour rows in flight per wave, waves and rows per
 * workgroup, non-temporal streams): measurements stored beside a benchmark (profiles/traffic.json) carry it, so a
 * number taken with another build is recognised as stale.  No reference counterpart. */
const char *clane_build_info(void);
/* Diagnostic: out[w] = the XCD (0..7, HW_REG_XCC_ID) that workgroup w of a launch of n_blocks workgroups of
 * block_threads threads ran on.  clane_spmm_update_class_* / clane_edge_score_class_* get their speed -- not their
 * results -- from workgroup w running on XCD (w + c) % 8 with c the same for the whole launch; the GPU test suite checks that with this call. */
int clane_xcc_ids(int32_t *out, int64_t n_blocks, int32_t block_threads, void *stream);
/* Validates on the device what every gather kernel below takes on trust -- call it once per graph before the first
 * sweep: rowptr[0..nrows] must be non-decreasing within [0, n_edges], every colidx[e] (e < n_edges) a row of a table of
 * table_rows rows.  *status (device int32, zeroed by the caller) gets bit 0 for a bad rowptr entry, bit 1 for a column
 * out of range; nothing else is touched.  (The reference cannot go wrong here: torch.sparse validates its indices,
 * graph.py:104-110.  A HIP kernel gathering through a bad index faults the GPU.) */
int clane_check_csr(const int64_t *rowptr, const int32_t *colidx, int64_t nrows, int64_t n_edges, int64_t table_rows,
                    int32_t *status, void *stream);

/* Doubles written by clane_spmm_update_*(nrows) plus clane_spmm_update_long_*(n_long). */
int64_t clane_spmm_partials_len(int64_t nrows, int64_t n_long);
/* Doubles of scratch needed by clane_degree_weighted_sums_* / clane_pair_cosine_*. */
int64_t clane_reduce_ws_len(void);

/* ---- Device memory that another process can map.  Everything else in this library works on memory the caller
 * owns; these five exist because a halo table that other GPUs store into (clane_mirror_t with several buffers)
 * must be its own allocation (hipIpcGetMemHandle works on allocation bases) and must be opened with peer access
 * from the device that is current in the calling process.  No reference counterpart (one process upstream).
 *   clane_device_alloc / _free : hipMalloc / hipFree on the current device.
 *   clane_ipc_export(ptr, handle)  : handle = CLANE_IPC_HANDLE_BYTES bytes to hand to the other process.
 *   clane_ipc_open(handle, &ptr)   : map it (hipIpcMemLazyEnablePeerAccess); clane_ipc_close(ptr) unmaps. */
#define CLANE_IPC_HANDLE_BYTES 64
int clane_device_alloc(int64_t bytes, void **ptr);
/* The same with PHYSICALLY CONTIGUOUS backing (hipExtMallocWithFlags, hipDeviceMallocContiguous): for the big gather
 * tables -- a random row gather over a 2 GB table runs up to 7 % slower when the driver backs it with scattered pages
 * (profiles/r04_placement_probe_*.jsonl).  Fails (CLANE_ERR_LAUNCH) when no contiguous range is free: fall back to
 * clane_device_alloc.  Freed with clane_device_free. */
int clane_device_alloc_contiguous(int64_t bytes, void **ptr);
int clane_device_free(void *ptr);
int clane_ipc_export(void *ptr, void *handle64);
int clane_ipc_open(const void *handle64, void **ptr);
int clane_ipc_close(void *ptr);

/* ---- K0: sq[v] = sum_k Z[v,k]^2.  Replaces the two `pow(2).sum()` of similarity.py:37
 * (together with clane_degree_weighted_sums_*). */
int clane_row_sqnorm_f32(const float *Z, int64_t nrows, int32_t d, int64_t ldz, float *sq, void *stream);
int clane_row_sqnorm_f64(const double *Z, int64_t nrows, int32_t d, int64_t ldz, double *sq, void *stream);
int clane_row_sqnorm_bf16(const uint16_t *Z, int64_t nrows, int32_t d, int64_t ldz, float *sq, void *stream);

/* out2[0] = sum_v outdeg_v*sq_v, out2[1] = sum_v indeg_v*sq_v over the local rows, in double
 * (= ||Z[src_all]||_F^2 and ||Z[dst_all]||_F^2 of similarity.py:37 for the edges built at
 * graph.py:119-120).  outdeg_v = rowptr[v+1]-rowptr[v]; indeg is caller-provided [nrows].
 * ws: clane_reduce_ws_len() doubles.  Multi-GPU: all-reduce out2 (sum) afterwards. */
int clane_degree_weighted_sums_f32(const float *sq, const int64_t *rowptr, const int32_t *indeg, int64_t nrows,
                                   double *ws, double *out2, void *stream);
int clane_degree_weighted_sums_f64(const double *sq, const int64_t *rowptr, const int32_t *indeg, int64_t nrows,
                                   double *ws, double *out2, void *stream);

/* ---- K1: per-edge similarity score in CSR order.  Replaces the gather + similarity call
 * of graph.py:119-121 and CosineSimilarity.__call__ (similarity.py:26-37) without
 * materialising Z[edges].  Source row of local row i is Z[row0+i].
 *   mode REFERENCE: sums2 = the (all-reduced) pair from clane_degree_weighted_sums_*; sq unused.
 *   mode PER_EDGE : sq = squared norms of ALL rows of Z; sums2 unused.
 *   mode RAW_DOT  : both unused.
 * Rows with more than `long_threshold` edges (0 = never) are cut into per-wave slices by a
 * second launch over `long_rows` (local row ids, n_long of them; one 16-wave workgroup per row);
 * pass n_long = 0 to have every row walked by a single wave.
 * flags & CLANE_SCORE_FUSE_SOFTMAX: every row this call scores is soft-maxed by it as well (graph.py:122-123):
 * a row walked by one wave in registers or with a running max / sum, a listed row by its workgroup ({max, sum}
 * per wave combined through LDS in wave order).  No clane_segment_softmax_* call is needed afterwards. */
int clane_edge_score_f32(const int64_t *rowptr, const int32_t *colidx, int64_t nrows, int64_t row0, const float *Z,
                         int64_t ldz, int32_t d, int32_t mode, const double *sums2, const float *sq, float *scores,
                         int32_t flags, int64_t long_threshold, const int32_t *long_rows, int64_t n_long, void *stream);
int clane_edge_score_f64(const int64_t *rowptr, const int32_t *colidx, int64_t nrows, int64_t row0, const double *Z,
                         int64_t ldz, int32_t d, int32_t mode, const double *sums2, const double *sq, double *scores,
                         int32_t flags, int64_t long_threshold, const int32_t *long_rows, int64_t n_long, void *stream);
int clane_edge_score_bf16(const int64_t *rowptr, const int32_t *colidx, int64_t nrows, int64_t row0,
                          const uint16_t *Z, int64_t ldz, int32_t d, int32_t mode, const double *sums2,
                          const float *sq, float *scores, int32_t flags, int64_t long_threshold,
                          const int32_t *long_rows, int64_t n_long, void *stream);

/* ---- K1b (column-split multi-GPU runs): scores holds RAW_DOT results summed over the GPUs (each GPU scored
 * its own columns); divide them by the denominators of `mode` exactly as clane_edge_score_* would have
 * (similarity.py:37).  sums2 / sq as for clane_edge_score_*, already summed over the GPUs.  RAW_DOT: no-op. */
int clane_edge_score_finalize_f32(const int64_t *rowptr, const int32_t *colidx, int64_t nrows, int64_t row0,
                                  int32_t mode, const double *sums2, const float *sq, float *scores, void *stream);
int clane_edge_score_finalize_f64(const int64_t *rowptr, const int32_t *colidx, int64_t nrows, int64_t row0,
                                  int32_t mode, const double *sums2, const double *sq, double *scores, void *stream);

/* ---- K2: in-place softmax of vals within each CSR row.  Replaces the per-row boolean-mask
 * loop of graph.py:122-123.  Rows with min_degree < deg <= max_degree (max_degree 0 = no upper
 * limit) are normalised by one wave each; the rows listed in `long_rows` (deg > min_degree) by one
 * 16-wave workgroup each.  Empty rows are skipped; max_degree <= min_degree (both > 0) disables the
 * one-wave pass.  Not needed after clane_edge_score_* with CLANE_SCORE_FUSE_SOFTMAX; used after
 * clane_edge_score_finalize_* and for plug-in scores. */
int clane_segment_softmax_f32(const int64_t *rowptr, int64_t nrows, float *vals, int64_t min_degree,
                              int64_t max_degree, const int32_t *long_rows, int64_t n_long, void *stream);
int clane_segment_softmax_f64(const int64_t *rowptr, int64_t nrows, double *vals, int64_t min_degree,
                              int64_t max_degree, const int32_t *long_rows, int64_t n_long, void *stream);

/* ---- K3: one Jacobi sweep over the local rows, fused with the L1 delta.  Replaces the
 * per-vertex loop embedder.py:84-92 and the reduction embedder.py:94:
 *     Z_new[i,:] = X[i,:] + gamma * sum_e P[e] * Z_old[colidx[e],:]     rowptr[i] <= e < rowptr[i+1]
 *     Z_new[i,:] = Z_old[row0+i,:]                                       if the row has no out-edge (embedder.py:88-89)
 *     delta_partials[b] = partial sums of |Z_new[i,:] - Z_old[row0+i,:]| (fixed order; reduce with clane_reduce_partials)
 * Z_new must not alias Z_old.
 *  clane_spmm_update_*      : one wave per row; rows with more than `long_threshold` edges (0 = never)
 *                             are skipped.  Writes clane_spmm_partials_len(nrows, 0) doubles.
 *                             flags & CLANE_SPMM_SINKS_UNTOUCHED: rows without out-edges are neither read
 *                             nor written -- the caller guarantees Z_new already equals Z_old there (they
 *                             never change, so two ping-pong buffers initialised alike stay alike).
 *  clane_spmm_update_long_* : the skipped rows, one workgroup of `waves_per_row` (4 or 16) waves per row,
 *                             each wave gathering a 64-aligned slice of the row, slices folded in wave
 *                             order; `long_rows` holds the local row ids (the caller bins rows by degree
 *                             once per graph: 4 waves suit rows of up to a few hundred edges, 16 the
 *                             hubs).  Writes n_long doubles.
 * Calls touch disjoint rows of Z_new and may run on different streams. */
int clane_spmm_update_f32(const int64_t *rowptr, const int32_t *colidx, const float *P, int64_t nrows, int64_t row0,
                          const float *Z_old, int64_t ldz, const float *X, int64_t ldx, float gamma, float *Z_new,
                          int64_t ldo, int32_t d, int64_t long_threshold, int32_t flags, const clane_mirror_t *mirror,
                          double *delta_partials, void *stream);
int clane_spmm_update_f64(const int64_t *rowptr, const int32_t *colidx, const double *P, int64_t nrows, int64_t row0,
                          const double *Z_old, int64_t ldz, const double *X, int64_t ldx, double gamma, double *Z_new,
                          int64_t ldo, int32_t d, int64_t long_threshold, int32_t flags, const clane_mirror_t *mirror,
                          double *delta_partials, void *stream);
int clane_spmm_update_bf16(const int64_t *rowptr, const int32_t *colidx, const float *P, int64_t nrows, int64_t row0,
                           const uint16_t *Z_old, int64_t ldz, const uint16_t *X, int64_t ldx, float gamma,
                           uint16_t *Z_new, int64_t ldo, int32_t d, int64_t long_threshold, int32_t flags,
                           const clane_mirror_t *mirror, double *delta_partials, void *stream);
int clane_spmm_update_long_f32(const int64_t *rowptr, const int32_t *colidx, const float *P, const int32_t *long_rows,
                               int64_t n_long, int32_t waves_per_row, int64_t row0, const float *Z_old, int64_t ldz, const float *X,
                               int64_t ldx, float gamma, float *Z_new, int64_t ldo, int32_t d,
                               const clane_mirror_t *mirror, double *delta_partials, void *stream);
int clane_spmm_update_long_f64(const int64_t *rowptr, const int32_t *colidx, const double *P,
                               const int32_t *long_rows, int64_t n_long, int32_t waves_per_row, int64_t row0, const double *Z_old,
                               int64_t ldz, const double *X, int64_t ldx, double gamma, double *Z_new, int64_t ldo,
                               int32_t d, const clane_mirror_t *mirror, double *delta_partials, void *stream);
int clane_spmm_update_long_bf16(const int64_t *rowptr, const int32_t *colidx, const float *P,
                                const int32_t *long_rows, int64_t n_long, int32_t waves_per_row, int64_t row0, const uint16_t *Z_old,
                                int64_t ldz, const uint16_t *X, int64_t ldx, float gamma, uint16_t *Z_new,
                                int64_t ldo, int32_t d, const clane_mirror_t *mirror, double *delta_partials,
                                void *stream);

/*  clane_spmm_update_split_* : hub rows, each cut into segments of `edges_per_segment` edges (a multiple of 64)
 *                             that are gathered by separate 16-wave workgroups; segment sums go to `slab`
 *                             (clane_spmm_split_slab_len(n_segments, d) accumulate-type elements) and are added
 *                             per row in segment order, then the usual epilogue.  split_rows[n_split] local row
 *                             ids; seg_ptr[n_split+1] prefix sums of the rows' segment counts
 *                             (ceil(deg / edges_per_segment)); seg_row[n_segments] = index into split_rows of the
 *                             row each segment belongs to.  Writes n_split doubles to delta_partials. */
int64_t clane_spmm_split_slab_len(int64_t n_segments, int32_t d);
int clane_spmm_update_split_f32(const int64_t *rowptr, const int32_t *colidx, const float *P, const int32_t *split_rows,
                                const int64_t *seg_ptr, const int32_t *seg_row, int64_t n_split, int64_t n_segments,
                                int64_t edges_per_segment, int64_t row0, const float *Z_old, int64_t ldz,
                                const float *X, int64_t ldx, float gamma, float *Z_new, int64_t ldo, int32_t d,
                                float *slab, const clane_mirror_t *mirror, double *delta_partials, void *stream);
int clane_spmm_update_split_f64(const int64_t *rowptr, const int32_t *colidx, const double *P,
                                const int32_t *split_rows, const int64_t *seg_ptr, const int32_t *seg_row,
                                int64_t n_split, int64_t n_segments, int64_t edges_per_segment, int64_t row0,
                                const double *Z_old, int64_t ldz, const double *X, int64_t ldx, double gamma,
                                double *Z_new, int64_t ldo, int32_t d, double *slab, const clane_mirror_t *mirror,
                                double *delta_partials, void *stream);
int clane_spmm_update_split_bf16(const int64_t *rowptr, const int32_t *colidx, const float *P,
                                 const int32_t *split_rows, const int64_t *seg_ptr, const int32_t *seg_row,
                                 int64_t n_split, int64_t n_segments, int64_t edges_per_segment, int64_t row0,
                                 const uint16_t *Z_old, int64_t ldz, const uint16_t *X, int64_t ldx, float gamma,
                                 uint16_t *Z_new, int64_t ldo, int32_t d, float *slab, const clane_mirror_t *mirror,
                                 double *delta_partials, void *stream);

/*  clane_edge_score_class_* : K1 (graph.py:119-123 + similarity.py:26-37) over the listed long rows with XCD-affine
 *                             gathers -- the build_P counterpart of clane_spmm_update_class_* below, over the SAME item
 *                             arrays plus item_row (local row id of each item's source row).  A wave scores the edges
 *                             of one item (modes and `sums2` / `sq` as clane_edge_score_*).  With
 *                             CLANE_SCORE_FUSE_SOFTMAX every item leaves {max, sum exp} in stats[2 * slot] (2 *
 *                             n_slots accumulate-type elements) and each listed row is then soft-maxed from its
 *                             slots, combined in slot order; without it rowptr / class_rows / slot_ptr / stats may be
 *                             NULL and the raw scores stay (column-split runs all-reduce them first). */
int clane_edge_score_class_f32(const int64_t *rowptr, const int32_t *colidx, const int64_t *item_e0,
                               const int32_t *item_len, const int32_t *item_slot, const int32_t *item_row,
                               int64_t n_blocks, int32_t items_per_block, const int32_t *class_rows,
                               const int64_t *slot_ptr, int64_t n_rows, int64_t row0, const float *Z, int64_t ldz,
                               int32_t d, int32_t mode, const double *sums2, const float *sq, float *scores,
                               int32_t flags, float *stats, void *stream);
int clane_edge_score_class_f64(const int64_t *rowptr, const int32_t *colidx, const int64_t *item_e0,
                               const int32_t *item_len, const int32_t *item_slot, const int32_t *item_row,
                               int64_t n_blocks, int32_t items_per_block, const int32_t *class_rows,
                               const int64_t *slot_ptr, int64_t n_rows, int64_t row0, const double *Z, int64_t ldz,
                               int32_t d, int32_t mode, const double *sums2, const double *sq, double *scores,
                               int32_t flags, double *stats, void *stream);
int clane_edge_score_class_bf16(const int64_t *rowptr, const int32_t *colidx, const int64_t *item_e0,
                                const int32_t *item_len, const int32_t *item_slot, const int32_t *item_row,
                                int64_t n_blocks, int32_t items_per_block, const int32_t *class_rows,
                                const int64_t *slot_ptr, int64_t n_rows, int64_t row0, const uint16_t *Z, int64_t ldz,
                                int32_t d, int32_t mode, const double *sums2, const float *sq, float *scores,
                                int32_t flags, float *stats, void *stream);

/*  clane_spmm_update_class_* : long rows whose gathers are kept XCD-affine (no reference counterpart: the reference's
 *                             loop is embedder.py:84-92 for every row alike).  MI355X has 8 XCDs with a private 4 MiB
 *                             L2 each and deals workgroups to them round-robin (workgroup w -> XCD (w + c) % 8, c fixed within a launch).  The caller
 *                             gives every table row a CLASS 0..7 (the engine: an xor-fold of the row number's 3-bit groups -- not row % 8,
 *                             which would pin low address bits and use only part of an L2), sorts the edges of each listed row by
 *                             (class of the column, column) and cuts every class segment
 *                             into ITEMS of a few hundred edges; item arrays are laid out in blocks of
 *                             `items_per_block` (4..64) items of ONE class, block j of class b at block index
 *                             8 j + b (n_blocks blocks, padding items have item_len = 0), so XCD b only gathers rows
 *                             of class b and the eight L2s cache different eighths of the hot rows.
 *                             item_e0 / item_len: edge range of an item in colidx / P; item_slot: where its partial
 *                             sum goes in `slab` (clane_spmm_class_slab_len(n_slots, d) accumulate-type elements,
 *                             16-byte aligned).  The ORDER of the blocks is the caller's too: the engine puts the
 *                             pieces of its heaviest rows first, sub-class by sub-class, so that an XCD's hot
 *                             working set at any moment is a fraction of its class (clane_amd/xcd.py).
 *                             class_rows[n_rows] local row ids; slot_ptr[n_rows+1]: the slots of
 *                             row i are [slot_ptr[i], slot_ptr[i+1]) and are added in that order (reproducible),
 *                             then the usual epilogue.  Writes n_rows doubles to delta_partials.
 *                             flags: CLANE_SPMM_TABLE_BEYOND_CACHE or 0. */
int64_t clane_spmm_class_slab_len(int64_t n_slots, int32_t d);
int clane_spmm_update_class_f32(const int32_t *colidx, const float *P, const int64_t *item_e0, const int32_t *item_len,
                                const int32_t *item_slot, int64_t n_blocks, int32_t items_per_block,
                                const int32_t *class_rows, const int64_t *slot_ptr, int64_t n_rows, int64_t row0,
                                const float *Z_old, int64_t ldz, const float *X, int64_t ldx, float gamma, float *Z_new,
                                int64_t ldo, int32_t d, int32_t flags, float *slab, const clane_mirror_t *mirror,
                                double *delta_partials, void *stream);
int clane_spmm_update_class_f64(const int32_t *colidx, const double *P, const int64_t *item_e0, const int32_t *item_len,
                                const int32_t *item_slot, int64_t n_blocks, int32_t items_per_block,
                                const int32_t *class_rows, const int64_t *slot_ptr, int64_t n_rows, int64_t row0,
                                const double *Z_old, int64_t ldz, const double *X, int64_t ldx, double gamma,
                                double *Z_new, int64_t ldo, int32_t d, int32_t flags, double *slab, const clane_mirror_t *mirror,
                                double *delta_partials, void *stream);
int clane_spmm_update_class_bf16(const int32_t *colidx, const float *P, const int64_t *item_e0, const int32_t *item_len,
                                 const int32_t *item_slot, int64_t n_blocks, int32_t items_per_block,
                                 const int32_t *class_rows, const int64_t *slot_ptr, int64_t n_rows, int64_t row0,
                                 const uint16_t *Z_old, int64_t ldz, const uint16_t *X, int64_t ldx, float gamma,
                                 uint16_t *Z_new, int64_t ldo, int32_t d, int32_t flags, float *slab, const clane_mirror_t *mirror,
                                 double *delta_partials, void *stream);

/* out[0] = sum of partials[0..n) in a fixed order (bitwise reproducible).  Finishes embedder.py:94 / :60.
 * ws: clane_reduce_ws_len() doubles. */
int clane_reduce_partials(const double *partials, int64_t n, double *ws, double *out, void *stream);

/* sum|A - B| over an [nrows, d] matrix pair -> out[0] (outer-loop delta, embedder.py:60).
 * sq_a (accumulate type, [nrows]; NULL: not wanted): also |A[i,:]|^2 for every row, bit for bit what clane_row_sqnorm_*
 * gives on A -- the pass reads every row of A anyway, so with A = the new embeddings the next build_P of the outer loop
 * (embedder.py:59 after :60) gets its norms (similarity.py:37) for no extra traffic.
 * ws: clane_reduce_ws_len() doubles. */
int clane_l1_distance_f32(const float *A, int64_t lda, const float *B, int64_t ldb, int64_t nrows, int32_t d,
                          float *sq_a, double *ws, double *out, void *stream);
int clane_l1_distance_f64(const double *A, int64_t lda, const double *B, int64_t ldb, int64_t nrows, int32_t d,
                          double *sq_a, double *ws, double *out, void *stream);
int clane_l1_distance_bf16(const uint16_t *A, int64_t lda, const uint16_t *B, int64_t ldb, int64_t nrows, int32_t d,
                           float *sq_a, double *ws, double *out, void *stream);

/* dst[i,:] = src[idx[i],:] for i < n: packs the rows other ranks read into the send buffer of the
 * multi-GPU halo exchange (no counterpart in the single-process reference). */
int clane_gather_rows_f32(const float *src, int64_t lds, const int32_t *idx, int64_t n, int32_t d, float *dst,
                          int64_t ldd, void *stream);
int clane_gather_rows_f64(const double *src, int64_t lds, const int32_t *idx, int64_t n, int32_t d, double *dst,
                          int64_t ldd, void *stream);
int clane_gather_rows_bf16(const uint16_t *src, int64_t lds, const int32_t *idx, int64_t n, int32_t d, uint16_t *dst,
                           int64_t ldd, void *stream);

/* ---- CosineSimilarity.__call__ on explicit pairs (similarity.py:26-37):
 *   out[i] = dot(A[i,:], B[i,:]) / (||A||_F * ||B||_F)      -- global denominators.
 * ws: clane_reduce_ws_len() doubles. */
int clane_pair_cosine_f32(const float *A, int64_t lda, const float *B, int64_t ldb, int64_t nrows, int32_t d,
                          float *out, double *ws, void *stream);
int clane_pair_cosine_f64(const double *A, int64_t lda, const double *B, int64_t ldb, int64_t nrows, int32_t d,
                          double *out, double *ws, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* CLANE_HIP_H_ */
