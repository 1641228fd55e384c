// extern "C" surface of libclane_hip.so (see include/clane_hip.h): argument validation,
// lane-layout selection and kernel launches.  No allocation, no synchronisation, no state.
#include "../../include/clane_hip.h"

#include <hip/hip_runtime.h>

#include <climits>
#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "build_p.h"
#include "edge_score.h"
#include "device_utils.h"
#include "spmm_update.h"

namespace {

using namespace clane;

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int check_launch(const char *what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(CLANE_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return CLANE_OK;
}

constexpr int kReduceGrid = 1024;              // partial blocks of the two-stage reductions
constexpr int64_t kReduceWs = 2 * kReduceGrid + 2;  // + the two finished sums
#ifndef CLANE_LONG_U
#define CLANE_LONG_U CLANE_SPMM_U    // neighbour-row loads in flight per wave of the workgroup-per-row kernels
#endif
#ifndef CLANE_SUBROW_U32
#define CLANE_SUBROW_U32 4        // CLANE_SPMM_TABLE_BEYOND_CACHE: row loads in flight in spmm_update_subrow_kernel, two fp32 rows per instruction
#endif
#ifndef CLANE_CLASS_U16B
#define CLANE_CLASS_U16B 4        // ... and in its instance with four bf16 rows per instruction (config 4)
#endif
#ifndef CLANE_CLASS_U32
#define CLANE_CLASS_U32 6         // ... and in spmm_class_chunk_kernel (72 registers: 7 waves per SIMD instead of 5)
#endif
#ifndef CLANE_LONG_WAVES
#define CLANE_LONG_WAVES 16         // waves of the workgroup-per-row kernels (A/B builds: 8)
#endif
constexpr int kLongWaves = CLANE_LONG_WAVES;
#ifndef CLANE_ROWS_PER_BLOCK
#define CLANE_ROWS_PER_BLOCK 32   // minimum consecutive rows per workgroup of the row kernels
#endif

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Row kernels (K1, K2, K3): a workgroup owns `rows_per_block` consecutive rows.  The grid aims at ~32k workgroups
// (>> the 2048 resident ones, so the dispatcher load-balances skewed rows) of at least CLANE_ROWS_PER_BLOCK rows:
// a workgroup's start-up -- rowptr slice to LDS, barrier, first claims -- is a chain of dependent round trips, and
// with 8 rows per wave-instruction (128-byte rows) 32 rows are ONE row per sub-wave.  Measured
// (profiles/r02_ab_rows_per_block.md): 2M rows x 128 B 0.760 -> 0.712 ms at 64 rows; 10M rows bf16 8.6 -> 8.2 ms at
// 256; a 200k-row graph loses 12 % above 32 (too few workgroups), 2M rows lose 30 % at 256.
#ifndef CLANE_TARGET_GRID
#define CLANE_TARGET_GRID 32768
#endif
inline int rows_per_block(int64_t nrows) {
    const int64_t r = ceil_div(ceil_div(nrows, CLANE_TARGET_GRID), kWavesPerBlock) * kWavesPerBlock;
    return int(r < CLANE_ROWS_PER_BLOCK ? CLANE_ROWS_PER_BLOCK : r > kMaxRowsPerBlock ? kMaxRowsPerBlock : r);
}
inline int row_grid(int64_t nrows) { return int(ceil_div(nrows > 0 ? nrows : 1, rows_per_block(nrows))); }

inline int grid_for_waves(int64_t nwaves_wanted) {
    int64_t g = ceil_div(nwaves_wanted, kWavesPerBlock);
    if (g > kMaxGrid) g = kMaxGrid;
    if (g < 1) g = 1;
    return int(g);
}

// Lane layout: 16-byte packs when every operand allows it, else one element per lane.
struct Layout {
    bool vec;  // 16-byte path
    int lpr;   // lanes per row (power of two)
};

template <typename T>
Layout pick_layout(int d, std::initializer_list<const void *> ptrs, std::initializer_list<int64_t> lds) {
    constexpr int KV = Elem<T>::kVec;
    bool vec = true;
    for (const void *p : ptrs) vec = vec && aligned16(p);
    for (int64_t ld : lds) vec = vec && (ld % KV == 0);
    Layout L;
    L.vec = vec;
    if (vec) {
        const int packs = int(ceil_div(d, KV));
        L.lpr = packs <= 8 ? 8 : packs <= 16 ? 16 : packs <= 32 ? 32 : 64;
    } else {
        L.lpr = d <= 4 ? 4 : d <= 16 ? 16 : 64;
    }
    return L;
}

// Calls f.template operator()<VEC, LPR>() for the chosen layout.
template <typename T, typename F>
void dispatch_layout(const Layout &L, F &&f) {
    constexpr int KV = Elem<T>::kVec;
    if (L.vec) {
        switch (L.lpr) {
            case 8: f.template operator()<KV, 8>(); break;
            case 16: f.template operator()<KV, 16>(); break;
            case 32: f.template operator()<KV, 32>(); break;
            default: f.template operator()<KV, 64>(); break;
        }
    } else {
        switch (L.lpr) {
            case 4: f.template operator()<1, 4>(); break;
            case 16: f.template operator()<1, 16>(); break;
            default: f.template operator()<1, 64>(); break;
        }
    }
}

template <typename T>
Mirror<T> make_mirror(const clane_mirror_t *m) {
    if (m == nullptr || m->row_ptr == nullptr) return Mirror<T>{nullptr, nullptr, nullptr, 0};
    return Mirror<T>{m->row_ptr, m->slot, reinterpret_cast<T *const *>(m->bufs), m->ld};
}
// The bases live in device memory: the caller vouches for their alignment; a mirror that is not 16-byte
// aligned sends the call down the scalar path (an odd address in the alignment check).
inline const void *mirror_alignment_probe(const clane_mirror_t *m, const void *aligned) {
    return (m && m->row_ptr && !m->aligned16) ? reinterpret_cast<const void *>(uintptr_t(1)) : aligned;
}
inline bool mirror_ok(const clane_mirror_t *m, int d) {
    return m == nullptr || m->row_ptr == nullptr || (m->slot != nullptr && m->bufs != nullptr && m->ld >= d);
}

#define REQUIRE(cond, ...) \
    do {                   \
        if (!(cond)) return fail(CLANE_ERR_INVALID_ARGUMENT, __VA_ARGS__); \
    } while (0)

// ---------------------------------------------------------------------------------------------
template <typename T>
int row_sqnorm(const T *Z, int64_t nrows, int32_t d, int64_t ldz, typename Elem<T>::acc_t *sq, void *stream) {
    REQUIRE(nrows >= 0 && d > 0 && ldz >= d, "row_sqnorm: bad shape nrows=%lld d=%d ldz=%lld", (long long)nrows, d,
            (long long)ldz);
    if (nrows == 0) return CLANE_OK;
    REQUIRE(Z && sq, "row_sqnorm: null pointer");
    const Layout L = pick_layout<T>(d, {Z}, {ldz});
    dispatch_layout<T>(L, [&]<int VEC, int LPR>() {
        const int grid = grid_for_waves(ceil_div(nrows, (kWave / LPR) * stream_rows(LPR)));
        row_sqnorm_kernel<T, VEC, LPR><<<grid, kBlock, 0, (hipStream_t)stream>>>(Z, nrows, d, ldz, sq);
    });
    return check_launch("row_sqnorm");
}

template <typename A>
int degree_weighted_sums(const A *sq, const int64_t *rowptr, const int32_t *indeg, int64_t nrows, double *ws,
                         double *out2, void *stream) {
    REQUIRE(nrows >= 0, "degree_weighted_sums: nrows < 0");
    REQUIRE(ws && out2, "degree_weighted_sums: null workspace/output");
    REQUIRE(nrows == 0 || (sq && rowptr && indeg), "degree_weighted_sums: null pointer");
    int grid = int(ceil_div(nrows > 0 ? nrows : 1, kBlock));
    if (grid > kReduceGrid) grid = kReduceGrid;
    degree_weighted_kernel<A><<<grid, kBlock, 0, (hipStream_t)stream>>>(sq, rowptr, indeg, nrows, ws);
    reduce_fixed_kernel<<<1, 1024, 0, (hipStream_t)stream>>>(ws, grid, grid, 2, out2);
    return check_launch("degree_weighted_sums");
}

template <typename T>
int edge_score(const int64_t *rowptr, const int32_t *colidx, int64_t nrows, int64_t row0, const T *Z, int64_t ldz,
               int32_t d, int32_t mode, const double *sums2, const typename Elem<T>::acc_t *sq,
               typename Elem<T>::acc_t *scores, int32_t flags, int64_t long_threshold, const int32_t *long_rows,
               int64_t n_long, void *stream) {
    REQUIRE(nrows >= 0 && row0 >= 0 && d > 0 && ldz >= d, "edge_score: bad shape");
    REQUIRE(mode == CLANE_SCORE_REFERENCE || mode == CLANE_SCORE_PER_EDGE || mode == CLANE_SCORE_RAW_DOT,
            "edge_score: unknown mode %d", mode);
    REQUIRE(long_threshold >= 0 && n_long >= 0 && n_long <= INT32_MAX, "edge_score: bad long-row parameters");
    REQUIRE(n_long == 0 || (long_rows && long_threshold > 0), "edge_score: long_rows needs a list and a threshold");
    if (nrows == 0) return CLANE_OK;
    REQUIRE(rowptr && colidx && Z && scores, "edge_score: null pointer");
    REQUIRE(mode != CLANE_SCORE_REFERENCE || sums2, "edge_score: mode REFERENCE needs sums2");
    REQUIRE(mode != CLANE_SCORE_PER_EDGE || sq, "edge_score: mode PER_EDGE needs sq");
    const Layout L = pick_layout<T>(d, {Z}, {ldz});
    const bool fuse = (flags & CLANE_SCORE_FUSE_SOFTMAX) != 0;
    dispatch_layout<T>(L, [&]<int VEC, int LPR>() {
        constexpr int U = VEC > 1 ? 8 : 4;
        if constexpr (LPR < kWave && VEC > 1)      // narrow rows: one sub-wave per source row
            edge_score_subrow_kernel<T, VEC, LPR, U><<<row_grid(nrows), kBlock, 0, (hipStream_t)stream>>>(
                rowptr, colidx, nrows, row0, Z, ldz, d, mode, sums2, sq, scores, long_threshold, fuse,
                rows_per_block(nrows));
        else
            edge_score_kernel<T, VEC, LPR, U><<<row_grid(nrows), kBlock, 0, (hipStream_t)stream>>>(
                rowptr, colidx, nrows, row0, Z, ldz, d, mode, sums2, sq, scores, long_threshold, fuse,
                rows_per_block(nrows));
        if (n_long > 0)
            edge_score_long_kernel<T, VEC, LPR, U, kLongWaves>
                <<<unsigned(n_long), kLongWaves * kWave, 0, (hipStream_t)stream>>>(
                    rowptr, colidx, long_rows, row0, Z, ldz, d, mode, sums2, sq, scores, fuse);
    });
    return check_launch("edge_score");
}

template <typename T>
int edge_score_class(const int64_t *rowptr, const int32_t *colidx, const int64_t *item_e0, const int32_t *item_len,
                     const int32_t *item_slot, const int32_t *item_row, int64_t n_blocks, int32_t items_per_block,
                     const int32_t *class_rows, const int64_t *slot_ptr, int64_t n_rows, int64_t row0, const T *Z,
                     int64_t ldz, int32_t d, int32_t mode, const double *sums2, const typename Elem<T>::acc_t *sq,
                     typename Elem<T>::acc_t *scores, int32_t flags, typename Elem<T>::acc_t *stats, void *stream) {
    REQUIRE(n_blocks >= 0 && n_blocks <= INT32_MAX && n_rows >= 0 && n_rows <= INT32_MAX && row0 >= 0 && d > 0 &&
                ldz >= d,
            "edge_score_class: bad shape");
    REQUIRE(items_per_block >= kWavesPerBlock && items_per_block <= kMaxItemsPerBlock,
            "edge_score_class: items_per_block must be in [%d, %d]", kWavesPerBlock, kMaxItemsPerBlock);
    REQUIRE(mode == CLANE_SCORE_REFERENCE || mode == CLANE_SCORE_PER_EDGE || mode == CLANE_SCORE_RAW_DOT,
            "edge_score_class: unknown mode %d", mode);
    if (n_rows == 0 || n_blocks == 0) return CLANE_OK;
    const bool fuse = (flags & CLANE_SCORE_FUSE_SOFTMAX) != 0;
    REQUIRE(colidx && item_e0 && item_len && item_slot && item_row && Z && scores, "edge_score_class: null pointer");
    REQUIRE(!fuse || (rowptr && class_rows && slot_ptr && stats),
            "edge_score_class: CLANE_SCORE_FUSE_SOFTMAX needs rowptr, class_rows, slot_ptr and stats");
    REQUIRE(mode != CLANE_SCORE_REFERENCE || sums2, "edge_score_class: mode REFERENCE needs sums2");
    REQUIRE(mode != CLANE_SCORE_PER_EDGE || sq, "edge_score_class: mode PER_EDGE needs sq");
    const Layout L = pick_layout<T>(d, {Z}, {ldz});
    dispatch_layout<T>(L, [&]<int VEC, int LPR>() {
        constexpr int U = VEC > 1 ? 8 : 4;
        edge_score_class_kernel<T, VEC, LPR, U><<<unsigned(n_blocks), kBlock, 0, (hipStream_t)stream>>>(
            colidx, item_e0, item_len, item_slot, item_row, items_per_block, row0, Z, ldz, d, mode, sums2, sq, scores,
            fuse ? stats : nullptr);
    });
    if (fuse) {
        int parts = (flags >> 8) & 0xff;        // CLANE_SCORE_ROW_PARTS(n): workgroups per row of the rescale pass
        if (parts < 1) parts = 1;
        edge_softmax_class_kernel<typename Elem<T>::acc_t>
            <<<dim3(unsigned(n_rows), unsigned(parts)), kBlock, 0, (hipStream_t)stream>>>(rowptr, class_rows, slot_ptr,
                                                                                            stats, scores);
    }
    return check_launch("edge_score_class");
}

template <typename A>
int segment_softmax(const int64_t *rowptr, int64_t nrows, A *vals, int64_t min_degree, int64_t max_degree,
                    const int32_t *long_rows, int64_t n_long, void *stream) {
    REQUIRE(nrows >= 0 && min_degree >= 0 && max_degree >= 0 && n_long >= 0 && n_long <= INT32_MAX,
            "segment_softmax: negative argument");
    REQUIRE(n_long == 0 || (long_rows && max_degree > 0), "segment_softmax: long_rows needs a list and max_degree");
    if (nrows == 0) return CLANE_OK;
    REQUIRE(rowptr && vals, "segment_softmax: null pointer");
    if (!(max_degree > 0 && max_degree <= min_degree))
        segment_softmax_kernel<A><<<row_grid(nrows), kBlock, 0, (hipStream_t)stream>>>(
            rowptr, nrows, vals, min_degree, max_degree, rows_per_block(nrows));
    if (n_long > 0)
        segment_softmax_long_kernel<A, kLongWaves><<<unsigned(n_long), kLongWaves * kWave, 0, (hipStream_t)stream>>>(
            rowptr, long_rows, vals, min_degree);
    return check_launch("segment_softmax");
}

template <typename A>
int edge_score_finalize(const int64_t *rowptr, const int32_t *colidx, int64_t nrows, int64_t row0, int32_t mode,
                        const double *sums2, const A *sq, A *scores, void *stream) {
    REQUIRE(nrows >= 0 && row0 >= 0, "edge_score_finalize: bad shape");
    REQUIRE(mode == CLANE_SCORE_REFERENCE || mode == CLANE_SCORE_PER_EDGE || mode == CLANE_SCORE_RAW_DOT,
            "edge_score_finalize: unknown mode %d", mode);
    if (nrows == 0 || mode == CLANE_SCORE_RAW_DOT) return CLANE_OK;
    REQUIRE(rowptr && colidx && scores, "edge_score_finalize: null pointer");
    REQUIRE(mode != CLANE_SCORE_REFERENCE || sums2, "edge_score_finalize: reference mode needs sums2");
    REQUIRE(mode != CLANE_SCORE_PER_EDGE || sq, "edge_score_finalize: per_edge mode needs sq");
    edge_score_finalize_kernel<A><<<row_grid(nrows), kBlock, 0, (hipStream_t)stream>>>(
        rowptr, colidx, nrows, row0, mode, sums2, sq, scores, rows_per_block(nrows));
    return check_launch("edge_score_finalize");
}

inline int64_t spmm_main_grid(int64_t nrows) { return row_grid(nrows); }

template <typename T, typename PT>
int spmm_update(const int64_t *rowptr, const int32_t *colidx, const PT *P, int64_t nrows, int64_t row0,
                const T *Z_old, int64_t ldz, const T *X, int64_t ldx, typename Elem<T>::acc_t gamma, T *Z_new,
                int64_t ldo, int32_t d, int64_t long_threshold, int32_t flags, const clane_mirror_t *mirror,
                double *delta_partials, void *stream) {
    REQUIRE(nrows >= 0 && row0 >= 0 && d > 0, "spmm_update: bad shape nrows=%lld row0=%lld d=%d", (long long)nrows,
            (long long)row0, d);
    REQUIRE(mirror_ok(mirror, d), "spmm_update: incomplete mirror descriptor");
    REQUIRE(ldz >= d && ldx >= d && ldo >= d, "spmm_update: leading dimension < d");
    REQUIRE(long_threshold >= 0, "spmm_update: negative long_threshold");
    REQUIRE(delta_partials, "spmm_update: null delta_partials");
    if (nrows == 0) return CLANE_OK;
    REQUIRE(rowptr && colidx && P && Z_old && X && Z_new, "spmm_update: null pointer");
    REQUIRE((const void *)Z_new != (const void *)Z_old, "spmm_update: Z_new must not alias Z_old (Jacobi sweep)");
    const Layout L = pick_layout<T>(d, {Z_old, X, Z_new, mirror_alignment_probe(mirror, Z_new)},
                                    {ldz, ldx, ldo, mirror && mirror->row_ptr ? mirror->ld : ldo});
    const int grid = int(spmm_main_grid(nrows));
    const Mirror<T> mir = make_mirror<T>(mirror);
    dispatch_layout<T>(L, [&]<int VEC, int LPR>() {
        constexpr int U = VEC > 1 ? CLANE_SPMM_U : 4;
        auto launch = [&]<bool MIRRORED>() {          // the mirrored instances prefetch a row's places (spmm_update.h)
            if constexpr (LPR == 32 && VEC == 4 && sizeof(T) == 4) {
                // two fp32 rows per instruction (512-byte rows: the column tiles, config 2, the N = 2 column slices): with the
                // table far beyond the caches 4 row loads in flight and 8 waves per SIMD beat 8 and 6 (config 3 in two
                // tiles: row pass 1.58 -> 1.52 ms); a cache-resident table wants the 8 (config 2: 0.103 vs 0.120 ms)
                if (flags & CLANE_SPMM_TABLE_BEYOND_CACHE) {
                    spmm_update_subrow_kernel<T, PT, VEC, LPR, CLANE_SUBROW_U32, MIRRORED>
                        <<<grid, kBlock, 0, (hipStream_t)stream>>>(
                            rowptr, colidx, P, nrows, row0, Z_old, ldz, X, ldx, gamma, Z_new, ldo, d, long_threshold,
                            (flags & CLANE_SPMM_SINKS_UNTOUCHED) != 0, rows_per_block(nrows), mir,
                            (flags & CLANE_SPMM_TABLE_BEYOND_CACHE) != 0, delta_partials);
                    return;
                }
            }
            if constexpr (LPR < kWave && VEC > 1)      // short rows of narrow matrices: one sub-wave per row
                spmm_update_subrow_kernel<T, PT, VEC, LPR, U, MIRRORED><<<grid, kBlock, 0, (hipStream_t)stream>>>(
                    rowptr, colidx, P, nrows, row0, Z_old, ldz, X, ldx, gamma, Z_new, ldo, d, long_threshold,
                    (flags & CLANE_SPMM_SINKS_UNTOUCHED) != 0, rows_per_block(nrows), mir,
                            (flags & CLANE_SPMM_TABLE_BEYOND_CACHE) != 0, delta_partials);
            else
                spmm_update_kernel<T, PT, VEC, LPR, U, MIRRORED><<<grid, kBlock, 0, (hipStream_t)stream>>>(
                    rowptr, colidx, P, nrows, row0, Z_old, ldz, X, ldx, gamma, Z_new, ldo, d, long_threshold,
                    (flags & CLANE_SPMM_SINKS_UNTOUCHED) != 0, rows_per_block(nrows), mir,
                            (flags & CLANE_SPMM_TABLE_BEYOND_CACHE) != 0, delta_partials);
        };
        if (mir.row_ptr != nullptr) launch.template operator()<true>();
        else launch.template operator()<false>();
    });
    return check_launch("spmm_update");
}

template <typename T, typename PT>
int spmm_update_long(const int64_t *rowptr, const int32_t *colidx, const PT *P, const int32_t *long_rows,
                     int64_t n_long, int32_t waves_per_row, int64_t row0, const T *Z_old, int64_t ldz, const T *X, int64_t ldx,
                     typename Elem<T>::acc_t gamma, T *Z_new, int64_t ldo, int32_t d, const clane_mirror_t *mirror,
                     double *delta_partials, void *stream) {
    REQUIRE(n_long >= 0 && n_long <= INT32_MAX && row0 >= 0 && d > 0, "spmm_update_long: bad shape");
    REQUIRE(mirror_ok(mirror, d), "spmm_update_long: incomplete mirror descriptor");
    REQUIRE(ldz >= d && ldx >= d && ldo >= d, "spmm_update_long: leading dimension < d");
    if (n_long == 0) return CLANE_OK;
    REQUIRE(rowptr && colidx && P && long_rows && Z_old && X && Z_new && delta_partials,
            "spmm_update_long: null pointer");
    REQUIRE((const void *)Z_new != (const void *)Z_old, "spmm_update_long: Z_new must not alias Z_old");
    const Layout L = pick_layout<T>(d, {Z_old, X, Z_new, mirror_alignment_probe(mirror, Z_new)},
                                    {ldz, ldx, ldo, mirror && mirror->row_ptr ? mirror->ld : ldo});
    REQUIRE(waves_per_row == 4 || waves_per_row == 16, "spmm_update_long: waves_per_row must be 4 or 16");
    const Mirror<T> mir = make_mirror<T>(mirror);
    dispatch_layout<T>(L, [&]<int VEC, int LPR>() {
        constexpr int U = VEC > 1 ? CLANE_LONG_U : 4;
        if (waves_per_row == 4)
            spmm_long_kernel<T, PT, VEC, LPR, U, 4><<<int(n_long), 4 * kWave, 0, (hipStream_t)stream>>>(
                rowptr, colidx, P, long_rows, row0, Z_old, ldz, X, ldx, gamma, Z_new, ldo, d, mir, delta_partials);
        else
            spmm_long_kernel<T, PT, VEC, LPR, U, kLongWaves>
                <<<int(n_long), kLongWaves * kWave, 0, (hipStream_t)stream>>>(
                    rowptr, colidx, P, long_rows, row0, Z_old, ldz, X, ldx, gamma, Z_new, ldo, d, mir, delta_partials);
    });
    return check_launch("spmm_update_long");
}

template <typename T, typename PT>
int spmm_update_split(const int64_t *rowptr, const int32_t *colidx, const PT *P, const int32_t *split_rows,
                      const int64_t *seg_ptr, const int32_t *seg_row, int64_t n_split, int64_t n_segments,
                      int64_t edges_per_segment, int64_t row0, const T *Z_old, int64_t ldz, const T *X, int64_t ldx,
                      typename Elem<T>::acc_t gamma, T *Z_new, int64_t ldo, int32_t d,
                      typename Elem<T>::acc_t *slab, const clane_mirror_t *mirror, double *delta_partials,
                      void *stream) {
    REQUIRE(mirror_ok(mirror, d), "spmm_update_split: incomplete mirror descriptor");
    REQUIRE(n_split >= 0 && n_split <= INT32_MAX && n_segments >= n_split && n_segments <= INT32_MAX && row0 >= 0 &&
                d > 0 && edges_per_segment >= kWave && edges_per_segment % kWave == 0,
            "spmm_update_split: bad shape (edges_per_segment must be a positive multiple of 64)");
    REQUIRE(ldz >= d && ldx >= d && ldo >= d, "spmm_update_split: leading dimension < d");
    if (n_split == 0) return CLANE_OK;
    REQUIRE(rowptr && colidx && P && split_rows && seg_ptr && seg_row && Z_old && X && Z_new && slab && delta_partials,
            "spmm_update_split: null pointer");
    REQUIRE((const void *)Z_new != (const void *)Z_old, "spmm_update_split: Z_new must not alias Z_old");
    REQUIRE(aligned16(slab), "spmm_update_split: slab must be 16-byte aligned");
    const Layout L = pick_layout<T>(d, {Z_old, X, Z_new, mirror_alignment_probe(mirror, Z_new)},
                                    {ldz, ldx, ldo, mirror && mirror->row_ptr ? mirror->ld : ldo});
    const int64_t ld_slab = ceil_div(d, 8) * 8;
    const Mirror<T> mir = make_mirror<T>(mirror);
    dispatch_layout<T>(L, [&]<int VEC, int LPR>() {
        constexpr int U = VEC > 1 ? CLANE_LONG_U : 4;
        spmm_split_segment_kernel<T, PT, VEC, LPR, U, kLongWaves>
            <<<unsigned(n_segments), kLongWaves * kWave, 0, (hipStream_t)stream>>>(
                rowptr, colidx, P, split_rows, seg_ptr, seg_row, edges_per_segment, Z_old, ldz, d, slab, ld_slab);
        // a row's segments sit in the slab like a class row's slots: the same fixed-order combine + epilogue
        spmm_class_combine_kernel<T, VEC, LPR>
            <<<unsigned(ceil_div(n_split, kWave / LPR)), kCombineWaves * kWave, 0, (hipStream_t)stream>>>(
            split_rows, seg_ptr, n_split, row0, slab, ld_slab, Z_old, ldz, X, ldx, gamma, Z_new, ldo, d, mir, false, delta_partials);
    });
    return check_launch("spmm_update_split");
}

template <typename T, typename PT>
int spmm_update_class(const int32_t *colidx, const PT *P, const int64_t *item_e0, const int32_t *item_len,
                      const int32_t *item_slot, int64_t n_blocks, int32_t items_per_block, const int32_t *class_rows,
                      const int64_t *slot_ptr, int64_t n_rows, int64_t row0, const T *Z_old, int64_t ldz, const T *X,
                      int64_t ldx, typename Elem<T>::acc_t gamma, T *Z_new, int64_t ldo, int32_t d, int32_t flags,
                      typename Elem<T>::acc_t *slab, const clane_mirror_t *mirror, double *delta_partials,
                      void *stream) {
    REQUIRE(mirror_ok(mirror, d), "spmm_update_class: incomplete mirror descriptor");
    REQUIRE(n_blocks >= 0 && n_blocks <= INT32_MAX && n_rows >= 0 && n_rows <= INT32_MAX && row0 >= 0 && d > 0,
            "spmm_update_class: bad shape");
    REQUIRE(items_per_block >= kWavesPerBlock && items_per_block <= kMaxItemsPerBlock,
            "spmm_update_class: items_per_block must be in [%d, %d]", kWavesPerBlock, kMaxItemsPerBlock);
    REQUIRE(ldz >= d && ldx >= d && ldo >= d, "spmm_update_class: leading dimension < d");
    if (n_rows == 0) return CLANE_OK;
    REQUIRE(colidx && P && item_e0 && item_len && item_slot && class_rows && slot_ptr && Z_old && X && Z_new && slab &&
                delta_partials,
            "spmm_update_class: null pointer");
    REQUIRE((const void *)Z_new != (const void *)Z_old, "spmm_update_class: Z_new must not alias Z_old");
    REQUIRE(aligned16(slab), "spmm_update_class: slab must be 16-byte aligned");
    const Layout L = pick_layout<T>(d, {Z_old, X, Z_new, mirror_alignment_probe(mirror, Z_new)},
                                    {ldz, ldx, ldo, mirror && mirror->row_ptr ? mirror->ld : ldo});
    const int64_t ld_slab = ceil_div(d, 8) * 8;
    const Mirror<T> mir = make_mirror<T>(mirror);
    dispatch_layout<T>(L, [&]<int VEC, int LPR>() {
        constexpr int U = VEC > 1 ? CLANE_LONG_U : 4;
        // CLANE_SPMM_TABLE_BEYOND_CACHE: fewer row loads in flight, more waves -- where it was measured to pay (see
        // spmm_update): two fp32 rows per instruction (6: config 3 in tiles), four bf16 rows (4: config 4's class pass
        // 2.48 -> 2.36 ms, 98 -> 74 registers)
        constexpr int kDeepU = (LPR == 32 && VEC == 4 && sizeof(T) == 4) ? CLANE_CLASS_U32
                               : (LPR == 16 && VEC == 8 && sizeof(T) == 2) ? CLANE_CLASS_U16B : 0;
        if (n_blocks > 0 && kDeepU > 0 && (flags & CLANE_SPMM_TABLE_BEYOND_CACHE))
            spmm_class_chunk_kernel<T, PT, VEC, LPR, (kDeepU > 0 ? kDeepU : U)>
                <<<unsigned(n_blocks), kBlock, 0, (hipStream_t)stream>>>(
                    colidx, P, item_e0, item_len, item_slot, items_per_block, Z_old, ldz, d, slab, ld_slab);
        else if (n_blocks > 0)
            spmm_class_chunk_kernel<T, PT, VEC, LPR, U><<<unsigned(n_blocks), kBlock, 0, (hipStream_t)stream>>>(
                colidx, P, item_e0, item_len, item_slot, items_per_block, Z_old, ldz, d, slab, ld_slab);
        spmm_class_combine_kernel<T, VEC, LPR>
            <<<unsigned(ceil_div(n_rows, kWave / LPR)), kCombineWaves * kWave, 0, (hipStream_t)stream>>>(
            class_rows, slot_ptr, n_rows, row0, slab, ld_slab, Z_old, ldz, X, ldx, gamma, Z_new, ldo, d, mir,
            (flags & CLANE_SPMM_TABLE_BEYOND_CACHE) != 0, delta_partials);
    });
    return check_launch("spmm_update_class");
}

template <typename T>
int l1_distance(const T *A, int64_t lda, const T *B, int64_t ldb, int64_t nrows, int32_t d,
                typename Elem<T>::acc_t *sq_a, double *ws, double *out, void *stream) {
    REQUIRE(nrows >= 0 && d > 0 && lda >= d && ldb >= d, "l1_distance: bad shape");
    REQUIRE(ws && out, "l1_distance: null workspace/output");
    REQUIRE(nrows == 0 || (A && B), "l1_distance: null pointer");
    const Layout L = pick_layout<T>(d, {A, B}, {lda, ldb});
    if (sq_a != nullptr) {      // sq_a must be bit for bit row_sqnorm's, whose lane layout follows from A alone
        const Layout LA = pick_layout<T>(d, {A}, {lda});
        REQUIRE(LA.vec == L.vec && LA.lpr == L.lpr,
                "l1_distance: sq_a needs B as aligned as A (the lane layout of row_sqnorm on A)");
    }
    int grid = 1;
    dispatch_layout<T>(L, [&]<int VEC, int LPR>() {
        grid = grid_for_waves(ceil_div(nrows > 0 ? nrows : 1, (kWave / LPR) * stream_rows(LPR)));
        if (grid > kReduceGrid) grid = kReduceGrid;
        l1_distance_kernel<T, VEC, LPR><<<grid, kBlock, 0, (hipStream_t)stream>>>(A, lda, B, ldb, nrows, d, sq_a, ws);
    });
    reduce_fixed_kernel<<<1, 1024, 0, (hipStream_t)stream>>>(ws, grid, grid, 1, out);
    return check_launch("l1_distance");
}

template <typename T>
int gather_rows(const T *src, int64_t lds, const int32_t *idx, int64_t n, int32_t d, T *dst, int64_t ldd, void *stream) {
    REQUIRE(n >= 0 && d > 0 && lds >= d && ldd >= d, "gather_rows: bad shape");
    if (n == 0) return CLANE_OK;
    REQUIRE(src && idx && dst, "gather_rows: null pointer");
    const Layout L = pick_layout<T>(d, {src, dst}, {lds, ldd});
    dispatch_layout<T>(L, [&]<int VEC, int LPR>() {
        const int grid = grid_for_waves(ceil_div(n, kWave / LPR));
        gather_rows_kernel<T, VEC, LPR><<<grid, kBlock, 0, (hipStream_t)stream>>>(src, lds, idx, n, d, dst, ldd);
    });
    return check_launch("gather_rows");
}

template <typename T>
int pair_cosine(const T *A, int64_t lda, const T *B, int64_t ldb, int64_t nrows, int32_t d,
                typename Elem<T>::acc_t *out, double *ws, void *stream) {
    REQUIRE(nrows >= 0 && d > 0 && lda >= d && ldb >= d, "pair_cosine: bad shape");
    if (nrows == 0) return CLANE_OK;
    REQUIRE(A && B && out && ws, "pair_cosine: null pointer");
    const Layout L = pick_layout<T>(d, {A, B}, {lda, ldb});
    int grid = 1;
    dispatch_layout<T>(L, [&]<int VEC, int LPR>() {
        grid = grid_for_waves(ceil_div(nrows, kWave / LPR));
        if (grid > kReduceGrid) grid = kReduceGrid;
        pair_dot_kernel<T, VEC, LPR><<<grid, kBlock, 0, (hipStream_t)stream>>>(A, lda, B, ldb, nrows, d, out, ws);
    });
    double *sums2 = ws + 2 * kReduceGrid;
    reduce_fixed_kernel<<<1, 1024, 0, (hipStream_t)stream>>>(ws, grid, grid, 2, sums2);
    int sgrid = int(ceil_div(nrows, kBlock));
    if (sgrid > kMaxGrid) sgrid = kMaxGrid;
    pair_scale_kernel<typename Elem<T>::acc_t><<<sgrid, kBlock, 0, (hipStream_t)stream>>>(out, nrows, sums2);
    return check_launch("pair_cosine");
}

}  // namespace

// ---------------------------------------------------------------------------------------------
extern "C" {

int clane_abi_version(void) { return CLANE_ABI_VERSION; }
const char *clane_last_error(void) { return g_err; }
#define CLANE_STR2(x) #x
#define CLANE_STR(x) CLANE_STR2(x)
const char *clane_build_info(void) {
    return "arch=gfx950;SPMM_U=" CLANE_STR(CLANE_SPMM_U) ";LONG_U=" CLANE_STR(CLANE_LONG_U) ";LONG_WAVES=" CLANE_STR(
        CLANE_LONG_WAVES) ";ROWS_PER_BLOCK=" CLANE_STR(CLANE_ROWS_PER_BLOCK) ";NT_STREAM=" CLANE_STR(CLANE_NT_STREAM) ";NT_SLAB=" CLANE_STR(CLANE_NT_SLAB)
        ";TARGET_GRID=" CLANE_STR(CLANE_TARGET_GRID) ";SPMM_DYNAMIC=" CLANE_STR(CLANE_SPMM_DYNAMIC) ";SPMM_PREFETCH=" CLANE_STR(CLANE_SPMM_PREFETCH)
        ";COMBINE_WAVES=" CLANE_STR(CLANE_COMBINE_WAVES) ";XOR_DPP=" CLANE_STR(CLANE_XOR_DPP) ";COMBINE_LOADS=" CLANE_STR(CLANE_COMBINE_LOADS) ";SUBROW_U32=" CLANE_STR(CLANE_SUBROW_U32) ";CLASS_U32=" CLANE_STR(CLANE_CLASS_U32) ";CLASS_U16B=" CLANE_STR(CLANE_CLASS_U16B);
}

int clane_xcc_ids(int32_t *out, int64_t n_blocks, int32_t block_threads, void *stream) {
    if (!out || n_blocks <= 0 || n_blocks > INT32_MAX || block_threads < 64 || block_threads > 1024 || block_threads % 64)
        return fail(CLANE_ERR_INVALID_ARGUMENT, "xcc_ids: bad arguments");
    xcc_ids_kernel<<<unsigned(n_blocks), unsigned(block_threads), 0, (hipStream_t)stream>>>(out);
    return check_launch("xcc_ids");
}

int clane_check_csr(const int64_t *rowptr, const int32_t *colidx, int64_t nrows, int64_t n_edges, int64_t table_rows,
                    int32_t *status, void *stream) {
    if (!rowptr || !status || nrows < 0 || n_edges < 0 || table_rows < 0 || (n_edges > 0 && !colidx))
        return fail(CLANE_ERR_INVALID_ARGUMENT, "check_csr: bad arguments");
    const int64_t work = n_edges > nrows + 1 ? n_edges : nrows + 1;
    const int64_t blocks = ceil_div(work, int64_t(kBlock) * 8);
    check_csr_kernel<<<unsigned(blocks < 1 ? 1 : blocks > 65536 ? 65536 : blocks), kBlock, 0, (hipStream_t)stream>>>(
        rowptr, colidx, nrows, n_edges, table_rows, status);
    return check_launch("check_csr");
}

int64_t clane_spmm_partials_len(int64_t nrows, int64_t n_long) {
    return spmm_main_grid(nrows > 0 ? nrows : 1) + (n_long > 0 ? n_long : 0);
}
int64_t clane_reduce_ws_len(void) { return kReduceWs; }

// ---- device memory that other processes can map (peer-to-peer halo rows) -----------------------------------
#define HIP_REQUIRE(call, what)                                                                  \
    do {                                                                                         \
        const hipError_t e_ = (call);                                                            \
        if (e_ != hipSuccess) {                                                                  \
            return fail(CLANE_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e_));                \
        }                                                                                        \
    } while (0)

int clane_device_alloc(int64_t bytes, void **ptr) {
    REQUIRE(bytes > 0 && ptr, "device_alloc: bad arguments");
    HIP_REQUIRE(hipMalloc(ptr, size_t(bytes)), "hipMalloc");
    return CLANE_OK;
}
int clane_device_alloc_contiguous(int64_t bytes, void **ptr) {
    REQUIRE(bytes > 0 && ptr, "device_alloc_contiguous: bad arguments");
    HIP_REQUIRE(hipExtMallocWithFlags(ptr, size_t(bytes), hipDeviceMallocContiguous), "hipExtMallocWithFlags(contiguous)");
    return CLANE_OK;
}
int clane_device_free(void *ptr) {
    HIP_REQUIRE(hipFree(ptr), "hipFree");
    return CLANE_OK;
}
int clane_ipc_export(void *ptr, void *handle64) {
    static_assert(sizeof(hipIpcMemHandle_t) == CLANE_IPC_HANDLE_BYTES, "handle size");
    REQUIRE(ptr && handle64, "ipc_export: null pointer");
    hipIpcMemHandle_t h;
    HIP_REQUIRE(hipIpcGetMemHandle(&h, ptr), "hipIpcGetMemHandle");
    std::memcpy(handle64, &h, sizeof h);
    return CLANE_OK;
}
int clane_ipc_open(const void *handle64, void **ptr) {
    REQUIRE(ptr && handle64, "ipc_open: null pointer");
    hipIpcMemHandle_t h;
    std::memcpy(&h, handle64, sizeof h);
    HIP_REQUIRE(hipIpcOpenMemHandle(ptr, h, hipIpcMemLazyEnablePeerAccess), "hipIpcOpenMemHandle");
    return CLANE_OK;
}
int clane_ipc_close(void *ptr) {
    HIP_REQUIRE(hipIpcCloseMemHandle(ptr), "hipIpcCloseMemHandle");
    return CLANE_OK;
}

int clane_row_sqnorm_f32(const float *Z, int64_t nrows, int32_t d, int64_t ldz, float *sq, void *stream) {
    return row_sqnorm<float>(Z, nrows, d, ldz, sq, stream);
}
int clane_row_sqnorm_f64(const double *Z, int64_t nrows, int32_t d, int64_t ldz, double *sq, void *stream) {
    return row_sqnorm<double>(Z, nrows, d, ldz, sq, stream);
}
int clane_row_sqnorm_bf16(const uint16_t *Z, int64_t nrows, int32_t d, int64_t ldz, float *sq, void *stream) {
    return row_sqnorm<bf16_t>(reinterpret_cast<const bf16_t *>(Z), nrows, d, ldz, sq, stream);
}

int clane_degree_weighted_sums_f32(const float *sq, const int64_t *rowptr, const int32_t *indeg, int64_t nrows,
                                   double *ws, double *out2, void *stream) {
    return degree_weighted_sums<float>(sq, rowptr, indeg, nrows, ws, out2, stream);
}
int clane_degree_weighted_sums_f64(const double *sq, const int64_t *rowptr, const int32_t *indeg, int64_t nrows,
                                   double *ws, double *out2, void *stream) {
    return degree_weighted_sums<double>(sq, rowptr, indeg, nrows, ws, out2, stream);
}

#define CLANE_EDGE_SCORE_WRAPPER(SUF, CT, T, AT)                                                                      \
    int clane_edge_score_##SUF(const int64_t *rowptr, const int32_t *colidx, int64_t nrows, int64_t row0,             \
                               const CT *Z, int64_t ldz, int32_t d, int32_t mode, const double *sums2, const AT *sq,  \
                               AT *scores, int32_t flags, int64_t long_threshold, const int32_t *long_rows,           \
                               int64_t n_long, void *stream) {                                                        \
        return edge_score<T>(rowptr, colidx, nrows, row0, reinterpret_cast<const T *>(Z), ldz, d, mode, sums2, sq,    \
                             scores, flags, long_threshold, long_rows, n_long, stream);                               \
    }
CLANE_EDGE_SCORE_WRAPPER(f32, float, float, float)
CLANE_EDGE_SCORE_WRAPPER(f64, double, double, double)
CLANE_EDGE_SCORE_WRAPPER(bf16, uint16_t, bf16_t, float)
#undef CLANE_EDGE_SCORE_WRAPPER
#define CLANE_EDGE_SCORE_CLASS_WRAPPER(SUF, CT, T, AT)                                                                \
    int clane_edge_score_class_##SUF(const int64_t *rowptr, const int32_t *colidx, const int64_t *item_e0,            \
                                     const int32_t *item_len, const int32_t *item_slot, const int32_t *item_row,      \
                                     int64_t n_blocks, int32_t items_per_block, const int32_t *class_rows,            \
                                     const int64_t *slot_ptr, int64_t n_rows, int64_t row0, const CT *Z, int64_t ldz, \
                                     int32_t d, int32_t mode, const double *sums2, const AT *sq, AT *scores,          \
                                     int32_t flags, AT *stats, void *stream) {                                        \
        return edge_score_class<T>(rowptr, colidx, item_e0, item_len, item_slot, item_row, n_blocks, items_per_block, \
                                   class_rows, slot_ptr, n_rows, row0, reinterpret_cast<const T *>(Z), ldz, d, mode,  \
                                   sums2, sq, scores, flags, stats, stream);                                          \
    }
CLANE_EDGE_SCORE_CLASS_WRAPPER(f32, float, float, float)
CLANE_EDGE_SCORE_CLASS_WRAPPER(f64, double, double, double)
CLANE_EDGE_SCORE_CLASS_WRAPPER(bf16, uint16_t, bf16_t, float)
#undef CLANE_EDGE_SCORE_CLASS_WRAPPER

int clane_edge_score_finalize_f32(const int64_t *rowptr, const int32_t *colidx, int64_t nrows, int64_t row0,
                                  int32_t mode, const double *sums2, const float *sq, float *scores, void *stream) {
    return edge_score_finalize<float>(rowptr, colidx, nrows, row0, mode, sums2, sq, scores, stream);
}
int clane_edge_score_finalize_f64(const int64_t *rowptr, const int32_t *colidx, int64_t nrows, int64_t row0,
                                  int32_t mode, const double *sums2, const double *sq, double *scores, void *stream) {
    return edge_score_finalize<double>(rowptr, colidx, nrows, row0, mode, sums2, sq, scores, stream);
}
int clane_segment_softmax_f32(const int64_t *rowptr, int64_t nrows, float *vals, int64_t min_degree,
                              int64_t max_degree, const int32_t *long_rows, int64_t n_long, void *stream) {
    return segment_softmax<float>(rowptr, nrows, vals, min_degree, max_degree, long_rows, n_long, stream);
}
int clane_segment_softmax_f64(const int64_t *rowptr, int64_t nrows, double *vals, int64_t min_degree,
                              int64_t max_degree, const int32_t *long_rows, int64_t n_long, void *stream) {
    return segment_softmax<double>(rowptr, nrows, vals, min_degree, max_degree, long_rows, n_long, stream);
}

#define CLANE_SPMM_WRAPPERS(SUF, CT, T, PT, GT)                                                                        \
    int clane_spmm_update_##SUF(const int64_t *rowptr, const int32_t *colidx, const PT *P, int64_t nrows,             \
                                int64_t row0, const CT *Z_old, int64_t ldz, const CT *X, int64_t ldx, GT gamma,       \
                                CT *Z_new, int64_t ldo, int32_t d, int64_t long_threshold, int32_t flags,             \
                                const clane_mirror_t *mirror, double *delta_partials, void *stream) {                \
        return spmm_update<T, PT>(rowptr, colidx, P, nrows, row0, reinterpret_cast<const T *>(Z_old), ldz,            \
                                  reinterpret_cast<const T *>(X), ldx, gamma, reinterpret_cast<T *>(Z_new), ldo, d,   \
                                  long_threshold, flags, mirror, delta_partials, stream);                             \
    }                                                                                                                 \
    int clane_spmm_update_long_##SUF(const int64_t *rowptr, const int32_t *colidx, const PT *P,                       \
                                     const int32_t *long_rows, int64_t n_long, int32_t waves_per_row, int64_t row0, const CT *Z_old,         \
                                     int64_t ldz, const CT *X, int64_t ldx, GT gamma, CT *Z_new, int64_t ldo,         \
                                     int32_t d, const clane_mirror_t *mirror, double *delta_partials, void *stream) { \
        return spmm_update_long<T, PT>(rowptr, colidx, P, long_rows, n_long, waves_per_row, row0,                     \
                                       reinterpret_cast<const T *>(Z_old), ldz, reinterpret_cast<const T *>(X), ldx,  \
                                       gamma, reinterpret_cast<T *>(Z_new), ldo, d, mirror, delta_partials, stream);  \
    }
#define CLANE_SPLIT_WRAPPER(SUF, CT, T, PT, GT)                                                                         \
    int clane_spmm_update_split_##SUF(const int64_t *rowptr, const int32_t *colidx, const PT *P,                      \
                                      const int32_t *split_rows, const int64_t *seg_ptr, const int32_t *seg_row,      \
                                      int64_t n_split, int64_t n_segments, int64_t edges_per_segment, int64_t row0,   \
                                      const CT *Z_old, int64_t ldz, const CT *X, int64_t ldx, GT gamma, CT *Z_new,    \
                                      int64_t ldo, int32_t d, GT *slab, const clane_mirror_t *mirror,                 \
                                      double *delta_partials, void *stream) {                                         \
        return spmm_update_split<T, PT>(rowptr, colidx, P, split_rows, seg_ptr, seg_row, n_split, n_segments,         \
                                        edges_per_segment, row0, reinterpret_cast<const T *>(Z_old), ldz,             \
                                        reinterpret_cast<const T *>(X), ldx, gamma, reinterpret_cast<T *>(Z_new),     \
                                        ldo, d, slab, mirror, delta_partials, stream);                                \
    }
CLANE_SPLIT_WRAPPER(f32, float, float, float, float)
CLANE_SPLIT_WRAPPER(f64, double, double, double, double)
CLANE_SPLIT_WRAPPER(bf16, uint16_t, bf16_t, float, float)
#undef CLANE_SPLIT_WRAPPER
#define CLANE_CLASS_WRAPPER(SUF, CT, T, PT, GT)                                                                        \
    int clane_spmm_update_class_##SUF(const int32_t *colidx, const PT *P, const int64_t *item_e0,                     \
                                      const int32_t *item_len, const int32_t *item_slot, int64_t n_blocks,            \
                                      int32_t items_per_block, const int32_t *class_rows, const int64_t *slot_ptr,    \
                                      int64_t n_rows, int64_t row0, const CT *Z_old, int64_t ldz, const CT *X,        \
                                      int64_t ldx, GT gamma, CT *Z_new, int64_t ldo, int32_t d, int32_t flags,        \
                                      GT *slab, const clane_mirror_t *mirror, double *delta_partials, void *stream) { \
        return spmm_update_class<T, PT>(colidx, P, item_e0, item_len, item_slot, n_blocks, items_per_block,           \
                                        class_rows, slot_ptr, n_rows, row0, reinterpret_cast<const T *>(Z_old), ldz,  \
                                        reinterpret_cast<const T *>(X), ldx, gamma, reinterpret_cast<T *>(Z_new),     \
                                        ldo, d, flags, slab, mirror, delta_partials, stream);                         \
    }
CLANE_CLASS_WRAPPER(f32, float, float, float, float)
CLANE_CLASS_WRAPPER(f64, double, double, double, double)
CLANE_CLASS_WRAPPER(bf16, uint16_t, bf16_t, float, float)
#undef CLANE_CLASS_WRAPPER
int64_t clane_spmm_class_slab_len(int64_t n_slots, int32_t d) { return n_slots * (ceil_div(d, 8) * 8); }
int64_t clane_spmm_split_slab_len(int64_t n_segments, int32_t d) { return n_segments * (ceil_div(d, 8) * 8); }

CLANE_SPMM_WRAPPERS(f32, float, float, float, float)
CLANE_SPMM_WRAPPERS(f64, double, double, double, double)
CLANE_SPMM_WRAPPERS(bf16, uint16_t, bf16_t, float, float)
#undef CLANE_SPMM_WRAPPERS

int clane_reduce_partials(const double *partials, int64_t n, double *ws, double *out, void *stream) {
    if (n < 0 || !out || !ws || (n > 0 && !partials))
        return fail(CLANE_ERR_INVALID_ARGUMENT, "reduce_partials: bad arguments");
    if (n <= 8192) {
        reduce_fixed_kernel<<<1, 1024, 0, (hipStream_t)stream>>>(partials, n, n, 1, out);
    } else {  // two stages, both in a fixed order: kReduceGrid slice sums, then their sum
        const int64_t slice = ceil_div(n, kReduceGrid);
        const int grid = int(ceil_div(n, slice));
        reduce_slices_kernel<<<grid, kBlock, 0, (hipStream_t)stream>>>(partials, n, slice, ws);
        reduce_fixed_kernel<<<1, 1024, 0, (hipStream_t)stream>>>(ws, grid, grid, 1, out);
    }
    return check_launch("reduce_partials");
}

int clane_l1_distance_f32(const float *A, int64_t lda, const float *B, int64_t ldb, int64_t nrows, int32_t d,
                          float *sq_a, double *ws, double *out, void *stream) {
    return l1_distance<float>(A, lda, B, ldb, nrows, d, sq_a, ws, out, stream);
}
int clane_l1_distance_f64(const double *A, int64_t lda, const double *B, int64_t ldb, int64_t nrows, int32_t d,
                          double *sq_a, double *ws, double *out, void *stream) {
    return l1_distance<double>(A, lda, B, ldb, nrows, d, sq_a, ws, out, stream);
}
int clane_l1_distance_bf16(const uint16_t *A, int64_t lda, const uint16_t *B, int64_t ldb, int64_t nrows, int32_t d,
                           float *sq_a, double *ws, double *out, void *stream) {
    return l1_distance<bf16_t>(reinterpret_cast<const bf16_t *>(A), lda, reinterpret_cast<const bf16_t *>(B), ldb,
                               nrows, d, sq_a, ws, out, stream);
}

int clane_gather_rows_f32(const float *src, int64_t lds, const int32_t *idx, int64_t n, int32_t d, float *dst,
                          int64_t ldd, void *stream) {
    return gather_rows<float>(src, lds, idx, n, d, dst, ldd, stream);
}
int clane_gather_rows_f64(const double *src, int64_t lds, const int32_t *idx, int64_t n, int32_t d, double *dst,
                          int64_t ldd, void *stream) {
    return gather_rows<double>(src, lds, idx, n, d, dst, ldd, stream);
}
int clane_gather_rows_bf16(const uint16_t *src, int64_t lds, const int32_t *idx, int64_t n, int32_t d, uint16_t *dst,
                           int64_t ldd, void *stream) {
    return gather_rows<bf16_t>(reinterpret_cast<const bf16_t *>(src), lds, idx, n, d,
                               reinterpret_cast<bf16_t *>(dst), ldd, stream);
}

int clane_pair_cosine_f32(const float *A, int64_t lda, const float *B, int64_t ldb, int64_t nrows, int32_t d,
                          float *out, double *ws, void *stream) {
    return pair_cosine<float>(A, lda, B, ldb, nrows, d, out, ws, stream);
}
int clane_pair_cosine_f64(const double *A, int64_t lda, const double *B, int64_t ldb, int64_t nrows, int32_t d,
                          double *out, double *ws, void *stream) {
    return pair_cosine<double>(A, lda, B, ldb, nrows, d, out, ws, stream);
}

}  // extern "C"
