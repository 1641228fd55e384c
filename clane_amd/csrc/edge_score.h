// K1 -- per-edge similarity scores in CSR order, with the row softmax fused for short rows.
//
// Reference being replaced: the gather `self.Z[edges]` + one batched similarity call of
// clane/graph.py:119-121 with CosineSimilarity.__call__ (clane/similarity.py:26-37), and for rows
// of <= 64 edges also the per-row softmax of graph.py:122-123.  The [2, E, d] gather (82 GB at
// |E|=40M, d=256) is never materialised.
//
// Same lane layout and gather shape as K3 (spmm_update.h): one wave per source row, the source
// row stays in registers while the wave walks the neighbour list, U neighbour rows in flight.
//  * The U = 8 per-lane partial dots of a group are reduced TOGETHER by a transposed butterfly:
//    the first three exchange steps halve the number of live values (8 -> 4 -> 2 -> 1) while
//    doubling the lanes each value has absorbed, the remaining log2(LPR)-3 steps finish the one
//    survivor: 10 cross-lane ops per 8 edges at LPR = 64 instead of 48.  Every edge goes through
//    the same exchange tree whatever slot it sits in, so its score does not depend on where a
//    row is cut (the long-row kernel reproduces the one-wave scores bit for bit).
//  * The finished scores are handed to lane = (edge index in the 64-edge chunk) with one more
//    permute, so a chunk is stored with one coalesced instruction, and a row that fits one
//    chunk is soft-maxed in registers (max / exp / sum butterflies) before it is stored.
//  * Rows longer than `long_threshold` go to edge_score_long_kernel: edges are independent, so
//    a long row is simply cut into per-wave slices; K2 normalises rows of > 64 edges afterwards.
#pragma once

#include "device_utils.h"

namespace clane {

// The source row of a CSR row is read once per build_P: non-temporal (build_P at config 3 3.84 -> 3.80 / 3.83 ms, config 4
// 7.02 / 6.97 -> 6.93 / 6.96: small, free).
#ifndef CLANE_K1_NT_SRC
#define CLANE_K1_NT_SRC 1
#endif
template <typename T, int VEC>
__device__ __forceinline__ Pack<T, VEC> k1_src_load(const T *p) {
    if constexpr (CLANE_K1_NT_SRC && sizeof(Pack<T, VEC>) == 16) {
        const clane_u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const clane_u32x4 *>(p));
        Pack<T, VEC> out;
        __builtin_memcpy(&out, &v, 16);
        return out;
    } else {
        return load_pack<T, VEC>(p);
    }
}

constexpr int kScoreReference = 0;
constexpr int kScorePerEdge = 1;
constexpr int kScoreRawDot = 2;

// v[0..7]: per-lane partials of 8 independent sums over the LPR lanes of a sub-wave (LPR >= 8).
// Returns, in every lane, the finished sum number  u = sl / (LPR/8)  (the top three bits of the
// lane's position in its sub-wave).
template <int LPR, typename A>
__device__ __forceinline__ A transpose_reduce8(const A (&v)[8], int sl) {
    static_assert(LPR >= 8, "needs at least 8 lanes per row");
    constexpr int m0 = LPR / 2, m1 = LPR / 4, m2 = LPR / 8;
    const bool h0 = (sl & m0) != 0, h1 = (sl & m1) != 0, h2 = (sl & m2) != 0;
    A r[4], q[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = (h0 ? v[i + 4] : v[i]) + lane_xor<m0>(h0 ? v[i] : v[i + 4]);
#pragma unroll
    for (int i = 0; i < 2; ++i) q[i] = (h1 ? r[i + 2] : r[i]) + lane_xor<m1>(h1 ? r[i] : r[i + 2]);
    const A t = (h2 ? q[1] : q[0]) + lane_xor<m2>(h2 ? q[0] : q[1]);
    return group_sum<m2>(t);             // the remaining partners lane ^ m2/2 ... lane ^ 1
}

template <typename A>
__device__ __forceinline__ A finalize_score(A dot, int mode, A D, A nsrc, const A *__restrict__ sq, int col) {
    if (mode == kScoreReference) return dot / D;
    if (mode == kScorePerEdge) return dot / (nsrc * sqrt(sq[col]));
    return dot;
}

// Scores of edges [ea, eb) of the row whose source is global row `src_row` (a whole row, or one
// wave's slice of a long row).  `softmax`: [ea, eb) is a WHOLE row -- normalise it (in registers when it
// fits one 64-edge chunk, else with a running max / sum and a second pass over the stored scores).
// `want_stats` (a slice of a long row): leave the raw scores and hand back {max, sum of exp(score - max)} of the
// slice, for the workgroup to combine (returned by value: a pointer that may be null kept the pair in scratch).
// Must be called by all 64 lanes.
template <typename A>
struct RangeStats {
    A max, sum;
};

template <typename T, int VEC, int LPR, int U>
__device__ __forceinline__ RangeStats<typename Elem<T>::acc_t> score_edge_range(const int32_t *__restrict__ colidx, int64_t ea, int64_t eb,
                                                 int64_t src_row, const T *__restrict__ Z, int64_t ldz, int d,
                                                 int mode, typename Elem<T>::acc_t D,
                                                 const typename Elem<T>::acc_t *__restrict__ sq,
                                                 typename Elem<T>::acc_t *__restrict__ scores, bool softmax,
                                                 bool want_stats = false) {
    using A = typename Elem<T>::acc_t;
    constexpr int EPW = kWave / LPR;
    constexpr bool kTransposed = (U == 8 && LPR >= 8);
    static_assert(kTransposed || LPR >= U, "fallback delivery needs one lane per value in a sub-wave");
    const int lane = lane_id();
    const int sub = lane / LPR, sl = lane % LPR;
    const bool single = d <= LPR * VEC;  // whole row in one pack per lane: keep the source row in registers
    const T *zsrc = Z + src_row * ldz;
    const A nsrc = mode == kScorePerEdge ? sqrt(sq[src_row]) : A(0);
    Pack<T, VEC> s0{};
    if (single && sl * VEC < d) s0 = k1_src_load<T, VEC>(zsrc + sl * VEC);
    A run_m = -A(INFINITY), run_s = A(0);

    for (int64_t e = ea; e < eb; e += kWave) {
        const int64_t left = eb - e;
        const int n = left < kWave ? int(left) : kWave;
        // Lanes past the end of the row hold the chunk's FIRST column: every address formed below is
        // valid, so the gathers need no branch (see pin_loaded in device_utils.h for why that matters).
        int c = lane < n ? colidx[e + lane] : 0;
        c = lane < n ? c : lane_get_uniform(c, 0);
        A mine = A(0);  // score of edge e + lane
        for (int j = 0; j < n; j += EPW * U) {
            A part[U];
            int cj[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int idx = j + u * EPW + sub;
                if constexpr (LPR == kWave)
                    cj[u] = lane_get_uniform(c, idx & (kWave - 1));
                else
                    cj[u] = lane_get(c, idx & (kWave - 1));
                part[u] = A(0);
            }
            for (int t0 = 0; t0 < d; t0 += LPR * VEC) {
                const int c0 = t0 + sl * VEC;
                const bool ok = c0 < d;
                const int c0s = ok ? c0 : 0;  // out-of-range lanes read column 0 and multiply it by a zero source pack
                Pack<T, VEC> s = s0;
                if (!single) {
                    s = load_pack<T, VEC>(zsrc + c0s);
                    if (!ok) s = Pack<T, VEC>{};
                }
                Pack<T, VEC> z[U];
#pragma unroll
                for (int u = 0; u < U; ++u) z[u] = load_pack<T, VEC>(Z + int64_t(cj[u]) * ldz + c0s);
#pragma unroll
                for (int u = 0; u < U; ++u) {
#pragma unroll
                    for (int k = 0; k < VEC; ++k)
                        part[u] = fma(Elem<T>::to_acc(s.v[k]), Elem<T>::to_acc(z[u].v[k]), part[u]);
                }
            }
            // reduce over the row's lanes; `serve` = the finished dot this lane can hand out, u_serve = which
            A serve;
            int u_serve;
            if constexpr (kTransposed) {
                serve = transpose_reduce8<LPR>(part, sl);
                u_serve = sl / (LPR / 8);
            } else {
                u_serve = sl % U;
                serve = A(0);
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const A dot = group_sum<LPR>(part[u]);
                    if (u == u_serve) serve = dot;
                }
            }
            const int idx_serve = j + u_serve * EPW + sub;      // chunk-relative edge of `serve`
            int col_serve = 0;
            if (mode == kScorePerEdge) col_serve = lane_get(c, idx_serve & (kWave - 1));
            serve = finalize_score<A>(serve, mode, D, nsrc, sq, col_serve);
            // hand edge (j + rel) to lane (j + rel):  it sits in sub-wave rel % EPW, value slot rel / EPW
            const int rel = lane - j;
            const bool take = rel >= 0 && rel < EPW * U;
            int src = 0;
            if (take) src = (rel % EPW) * LPR + (kTransposed ? (rel / EPW) * (LPR / 8) : rel / EPW);
            const A got = lane_get(serve, src);
            if (take) mine = got;
        }
        if (softmax || want_stats) {  // graph.py:122-123; running max / sum over the row's chunks (online softmax)
            const bool in = lane < n;
            const A v = in ? mine : -A(INFINITY);
            const A new_m = fmax(run_m, group_max<kWave>(v));
            const A ex = in ? exp_acc<A>(v - new_m) : A(0);
            const A cs = group_sum<kWave>(ex);
            run_s = run_s * exp_acc<A>(run_m - new_m) + cs;
            run_m = new_m;
            if (softmax && eb - ea <= kWave) mine = ex / cs;  // the whole row is in this chunk: finished in registers
        }
        if (lane < n) scores[e + lane] = mine;
    }
    if (softmax && eb - ea > kWave) {  // second pass over this wave's own stores (each lane re-reads what it wrote)
        for (int64_t e = ea + lane; e < eb; e += kWave) scores[e] = exp_acc<A>(scores[e] - run_m) / run_s;
    }
    return RangeStats<A>{run_m, run_s};
}

template <typename A>
__device__ __forceinline__ A global_denominator(int mode, const double *__restrict__ sums2) {
    // sqrt(||Z[src_all]||_F^2) * sqrt(||Z[dst_all]||_F^2)  (similarity.py:37)
    return mode == kScoreReference ? sqrt(A(sums2[0])) * sqrt(A(sums2[1])) : A(1);
}

// (Round 3 tried K3's row-kernel structure here as well -- rowptr slice in LDS, rows claimed from an LDS counter, the next
// row's first columns and source pack requested before the current row's gathers: 1.378 -> 1.401 ms at config 3, the 14
// extra registers cost what the prefetch saved.  The static interleave stays.)
template <typename T, int VEC, int LPR, int U>
__global__ __launch_bounds__(kBlock) void edge_score_kernel(
    const int64_t *__restrict__ rowptr, const int32_t *__restrict__ colidx, int64_t nrows, int64_t row0,
    const T *__restrict__ Z, int64_t ldz, int d, int mode, const double *__restrict__ sums2,
    const typename Elem<T>::acc_t *__restrict__ sq, typename Elem<T>::acc_t *__restrict__ scores,
    int64_t long_threshold, bool fuse_softmax, int rows_per_block) {
    using A = typename Elem<T>::acc_t;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const int64_t row_begin = int64_t(blockIdx.x) * rows_per_block;
    const int64_t row_end = row_begin + rows_per_block < nrows ? row_begin + rows_per_block : nrows;
    const A D = global_denominator<A>(mode, sums2);
    for (int64_t r = row_begin + wave; r < row_end; r += kWavesPerBlock) {
        const int64_t e0 = rowptr[r];
        const int64_t e1 = rowptr[r + 1];
        if (e0 == e1 || (long_threshold > 0 && e1 - e0 > long_threshold)) continue;
        score_edge_range<T, VEC, LPR, U>(colidx, e0, e1, row0 + r, Z, ldz, d, mode, D, sq, scores, fuse_softmax);
    }
}

// Narrow rows (LPR < 64): one SUB-WAVE per source row, 64/LPR rows per wave in flight -- the K1 counterpart
// of spmm_update_subrow_kernel, with the same row claiming: every sub-wave takes its rows from the workgroup's
// LDS counter on its own and the wave advances all of them by one group of 8 edges per iteration, so a wave's
// time is the SUM of its rows' groups / (64/LPR) instead of the maximum over a fixed group of rows (r03: the
// statically grouped version ran at 4.0 TB/s on the 10M-vertex power-law graph where K3 reaches 6.1 over the
// same rows).  A sub-wave is in one of four states, uniform over its LPR lanes:
//   claim   -- take the next row of <= long_threshold edges, load its source pack;
//   score   -- LPR buffered edges (one column per lane), consumed 8 at a time: 8 branch-free neighbour-row
//              loads, 8 partial dots, the transposed butterfly over the LPR lanes, the finished scores handed
//              to lane = edge; after a chunk the running max / sum of the row and one coalesced store;
//   rescale -- a row of more than LPR edges is soft-maxed by a second pass over the sub-wave's own stores (each
//              lane re-reads what it wrote), except for its last chunk, which leaves normalised from the registers;
//              four chunks per visit (independent loads: one round trip), so a row of up to 5 chunks is done in
//              the visit that finished its last chunk, a longer one while the other sub-waves keep gathering;
//   done    -- no rows left in the block: the lanes carry on with zero-weight reads of table row 0 (an L1 hit).
// Every edge goes through the same fma chain and the same exchange tree as in score_edge_range: the scores are
// bit-identical to the one-wave and the long-row kernels'.
#ifndef CLANE_K1_MIN_WAVES
#define CLANE_K1_MIN_WAVES 7      // edge_score_subrow_kernel: __launch_bounds__ 2nd argument (waves per SIMD), 0 = unconstrained;
                                  // 7 = 72 VGPRs, no spills (config 4's shape: 4.61 -> 4.41 ms; 8 spills: 6.41, profiles/r03_k1_subrow.md)
#endif
#if CLANE_K1_MIN_WAVES > 0      // fp64 accumulators need the registers: unconstrained there (the bound made them spill 64 bytes a lane)
#define CLANE_K1_BOUNDS __launch_bounds__(kBlock, (sizeof(typename Elem<T>::acc_t) == 8 ? 1 : CLANE_K1_MIN_WAVES))
#else
#define CLANE_K1_BOUNDS __launch_bounds__(kBlock)
#endif
template <typename T, int VEC, int LPR, int U>
__global__ CLANE_K1_BOUNDS void edge_score_subrow_kernel(
    const int64_t *__restrict__ rowptr, const int32_t *__restrict__ colidx, int64_t nrows, int64_t row0,
    const T *__restrict__ Z, int64_t ldz, int d, int mode, const double *__restrict__ sums2,
    const typename Elem<T>::acc_t *__restrict__ sq, typename Elem<T>::acc_t *__restrict__ scores,
    int64_t long_threshold, bool fuse_softmax, int rows_per_block) {
    using A = typename Elem<T>::acc_t;
    static_assert(LPR < kWave && LPR >= 8 && U == 8 && LPR % U == 0, "sub-wave layout");
    constexpr int kClaim = 0, kScore = 1, kRescale = 2, kDone = 3;
#ifndef CLANE_K1_RESCALE_CHUNKS
#define CLANE_K1_RESCALE_CHUNKS 4
#endif
    constexpr int kRescaleChunks = CLANE_K1_RESCALE_CHUNKS;
    __shared__ int64_t s_rowptr[kMaxRowsPerBlock + 1];
    __shared__ int s_next;
    const int lane = lane_id();
    const int sub = lane / LPR, sl = lane % LPR, sub_base = sub * LPR;
    const int64_t row_begin = int64_t(blockIdx.x) * rows_per_block;
    const int nb = int((row_begin + rows_per_block < nrows ? row_begin + rows_per_block : nrows) - row_begin);
    const A D = global_denominator<A>(mode, sums2);
    const int c0 = sl * VEC;
    const bool col_ok = c0 < d;
    const int c0s = col_ok ? c0 : 0;

    for (int i = threadIdx.x; i <= nb; i += kBlock) s_rowptr[i] = rowptr[row_begin + i];
    if (threadIdx.x == 0) s_next = 0;
    __syncthreads();

    int phase = kClaim;
    int64_t e0 = 0;            // first edge of the open row
    int deg = 0, eb = 0;       // its edge count; edges before the buffered chunk
    int n = 0, j = 0;          // buffered edges (c: one column per lane) and how many of them are scored
    int rpos = 0, rend = 0;    // rescale: edges of the row already normalised, of those stored raw
    int c = 0;
    A mine = A(0);             // score of edge eb + sl
    A run_m = -A(INFINITY), run_s = A(0), nsrc = A(0);
    Pack<T, VEC> s0{};

    for (;;) {
        if (phase != kDone && (phase != kScore || j >= n)) {      // divergent between sub-waves, uniform inside one
            if (phase == kScore) {               // the buffered chunk is scored: running max / sum, store
                const bool in = sl < n;
                if (fuse_softmax) {              // graph.py:122-123, online over the row's chunks
                    const A v = in ? mine : -A(INFINITY);
                    const A new_m = fmax(run_m, group_max<LPR>(v));
                    const A ex = in ? exp_acc<A>(v - new_m) : A(0);
                    const A cs = group_sum<LPR>(ex);
                    if (eb == 0) run_s = cs;                     // first chunk: 0 * exp(-inf) + cs without the exp
                    else run_s = run_s * exp_acc<A>(run_m - new_m) + cs;
                    run_m = new_m;
                    // the row's LAST chunk leaves normalised straight from the registers (run_m, run_s are final:
                    // exp(v - run_m) / run_s); only the chunks before it take the second pass
                    if (eb + n >= deg) mine = ex / run_s;
                }
                if (in) scores[e0 + eb + sl] = mine;
                eb += n;
                if (eb >= deg) {
                    rend = deg - n;                              // edges stored raw, ahead of the last chunk
                    phase = (fuse_softmax && rend > 0) ? kRescale : kClaim;
                    rpos = 0;
                }
            }
            if (phase == kRescale) {             // up to kRescaleChunks chunks per visit: independent loads, one round trip
                A raw[kRescaleChunks];
#pragma unroll
                for (int r = 0; r < kRescaleChunks; ++r) {
                    const int e = rpos + r * LPR + sl;
                    raw[r] = e < rend ? scores[e0 + e] : A(0);
                }
#pragma unroll
                for (int r = 0; r < kRescaleChunks; ++r) {
                    const int e = rpos + r * LPR + sl;
                    if (e < rend) scores[e0 + e] = exp_acc<A>(raw[r] - run_m) / run_s;
                }
                rpos += kRescaleChunks * LPR;
                if (rpos >= rend) phase = kClaim;
            }
            if (phase == kClaim) {
                for (;;) {
                    int v = 0;
                    if (sl == 0) v = atomicAdd(&s_next, 1);
                    const int row = lane_get(v, sub_base);
                    if (row >= nb) {
                        phase = kDone;
                        break;
                    }
                    const int64_t dg = s_rowptr[row + 1] - s_rowptr[row];
                    if (dg == 0 || (long_threshold > 0 && dg > long_threshold)) continue;
                    e0 = s_rowptr[row];
                    deg = int(dg);
                    eb = 0;
                    const int64_t gsrc = row0 + row_begin + row;
                    s0 = k1_src_load<T, VEC>(Z + gsrc * ldz + c0s);
                    if (!col_ok) s0 = Pack<T, VEC>{};
                    nsrc = mode == kScorePerEdge ? sqrt(sq[gsrc]) : A(0);
                    run_m = -A(INFINITY);
                    run_s = A(0);
                    phase = kScore;
                    break;
                }
            }
            n = 0;
            j = 0;
            c = 0;
            mine = A(0);
            if (phase == kScore) {               // refill: the next LPR edges of the row, one per lane
                n = deg - eb < LPR ? deg - eb : LPR;
                if (sl < n) c = colidx[e0 + eb + sl];
                c = sl < n ? c : lane_get(c, sub_base);           // lanes past the row: the chunk's first column
            }
        }
        if (__all(phase == kDone)) break;
        // one group of U edges per sub-wave
        A part[U];
        Pack<T, VEC> z[U];
        // (Skipping the second half of a group when no sub-wave has more than 4 edges left -- most rows of a power-law
        // graph have 2..4 -- was tried and lost: the wave-uniform branch breaks the run of 8 loads, 4.67 -> 7.99 ms at
        // config 4's shape, profiles/r03_k1_subrow.md.)
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int cj = lane_get(c, sub_base + ((j + u) & (LPR - 1)));
            z[u] = load_pack<T, VEC>(Z + int64_t(cj) * ldz + c0s);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            part[u] = A(0);
#pragma unroll
            for (int k = 0; k < VEC; ++k) part[u] = fma(Elem<T>::to_acc(s0.v[k]), Elem<T>::to_acc(z[u].v[k]), part[u]);
        }
        A serve = transpose_reduce8<LPR>(part, sl);
        const int idx_serve = (j + sl / (LPR / 8)) & (LPR - 1);
        int col_serve = 0;
        if (mode == kScorePerEdge) col_serve = lane_get(c, sub_base + idx_serve);
        serve = finalize_score<A>(serve, mode, D, nsrc, sq, col_serve);
        const int rel = sl - j;                      // lane sl of the sub-wave takes edge eb + sl
        const bool take = rel >= 0 && rel < U;
        const A got = lane_get(serve, take ? sub_base + rel * (LPR / 8) : lane);
        if (take) mine = got;
        j += U;
    }
}

// One workgroup of WAVES waves per long row: wave w scores the 64-aligned slice w (same slicing as
// spmm_long_kernel); a wave with an empty slice leaves at once.  Edges are independent: no fold.  With
// `fuse_softmax` the row is soft-maxed here too: every wave keeps {max, sum exp} of its slice, the workgroup
// combines them through LDS in wave order, and each wave rescales the scores it stored itself.
template <typename T, int VEC, int LPR, int U, int WAVES>
__global__ __launch_bounds__(WAVES *kWave) void edge_score_long_kernel(
    const int64_t *__restrict__ rowptr, const int32_t *__restrict__ colidx, const int32_t *__restrict__ long_rows,
    int64_t row0, const T *__restrict__ Z, int64_t ldz, int d, int mode, const double *__restrict__ sums2,
    const typename Elem<T>::acc_t *__restrict__ sq, typename Elem<T>::acc_t *__restrict__ scores,
    bool fuse_softmax) {
    using A = typename Elem<T>::acc_t;
    __shared__ A s_max[WAVES], s_sum[WAVES];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const int64_t r = long_rows[blockIdx.x];
    const int64_t e0 = rowptr[r];
    const int64_t e1 = rowptr[r + 1];
    const int64_t seg = ceil_div(ceil_div(e1 - e0, WAVES), kWave) * kWave;
    const int64_t a = e0 + wave * seg;
    if (idle_wave_may_exit(a < e1)) return;  // before the barrier below: device_utils.h
    const int64_t b = a + seg < e1 ? a + seg : e1;
    const A D = global_denominator<A>(mode, sums2);
    const bool whole_row_here = e1 - e0 <= kWave;
    const bool combine = fuse_softmax && !whole_row_here;
    const RangeStats<A> stats = score_edge_range<T, VEC, LPR, U>(colidx, a, b, row0 + r, Z, ldz, d, mode, D, sq, scores,
                                                                 fuse_softmax && whole_row_here, combine);
    if (!combine) return;
    if (lane_id() == 0) {
        s_max[wave] = stats.max;
        s_sum[wave] = stats.sum;
    }
    __syncthreads();
    const int active = int(ceil_div(e1 - e0, seg));
    A m = s_max[0];
    for (int w = 1; w < active; ++w) m = fmax(m, s_max[w]);
    A total = A(0);
    for (int w = 0; w < active; ++w) total += s_sum[w] * exp_acc<A>(s_max[w] - m);
    for (int64_t e = a + lane_id(); e < b; e += kWave) scores[e] = exp_acc<A>(scores[e] - m) / total;
}

// ---- class-affine rows (see spmm_update.h: every gathered row is read through ONE XCD's L2) ------------------
// K1 over the same work items as K3's class pass: a wave scores the edges of one item -- a chunk of one row's edges
// whose columns all belong to one XCD class -- with the source row in registers, exactly like a slice of
// edge_score_long_kernel, and leaves {max, sum of exp(score - max)} of the chunk in stats[2 * slot].  Item blocks of
// class b sit at block index 8 j + b, so XCD b only gathers rows of class b.
template <typename T, int VEC, int LPR, int U>
__global__ __launch_bounds__(kBlock) void edge_score_class_kernel(
    const int32_t *__restrict__ colidx, const int64_t *__restrict__ item_e0, const int32_t *__restrict__ item_len,
    const int32_t *__restrict__ item_slot, const int32_t *__restrict__ item_row, int items_per_block, int64_t row0,
    const T *__restrict__ Z, int64_t ldz, int d, int mode, const double *__restrict__ sums2,
    const typename Elem<T>::acc_t *__restrict__ sq, typename Elem<T>::acc_t *__restrict__ scores,
    typename Elem<T>::acc_t *__restrict__ stats) {
    using A = typename Elem<T>::acc_t;
    __shared__ int s_next;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const int64_t base = int64_t(blockIdx.x) * items_per_block;
    if (threadIdx.x == 0) s_next = kWavesPerBlock;
    __syncthreads();
    const A D = global_denominator<A>(mode, sums2);
    int cur = wave;
    while (cur < items_per_block) {
        const int64_t e0 = item_e0[base + cur];
        const int len = item_len[base + cur];
        if (len > 0) {
            const RangeStats<A> st = score_edge_range<T, VEC, LPR, U>(
                colidx, e0, e0 + len, row0 + item_row[base + cur], Z, ldz, d, mode, D, sq, scores, false, stats != nullptr);
            if (stats && lane_id() == 0) {
                const int64_t slot = item_slot[base + cur];
                stats[2 * slot] = st.max;
                stats[2 * slot + 1] = st.sum;
            }
        }
        int v = 0;
        if (lane_id() == 0) v = atomicAdd(&s_next, 1);
        cur = __builtin_amdgcn_readfirstlane(v);
    }
}

// One workgroup per class row: the chunks' {max, sum} combined, then the row's scores become  exp(score - max) / total
// (graph.py:122-123).  Every wave forms the row's {max, total} on its own and all of them alike: lane l folds the slots
// l, l + 64, ... in slot order, then a fixed butterfly -- the association depends on the slot count alone (not on the
// thread count, not on which wave), and a 4 000-slot hub is a chain of 64 dependent steps instead of 4 000 (r03; every
// thread used to walk the whole list: 0.38 ms of config 4's build_P).
// gridDim.y workgroups share one row's rescale pass (CLANE_SCORE_ROW_PARTS: a 2M-edge row rescaled by ONE workgroup was
// 2.3 ms of a 6.6 ms build_P); each forms the row's {max, total} itself -- the same way, so all parts agree -- and a
// part whose share of the row is empty leaves before doing so.
template <typename A>
__global__ __launch_bounds__(kBlock) void edge_softmax_class_kernel(const int64_t *__restrict__ rowptr,
                                                                    const int32_t *__restrict__ class_rows,
                                                                    const int64_t *__restrict__ slot_ptr,
                                                                    const A *__restrict__ stats,
                                                                    A *__restrict__ scores) {
    const int i = blockIdx.x;
    const int lane = lane_id();
    const int64_t r = class_rows[i];
    const int64_t row_e0 = rowptr[r], row_e1 = rowptr[r + 1];
    const int64_t share = ceil_div(ceil_div(row_e1 - row_e0, int64_t(gridDim.y)), kBlock) * kBlock;
    const int64_t my_e0 = row_e0 + int64_t(blockIdx.y) * share;
    if (my_e0 >= row_e1) return;
    const int64_t my_e1 = my_e0 + share < row_e1 ? my_e0 + share : row_e1;
    const int64_t s0 = slot_ptr[i], s1 = slot_ptr[i + 1];
    // the first score of every thread is requested BEFORE the chunks' stats: it does not depend on them, and most class
    // rows are a few hundred edges -- one score per thread -- so the launch is a chain of dependent round trips, not a
    // stream (r04: 1.84 TB/s); this takes one of them out of the chain (config 3: 162 -> 155 us per build_P; config 4,
    // whose time is in the multi-million-edge hubs: unchanged)
    const int64_t e_first = my_e0 + threadIdx.x;
    const A first = e_first < my_e1 ? scores[e_first] : A(0);
    A m = -A(INFINITY);
    for (int64_t s = s0 + lane; s < s1; s += kWave) m = fmax(m, stats[2 * s]);
    m = group_max<kWave>(m);
    A total = A(0);
    for (int64_t s = s0 + lane; s < s1; s += kWave) total += stats[2 * s + 1] * exp_acc<A>(stats[2 * s] - m);
    total = group_sum<kWave>(total);
    if (e_first < my_e1) scores[e_first] = exp_acc<A>(first - m) / total;
    for (int64_t e = e_first + kBlock; e < my_e1; e += kBlock) scores[e] = exp_acc<A>(scores[e] - m) / total;
}

}  // namespace clane
