// K1 -- per-edge similarity scores in CSR order.
//
// Reference being replaced: the gather `self.Z[edges]` + one batched similarity call of
// clane/graph.py:119-121 with CosineSimilarity.__call__ (clane/similarity.py:26-37); the
// [2, E, d] gather (82 GB at |E|=40M, d=256) is never materialised here.
//
// Same lane layout and gather shape as K3 (spmm_update.h): one wave per source row, the source
// row stays in registers while the wave walks the neighbour list, U neighbour rows in flight.
// The per-edge dot is a butterfly over the LPR lanes of a row.  Rows longer than
// `long_threshold` go to edge_score_long_kernel: edges are independent, so a long row is
// simply cut into per-wave slices (no fold needed).
#pragma once

#include "device_utils.h"

namespace clane {

constexpr int kScoreReference = 0;
constexpr int kScorePerEdge = 1;
constexpr int kScoreRawDot = 2;

// Scores of edges [ea, eb) of the row whose source is global row `src_row` (a whole row, or one
// wave's slice of a long row).  Must be called by all 64 lanes.
template <typename T, int VEC, int LPR, int U>
__device__ __forceinline__ void score_edge_range(const int32_t *__restrict__ colidx, int64_t ea, int64_t eb,
                                                 int64_t src_row, const T *__restrict__ Z, int64_t ldz, int d,
                                                 int mode, typename Elem<T>::acc_t D,
                                                 const typename Elem<T>::acc_t *__restrict__ sq,
                                                 typename Elem<T>::acc_t *__restrict__ scores) {
    using A = typename Elem<T>::acc_t;
    constexpr int EPW = kWave / LPR;
    const int lane = lane_id();
    const int sub = lane / LPR, sl = lane % LPR;
    const bool single = d <= LPR * VEC;  // whole row in one pack per lane: keep the source row in registers
    const T *zsrc = Z + src_row * ldz;
    const A nsrc = mode == kScorePerEdge ? sqrt(sq[src_row]) : A(0);
    Pack<T, VEC> s0{};
    if (single && sl * VEC < d) s0 = load_pack<T, VEC>(zsrc + sl * VEC);

    for (int64_t e = ea; e < eb; e += kWave) {
        const int64_t left = eb - e;
        const int n = left < kWave ? int(left) : kWave;
        const int c = lane < n ? colidx[e + lane] : 0;
        A mine = A(0);  // LPR == 64: score of edge e + lane, stored coalesced once per 64 edges
        for (int j = 0; j < n; j += EPW * U) {
            A part[U];
            int cj[U];
            bool act[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int idx = j + u * EPW + sub;
                if constexpr (LPR == kWave)
                    cj[u] = lane_get_uniform(c, idx & (kWave - 1));
                else
                    cj[u] = lane_get(c, idx & (kWave - 1));
                act[u] = idx < n;
                part[u] = A(0);
            }
            for (int t0 = 0; t0 < d; t0 += LPR * VEC) {
                const int c0 = t0 + sl * VEC;
                const bool ok = c0 < d;
                Pack<T, VEC> s = s0;
                if (!single) {
                    s = Pack<T, VEC>{};
                    if (ok) s = load_pack<T, VEC>(zsrc + c0);
                }
                Pack<T, VEC> z[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    z[u] = Pack<T, VEC>{};
                    if (act[u] && ok) z[u] = load_pack<T, VEC>(Z + int64_t(cj[u]) * ldz + c0);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
#pragma unroll
                    for (int k = 0; k < VEC; ++k)
                        part[u] = fma(Elem<T>::to_acc(s.v[k]), Elem<T>::to_acc(z[u].v[k]), part[u]);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const A dot = group_sum<LPR>(part[u]);  // every lane of the sub-wave holds the total
                A score = dot;
                if (mode == kScoreReference)
                    score = dot / D;
                else if (mode == kScorePerEdge)
                    score = dot / (nsrc * sqrt(sq[act[u] ? cj[u] : 0]));
                if constexpr (LPR == kWave) {
                    if (lane == j + u) mine = score;
                } else {
                    if (act[u] && sl == 0) scores[e + j + u * EPW + sub] = score;
                }
            }
        }
        if constexpr (LPR == kWave) {
            if (lane < n) scores[e + lane] = mine;
        }
    }
}

template <typename A>
__device__ __forceinline__ A global_denominator(int mode, const double *__restrict__ sums2) {
    // sqrt(||Z[src_all]||_F^2) * sqrt(||Z[dst_all]||_F^2)  (similarity.py:37)
    return mode == kScoreReference ? sqrt(A(sums2[0])) * sqrt(A(sums2[1])) : A(1);
}

template <typename T, int VEC, int LPR, int U>
__global__ __launch_bounds__(kBlock) void edge_score_kernel(
    const int64_t *__restrict__ rowptr, const int32_t *__restrict__ colidx, int64_t nrows, int64_t row0,
    const T *__restrict__ Z, int64_t ldz, int d, int mode, const double *__restrict__ sums2,
    const typename Elem<T>::acc_t *__restrict__ sq, typename Elem<T>::acc_t *__restrict__ scores,
    int64_t long_threshold, int rows_per_block) {
    using A = typename Elem<T>::acc_t;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const int64_t row_begin = int64_t(blockIdx.x) * rows_per_block;
    const int64_t row_end = row_begin + rows_per_block < nrows ? row_begin + rows_per_block : nrows;
    const A D = global_denominator<A>(mode, sums2);
    for (int64_t r = row_begin + wave; r < row_end; r += kWavesPerBlock) {
        const int64_t e0 = rowptr[r];
        const int64_t e1 = rowptr[r + 1];
        if (e0 == e1 || (long_threshold > 0 && e1 - e0 > long_threshold)) continue;
        score_edge_range<T, VEC, LPR, U>(colidx, e0, e1, row0 + r, Z, ldz, d, mode, D, sq, scores);
    }
}

// grid = (n_long, slices): wave w of workgroup (i, y) scores edges
// [e0 + (y*WAVES + w)*edges_per_wave, +edges_per_wave) of long row i; edges_per_wave % 64 == 0.
template <typename T, int VEC, int LPR, int U, int WAVES>
__global__ __launch_bounds__(WAVES *kWave) void edge_score_long_kernel(
    const int64_t *__restrict__ rowptr, const int32_t *__restrict__ colidx, const int32_t *__restrict__ long_rows,
    int64_t row0, const T *__restrict__ Z, int64_t ldz, int d, int mode, const double *__restrict__ sums2,
    const typename Elem<T>::acc_t *__restrict__ sq, typename Elem<T>::acc_t *__restrict__ scores,
    int edges_per_wave) {
    using A = typename Elem<T>::acc_t;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const int64_t r = long_rows[blockIdx.x];
    const int64_t e0 = rowptr[r];
    const int64_t e1 = rowptr[r + 1];
    const int64_t a = e0 + (int64_t(blockIdx.y) * WAVES + wave) * edges_per_wave;
    if (a >= e1) return;
    const int64_t b = a + edges_per_wave < e1 ? a + edges_per_wave : e1;
    const A D = global_denominator<A>(mode, sums2);
    score_edge_range<T, VEC, LPR, U>(colidx, a, b, row0 + r, Z, ldz, d, mode, D, sq, scores);
}

}  // namespace clane
