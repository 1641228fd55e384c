// K0 / K1 / K2 -- the pieces of Graph.build_P (clane/graph.py:118-128) and
// CosineSimilarity.__call__ (clane/similarity.py:26-37):
//   K0  row_sqnorm_kernel          sq[v] = |z_v|^2
//       degree_weighted_kernel     sum_v outdeg_v sq_v , sum_v indeg_v sq_v   (the two GLOBAL
//                                  Frobenius norms of similarity.py:37, without gathering Z[edges])
//   K1  (edge_score.h)
//   K2  segment_softmax_kernel     per-row softmax in place (graph.py:122-123)
// All HBM-bound; K1 has the same gather shape as K3 (spmm_update.h) and the same lane layout.
#pragma once

#include "device_utils.h"

namespace clane {


// ---- K0 ------------------------------------------------------------------------------------
// LPR lanes per row; a wave covers 64/LPR rows with one load instruction and stream_rows(LPR) such groups per turn, all
// their loads issued before the first is used: a streaming pass needs ~8 MB in flight on this chip (2 us x 4+ TB/s),
// and one 16-byte load per lane per turn left the 1-KiB-row instance at 5.3 TB/s.  Measured per shape
// (profiles/r05_stream_kernels.md): a row per instruction wants 4 groups in flight (row_sqnorm 5.33 -> 5.73 TB/s,
// l1_distance + norms 5.44 -> 6.14), 2 or 4 rows per instruction want 2 (bf16 d=128: 5.30 -> 5.50; 4 groups there cost
// registers and 10-20 %), 8 per turn lose everywhere.
constexpr int stream_rows(int lpr) { return lpr >= kWave ? 4 : lpr >= 16 ? 2 : 1; }
template <typename T, int VEC, int LPR>
__global__ __launch_bounds__(kBlock) void row_sqnorm_kernel(const T *__restrict__ Z, int64_t nrows, int d, int64_t ldz,
                                                            typename Elem<T>::acc_t *__restrict__ sq) {
    using A = typename Elem<T>::acc_t;
    constexpr int RPW = kWave / LPR;
    constexpr int kStreamRows = stream_rows(LPR);
    const int lane = lane_id();
    const int sub = lane / LPR, sl = lane % LPR;
    const int64_t wave = int64_t(blockIdx.x) * kWavesPerBlock + threadIdx.x / kWave;
    const int64_t nwaves = int64_t(gridDim.x) * kWavesPerBlock;
    for (int64_t base = wave * (RPW * kStreamRows); base < nrows; base += nwaves * (RPW * kStreamRows)) {
        A s[kStreamRows];
#pragma unroll
        for (int j = 0; j < kStreamRows; ++j) s[j] = A(0);
        for (int c0 = sl * VEC; c0 < d; c0 += LPR * VEC) {      // a row's packs in the same order, whatever kStreamRows
            Pack<T, VEC> z[kStreamRows];
#pragma unroll
            for (int j = 0; j < kStreamRows; ++j) {
                const int64_t r = base + j * RPW + sub;
                z[j] = Pack<T, VEC>{};
                if (r < nrows) z[j] = load_pack<T, VEC>(Z + r * ldz + c0);
            }
#pragma unroll
            for (int j = 0; j < kStreamRows; ++j) {
#pragma unroll
                for (int k = 0; k < VEC; ++k) {
                    const A v = Elem<T>::to_acc(z[j].v[k]);
                    s[j] = fma(v, v, s[j]);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < kStreamRows; ++j) {
            const int64_t r = base + j * RPW + sub;
            const A t = group_sum<LPR>(s[j]);
            if (r < nrows && sl == 0) sq[r] = t;
        }
    }
}

// ws[b] / ws[G + b] <- block partials of outdeg*sq and indeg*sq  (G = gridDim.x)
template <typename A>
__global__ __launch_bounds__(kBlock) void degree_weighted_kernel(const A *__restrict__ sq,
                                                                 const int64_t *__restrict__ rowptr,
                                                                 const int32_t *__restrict__ indeg, int64_t nrows,
                                                                 double *__restrict__ ws) {
    __shared__ double smem[kWavesPerBlock];
    double a = 0.0, b = 0.0;
    for (int64_t v = int64_t(blockIdx.x) * kBlock + threadIdx.x; v < nrows; v += int64_t(gridDim.x) * kBlock) {
        const double s = double(sq[v]);
        a += double(rowptr[v + 1] - rowptr[v]) * s;
        b += double(indeg[v]) * s;
    }
    const double ta = block_sum_fixed(a, smem);
    const double tb = block_sum_fixed(b, smem);
    if (threadIdx.x == 0) {
        ws[blockIdx.x] = ta;
        ws[gridDim.x + blockIdx.x] = tb;
    }
}

// out[q] = sum_{i<n} in[q*stride + i], fixed order, for q < nout.  One workgroup.
__global__ __launch_bounds__(1024) void reduce_fixed_kernel(const double *__restrict__ in, int64_t n, int64_t stride,
                                                            int nout, double *__restrict__ out) {
    __shared__ double smem[1024 / kWave];
    for (int q = 0; q < nout; ++q) {
        double s = 0.0;
        for (int64_t i = threadIdx.x; i < n; i += blockDim.x) s += in[q * stride + i];
        const double t = block_sum_fixed(s, smem);
        if (threadIdx.x == 0) out[q] = t;
    }
}

// ws[b] = sum of in[b*slice, min(n, (b+1)*slice)) in a fixed order -- first stage of a long reduction.
__global__ __launch_bounds__(kBlock) void reduce_slices_kernel(const double *__restrict__ in, int64_t n, int64_t slice,
                                                               double *__restrict__ ws) {
    __shared__ double smem[kWavesPerBlock];
    const int64_t a = int64_t(blockIdx.x) * slice;
    const int64_t b = a + slice < n ? a + slice : n;
    double s = 0.0;
    for (int64_t i = a + threadIdx.x; i < b; i += kBlock) s += in[i];
    const double t = block_sum_fixed(s, smem);
    if (threadIdx.x == 0) ws[blockIdx.x] = t;
}

// ---- K2 ------------------------------------------------------------------------------------
// One wave per row; rows of <= 64 edges stay in registers, longer rows take three passes
// over their (L2-resident) segment.
template <typename A>
__global__ __launch_bounds__(kBlock) void segment_softmax_kernel(const int64_t *__restrict__ rowptr, int64_t nrows,
                                                                 A *__restrict__ vals, int64_t min_degree,
                                                                 int64_t max_degree, int rows_per_block) {
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const int64_t row_begin = int64_t(blockIdx.x) * rows_per_block;
    const int64_t row_end = row_begin + rows_per_block < nrows ? row_begin + rows_per_block : nrows;
    const A neg_inf = -A(INFINITY);
    for (int64_t r = row_begin + wave; r < row_end; r += kWavesPerBlock) {
        const int64_t e0 = rowptr[r];
        const int64_t e1 = rowptr[r + 1];
        const int64_t deg = e1 - e0;
        if (deg == 0 || deg <= min_degree) continue;  // short rows may already be normalised by K1
        if (max_degree > 0 && deg > max_degree) continue;  // long rows: segment_softmax_long_kernel
        if (deg <= kWave) {
            const bool in = lane < deg;
            const A v = in ? vals[e0 + lane] : neg_inf;
            const A m = group_max<kWave>(v);
            const A ex = in ? exp_acc<A>(v - m) : A(0);
            const A s = group_sum<kWave>(ex);
            if (in) vals[e0 + lane] = ex / s;
        } else {
            A m = neg_inf;
            for (int64_t e = e0 + lane; e < e1; e += kWave) {
                const A v = vals[e];
                m = v > m ? v : m;
            }
            m = group_max<kWave>(m);
            A s = A(0);
            for (int64_t e = e0 + lane; e < e1; e += kWave) s += exp_acc<A>(vals[e] - m);
            s = group_sum<kWave>(s);
            for (int64_t e = e0 + lane; e < e1; e += kWave) vals[e] = exp_acc<A>(vals[e] - m) / s;
        }
    }
}

// One workgroup per long row (a single wave would walk a 70k-edge hub three times on its own):
// workgroup max, workgroup sum of exp, normalise -- thread-strided passes over the (L2-resident)
// segment, fixed-order LDS folds.
template <typename A, int WAVES>
__global__ __launch_bounds__(WAVES *kWave) void segment_softmax_long_kernel(const int64_t *__restrict__ rowptr,
                                                                           const int32_t *__restrict__ long_rows,
                                                                           A *__restrict__ vals, int64_t min_degree) {
    __shared__ A smem[WAVES];
    __shared__ A bcast;
    const int64_t r = long_rows[blockIdx.x];
    const int64_t e0 = rowptr[r];
    const int64_t e1 = rowptr[r + 1];
    if (e1 - e0 <= min_degree) return;
    const int wave = threadIdx.x / kWave, lane = lane_id();
    constexpr int T = WAVES * kWave;
    A m = -A(INFINITY);
    for (int64_t e = e0 + threadIdx.x; e < e1; e += T) {
        const A v = vals[e];
        m = v > m ? v : m;
    }
    m = group_max<kWave>(m);
    if (lane == 0) smem[wave] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        A t = smem[0];
        for (int w = 1; w < WAVES; ++w) t = smem[w] > t ? smem[w] : t;
        bcast = t;
    }
    __syncthreads();
    m = bcast;
    A s = A(0);
    for (int64_t e = e0 + threadIdx.x; e < e1; e += T) s += exp_acc<A>(vals[e] - m);
    s = group_sum<kWave>(s);
    __syncthreads();
    if (lane == 0) smem[wave] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        A t = A(0);
        for (int w = 0; w < WAVES; ++w) t += smem[w];
        bcast = t;
    }
    __syncthreads();
    s = bcast;
    for (int64_t e = e0 + threadIdx.x; e < e1; e += T) vals[e] = exp_acc<A>(vals[e] - m) / s;
}

// ---- sum|A - B| (outer-loop delta, embedder.py:60) -------------------------------------------
// sq_a != nullptr: also sq_a[r] = |A[r,:]|^2, accumulated exactly as row_sqnorm_kernel (K0) does -- lane l takes the
// packs l, l + LPR, ..., one fma per element, the same butterfly -- so the pass that measures how far an outer round
// moved Z also leaves the norms the NEXT build_P needs (similarity.py:37), bit for bit K0's, for no extra traffic: it
// reads every row of A anyway.
template <typename T, int VEC, int LPR>
__global__ __launch_bounds__(kBlock) void l1_distance_kernel(const T *__restrict__ Am, int64_t lda,
                                                             const T *__restrict__ Bm, int64_t ldb, int64_t nrows,
                                                             int d, typename Elem<T>::acc_t *__restrict__ sq_a,
                                                             double *__restrict__ ws) {
    using A = typename Elem<T>::acc_t;
    __shared__ double smem[kWavesPerBlock];
    constexpr int RPW = kWave / LPR;
    constexpr int kStreamRows = stream_rows(LPR);
    const int lane = lane_id();
    const int sub = lane / LPR, sl = lane % LPR;
    const int64_t wave = int64_t(blockIdx.x) * kWavesPerBlock + threadIdx.x / kWave;
    const int64_t nwaves = int64_t(gridDim.x) * kWavesPerBlock;
    double dsum = 0.0;
    for (int64_t base = wave * (RPW * kStreamRows); base < nrows; base += nwaves * (RPW * kStreamRows)) {
        A s[kStreamRows], q[kStreamRows];                  // kStreamRows row groups per turn: 2 x kStreamRows loads in flight
#pragma unroll
        for (int j = 0; j < kStreamRows; ++j) s[j] = q[j] = A(0);
        for (int c0 = sl * VEC; c0 < d; c0 += LPR * VEC) {
            Pack<T, VEC> a[kStreamRows], b[kStreamRows];
#pragma unroll
            for (int j = 0; j < kStreamRows; ++j) {
                const int64_t r = base + j * RPW + sub;
                a[j] = b[j] = Pack<T, VEC>{};
                if (r < nrows) {
                    a[j] = load_pack<T, VEC>(Am + r * lda + c0);
                    b[j] = load_pack<T, VEC>(Bm + r * ldb + c0);
                }
            }
#pragma unroll
            for (int j = 0; j < kStreamRows; ++j) {
#pragma unroll
                for (int k = 0; k < VEC; ++k) {
                    const A av = Elem<T>::to_acc(a[j].v[k]);
                    s[j] += fabs(av - Elem<T>::to_acc(b[j].v[k]));
                    q[j] = fma(av, av, q[j]);               // K0's order: lane l takes the packs l, l + LPR, ...
                }
            }
        }
#pragma unroll
        for (int j = 0; j < kStreamRows; ++j) {
            const int64_t r = base + j * RPW + sub;
            dsum += double(s[j]);                           // rows past the end contribute exact zeros
            if (sq_a != nullptr) {                          // kernel argument: uniform
                const A t = group_sum<LPR>(q[j]);
                if (r < nrows && sl == 0) sq_a[r] = t;
            }
        }
    }
    const double t = block_sum_fixed(dsum, smem);
    if (threadIdx.x == 0) ws[blockIdx.x] = t;
}

// ---- dst[i, :] = src[idx[i], :]  -- packs the rows other ranks read before the halo exchange -------
template <typename T, int VEC, int LPR>
__global__ __launch_bounds__(kBlock) void gather_rows_kernel(const T *__restrict__ src, int64_t lds,
                                                             const int32_t *__restrict__ idx, int64_t n, int d,
                                                             T *__restrict__ dst, int64_t ldd) {
    constexpr int RPW = kWave / LPR;
    const int lane = lane_id();
    const int sub = lane / LPR, sl = lane % LPR;
    const int64_t wave = int64_t(blockIdx.x) * kWavesPerBlock + threadIdx.x / kWave;
    const int64_t nwaves = int64_t(gridDim.x) * kWavesPerBlock;
    for (int64_t base = wave * RPW; base < n; base += nwaves * RPW) {
        const int64_t r = base + sub;
        if (r >= n) continue;
        const int64_t row = idx[r];
        for (int c0 = sl * VEC; c0 < d; c0 += LPR * VEC)
            store_pack<T, VEC>(dst + r * ldd + c0, load_pack<T, VEC>(src + row * lds + c0));
    }
}

// ---- CosineSimilarity on explicit pairs (similarity.py:26-37) --------------------------------
// out[r] <- dot(A_r, B_r); ws[b], ws[G+b] <- block partials of |A_r|^2, |B_r|^2.
template <typename T, int VEC, int LPR>
__global__ __launch_bounds__(kBlock) void pair_dot_kernel(const T *__restrict__ Am, int64_t lda,
                                                          const T *__restrict__ Bm, int64_t ldb, int64_t nrows, int d,
                                                          typename Elem<T>::acc_t *__restrict__ out,
                                                          double *__restrict__ ws) {
    using A = typename Elem<T>::acc_t;
    __shared__ double smem[kWavesPerBlock];
    constexpr int RPW = kWave / LPR;
    const int lane = lane_id();
    const int sub = lane / LPR, sl = lane % LPR;
    const int64_t wave = int64_t(blockIdx.x) * kWavesPerBlock + threadIdx.x / kWave;
    const int64_t nwaves = int64_t(gridDim.x) * kWavesPerBlock;
    double qa = 0.0, qb = 0.0;
    for (int64_t base = wave * RPW; base < nrows; base += nwaves * RPW) {
        const int64_t r = base + sub;
        A dot = A(0), sa = A(0), sb = A(0);
        if (r < nrows) {
            for (int c0 = sl * VEC; c0 < d; c0 += LPR * VEC) {
                const Pack<T, VEC> a = load_pack<T, VEC>(Am + r * lda + c0);
                const Pack<T, VEC> b = load_pack<T, VEC>(Bm + r * ldb + c0);
#pragma unroll
                for (int k = 0; k < VEC; ++k) {
                    const A x = Elem<T>::to_acc(a.v[k]), y = Elem<T>::to_acc(b.v[k]);
                    dot = fma(x, y, dot);
                    sa = fma(x, x, sa);
                    sb = fma(y, y, sb);
                }
            }
        }
        dot = group_sum<LPR>(dot);
        if (r < nrows && sl == 0) out[r] = dot;
        qa += double(sa);
        qb += double(sb);
    }
    const double ta = block_sum_fixed(qa, smem);
    const double tb = block_sum_fixed(qb, smem);
    if (threadIdx.x == 0) {
        ws[blockIdx.x] = ta;
        ws[gridDim.x + blockIdx.x] = tb;
    }
}

template <typename A>
__global__ __launch_bounds__(kBlock) void pair_scale_kernel(A *__restrict__ out, int64_t nrows,
                                                            const double *__restrict__ sums2) {
    const A D = sqrt(A(sums2[0])) * sqrt(A(sums2[1]));
    for (int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x; i < nrows; i += int64_t(gridDim.x) * kBlock)
        out[i] = out[i] / D;
}

// Column-split runs (each GPU holds d/N columns of every row): clane_edge_score_* in RAW_DOT mode leaves the
// partial dot products of this GPU's columns; after they are summed over the GPUs this pass divides by the
// denominators exactly as finalize_score() in edge_score.h does (similarity.py:37).
template <typename A>
__global__ __launch_bounds__(kBlock) void edge_score_finalize_kernel(
    const int64_t *__restrict__ rowptr, const int32_t *__restrict__ colidx, int64_t nrows, int64_t row0, int mode,
    const double *__restrict__ sums2, const A *__restrict__ sq, A *__restrict__ scores, int rows_per_block) {
    if (mode == 0) {  // reference: one global denominator
        const A D = sqrt(A(sums2[0])) * sqrt(A(sums2[1]));
        const int64_t e_end = rowptr[nrows];
        for (int64_t e = rowptr[0] + int64_t(blockIdx.x) * kBlock + threadIdx.x; e < e_end;
             e += int64_t(gridDim.x) * kBlock)
            scores[e] = scores[e] / D;
        return;
    }
    const int wave = threadIdx.x / kWave, lane = lane_id();
    const int64_t row_begin = int64_t(blockIdx.x) * rows_per_block;
    const int64_t row_end = row_begin + rows_per_block < nrows ? row_begin + rows_per_block : nrows;
    for (int64_t r = row_begin + wave; r < row_end; r += kWavesPerBlock) {
        const int64_t e0 = rowptr[r], e1 = rowptr[r + 1];
        if (e0 == e1) continue;
        const A nsrc = sqrt(sq[row0 + r]);
        for (int64_t e = e0 + lane; e < e1; e += kWave) scores[e] = scores[e] / (nsrc * sqrt(sq[colidx[e]]));
    }
}

// out[w] = XCD that workgroup w of this launch ran on (diagnostic: see xcc_id()).
__global__ void xcc_ids_kernel(int32_t *__restrict__ out) {
    if (threadIdx.x == 0) out[blockIdx.x] = xcc_id();
}

// Validates what every gather kernel takes on trust: rowptr[0..nrows] non-decreasing within [0, n_edges] and every
// colidx[e] a row of the table.  status[0] |= 1: a bad rowptr entry, |= 2: a column out of range.  (A kernel that
// gathers through a bad index faults the GPU -- on this pool that can reset every GPU of the host.)
__global__ void check_csr_kernel(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ colidx, int64_t nrows,
                                 int64_t n_edges, int64_t table_rows, int32_t *__restrict__ status) {
    const int64_t stride = int64_t(gridDim.x) * blockDim.x;
    int bad = 0;
    for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i <= nrows; i += stride) {
        const int64_t a = rowptr[i];
        if (a < 0 || a > n_edges || (i < nrows && rowptr[i + 1] < a)) bad |= 1;
    }
    for (int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; e < n_edges; e += stride) {
        const int32_t c = colidx[e];
        if (c < 0 || c >= table_rows) bad |= 2;
    }
    if (bad) atomicOr(status, bad);
}

}  // namespace clane
