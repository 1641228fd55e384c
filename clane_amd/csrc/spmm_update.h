// K3 -- one Jacobi sweep  Z_new = X + gamma * P @ Z_old  over a block of CSR rows, fused with
// the L1 delta  sum|Z_new - Z_old|.
//
// Reference being replaced: the per-vertex Python loop of clane/embedder.py:84-92 (gather
// Z_old[nbrs], [1,deg]@[deg,d] mm, scale, add x) and the reduction at embedder.py:94.
//
// Mapping to gfx950 (HBM-bound, no MFMA: there is no dense contraction here):
//  * one wave64 owns one destination row.  A row of d elements is covered by LPR lanes that
//    each move one 16-byte pack (d=256 fp32: LPR=64, one 1-KiB row per wave-instruction);
//    when a row needs fewer than 64 lanes the wave gathers 64/LPR neighbour rows with ONE
//    instruction (sub-wave `sub` takes every (64/LPR)-th edge) and folds the sub-wave sums in
//    a fixed butterfly at the end of the row.
//  * colidx / P are read 64 edges at a time, coalesced, then broadcast lane->wave
//    (v_readlane when the edge index is wave-uniform, so the row base address is scalar).
//  * U row loads are issued back-to-back before the first FMA consumes one, so every wave
//    keeps U KiB of gathers in flight; edges past the end of a row are exec-masked loads
//    (no traffic), never branches.
//  * a workgroup owns `rows_per_block` CONSECUTIVE rows (its 4 waves interleave over them) and
//    there are many more workgroups than CUs, so the hardware dispatcher balances skewed rows
//    dynamically.  (A fixed grid striding rows by a power of two is pathological on R-MAT:
//    stride 2^13 hands one wave every id with 13 trailing zero bits, i.e. the heaviest hubs.)
//    One delta partial per workgroup, reduced later in index order: no float atomics anywhere.
//  * rows longer than `long_threshold` edges are skipped here and done by spmm_long_kernel
//    (one 16-wave workgroup per row, per-wave edge segments, fixed-order LDS fold).
#pragma once

#include "device_utils.h"

namespace clane {

// acc[k] += sum over edges e in [e0, e1) of P[e] * Z[colidx[e], col + k]  for this lane's pack.
// `zcol` already includes the lane's column offset.  Must be called by all 64 lanes.
template <typename T, typename PT, int VEC, int LPR, int U>
__device__ __forceinline__ void gather_accumulate(const int32_t *__restrict__ colidx, const PT *__restrict__ P,
                                                  int64_t e0, int64_t e1, const T *__restrict__ zcol, int64_t ldz,
                                                  bool col_ok, typename Elem<T>::acc_t (&acc)[VEC]) {
    using A = typename Elem<T>::acc_t;
    constexpr int EPW = kWave / LPR;  // neighbour rows fetched by one wave-instruction
    const int lane = lane_id();
    const int sub = lane / LPR;
    for (int64_t e = e0; e < e1; e += kWave) {
        const int64_t left = e1 - e;
        const int n = left < kWave ? int(left) : kWave;
        int c = 0;
        A p = A(0);
        if (lane < n) {
            c = colidx[e + lane];
            p = A(P[e + lane]);
        }
        for (int j = 0; j < n; j += EPW * U) {
            Pack<T, VEC> z[U];
            A pj[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int idx = j + u * EPW + sub;
                int cj;
                A pv;
                if constexpr (LPR == kWave) {
                    cj = lane_get_uniform(c, idx & (kWave - 1));
                    pv = lane_get_uniform(p, idx & (kWave - 1));
                } else {
                    cj = lane_get(c, idx & (kWave - 1));
                    pv = lane_get(p, idx & (kWave - 1));
                }
                const bool in_row = idx < n;
                pj[u] = in_row ? pv : A(0);
                z[u] = Pack<T, VEC>{};
                if (in_row && col_ok) z[u] = load_pack<T, VEC>(zcol + int64_t(cj) * ldz);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
#pragma unroll
                for (int k = 0; k < VEC; ++k) acc[k] = fma(pj[u], Elem<T>::to_acc(z[u].v[k]), acc[k]);
            }
        }
    }
}

// Fold the 64/LPR sub-wave partial sums: every lane ends with the row total for its column.
template <int LPR, typename A, int VEC>
__device__ __forceinline__ void fold_subwaves(A (&acc)[VEC]) {
#pragma unroll
    for (int m = LPR; m < kWave; m <<= 1) {
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] += __shfl_xor(acc[k], m, kWave);
    }
}

template <typename T, typename PT, int VEC, int LPR, int U>
__global__ __launch_bounds__(kBlock) void spmm_update_kernel(
    const int64_t *__restrict__ rowptr, const int32_t *__restrict__ colidx, const PT *__restrict__ P, int64_t nrows,
    int64_t row0, const T *__restrict__ Zold, int64_t ldz, const T *__restrict__ X, int64_t ldx,
    typename Elem<T>::acc_t gamma, T *__restrict__ Znew, int64_t ldo, int d, int64_t long_threshold,
    int rows_per_block, double *__restrict__ partials) {
    using A = typename Elem<T>::acc_t;
    __shared__ double smem[kWavesPerBlock];
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const int sub = lane / LPR;
    const int sl = lane % LPR;
    const int64_t row_begin = int64_t(blockIdx.x) * rows_per_block;
    const int64_t row_end = row_begin + rows_per_block < nrows ? row_begin + rows_per_block : nrows;
    double dsum = 0.0;

    for (int64_t r = row_begin + wave; r < row_end; r += kWavesPerBlock) {
        const int64_t e0 = rowptr[r];
        const int64_t e1 = rowptr[r + 1];
        if (long_threshold > 0 && e1 - e0 > long_threshold) continue;
        A rsum = A(0);
        for (int t0 = 0; t0 < d; t0 += LPR * VEC) {
            const int c0 = t0 + sl * VEC;
            const bool col_ok = c0 < d;
            const bool writer = col_ok && sub == 0;
            Pack<T, VEC> x{}, zo{};
            if (writer) {
                x = load_pack<T, VEC>(X + r * ldx + c0);
                zo = load_pack<T, VEC>(Zold + (row0 + r) * ldz + c0);
            }
            A acc[VEC];
#pragma unroll
            for (int k = 0; k < VEC; ++k) acc[k] = A(0);
            gather_accumulate<T, PT, VEC, LPR, U>(colidx, P, e0, e1, Zold + c0, ldz, col_ok, acc);
            fold_subwaves<LPR>(acc);
            if (writer) {
                Pack<T, VEC> out;
#pragma unroll
                for (int k = 0; k < VEC; ++k) {
                    const A zold = Elem<T>::to_acc(zo.v[k]);
                    const A znew = e1 > e0 ? Elem<T>::to_acc(x.v[k]) + gamma * acc[k] : zold;
                    out.v[k] = Elem<T>::from_acc(znew);
                    rsum += fabs(Elem<T>::to_acc(out.v[k]) - zold);
                }
                store_pack<T, VEC>(Znew + r * ldo + c0, out);
            }
        }
        dsum += double(rsum);
    }
    const double total = block_sum_fixed(dsum, smem);
    if (threadIdx.x == 0) partials[blockIdx.x] = total;
}

// One workgroup of WAVES waves per long row: wave w gathers edge segment w, the segment sums
// are folded through LDS in wave order (fixed order => reproducible), wave 0 writes the row.
template <typename T, typename PT, int VEC, int LPR, int U, int WAVES>
__global__ __launch_bounds__(WAVES *kWave) void spmm_long_kernel(
    const int64_t *__restrict__ rowptr, const int32_t *__restrict__ colidx, const PT *__restrict__ P,
    const int32_t *__restrict__ long_rows, int64_t row0, const T *__restrict__ Zold, int64_t ldz,
    const T *__restrict__ X, int64_t ldx, typename Elem<T>::acc_t gamma, T *__restrict__ Znew, int64_t ldo, int d,
    double *__restrict__ partials) {
    using A = typename Elem<T>::acc_t;
    __shared__ A red[WAVES][kWave][VEC];
    __shared__ double smem[WAVES];
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const int sub = lane / LPR;
    const int sl = lane % LPR;
    const int64_t r = long_rows[blockIdx.x];
    const int64_t e0 = rowptr[r];
    const int64_t e1 = rowptr[r + 1];
    const int64_t seg = ceil_div(ceil_div(e1 - e0, WAVES), kWave) * kWave;
    const int64_t a = e0 + wave * seg;
    const int64_t b = a + seg < e1 ? a + seg : e1;
    double dsum = 0.0;

    for (int t0 = 0; t0 < d; t0 += LPR * VEC) {
        const int c0 = t0 + sl * VEC;
        const bool col_ok = c0 < d;
        A acc[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] = A(0);
        gather_accumulate<T, PT, VEC, LPR, U>(colidx, P, a, b, Zold + c0, ldz, col_ok, acc);
        fold_subwaves<LPR>(acc);
#pragma unroll
        for (int k = 0; k < VEC; ++k) red[wave][lane][k] = acc[k];
        __syncthreads();
        if (wave == 0 && col_ok && sub == 0) {
            const Pack<T, VEC> x = load_pack<T, VEC>(X + r * ldx + c0);
            const Pack<T, VEC> zo = load_pack<T, VEC>(Zold + (row0 + r) * ldz + c0);
            Pack<T, VEC> out;
            A rsum = A(0);
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                A tot = A(0);
                for (int w = 0; w < WAVES; ++w) tot += red[w][lane][k];
                const A zold = Elem<T>::to_acc(zo.v[k]);
                const A znew = Elem<T>::to_acc(x.v[k]) + gamma * tot;
                out.v[k] = Elem<T>::from_acc(znew);
                rsum += fabs(Elem<T>::to_acc(out.v[k]) - zold);
            }
            store_pack<T, VEC>(Znew + r * ldo + c0, out);
            dsum += double(rsum);
        }
        __syncthreads();
    }
    const double total = block_sum_fixed(dsum, smem);
    if (threadIdx.x == 0) partials[blockIdx.x] = total;
}

}  // namespace clane
