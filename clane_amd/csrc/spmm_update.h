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
//  * a workgroup owns `rows_per_block` CONSECUTIVE rows and there are many more workgroups
//    than CUs, so the hardware dispatcher balances skewed rows dynamically.  (A fixed grid
//    striding rows by a power of two is pathological on R-MAT: stride 2^13 hands one wave
//    every id with 13 trailing zero bits, i.e. the heaviest hubs.)
//  * inside a workgroup: the block's rowptr slice is staged in LDS once; waves CLAIM rows from
//    an LDS counter (a wave stuck on a 1000-edge row does not hold the others back), and the
//    colidx / P of the next claimed row are requested before the current row's gathers, so a
//    row costs one dependent memory round trip (its gathers), not three.
//  * the L1 delta is kept per ROW in LDS and summed in row order at the end of the block, so
//    the result does not depend on which wave claimed which row: one partial per workgroup,
//    reduced later in index order -- bitwise reproducible, no float atomics anywhere.
//  * rows longer than `long_threshold` edges are skipped here and done by spmm_long_kernel
//    (one 16-wave workgroup per row, per-wave edge segments, fixed-order LDS fold).
#pragma once

#include "device_utils.h"

#ifndef CLANE_SPMM_U
#define CLANE_SPMM_U 8            // neighbour-row loads in flight per wave (16-byte path)
#endif
#ifndef CLANE_SPMM_DYNAMIC
#define CLANE_SPMM_DYNAMIC 1      // waves claim rows from an LDS counter (0: static interleave)
#endif
#ifndef CLANE_COMBINE_WAVES
#define CLANE_COMBINE_WAVES 2     // waves that share the slots of one class row in spmm_class_combine_kernel (profiles/r02_ab_combine_waves.md)
#endif
#ifndef CLANE_SPMM_PREFETCH
#define CLANE_SPMM_PREFETCH 1     // request the next row's colidx/P before gathering the current row
#endif
#ifndef CLANE_SPMM_MIN_WAVES
#define CLANE_SPMM_MIN_WAVES 0    // __launch_bounds__ 2nd argument (waves per SIMD), 0 = unconstrained
#endif

namespace clane {

// First (<= 64-edge) chunk of a row's colidx / P, one edge per lane.
template <typename A>
struct EdgeChunk {
    int c;
    A p;
};

template <typename A, typename PT>
__device__ __forceinline__ EdgeChunk<A> load_chunk(const int32_t *__restrict__ colidx, const PT *__restrict__ P,
                                                   int64_t e, int64_t e1) {
    EdgeChunk<A> ch{0, A(0)};
    if (e + lane_id() < e1) {
        ch.c = colidx[e + lane_id()];
        ch.p = A(P[e + lane_id()]);
    }
    return ch;
}

// acc[k] += sum over the n (<= 64) edges held in `ch` of p * Z[c, col + k] for this lane's pack.
// Two forms of the same loop:
//  * masked (fp32 / fp64): loads of edges past the end of the row are exec-masked -- no request at all;
//  * branch-free (bf16): hipcc puts phi copies of the unpacked values inside each conditional-load
//    block and waits for every load on its own (seen in the .s), so there every load is
//    unconditional from an always-valid address (edge past the end -> the chunk's first column,
//    column lane past d -> column 0) and the VALUE is zeroed instead.
template <typename T, int VEC, int LPR, int U>
__device__ __forceinline__ void accumulate_chunk(const EdgeChunk<typename Elem<T>::acc_t> &ch, int n,
                                                 const T *__restrict__ zcol, int64_t ldz, bool col_ok,
                                                 typename Elem<T>::acc_t (&acc)[VEC]) {
    using A = typename Elem<T>::acc_t;
    constexpr int EPW = kWave / LPR;  // neighbour rows fetched by one wave-instruction
    constexpr bool kBranchFree = sizeof(T) == 2;
    const int sub = lane_id() / LPR;
    int c_all = ch.c;
    if constexpr (kBranchFree) c_all = lane_id() < n ? ch.c : lane_get_uniform(ch.c, 0);
    for (int j = 0; j < n; j += EPW * U) {
        Pack<T, VEC> z[U];
        A pj[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int idx = j + u * EPW + sub;
            int cj;
            A pv;
            if constexpr (LPR == kWave) {
                cj = lane_get_uniform(c_all, idx & (kWave - 1));
                pv = lane_get_uniform(ch.p, idx & (kWave - 1));
            } else {
                cj = lane_get(c_all, idx & (kWave - 1));
                pv = lane_get(ch.p, idx & (kWave - 1));
            }
            const bool in_row = idx < n;
            pj[u] = in_row ? pv : A(0);
            if constexpr (kBranchFree) {
                z[u] = load_pack<T, VEC>(zcol + int64_t(cj) * ldz);  // zcol is column 0 for lanes past d (see callers)
            } else {
                z[u] = Pack<T, VEC>{};
                if (in_row && col_ok) z[u] = load_pack<T, VEC>(zcol + int64_t(cj) * ldz);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                A zv = Elem<T>::to_acc(z[u].v[k]);
                if constexpr (kBranchFree) zv = pj[u] != A(0) ? zv : A(0);  // keeps 0 * inf out of the sum
                acc[k] = fma(pj[u], zv, acc[k]);
            }
        }
    }
}

// acc[k] += sum over edges e in [e0, e1) of P[e] * Z[colidx[e], col + k].  `first` holds the
// chunk starting at e0 if `have_first` (prefetched by the caller).  Called by all 64 lanes.
template <typename T, typename PT, int VEC, int LPR, int U>
__device__ __forceinline__ void gather_accumulate(const int32_t *__restrict__ colidx, const PT *__restrict__ P,
                                                  int64_t e0, int64_t e1, const T *__restrict__ zcol, int64_t ldz,
                                                  bool col_ok, typename Elem<T>::acc_t (&acc)[VEC],
                                                  const EdgeChunk<typename Elem<T>::acc_t> &first, bool have_first) {
    using A = typename Elem<T>::acc_t;
    for (int64_t e = e0; e < e1; e += kWave) {
        const int64_t left = e1 - e;
        const int n = left < kWave ? int(left) : kWave;
        EdgeChunk<A> ch = first;
        if (!(have_first && e == e0)) ch = load_chunk<A, PT>(colidx, P, e, e1);
        accumulate_chunk<T, VEC, LPR, U>(ch, n, zcol, ldz, col_ok, acc);
    }
}

// Fold the 64/LPR sub-wave partial sums: every lane ends with the row total for its column.
template <int LPR, typename A, int VEC>
__device__ __forceinline__ void fold_subwaves(A (&acc)[VEC]) {
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
        if constexpr (LPR <= 4) acc[k] += lane_xor<4>(acc[k]);
        if constexpr (LPR <= 8) acc[k] += lane_xor<8>(acc[k]);
        if constexpr (LPR <= 16) acc[k] += lane_xor<16>(acc[k]);
        if constexpr (LPR <= 32) acc[k] += lane_xor<32>(acc[k]);
    }
}

// Optional further destinations of finished rows: row r (relative to the call's first row) is also stored at
// the places slot[row_ptr[r] .. row_ptr[r+1]); a place is (buffer << 28 | row) into `bufs`, a device array of up
// to 8 matrix base addresses.  One buffer: the send buffer of the halo exchange, packed by the kernel that
// produces the row instead of by a separate gather pass.  Several: the other GPUs' own tables, mapped into this
// process -- the row goes straight over xGMI, no exchange step at all.  row_ptr == nullptr: none.
constexpr int kMirrorRowBits = 28;
template <typename T>
struct Mirror {
    const int64_t *row_ptr;
    const int32_t *slot;
    T *const *bufs;
    int64_t ld;
    __device__ __forceinline__ T *row(int32_t place) const {
        return bufs[place >> kMirrorRowBits] + int64_t(place & ((1 << kMirrorRowBits) - 1)) * ld;
    }
};

// Epilogue of one row's column tile; returns this lane's share of sum|z_new - z_old| (of the STORED values).
template <typename T, int VEC>
__device__ __forceinline__ typename Elem<T>::acc_t finish_pack(const Pack<T, VEC> &x, const Pack<T, VEC> &zo,
                                                               const typename Elem<T>::acc_t (&acc)[VEC],
                                                               typename Elem<T>::acc_t gamma, bool has_edges,
                                                               T *__restrict__ dst, Pack<T, VEC> &out, bool nt_store) {
    using A = typename Elem<T>::acc_t;
    A rsum = A(0);
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
        const A zold = Elem<T>::to_acc(zo.v[k]);
        const A znew = has_edges ? Elem<T>::to_acc(x.v[k]) + gamma * acc[k] : zold;  // embedder.py:88-92
        out.v[k] = Elem<T>::from_acc(znew);
        const A stored = Elem<T>::to_acc(out.v[k]);
        rsum += fabs(stored - zold);
    }
    store_pack_stream<T, VEC>(dst, out, nt_store);
    return rsum;
}

// The further destinations of a finished row (Mirror above), stored by the GROUP lanes that cover the row TOGETHER: the
// row's places are read with one coalesced load (lane j of the group takes place j) and handed round lane to lane, so
// a row that goes to 7 readers costs one memory round trip before its stores, not a chain of 7 dependent ones (round
// 4: every lane used to walk slot[row_ptr[r] ..] on its own, load after load).  EVERY lane of the group must call this
// (`row` is uniform over the group); only `writer` lanes hold a pack and store it.  `base` = first lane of the group.
template <typename T, int VEC, int GROUP>
__device__ __forceinline__ void mirror_store(const Mirror<T> &mirror, int64_t row, int col, const Pack<T, VEC> &out,
                                             bool writer, int group_lane, int base) {
    if (mirror.row_ptr == nullptr) return;                       // kernel argument: uniform
    const int64_t s0 = mirror.row_ptr[row], s1 = mirror.row_ptr[row + 1];
    for (int64_t s = s0; s < s1; s += GROUP) {
        const int n = s1 - s < GROUP ? int(s1 - s) : GROUP;
        int32_t mine = 0;
        if (group_lane < n) mine = mirror.slot[s + group_lane];
        for (int j = 0; j < n; ++j) {
            int32_t place;
            if constexpr (GROUP == kWave) place = lane_get_uniform(mine, j);   // the whole wave works on one row
            else place = lane_get(mine, base + j);
            if (writer) store_pack<T, VEC>(mirror.row(place) + col, out);
        }
    }
}

// The same with the row's first GROUP places PREFETCHED: the row kernels know a row's place count when they claim the
// row (the block's slice of mirror.row_ptr sits in LDS beside its rowptr slice) and request the places then (`mine`:
// lane j of the group holds place j), so they arrive under the row's gathers and the finished row is stored without a
// single memory round trip of its own -- on one rank's halo kernels at N = 8 the two dependent loads per row (row_ptr,
// then slot) were 0.56 ms of a 1.36 ms pass (profiles/r04_rank_compute_halo_variants.jsonl).  Places beyond GROUP
// (more readers than lanes: never with <= 8 GPUs) take the loop above.
template <typename T, int VEC, int GROUP>
__device__ __forceinline__ void mirror_store_prefetched(const Mirror<T> &mirror, int64_t m0, int64_t m1, int32_t mine,
                                                        int col, const Pack<T, VEC> &out, bool writer, int group_lane,
                                                        int base) {
    if (mirror.row_ptr == nullptr) return;                       // kernel argument: uniform
    const int n = m1 - m0 < GROUP ? int(m1 - m0) : GROUP;
    for (int j = 0; j < n; ++j) {
        int32_t place;
        if constexpr (GROUP == kWave) place = lane_get_uniform(mine, j);
        else place = lane_get(mine, base + j);
        if (writer) store_pack<T, VEC>(mirror.row(place) + col, out);
    }
    for (int64_t s = m0 + GROUP; s < m1; s += GROUP) {           // the rest, if any, the plain way
        const int k = m1 - s < GROUP ? int(m1 - s) : GROUP;
        int32_t more = 0;
        if (group_lane < k) more = mirror.slot[s + group_lane];
        for (int j = 0; j < k; ++j) {
            int32_t place;
            if constexpr (GROUP == kWave) place = lane_get_uniform(more, j);
            else place = lane_get(more, base + j);
            if (writer) store_pack<T, VEC>(mirror.row(place) + col, out);
        }
    }
}

#if CLANE_SPMM_MIN_WAVES > 0
#define CLANE_SPMM_BOUNDS __launch_bounds__(kBlock, CLANE_SPMM_MIN_WAVES)
#else
#define CLANE_SPMM_BOUNDS __launch_bounds__(kBlock)
#endif

// MIRRORED: compiled in only for launches that have a mirror (multi-GPU halo / p2p): the prefetched places cost
// registers (the bf16 sub-wave instance would drop from 5 to 4 waves per SIMD) that a one-GPU sweep must not pay.
template <typename T, typename PT, int VEC, int LPR, int U, bool MIRRORED>
__global__ CLANE_SPMM_BOUNDS void spmm_update_kernel(
    const int64_t *__restrict__ rowptr, const int32_t *__restrict__ colidx, const PT *__restrict__ P, int64_t nrows,
    int64_t row0, const T *__restrict__ Zold, int64_t ldz, const T *__restrict__ X, int64_t ldx,
    typename Elem<T>::acc_t gamma, T *__restrict__ Znew, int64_t ldo, int d, int64_t long_threshold,
    bool skip_sinks, int rows_per_block, Mirror<T> mirror, bool nt_store, double *__restrict__ partials) {
    using A = typename Elem<T>::acc_t;
    __shared__ int64_t s_rowptr[kMaxRowsPerBlock + 1];
    __shared__ int64_t s_mptr[MIRRORED ? kMaxRowsPerBlock + 1 : 1];   // the block's slice of mirror.row_ptr
    __shared__ double s_rowsum[kMaxRowsPerBlock];
    __shared__ int s_next;
    __shared__ int s_done;

    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const int sub = lane / LPR;
    const int sl = lane % LPR;
    const int64_t row_begin = int64_t(blockIdx.x) * rows_per_block;
    const int nb = int((row_begin + rows_per_block < nrows ? row_begin + rows_per_block : nrows) - row_begin);

    for (int i = threadIdx.x; i <= nb; i += kBlock) {
        s_rowptr[i] = rowptr[row_begin + i];
        if constexpr (MIRRORED) s_mptr[i] = mirror.row_ptr[row_begin + i];
    }
    for (int i = threadIdx.x; i < nb; i += kBlock) s_rowsum[i] = 0.0;
    if (threadIdx.x == 0) {
        s_next = kWavesPerBlock;
        s_done = 0;
    }
    __syncthreads();
    // a row's first 64 mirror places, requested when the row is claimed (lane j: place j)
    auto places_of = [&]([[maybe_unused]] int row, [[maybe_unused]] int64_t &m0, [[maybe_unused]] int64_t &m1) -> int32_t {
        if constexpr (MIRRORED) {
            m0 = s_mptr[row];
            m1 = s_mptr[row + 1];
            return lane < m1 - m0 ? mirror.slot[m0 + lane] : 0;
        } else {
            return 0;
        }
    };

    // next row of this block for the calling wave (wave-uniform)
    auto claim = [&]([[maybe_unused]] int prev) -> int {
#if CLANE_SPMM_DYNAMIC
        int v = 0;
        if (lane == 0) v = atomicAdd(&s_next, 1);
        return __builtin_amdgcn_readfirstlane(v);
#else
        return prev + kWavesPerBlock;
#endif
    };

    int cur = wave;
    int64_t e0 = 0, e1 = 0, m0 = 0, m1 = 0;
    int32_t places = 0;
    EdgeChunk<A> ch{0, A(0)};
    if (cur < nb) {
        e0 = s_rowptr[cur];
        e1 = s_rowptr[cur + 1];
        ch = load_chunk<A, PT>(colidx, P, e0, e1);
        places = places_of(cur, m0, m1);
    }
    while (cur < nb) {
        const int nxt = claim(cur);
        int64_t n0 = 0, n1 = 0, mn0 = 0, mn1 = 0;
        int32_t places_n = 0;
        EdgeChunk<A> chn{0, A(0)};
        if (nxt < nb) {
            n0 = s_rowptr[nxt];
            n1 = s_rowptr[nxt + 1];
#if CLANE_SPMM_PREFETCH
            chn = load_chunk<A, PT>(colidx, P, n0, n1);
#endif
            places_n = places_of(nxt, mn0, mn1);
        }
        if (!(long_threshold > 0 && e1 - e0 > long_threshold) && !(skip_sinks && e1 == e0)) {
            const int64_t r = row_begin + cur;
            A rsum = A(0);
            for (int t0 = 0; t0 < d; t0 += LPR * VEC) {
                const int c0 = t0 + sl * VEC;
                const bool col_ok = c0 < d;
                const bool writer = col_ok && sub == 0;
                Pack<T, VEC> x{}, zo{};
                if (writer) {
                    x = load_pack_stream<T, VEC>(X + r * ldx + c0);
                    zo = load_pack_stream<T, VEC>(Zold + (row0 + r) * ldz + c0);
                }
                A acc[VEC];
#pragma unroll
                for (int k = 0; k < VEC; ++k) acc[k] = A(0);
                gather_accumulate<T, PT, VEC, LPR, U>(colidx, P, e0, e1, Zold + (col_ok ? c0 : 0), ldz, col_ok, acc, ch, true);
                fold_subwaves<LPR>(acc);
                Pack<T, VEC> out{};
                if (writer) rsum += finish_pack<T, VEC>(x, zo, acc, gamma, e1 > e0, Znew + r * ldo + c0, out, nt_store);
                if constexpr (MIRRORED)
                    mirror_store_prefetched<T, VEC, kWave>(mirror, m0, m1, places, c0, out, writer, lane, 0);
            }
            rsum = group_sum<kWave>(rsum);
            if (lane == 0) s_rowsum[cur] = double(rsum);
        }
        cur = nxt;
        e0 = n0;
        e1 = n1;
        m0 = mn0;
        m1 = mn1;
        places = places_n;
#if CLANE_SPMM_PREFETCH
        ch = chn;
#else
        if (cur < nb) ch = load_chunk<A, PT>(colidx, P, e0, e1);
#endif
    }
    // No closing barrier: a wave that runs out of rows leaves (its slot goes to another workgroup);
    // the LAST wave to arrive sums the per-row deltas in row order (see last_wave_of_block).
    if (!last_wave_of_block(&s_done)) return;
    double dsum = 0.0;
    for (int i = lane; i < nb; i += kWave) dsum += s_rowsum[i];
    dsum = group_sum<kWave>(dsum);
    if (lane == 0) partials[blockIdx.x] = dsum;
}

// Rows that need fewer than 64 lanes (LPR < 64; e.g. d=128 bf16: 16 lanes x 16 B, or the column slices of a
// multi-GPU run): one SUB-WAVE per destination row, so a wave keeps 64/LPR rows in flight and nothing has to
// be folded across sub-waves.  (Splitting ONE short row's edges over the sub-waves, as the long-row kernel
// does, leaves most lanes of a 5-edge row idle: 3.5 TB/s on the 10M-vertex power-law graph.)
// Same workgroup structure as spmm_update_kernel: consecutive rows, rowptr slice in LDS, per-row deltas summed
// in row order by the last wave.  Every SUB-WAVE claims its rows on its own from the LDS counter and the wave
// advances all of them by one group of U edges per iteration: a sub-wave whose row is finished writes it and
// claims the next one while its neighbours carry on, so a wave's time is the SUM of its rows' groups / (64/LPR)
// instead of the maximum over a fixed group of rows (out-degrees are skewed: the maximum of 8 power-law rows
// is several times their mean).  Gathers are branch-free (see accumulate_chunk): a sub-wave with fewer than U
// edges left re-reads its last neighbour row (an L1 hit) with weight zero.
#ifndef CLANE_SUBROW_MIN_WAVES
#define CLANE_SUBROW_MIN_WAVES 0  // spmm_update_subrow_kernel: __launch_bounds__ 2nd argument (waves per SIMD), 0 = unconstrained
#endif
#if CLANE_SUBROW_MIN_WAVES > 0
#define CLANE_SUBROW_BOUNDS __launch_bounds__(kBlock, (sizeof(typename Elem<T>::acc_t) == 8 ? 1 : CLANE_SUBROW_MIN_WAVES))
#else
#define CLANE_SUBROW_BOUNDS __launch_bounds__(kBlock)
#endif
template <typename T, typename PT, int VEC, int LPR, int U, bool MIRRORED>
__global__ CLANE_SUBROW_BOUNDS void spmm_update_subrow_kernel(
    const int64_t *__restrict__ rowptr, const int32_t *__restrict__ colidx, const PT *__restrict__ P, int64_t nrows,
    int64_t row0, const T *__restrict__ Zold, int64_t ldz, const T *__restrict__ X, int64_t ldx,
    typename Elem<T>::acc_t gamma, T *__restrict__ Znew, int64_t ldo, int d, int64_t long_threshold,
    bool skip_sinks, int rows_per_block, Mirror<T> mirror, bool nt_store, double *__restrict__ partials) {
    using A = typename Elem<T>::acc_t;
    static_assert(LPR < kWave, "use spmm_update_kernel for rows that fill a wave");
    static_assert(LPR % U == 0, "a sub-wave's edge buffer is consumed in whole groups of U");
    __shared__ int64_t s_rowptr[kMaxRowsPerBlock + 1];
    __shared__ int64_t s_mptr[MIRRORED ? kMaxRowsPerBlock + 1 : 1];   // the block's slice of mirror.row_ptr
    __shared__ double s_rowsum[kMaxRowsPerBlock];
    __shared__ int s_next;
    __shared__ int s_done;

    const int lane = lane_id();
    const int sub = lane / LPR;
    const int sl = lane % LPR;
    const int sub_base = sub * LPR;
    const int64_t row_begin = int64_t(blockIdx.x) * rows_per_block;
    const int nb = int((row_begin + rows_per_block < nrows ? row_begin + rows_per_block : nrows) - row_begin);
    const int c0 = sl * VEC;                 // LPR * VEC >= d: a row is one pack per lane
    const bool col_ok = c0 < d;
    const int c0s = col_ok ? c0 : 0;

    for (int i = threadIdx.x; i <= nb; i += kBlock) {
        s_rowptr[i] = rowptr[row_begin + i];
        if constexpr (MIRRORED) s_mptr[i] = mirror.row_ptr[row_begin + i];
    }
    for (int i = threadIdx.x; i < nb; i += kBlock) s_rowsum[i] = 0.0;
    if (threadIdx.x == 0) {
        s_next = 0;
        s_done = 0;
    }
    __syncthreads();

    // per sub-wave (uniform over its LPR lanes): the row it is walking and the LPR edges it has buffered
    int mine = 0;              // row within the block
    bool have = false;         // a row is open: x / zo / acc are live
    bool done = false;         // no rows left in this block
    int64_t e_next = 0;        // first edge of the row not yet buffered
    int64_t e_end = 0;
    int nbuf = 0, pos = 0;     // buffered edges (one per lane: c, p) and how many of them are consumed
    int c = 0;
    A p = A(0);
    [[maybe_unused]] int64_t m0 = 0, m1 = 0;   // the open row's mirror places: range in mirror.slot, the first LPR of
    [[maybe_unused]] int32_t places = 0;       // them one per lane, requested when the row is claimed
    Pack<T, VEC> x{}, zo{};
    A acc[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) acc[k] = A(0);

    for (;;) {
        if (pos >= nbuf && !done) {          // divergent between sub-waves, uniform inside one
            if (e_next == e_end) {           // row finished (or none yet): write it, claim the next one with work
                if (have) {
                    const int64_t r = row_begin + mine;
                    A rsum = A(0);
                    Pack<T, VEC> out{};
                    if (col_ok) rsum = finish_pack<T, VEC>(x, zo, acc, gamma, true, Znew + r * ldo + c0, out, nt_store);
                    if constexpr (MIRRORED)
                        mirror_store_prefetched<T, VEC, LPR>(mirror, m0, m1, places, c0, out, col_ok, sl, sub_base);
                    rsum = group_sum<LPR>(rsum);
                    if (sl == 0) s_rowsum[mine] = double(rsum);
                    have = false;
                }
                for (;;) {
                    int v = 0;
                    if (sl == 0) v = atomicAdd(&s_next, 1);
                    mine = lane_get(v, sub_base);
                    if (mine >= nb) {
                        done = true;
                        break;
                    }
                    const int64_t e0 = s_rowptr[mine];
                    const int64_t dg = s_rowptr[mine + 1] - e0;
                    if (long_threshold > 0 && dg > long_threshold) continue;      // the row-split kernels own it
                    if (dg == 0 && skip_sinks) continue;
                    const int64_t r = row_begin + mine;
                    if (dg == 0) {           // row without out-edges, sinks not skipped: z stays (embedder.py:88-89)
                        if (col_ok) store_pack<T, VEC>(Znew + r * ldo + c0, load_pack<T, VEC>(Zold + (row0 + r) * ldz + c0));
                        continue;
                    }
                    e_next = e0;
                    e_end = e0 + dg;
                    if constexpr (MIRRORED) {
                        m0 = s_mptr[mine];
                        m1 = s_mptr[mine + 1];
                        places = sl < m1 - m0 ? mirror.slot[m0 + sl] : 0;
                    }
                    if (col_ok) {
                        x = load_pack_stream<T, VEC>(X + r * ldx + c0);
                        zo = load_pack_stream<T, VEC>(Zold + (row0 + r) * ldz + c0);
                    }
#pragma unroll
                    for (int k = 0; k < VEC; ++k) acc[k] = A(0);
                    have = true;
                    break;
                }
            }
            pos = 0;
            nbuf = 0;
            c = 0;
            p = A(0);
            if (!done) {                     // refill: the next LPR edges of the row, one per lane
                const int64_t left = e_end - e_next;
                nbuf = left < LPR ? int(left) : LPR;
                if (sl < nbuf) {
                    c = colidx[e_next + sl];
                    p = A(P[e_next + sl]);
                }
                e_next += nbuf;
            }
        }
        if (__all(done)) break;
        // one group of U edges per sub-wave; lanes past the buffered edges hold c = 0 (row 0 is valid), weight 0
        Pack<T, VEC> z[U];
        A pj[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            // pos is a multiple of U and below LPR while a row is open; lanes past nbuf hold c = 0, p = 0
            const int src = sub_base + ((pos + u) & (LPR - 1));
            const int cj = lane_get(c, src);
            pj[u] = lane_get(p, src);
            z[u] = load_pack<T, VEC>(Zold + int64_t(cj) * ldz + c0s);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                A zv = Elem<T>::to_acc(z[u].v[k]);
                zv = pj[u] != A(0) ? zv : A(0);   // keeps 0 * inf (and the zero-weight re-reads) out of the sum
                acc[k] = fma(pj[u], zv, acc[k]);
            }
        }
        pos += U;
    }
    if (!last_wave_of_block(&s_done)) return;
    double dsum = 0.0;
    for (int i = lane; i < nb; i += kWave) dsum += s_rowsum[i];
    dsum = group_sum<kWave>(dsum);
    if (lane == 0) partials[blockIdx.x] = dsum;
}

// One workgroup of WAVES waves per long row.  Wave w gathers the 64-aligned edge slice w; slices are
// folded through LDS in wave order (fixed order => reproducible) and wave 0 writes the row.  A wave
// whose slice is empty leaves at once: on AMD hardware s_barrier waits only for the waves of the
// workgroup that have not terminated, so a 100-edge row costs two working waves, not sixteen
// parked ones -- the same kernel serves 65-edge rows and 70k-edge hubs.
template <typename T, typename PT, int VEC, int LPR, int U, int WAVES>
__global__ __launch_bounds__(WAVES *kWave) void spmm_long_kernel(
    const int64_t *__restrict__ rowptr, const int32_t *__restrict__ colidx, const PT *__restrict__ P,
    const int32_t *__restrict__ long_rows, int64_t row0, const T *__restrict__ Zold, int64_t ldz,
    const T *__restrict__ X, int64_t ldx, typename Elem<T>::acc_t gamma, T *__restrict__ Znew, int64_t ldo, int d,
    Mirror<T> mirror, double *__restrict__ partials) {
    using A = typename Elem<T>::acc_t;
    __shared__ A red[WAVES][kWave][VEC];
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const int sub = lane / LPR;
    const int sl = lane % LPR;
    const int64_t r = long_rows[blockIdx.x];
    const int64_t e0 = rowptr[r];
    const int64_t e1 = rowptr[r + 1];
    const int64_t seg = ceil_div(ceil_div(e1 - e0, WAVES), kWave) * kWave;
    const int active = e1 > e0 ? int(ceil_div(e1 - e0, seg)) : 1;  // waves with a non-empty slice (wave 0 always stays)
    if (idle_wave_may_exit(wave < active)) return;  // before the barriers below: device_utils.h
    const int64_t a = e0 + wave * seg;
    const int64_t b = a + seg < e1 ? a + seg : e1;
    const EdgeChunk<A> none{0, A(0)};
    A rsum = A(0);

    for (int t0 = 0; t0 < d; t0 += LPR * VEC) {
        const int c0 = t0 + sl * VEC;
        const bool col_ok = c0 < d;
        const bool writer = wave == 0 && col_ok && sub == 0;
        Pack<T, VEC> x{}, zo{};
        if (writer) {  // requested before the gathers, consumed after the fold
            x = load_pack_stream<T, VEC>(X + r * ldx + c0);
            zo = load_pack_stream<T, VEC>(Zold + (row0 + r) * ldz + c0);
        }
        A acc[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] = A(0);
        gather_accumulate<T, PT, VEC, LPR, U>(colidx, P, a, b, Zold + (col_ok ? c0 : 0), ldz, col_ok, acc, none, false);
        fold_subwaves<LPR>(acc);
        if (wave > 0) {
#pragma unroll
            for (int k = 0; k < VEC; ++k) red[wave][lane][k] = acc[k];
        }
        __syncthreads();
        if (wave == 0) {
            Pack<T, VEC> out{};
            if (writer) {
                for (int w = 1; w < active; ++w) {
#pragma unroll
                    for (int k = 0; k < VEC; ++k) acc[k] += red[w][lane][k];
                }
                rsum += finish_pack<T, VEC>(x, zo, acc, gamma, e1 > e0, Znew + r * ldo + c0, out, false);
            }
            mirror_store<T, VEC, kWave>(mirror, r, c0, out, writer, lane, 0);
        }
        if (t0 + LPR * VEC < d) __syncthreads();  // red[] is reused by the next column tile
    }
    if (wave == 0) {
        rsum = group_sum<kWave>(rsum);
        if (lane == 0) partials[blockIdx.x] = double(rsum);
    }
}

}  // namespace clane

namespace clane {

// ---- hub rows: one row over SEVERAL workgroups ------------------------------------------------------
// A 70k-edge row done by one 16-wave workgroup is 4.4k edges per wave in sequence -- a ~0.25 ms tail
// on every launch, which dominates once a sweep is cut into chunks (multi-GPU) or the graph is small.
// Rows above `edges_per_segment` are cut into segments; workgroup s gathers segment s exactly like
// spmm_long_kernel and leaves its partial sum (accumulate type) in slab[s]; spmm_class_combine_kernel
// then adds a row's segments in a fixed order (reproducible) and runs the usual epilogue.
template <typename T, typename PT, int VEC, int LPR, int U, int WAVES>
__global__ __launch_bounds__(WAVES *kWave) void spmm_split_segment_kernel(
    const int64_t *__restrict__ rowptr, const int32_t *__restrict__ colidx, const PT *__restrict__ P,
    const int32_t *__restrict__ split_rows, const int64_t *__restrict__ seg_ptr, const int32_t *__restrict__ seg_row,
    int64_t edges_per_segment, const T *__restrict__ Zold, int64_t ldz, int d,
    typename Elem<T>::acc_t *__restrict__ slab, int64_t ld_slab) {
    using A = typename Elem<T>::acc_t;
    __shared__ A red[WAVES][kWave][VEC];
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const int sub = lane / LPR;
    const int sl = lane % LPR;
    const int64_t s = blockIdx.x;
    const int i = seg_row[s];
    const int64_t r = split_rows[i];
    const int64_t e0 = rowptr[r] + (s - seg_ptr[i]) * edges_per_segment;
    const int64_t e1 = e0 + edges_per_segment < rowptr[r + 1] ? e0 + edges_per_segment : rowptr[r + 1];
    const int64_t seg = ceil_div(ceil_div(e1 - e0, WAVES), kWave) * kWave;
    const int active = e1 > e0 ? int(ceil_div(e1 - e0, seg)) : 1;
    if (idle_wave_may_exit(wave < active)) return;  // before the barriers below: device_utils.h
    const int64_t a = e0 + wave * seg;
    const int64_t b = a + seg < e1 ? a + seg : e1;
    const EdgeChunk<A> none{0, A(0)};
    for (int t0 = 0; t0 < d; t0 += LPR * VEC) {
        const int c0 = t0 + sl * VEC;
        const bool col_ok = c0 < d;
        A acc[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] = A(0);
        gather_accumulate<T, PT, VEC, LPR, U>(colidx, P, a, b, Zold + (col_ok ? c0 : 0), ldz, col_ok, acc, none, false);
        fold_subwaves<LPR>(acc);
        if (wave > 0) {
#pragma unroll
            for (int k = 0; k < VEC; ++k) red[wave][lane][k] = acc[k];
        }
        __syncthreads();
        if (wave == 0 && col_ok && sub == 0) {
            for (int w = 1; w < active; ++w) {
#pragma unroll
                for (int k = 0; k < VEC; ++k) acc[k] += red[w][lane][k];
            }
            Pack<A, VEC> o;                    // whole packs (the pad columns of Z are zero, so are their sums):
#pragma unroll
            for (int k = 0; k < VEC; ++k) o.v[k] = acc[k];   // the combine reads packs
            store_pack<A, VEC>(slab + s * ld_slab + c0, o);
        }
        if (t0 + LPR * VEC < d) __syncthreads();
    }
}

// The segments of a row are then added IN ORDER and the usual epilogue runs: spmm_class_combine_kernel below (a split
// row's segments are laid out in the slab exactly like a class row's slots).
}  // namespace clane

namespace clane {

// ---- class-affine rows: every gathered row is read through ONE XCD's L2 --------------------------------------
// The 256 CUs sit in 8 XCDs with a private 4 MiB L2 each, and workgroups are dealt to the XCDs round-robin
// (workgroup w runs on XCD (w + c) % 8, c fixed within a launch).  When any workgroup may gather any row, all eight L2s end up caching the SAME
// few thousand hottest rows.  Here every table row has a class 0..7 (host side: clane_amd/xcd.py), the edges of
// a long row are sorted by (class of the column, column) and cut into CHUNKS of at most
// a few hundred edges of one class, and the chunks of class b are dealt to the workgroups w = 8 j + b: XCD b only
// ever gathers rows of class b, so the eight L2s cache eight DIFFERENT eighths of the hot rows (Z is laid out
// hottest rows first: every class gets its share of the heat).  Measured (profiles/r02_gather_rows_ceiling.md,
// r02_class_threshold_sweep.md): pure 1-KiB-row gathers 9.1 -> 11.5..14.3 TB/s, 256-byte rows 11.3 -> 19.5 TB/s;
// config 3's sweep 5.36 -> 4.38 ms.
// A chunk is done by one wave -- its partial sum goes to slab[slot] -- and spmm_class_combine_kernel then adds a
// row's slots IN ORDER (reproducible: no atomics) and runs the usual epilogue.  A workgroup owns
// `items_per_block` consecutive chunks of its class (padding chunks have length 0); their descriptors are staged
// in LDS, waves claim them from an LDS counter and request the next chunk's colidx / P before gathering the
// current one, exactly like spmm_update_kernel does with rows.

// The slab is written once by the chunk kernel and read once by the combine, a whole launch later: its lines only take
// L2 / Infinity-Cache space from the rows being gathered.  CLANE_NT_SLAB bit 0 = non-temporal stores (default: config 3's
// class pass 2.19 -> 2.14 ms, config 2 unchanged), bit 1 = non-temporal loads in the combine (nothing: 3.714 -> 3.701 with
// the loads alone, 3.678 / 3.665 with both against 3.663 with the stores alone; profiles/r05_nt_streams.md).
#ifndef CLANE_NT_SLAB
#define CLANE_NT_SLAB 1
#endif
template <typename A, int VEC>
__device__ __forceinline__ void slab_store(A *p, const Pack<A, VEC> &v) {
    if constexpr ((CLANE_NT_SLAB & 1) && sizeof(Pack<A, VEC>) == 16) {
        clane_u32x4 w;
        __builtin_memcpy(&w, &v, 16);
        __builtin_nontemporal_store(w, reinterpret_cast<clane_u32x4 *>(p));
    } else {
        store_pack<A, VEC>(p, v);
    }
}
template <typename A, int VEC>
__device__ __forceinline__ Pack<A, VEC> slab_load(const A *p) {
    if constexpr ((CLANE_NT_SLAB & 2) && sizeof(Pack<A, VEC>) == 16) {
        const clane_u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const clane_u32x4 *>(p));
        Pack<A, VEC> out;
        __builtin_memcpy(&out, &v, 16);
        return out;
    } else {
        return load_pack<A, VEC>(p);
    }
}

template <typename T, typename PT, int VEC, int LPR, int U>
__global__ __launch_bounds__(kBlock) void spmm_class_chunk_kernel(
    const int32_t *__restrict__ colidx, const PT *__restrict__ P, const int64_t *__restrict__ item_e0,
    const int32_t *__restrict__ item_len, const int32_t *__restrict__ item_slot, int items_per_block,
    const T *__restrict__ Zold, int64_t ldz, int d, typename Elem<T>::acc_t *__restrict__ slab, int64_t ld_slab) {
    using A = typename Elem<T>::acc_t;
    __shared__ int64_t s_e0[kMaxItemsPerBlock];
    __shared__ int s_len[kMaxItemsPerBlock];
    __shared__ int s_slot[kMaxItemsPerBlock];
    __shared__ int s_next;
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const int sub = lane / LPR;
    const int sl = lane % LPR;
    const int64_t base = int64_t(blockIdx.x) * items_per_block;
    for (int i = threadIdx.x; i < items_per_block; i += kBlock) {
        s_e0[i] = item_e0[base + i];
        s_len[i] = item_len[base + i];
        s_slot[i] = item_slot[base + i];
    }
    if (threadIdx.x == 0) s_next = kWavesPerBlock;
    __syncthreads();

    auto claim = [&]() -> int {
        int v = 0;
        if (lane == 0) v = atomicAdd(&s_next, 1);
        return __builtin_amdgcn_readfirstlane(v);
    };
    int cur = wave;
    int64_t e0 = 0, e1 = 0;
    EdgeChunk<A> ch{0, A(0)};
    if (cur < items_per_block) {
        e0 = s_e0[cur];
        e1 = e0 + s_len[cur];
        ch = load_chunk<A, PT>(colidx, P, e0, e1);
    }
    while (cur < items_per_block) {
        const int nxt = claim();
        int64_t n0 = 0, n1 = 0;
        EdgeChunk<A> chn{0, A(0)};
        if (nxt < items_per_block) {
            n0 = s_e0[nxt];
            n1 = n0 + s_len[nxt];
            chn = load_chunk<A, PT>(colidx, P, n0, n1);
        }
        if (e1 > e0) {
            A *out = slab + int64_t(s_slot[cur]) * ld_slab;
            for (int t0 = 0; t0 < d; t0 += LPR * VEC) {
                const int c0 = t0 + sl * VEC;
                const bool col_ok = c0 < d;
                A acc[VEC];
#pragma unroll
                for (int k = 0; k < VEC; ++k) acc[k] = A(0);
                gather_accumulate<T, PT, VEC, LPR, U>(colidx, P, e0, e1, Zold + (col_ok ? c0 : 0), ldz, col_ok, acc, ch, true);
                fold_subwaves<LPR>(acc);
                if (col_ok && sub == 0) {
                    Pack<A, VEC> o;
#pragma unroll
                    for (int k = 0; k < VEC; ++k) o.v[k] = acc[k];
                    slab_store<A, VEC>(out + c0, o);
                }
            }
        }
        cur = nxt;
        e0 = n0;
        e1 = n1;
        ch = chn;
    }
}

// One workgroup of kCombineWaves waves per class row.  Wave w adds the w-th contiguous share of the row's slots IN ORDER;
// wave 0 then adds the waves' sums in wave order (through LDS) and runs the usual epilogue:  z = x + gamma * sum,
// delta, store.  The association is fixed by (slot count, kCombineWaves) alone -- reproducible, no atomics -- and a
// heavy row's chain of dependent loads is a kCombineWaves-th of what one wave would walk (config 3's heaviest row:
// 280 slots; the 2.4e9-edge test's hubs: 4 688).
constexpr int kCombineWaves = CLANE_COMBINE_WAVES;
#ifndef CLANE_COMBINE_LOADS
#define CLANE_COMBINE_LOADS 16    // slot loads in flight per wave of the combine (rows of fewer slots take the plain loop)
#endif
constexpr int kCombineLoads = CLANE_COMBINE_LOADS;

template <typename T, int VEC, int LPR>
__global__ __launch_bounds__(kCombineWaves *kWave) void spmm_class_combine_kernel(
    const int32_t *__restrict__ class_rows, const int64_t *__restrict__ slot_ptr, int64_t n_rows, int64_t row0,
    const typename Elem<T>::acc_t *__restrict__ slab, int64_t ld_slab, const T *__restrict__ Zold, int64_t ldz,
    const T *__restrict__ X, int64_t ldx, typename Elem<T>::acc_t gamma, T *__restrict__ Znew, int64_t ldo, int d,
    Mirror<T> mirror, bool nt_store, double *__restrict__ partials) {
    using A = typename Elem<T>::acc_t;
    // A row needs LPR lanes; a workgroup takes 64 / LPR rows, one per lane group (round 5: at 512-byte rows -- the column
    // tiles -- half the lanes of the one-row form had nothing to do).  Each group walks its own row's slots: the
    // association of a row's sum depends on its slot count and kCombineWaves alone, as before.
    constexpr int kRows = kWave / LPR;
    __shared__ A s_part[kCombineWaves > 1 ? kCombineWaves - 1 : 1][kWave][VEC];
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const int sub = lane / LPR, sl = lane % LPR;
    const int64_t i = int64_t(blockIdx.x) * kRows + sub;
    const bool live = i < n_rows;
    const int64_t r = live ? class_rows[i] : 0;
    const int64_t s0 = live ? slot_ptr[i] : 0, s1 = live ? slot_ptr[i + 1] : 0;
    const int64_t share = ceil_div(s1 - s0, int64_t(kCombineWaves));
    const int64_t a = s0 + wave * share < s1 ? s0 + wave * share : s1;
    const int64_t b = a + share < s1 ? a + share : s1;
    A rsum = A(0);
    for (int t0 = 0; t0 < d; t0 += LPR * VEC) {         // the same trip count in every wave (barriers inside)
        const int c0 = t0 + sl * VEC;
        const bool ok = live && c0 < d;
        A acc[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] = A(0);
        if (ok) {
            // kCombineLoads slots requested back to back, then added in slot order: a mega-hub's share is thousands of
            // slots (a 2M-edge row: 3 920 per wave), and with 4 loads in flight a wave streamed them at 8 GB/s
            int64_t s = a;
            for (; s + kCombineLoads <= b; s += kCombineLoads) {
                Pack<A, VEC> part[kCombineLoads];
#pragma unroll
                for (int u = 0; u < kCombineLoads; ++u) part[u] = slab_load<A, VEC>(slab + (s + u) * ld_slab + c0);
#pragma unroll
                for (int u = 0; u < kCombineLoads; ++u) {
#pragma unroll
                    for (int k = 0; k < VEC; ++k) acc[k] += part[u].v[k];
                }
            }
#pragma unroll 4
            for (; s < b; ++s) {
                const Pack<A, VEC> part = slab_load<A, VEC>(slab + s * ld_slab + c0);
#pragma unroll
                for (int k = 0; k < VEC; ++k) acc[k] += part.v[k];
            }
        }
        if constexpr (kCombineWaves > 1) {
            if (t0 > 0) __syncthreads();                 // wave 0 has read the previous pass's sums
            if (wave > 0) {
#pragma unroll
                for (int k = 0; k < VEC; ++k) s_part[wave - 1][lane][k] = acc[k];
            }
            __syncthreads();
        }
        if (wave == 0) {
            Pack<T, VEC> out{};
            if (ok) {
                if constexpr (kCombineWaves > 1) {
#pragma unroll
                    for (int w = 1; w < kCombineWaves; ++w) {
#pragma unroll
                        for (int k = 0; k < VEC; ++k) acc[k] += s_part[w - 1][lane][k];
                    }
                }
                const Pack<T, VEC> x = load_pack_stream<T, VEC>(X + r * ldx + c0);
                const Pack<T, VEC> zo = load_pack_stream<T, VEC>(Zold + (row0 + r) * ldz + c0);
                rsum += finish_pack<T, VEC>(x, zo, acc, gamma, true, Znew + r * ldo + c0, out, nt_store);
            }
            if (live) mirror_store<T, VEC, LPR>(mirror, r, c0, out, ok, sl, sub * LPR);     // uniform over the row's lanes
        }
    }
    if (wave == 0) {
        rsum = group_sum<LPR>(rsum);
        if (live && sl == 0) partials[i] = double(rsum);
    }
}

}  // namespace clane

