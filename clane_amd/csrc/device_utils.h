// Shared device helpers for the CLANE gfx950 kernels: element traits (storage type -> accumulate
// type, 16-byte packs), wave64 broadcast / reduction primitives, and the block-level
// fixed-order reduction used for every value that feeds host control flow.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace clane {

constexpr int kWave = 64;          // gfx950 wavefront
constexpr int kBlock = 256;        // 4 waves per workgroup
constexpr int kWavesPerBlock = kBlock / kWave;
constexpr int kMaxRowsPerBlock = 256;   // rows a workgroup of the row kernels stages in LDS (rowptr slice, per-row deltas)
constexpr int kMaxGrid = 256 * 8;  // 256 CUs x 8 resident 256-thread workgroups
constexpr int kMaxItemsPerBlock = 64;  // class-affine passes: chunk descriptors a workgroup stages in LDS

struct bf16_t {
    uint16_t bits;
};

template <typename T>
struct Elem;
template <>
struct Elem<float> {
    using acc_t = float;
    static constexpr int kVec = 4;
    static __device__ __forceinline__ float to_acc(float v) { return v; }
    static __device__ __forceinline__ float from_acc(float v) { return v; }
};
template <>
struct Elem<double> {
    using acc_t = double;
    static constexpr int kVec = 2;
    static __device__ __forceinline__ double to_acc(double v) { return v; }
    static __device__ __forceinline__ double from_acc(double v) { return v; }
};
template <>
struct Elem<bf16_t> {
    using acc_t = float;
    static constexpr int kVec = 8;
    static __device__ __forceinline__ float to_acc(bf16_t v) { return __uint_as_float(uint32_t(v.bits) << 16); }
    static __device__ __forceinline__ bf16_t from_acc(float f) {
        // round-to-nearest-even; NaN stays NaN (quiet bit forced)
        uint32_t u = __float_as_uint(f);
        if ((u & 0x7fffffffu) > 0x7f800000u) return bf16_t{uint16_t((u >> 16) | 0x0040u)};
        return bf16_t{uint16_t((u + 0x7fffu + ((u >> 16) & 1u)) >> 16)};
    }
};

// VEC elements moved as one access (16 B when VEC == Elem<T>::kVec, sizeof(T) when VEC == 1).
template <typename T, int VEC>
struct alignas(sizeof(T) * VEC) Pack {
    T v[VEC];
};

template <typename T, int VEC>
__device__ __forceinline__ Pack<T, VEC> load_pack(const T *p) {
    return *reinterpret_cast<const Pack<T, VEC> *>(p);
}
template <typename T, int VEC>
__device__ __forceinline__ void store_pack(T *p, const Pack<T, VEC> &v) {
    *reinterpret_cast<Pack<T, VEC> *>(p) = v;
}

// Data touched once per sweep (a row's own x / z_old, the z_new it writes) can be marked non-temporal so that it does
// not push gathered rows out of the L2s / the Infinity Cache.  Measured with round 5's kernels
// (profiles/r05_nt_streams.md): non-temporal LOADS gain everywhere (config 3 -0.7 %, config 4 -1.8 %, config 2 -1 %):
// on by default (bit 0).  Non-temporal STORES of z_new gain on tables far beyond the caches (config 3 another -0.8 %,
// the 16M-vertex run -1 %) and LOSE on a cache-resident table (config 2 +4 %: the next sweep gathers what this one
// wrote): a run-time choice of the caller (`nt`: CLANE_SPMM_TABLE_BEYOND_CACHE), bit 1 forces them.
#ifndef CLANE_NT_STREAM
#define CLANE_NT_STREAM 1          // bit 0: streamed loads non-temporal, bit 1: streamed stores always non-temporal
#endif
typedef uint32_t clane_u32x4 __attribute__((ext_vector_type(4)));
template <typename T, int VEC>
__device__ __forceinline__ Pack<T, VEC> load_pack_stream(const T *p) {
    if constexpr ((CLANE_NT_STREAM & 1) && sizeof(Pack<T, VEC>) == 16) {
        const clane_u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const clane_u32x4 *>(p));
        Pack<T, VEC> out;
        __builtin_memcpy(&out, &v, 16);
        return out;
    } else {
        return load_pack<T, VEC>(p);
    }
}
template <typename T, int VEC>
__device__ __forceinline__ void store_pack_stream(T *p, const Pack<T, VEC> &v, bool nt) {
    if constexpr (sizeof(Pack<T, VEC>) == 16) {
        if ((CLANE_NT_STREAM & 2) || nt) {           // `nt`: a kernel argument, uniform
            clane_u32x4 w;
            __builtin_memcpy(&w, &v, 16);
            __builtin_nontemporal_store(w, reinterpret_cast<clane_u32x4 *>(p));
            return;
        }
    }
    store_pack<T, VEC>(p, v);
}

// ---- wave64 cross-lane -------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return threadIdx.x & (kWave - 1); }

// Value of lane `src` (any per-lane index) -- ds_bpermute.
__device__ __forceinline__ int lane_get(int v, int src) { return __shfl(v, src, kWave); }
__device__ __forceinline__ float lane_get(float v, int src) { return __shfl(v, src, kWave); }
__device__ __forceinline__ double lane_get(double v, int src) { return __shfl(v, src, kWave); }

// Value of lane `src` where `src` is wave-uniform -- v_readlane_b32 into an SGPR, so the
// dependent row address is formed on the scalar unit.
__device__ __forceinline__ int lane_get_uniform(int v, int src) { return __builtin_amdgcn_readlane(v, src); }
__device__ __forceinline__ float lane_get_uniform(float v, int src) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}
__device__ __forceinline__ double lane_get_uniform(double v, int src) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane(int(b), src);
    const int hi = __builtin_amdgcn_readlane(int(b >> 32), src);
    return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo));
}

// Value of lane (lane ^ M), M a compile-time power of two -- WITHOUT the LDS crossbar: DPP moves inside a row of 16
// lanes (quad_perm for 1 and 2; row_half_mirror + quad reversal for 4; row_ror:8 for 8), gfx950's
// v_permlane16_swap / v_permlane32_swap across rows (16, 32).  A ds_bpermute (what __shfl_xor compiles to) is an
// LDS-pipe round trip of ~100 cycles; the butterflies below are chains of 3..6 of them, sitting between a group's
// gathers and the next group's in K1 and inside the sub-wave service blocks.  Pure data movement: results are
// bit-identical to the __shfl_xor form.  Partner lanes must be active (they are: every caller reduces over groups
// of lanes that execute together).  All six verified on the card against lane ^ M (tools/lane_xor_check.hip).
#ifndef CLANE_XOR_DPP
#define CLANE_XOR_DPP 1
#endif
template <int CTRL>
__device__ __forceinline__ int dpp_mov(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false);
}
template <int M>
__device__ __forceinline__ int lane_xor(int v) {
    static_assert(M == 1 || M == 2 || M == 4 || M == 8 || M == 16 || M == 32, "lane_xor: M must be a power of two below 64");
#if CLANE_XOR_DPP
    if constexpr (M == 1) return dpp_mov<0xB1>(v);                       // quad_perm [1,0,3,2]
    else if constexpr (M == 2) return dpp_mov<0x4E>(v);                  // quad_perm [2,3,0,1]
    else if constexpr (M == 4) return dpp_mov<0x1B>(dpp_mov<0x141>(v));  // row_half_mirror (i -> 7 - i), then quad_perm [3,2,1,0]
    else if constexpr (M == 8) return dpp_mov<0x128>(v);                 // row_ror:8
    else if constexpr (M == 16) {   // {rows 0 and 2 of a with rows 0 and 2 of b, ...}: r[0] = (a0, b0, a2, b2), r[1] = (a1, b1, a3, b3)
        const auto r = __builtin_amdgcn_permlane16_swap(unsigned(v), unsigned(v), false, false);
        return int((lane_id() & 16) ? r[0] : r[1]);
    } else {
        const auto r = __builtin_amdgcn_permlane32_swap(unsigned(v), unsigned(v), false, false);
        return int((lane_id() & 32) ? r[0] : r[1]);
    }
#else
    return __shfl_xor(v, M, kWave);
#endif
}
template <int M>
__device__ __forceinline__ float lane_xor(float v) {
    return __int_as_float(lane_xor<M>(__float_as_int(v)));
}
template <int M>
__device__ __forceinline__ double lane_xor(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = lane_xor<M>(int(b));
    const int hi = lane_xor<M>(int(b >> 32));
    return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo));
}

// Butterfly sum over aligned groups of WIDTH lanes (partners lane ^ WIDTH/2, ..., lane ^ 1, in that order); every
// lane of the group gets the total, bit for bit the same one.
template <int WIDTH, typename A>
__device__ __forceinline__ A group_sum(A v) {
    if constexpr (WIDTH >= 64) v += lane_xor<32>(v);
    if constexpr (WIDTH >= 32) v += lane_xor<16>(v);
    if constexpr (WIDTH >= 16) v += lane_xor<8>(v);
    if constexpr (WIDTH >= 8) v += lane_xor<4>(v);
    if constexpr (WIDTH >= 4) v += lane_xor<2>(v);
    if constexpr (WIDTH >= 2) v += lane_xor<1>(v);
    return v;
}
template <int M, typename A>
__device__ __forceinline__ A max_step(A v) {
    const A o = lane_xor<M>(v);
    return o > v ? o : v;
}
template <int WIDTH, typename A>
__device__ __forceinline__ A group_max(A v) {
    if constexpr (WIDTH >= 64) v = max_step<32>(v);
    if constexpr (WIDTH >= 32) v = max_step<16>(v);
    if constexpr (WIDTH >= 16) v = max_step<8>(v);
    if constexpr (WIDTH >= 8) v = max_step<4>(v);
    if constexpr (WIDTH >= 4) v = max_step<2>(v);
    if constexpr (WIDTH >= 2) v = max_step<1>(v);
    return v;
}

// Sum of `v` over the workgroup in a fixed order (lane butterfly, then waves 0..n-1 in turn);
// valid in thread 0.  `smem` holds one double per wave.
__device__ __forceinline__ double block_sum_fixed(double v, double *smem) {
    v = group_sum<kWave>(v);
    const int wave = threadIdx.x / kWave;
    if (lane_id() == 0) smem[wave] = v;
    __syncthreads();
    double s = 0.0;
    if (threadIdx.x == 0) {
        const int nw = blockDim.x / kWave;
        for (int w = 0; w < nw; ++w) s += smem[w];
    }
    __syncthreads();
    return s;
}

// This library is written for gfx950 (MI355X) only.  Two things below lean on the gfx9 ISA rather than on the HIP
// programming model, so any other target must fail to compile instead of hanging at run time:
//  * kernels that give one workgroup to one row let waves WITHOUT work terminate before a later s_barrier
//    (`idle_wave_may_exit`): on gfx9 "s_barrier" counts only the waves of the workgroup that have not ended
//    (CDNA3/4 ISA guide, s_barrier / s_endpgm), so a 40-edge row costs one working wave, not 16 parked ones
//    holding their registers;
//  * a wave executes in lock step, so one release/acquire by lane 0 publishes / observes the LDS accesses of all
//    64 lanes once the wave has waited for its own LDS counter (`last_wave_of_block`).
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "clane_amd's kernels are written for gfx950 only (wave64, gfx9 s_barrier semantics): compile with --offload-arch=gfx950"
#endif

// True for the waves of a one-workgroup-per-row kernel that have nothing to do and may end before the
// workgroup's barriers.  `has_work` must be wave-uniform.
__device__ __forceinline__ bool idle_wave_may_exit(bool has_work) { return !has_work; }

// Ticket taken by every wave of a kBlock-thread workgroup when it has finished its rows; true in the LAST wave to
// arrive, which may then read what all the others wrote to LDS.  All 64 lanes first wait for their own LDS stores
// (release fence: s_waitcnt lgkmcnt(0) on gfx9), lane 0 takes the ticket with acquire-release ordering at
// workgroup scope, and the reads of the last wave are kept behind it by an acquire fence.
__device__ __forceinline__ bool last_wave_of_block(int *s_done) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    int ticket = 0;
    if (lane_id() == 0) ticket = __hip_atomic_fetch_add(s_done, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
    ticket = __builtin_amdgcn_readfirstlane(ticket);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    return ticket == kWavesPerBlock - 1;
}

// XCD (accelerator complex die) the calling wave runs on: HW_REG_XCC_ID[3:0].  The class-affine kernels do not
// read it -- they rely on workgroup w of a launch running on XCD (w + c) % 8, c fixed for the launch -- but tests do, so that the assumption is
// checked on every box (clane_xcc_ids).
__device__ __forceinline__ int xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 0xf; }

template <typename A>
__device__ __forceinline__ A exp_acc(A v);
template <>
__device__ __forceinline__ float exp_acc<float>(float v) {
    return expf(v);
}
template <>
__device__ __forceinline__ double exp_acc<double>(double v) {
    return exp(v);
}

__host__ __device__ __forceinline__ int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace clane
