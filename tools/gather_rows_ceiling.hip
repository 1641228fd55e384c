// Micro-benchmark (not part of the product): the practical ceiling on this MI355X for gathering whole rows of
// 128 / 256 / 512 / 1024 bytes from a 2M-row table with the R-MAT read skew of config 3, rows laid out hottest
// first (the engine's layout) -- i.e. what one rank's column slice of the N-GPU split can reach at
// best -- and what XCD-AFFINE reads would add: workgroup i runs on XCD i % 8 (round-robin dispatch), so if it only
// reads rows r with r % 8 == i % 8, each XCD's 4 MiB L2 caches a DIFFERENT eighth of the hot rows (32 MiB of
// distinct hot rows on the chip instead of 8 copies of the same 4 MiB).
// Also in one run: three ways of forming the classes (row % 8 pins address bits and wastes L2), the cache-policy bits
// of the cold rows' loads, and reads that are XCD-affine in space AND phased in time.  Results: profiles/
// r02_gather_rows_ceiling.md.
//   hipcc --offload-arch=gfx950 -O3 -o build/gather_rows_ceiling tools/gather_rows_ceiling.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>
#include <vector>

#define CK(x)                                                     \
    do {                                                          \
        hipError_t e = (x);                                       \
        if (e != hipSuccess) {                                    \
            printf("%s: %s\n", #x, hipGetErrorString(e));         \
            exit(1);                                              \
        }                                                         \
    } while (0)

// LPR lanes cover one row with 16 B each; a wave reads 64/LPR rows per instruction, U instructions in flight.
// Work is cut into groups of 64 indices; group g belongs to "bucket" g % nb, and workgroup w only takes groups of
// bucket w % nb (nb = 1: no affinity; nb = 8 with bucketed index lists: XCD-affine).
typedef float nt_f4 __attribute__((ext_vector_type(4)));
// nt_from >= 0: a group whose FIRST index is >= nt_from is read with non-temporal loads (cold rows should not push
// the hot ones out of L2); groups are pure (all hot or all cold) in the lists used with it.
template <int LPR, int U>
__global__ __launch_bounds__(256) void gather(const float4 *__restrict__ Z, const int *__restrict__ idx,
                                              long n_groups, int nb, float4 *__restrict__ out, int nt_from = -1, int policy = 1) {
    constexpr int EPW = 64 / LPR;
    const int lane = threadIdx.x & 63, sub = lane / LPR, sl = lane % LPR;
    const long wave_in_bucket = (long)(blockIdx.x / nb) * 4 + (threadIdx.x >> 6);
    const long waves_per_bucket = (long)(gridDim.x / nb) * 4;
    const int bucket = blockIdx.x % nb;
    float4 acc = {0, 0, 0, 0};
    for (long k = wave_in_bucket; k * nb + bucket < n_groups; k += waves_per_bucket) {
        const long g = k * nb + bucket;
        const int my = idx[g * 64 + lane];
        const bool cold = nt_from >= 0 && __builtin_amdgcn_readfirstlane(my) >= nt_from;
        for (int j = 0; j < 64; j += EPW * U) {
            float4 z[U];
            if (cold) {
                const float4 *ptr[U];
#pragma unroll
                for (int u = 0; u < U; ++u) ptr[u] = Z + (long)__shfl(my, j + u * EPW + sub, 64) * LPR + sl;
                // cache policy of the cold gathers: the bits are instruction modifiers, hence inline asm (the compiler
                // does not count these loads: explicit s_waitcnt below)
#define CLANE_COLD_LOADS(MOD)                                                                              \
    _Pragma("unroll") for (int u = 0; u < U; ++u) asm volatile("global_load_dwordx4 %0, %1, off " MOD     \
                                                               : "=v"(z[u])                               \
                                                               : "v"(ptr[u])                              \
                                                               : "memory");
                if (policy == 1) { CLANE_COLD_LOADS("nt") }
                else if (policy == 2) { CLANE_COLD_LOADS("sc0") }
                else if (policy == 3) { CLANE_COLD_LOADS("sc1") }
                else if (policy == 4) { CLANE_COLD_LOADS("sc0 sc1") }
                else if (policy == 5) { CLANE_COLD_LOADS("sc1 nt") }
                else { CLANE_COLD_LOADS("sc0 sc1 nt") }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int c = __shfl(my, j + u * EPW + sub, 64);
                    z[u] = Z[(long)c * LPR + sl];
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                acc.x += z[u].x;
                acc.y += z[u].y;
                acc.z += z[u].z;
                acc.w += z[u].w;
            }
        }
    }
    if (acc.x == 1.2345f) out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 64 + lane] = acc;
}

template <int LPR>
float run(const float4 *Z, const int *idx, long n_groups, int nb, float4 *out, int grid, int nt_from = -1, int policy = 1) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    float best = 1e9;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(a));
        gather<LPR, 8><<<grid, 256>>>(Z, idx, n_groups, nb, out, nt_from, policy);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    return best;
}

static int g_class_mode = 0;
static inline int cls_of(int r) {
    if (g_class_mode == 0) return r & 7;                                   // row % 8
    if (g_class_mode == 1) return (r ^ (r >> 3) ^ (r >> 6) ^ (r >> 9)) & 7;  // xor-folded: every class holds every value of the low bits
    return (r >> 3) & 7;                                                   // blocks of 8 consecutive rows
}

int run_all();
int main() {
    for (g_class_mode = 0; g_class_mode < 3; ++g_class_mode) {
        printf("== class of a row: %s\n", g_class_mode == 0 ? "r % 8" : g_class_mode == 1 ? "xor-fold(r) % 8" : "(r / 8) % 8");
        run_all();
    }
    return 0;
}

int run_all() {
    const long V = 2000000;
    const long n_idx = 40000000 / 512 * 512;
    std::vector<int> h(n_idx);
    std::mt19937_64 rng(1);
    for (long i = 0; i < n_idx; ++i) {  // R-MAT-like skew (a + c = 0.76 per destination bit)
        unsigned v = 0;
        for (int l = 0; l < 21; ++l) v = (v << 1) | ((rng() & 0xffff) >= 0.76 * 65536);
        h[i] = v % V;
    }
    // hottest rows first (the engine's layout)
    std::vector<int> cnt(V, 0), order(V), rank_of(V);
    for (long i = 0; i < n_idx; ++i) cnt[h[i]]++;
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return cnt[a] > cnt[b]; });
    for (int r = 0; r < V; ++r) rank_of[order[r]] = r;
    for (long i = 0; i < n_idx; ++i) h[i] = rank_of[h[i]];
    // bucketed copy: group g (64 indices) holds only rows with row % 8 == g % 8 (equal-sized buckets by construction
    // of the tail: leftovers are dropped from both lists so that both variants read the same multiset)
    std::vector<std::vector<int>> by(8);
    for (long i = 0; i < n_idx; ++i) by[cls_of(h[i])].push_back(h[i]);
    size_t per = by[0].size();
    for (auto &b : by) per = std::min(per, b.size());
    per = per / 64 * 64;
    std::vector<int> plain, bucketed(per * 8);
    plain.reserve(per * 8);
    for (int b = 0; b < 8; ++b) plain.insert(plain.end(), by[b].begin(), by[b].begin() + per);
    std::shuffle(plain.begin(), plain.end(), rng);  // same multiset, no affinity
    for (size_t g = 0; g < per / 64; ++g)
        for (int b = 0; b < 8; ++b)
            std::copy(by[b].begin() + g * 64, by[b].begin() + (g + 1) * 64, bucketed.begin() + (g * 8 + b) * 64);
    const long n = (long)per * 8, n_groups = n / 64;
    float4 *Z, *out;
    int *idx;
    CK(hipMalloc(&Z, V * 1024));
    CK(hipMemset(Z, 0, V * 1024));
    CK(hipMalloc(&idx, n * sizeof(int)));
    CK(hipMalloc(&out, 1 << 26));
    printf("%ld row reads, hottest-first layout; share of reads to the top 4k / 32k / 250k rows: ", n);
    {
        long c4 = 0, c32 = 0, c250 = 0;
        for (long i = 0; i < n; ++i) {
            c4 += plain[i] < 4096;
            c32 += plain[i] < 32768;
            c250 += plain[i] < 250000;
        }
        printf("%.3f %.3f %.3f\n", double(c4) / n, double(c32) / n, double(c250) / n);
    }
    // XCD-affine with PURE groups: per class, the hot indices (row < H) first, then the cold ones, each kind padded to
    // whole groups by repeating its last index (a few extra reads, counted); groups of the classes interleaved as above.
    const int H = 32768;
    std::vector<int> pure;
    {
        std::vector<std::vector<int>> hot(8), cold(8);
        for (int b = 0; b < 8; ++b)
            for (size_t i = 0; i < per; ++i) (by[b][i] < H ? hot[b] : cold[b]).push_back(by[b][i]);
        size_t gh = 0, gc = 0;
        for (int b = 0; b < 8; ++b) {
            gh = std::max(gh, (hot[b].size() + 63) / 64);
            gc = std::max(gc, (cold[b].size() + 63) / 64);
        }
        for (int b = 0; b < 8; ++b) {
            hot[b].resize(gh * 64, hot[b].back());
            cold[b].resize(gc * 64, cold[b].back());
            std::shuffle(cold[b].begin(), cold[b].end(), rng);
            std::shuffle(hot[b].begin(), hot[b].end(), rng);
        }
        // interleave hot and cold groups in time (as the rows of a sweep would), class-affine in space
        pure.resize((gh + gc) * 8 * 64);
        size_t ih = 0, ic = 0, g = 0;
        while (ih < gh || ic < gc) {
            const bool take_hot = ic >= gc || (ih < gh && ih * gc <= ic * gh);
            for (int b = 0; b < 8; ++b) {
                const std::vector<int> &src = take_hot ? hot[b] : cold[b];
                const size_t k = take_hot ? ih : ic;
                std::copy(src.begin() + k * 64, src.begin() + (k + 1) * 64, pure.begin() + (g * 8 + b) * 64);
            }
            take_hot ? ++ih : ++ic;
            ++g;
        }
    }
    int *idx_pure;
    CK(hipMalloc(&idx_pure, pure.size() * sizeof(int)));
    CK(hipMemcpy(idx_pure, pure.data(), pure.size() * sizeof(int), hipMemcpyHostToDevice));
    for (int pol = 0; pol <= (g_class_mode == 1 ? 6 : -1); ++pol) {   // cache-policy table: once, with the adopted class
        const int nt = pol == 0 ? -1 : H;
        const long np = (long)pure.size(), ng = np / 64;
        const int grid = 8192;
        const float t256 = run<16>(Z, idx_pure, ng, 8, out, grid, nt, pol), t1k = run<64>(Z, idx_pure, ng, 8, out, grid, nt, pol);
        const char *names[] = {"default", "nt", "sc0", "sc1", "sc0 sc1", "sc1 nt", "sc0 sc1 nt"};
        printf("affine, pure groups, cold rows read with [%-10s] | 256 B %6.3f ms %7.1f GB/s | 1 KiB %6.3f ms %7.1f GB/s\n",
               names[pol], t256, np * 256.0 / t256 / 1e6, t1k, np * 1024.0 / t1k / 1e6);
        fflush(stdout);
    }
    // XCD-affine AND phased in time: the rows of an XCD class are split into P sub-classes ((r / 64) % P) and all the
    // reads of sub-class 0 come before those of sub-class 1, ...: at any moment an XCD's hot working set is 1/P of its
    // class (1-KiB rows: the 4096 hottest rows of a class are the WHOLE 4 MiB L2 -- with P = 2 they are half of it).
    for (int P : {1, 2, 4, 8}) {
        if (g_class_mode != 1) break;                                // phased table: once, with the adopted class
        std::vector<std::vector<int>> lists(8 * P);
        for (int b = 0; b < 8; ++b)
            for (size_t i = 0; i < per; ++i) lists[((by[b][i] >> 6) % P) * 8 + b].push_back(by[b][i]);
        std::vector<int> phased;
        for (int ph = 0; ph < P; ++ph) {
            size_t g = 0;
            for (int b = 0; b < 8; ++b) g = std::max(g, (lists[ph * 8 + b].size() + 63) / 64);
            for (int b = 0; b < 8; ++b) lists[ph * 8 + b].resize(g * 64, lists[ph * 8 + b].back());
            for (size_t k = 0; k < g; ++k)
                for (int b = 0; b < 8; ++b)
                    phased.insert(phased.end(), lists[ph * 8 + b].begin() + k * 64, lists[ph * 8 + b].begin() + (k + 1) * 64);
        }
        int *idx_ph;
        CK(hipMalloc(&idx_ph, phased.size() * sizeof(int)));
        CK(hipMemcpy(idx_ph, phased.data(), phased.size() * sizeof(int), hipMemcpyHostToDevice));
        const long np = (long)phased.size(), ng = np / 64;
        for (int grid : {2048, 8192}) {
            const float t256 = run<16>(Z, idx_ph, ng, 8, out, grid), t512 = run<32>(Z, idx_ph, ng, 8, out, grid);
            const float t1k = run<64>(Z, idx_ph, ng, 8, out, grid);
            printf("XCD-affine, %d phase(s), grid %5d | 256 B %6.3f ms %7.1f GB/s | 512 B %6.3f ms %7.1f | 1 KiB %6.3f ms %7.1f\n",
                   P, grid, t256, np * 256.0 / t256 / 1e6, t512, np * 512.0 / t512 / 1e6, t1k, np * 1024.0 / t1k / 1e6);
            fflush(stdout);
        }
        CK(hipFree(idx_ph));
    }
    for (int variant = 0; variant < 2; ++variant) {
        CK(hipMemcpy(idx, variant ? bucketed.data() : plain.data(), n * sizeof(int), hipMemcpyHostToDevice));
        const int nb = variant ? 8 : 1;
        for (int grid : {8192}) {
            const float t128 = run<8>(Z, idx, n_groups, nb, out, grid);
            const float t256 = run<16>(Z, idx, n_groups, nb, out, grid);
            const float t512 = run<32>(Z, idx, n_groups, nb, out, grid);
            const float t1k = run<64>(Z, idx, n_groups, nb, out, grid);
            printf("%-12s grid %5d | 128 B rows %6.3f ms %7.1f GB/s | 256 B %6.3f ms %7.1f | 512 B %6.3f ms %7.1f | "
                   "1 KiB %6.3f ms %7.1f\n",
                   variant ? "XCD-affine" : "no affinity", grid, t128, n * 128.0 / t128 / 1e6, t256,
                   n * 256.0 / t256 / 1e6, t512, n * 512.0 / t512 / 1e6, t1k, n * 1024.0 / t1k / 1e6);
            fflush(stdout);
        }
    }
    CK(hipFree(Z));
    CK(hipFree(idx));
    CK(hipFree(out));
    CK(hipFree(idx_pure));
    return 0;
}
