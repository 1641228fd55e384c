// Experiment: does marking COLD-row gathers non-temporal keep the HOT (hub) rows resident in L2/MALL?
// Rows are relabelled by popularity (row id = popularity rank), indices follow an R-MAT-like skew.
// Variants: plain loads; NT loads for rows >= H.  Prints GB/s.  Not part of the product.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int U, bool NT>
__global__ __launch_bounds__(256) void gather(const float4* __restrict__ Z, const int* __restrict__ idx, long n_groups,
                                              int H, float4* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long nw = (long)gridDim.x * 4;
    float4 acc = {0, 0, 0, 0};
    for (long g = wave; g < n_groups; g += nw) {
        const int my = idx[g * 64 + lane];
        for (int j = 0; j < 64; j += U) {
            float4 z[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int c = __builtin_amdgcn_readlane(my, j + u);
                typedef float v4 __attribute__((ext_vector_type(4)));
                const v4* p = reinterpret_cast<const v4*>(Z + (long)c * 64 + lane);
                v4 t;
                if (NT && c >= H) t = __builtin_nontemporal_load(p); else t = *p;
                z[u] = make_float4(t.x, t.y, t.z, t.w);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) { acc.x += z[u].x; acc.y += z[u].y; acc.z += z[u].z; acc.w += z[u].w; }
        }
    }
    if (acc.x == 1.2345f) out[wave * 64 + lane] = acc;
}

int main() {
    const long V = 2000000, n_idx = 40000000 / 64 * 64;
    float4* Z; int* idx; float4* out;
    CK(hipMalloc(&Z, V * 64 * sizeof(float4))); CK(hipMemset(Z, 0, V * 64 * sizeof(float4)));
    CK(hipMalloc(&idx, n_idx * sizeof(int))); CK(hipMalloc(&out, 1 << 24));
    std::vector<int> h(n_idx); std::mt19937_64 rng(1);
    for (long i = 0; i < n_idx; ++i) { unsigned v = 0; for (int l = 0; l < 21; ++l) v = (v << 1) | ((rng() & 0xffff) >= 0.76 * 65536); h[i] = v % V; }
    // relabel by popularity
    std::vector<int> cnt(V, 0); for (long i = 0; i < n_idx; ++i) cnt[h[i]]++;
    std::vector<int> order(V); std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return cnt[a] > cnt[b]; });
    std::vector<int> rank(V); for (int r = 0; r < V; ++r) rank[order[r]] = r;
    long c4k = 0, c32k = 0, c250k = 0; for (int r = 0; r < V; ++r) { if (r < 4096) c4k += cnt[order[r]]; if (r < 32768) c32k += cnt[order[r]]; if (r < 250000) c250k += cnt[order[r]]; }
    printf("share of reads to top 4k / 32k / 250k rows: %.3f %.3f %.3f\n", (double)c4k / n_idx, (double)c32k / n_idx, (double)c250k / n_idx);
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int relabel = 0; relabel < 2; ++relabel) {
        std::vector<int> hh(n_idx); for (long i = 0; i < n_idx; ++i) hh[i] = relabel ? rank[h[i]] : h[i];
        CK(hipMemcpy(idx, hh.data(), n_idx * sizeof(int), hipMemcpyHostToDevice));
        for (int H : {0, 1024, 4096, 32768, 250000}) {
            if (!relabel && H) continue;
            float best = 1e9;
            for (int rep = 0; rep < 4; ++rep) {
                CK(hipEventRecord(a));
                if (H == 0) gather<8, false><<<4096, 256>>>(Z, idx, n_idx / 64, 0, out);
                else gather<8, true><<<4096, 256>>>(Z, idx, n_idx / 64, H, out);
                CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
                float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
            }
            printf("relabel %d  NT for rows >= %6d : %7.3f ms  %7.1f GB/s\n", relabel, H, best, n_idx * 1024.0 / best / 1e6);
            fflush(stdout);
        }
    }
    return 0;
}
