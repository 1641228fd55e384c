// Micro-benchmark: what is the practical ceiling for gathering whole 1-KiB rows (d=256 fp32) of a
// 2 GB table on this MI355X?  One wave per "destination", U rows in flight, uniform-random or
// sequential indices, W workgroups.  Prints GB/s per configuration.  Not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int U>
__global__ __launch_bounds__(256) void gather(const float4* __restrict__ Z, const int* __restrict__ idx,
                                              long n_groups, float4* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long nw = (long)gridDim.x * 4;
    float4 acc = {0, 0, 0, 0};
    for (long g = wave; g < n_groups; g += nw) {
        const int my = idx[g * 64 + lane];           // 64 indices per group, U consumed per step
        for (int j = 0; j < 64; j += U) {
            float4 z[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int c = __builtin_amdgcn_readlane(my, j + u);
                z[u] = Z[(long)c * 64 + lane];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) { acc.x += z[u].x; acc.y += z[u].y; acc.z += z[u].z; acc.w += z[u].w; }
        }
    }
    if (acc.x == 1.2345f) out[wave * 64 + lane] = acc;
}

int main() {
    const long V = 2000000, d4 = 64;          // 2M rows x 1 KiB
    const long n_idx = 40000000 / 64 * 64;    // 40M row reads = 40.96 GB
    float4* Z; int* idx; float4* out;
    CK(hipMalloc(&Z, V * d4 * sizeof(float4)));
    CK(hipMemset(Z, 0, V * d4 * sizeof(float4)));
    CK(hipMalloc(&idx, n_idx * sizeof(int)));
    CK(hipMalloc(&out, 1 << 24));
    std::vector<int> h(n_idx);
    std::mt19937_64 rng(1);
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int pattern = 0; pattern < 3; ++pattern) {
        if (pattern == 0) for (long i = 0; i < n_idx; ++i) h[i] = (int)(rng() % V);
        if (pattern == 1) for (long i = 0; i < n_idx; ++i) h[i] = (int)(i % V);
        if (pattern == 2) for (long i = 0; i < n_idx; ++i) {                     // R-MAT-like skew (a+c=0.76)
            unsigned v = 0; for (int l = 0; l < 21; ++l) v = (v << 1) | ((rng() & 0xffff) >= 0.76 * 65536); h[i] = v % V; }
        CK(hipMemcpy(idx, h.data(), n_idx * sizeof(int), hipMemcpyHostToDevice));
        for (int grid : {1024, 2048, 4096, 16384}) {
            for (int U : {4, 8, 16}) {
                float best = 1e9;
                for (int rep = 0; rep < 3; ++rep) {
                    CK(hipEventRecord(a));
                    if (U == 4) gather<4><<<grid, 256>>>(Z, idx, n_idx / 64, out);
                    if (U == 8) gather<8><<<grid, 256>>>(Z, idx, n_idx / 64, out);
                    if (U == 16) gather<16><<<grid, 256>>>(Z, idx, n_idx / 64, out);
                    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
                    float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
                }
                printf("pattern %s grid %5d U %2d : %7.3f ms  %7.1f GB/s\n",
                       pattern == 0 ? "uniform" : pattern == 1 ? "sequential" : "rmat-skew", grid, U, best,
                       n_idx * 1024.0 / best / 1e6);
                fflush(stdout);
            }
        }
    }
    return 0;
}
