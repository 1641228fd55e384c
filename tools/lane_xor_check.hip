// Checks, on the card, the six cross-lane exchanges behind clane::lane_xor<M> (csrc/device_utils.h) against lane ^ M:
// DPP quad_perm / row_half_mirror / row_ror inside a row of 16 lanes, v_permlane16_swap / v_permlane32_swap (gfx950)
// across rows -- for int, float and double payloads, and the butterflies built on them.
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 -I clane_amd/csrc -o build/lane_xor_check tools/lane_xor_check.hip && build/lane_xor_check
#include <hip/hip_runtime.h>

#include <cstdio>

#include "device_utils.h"

using namespace clane;

__global__ void exchange(double *out) {
    const int l = threadIdx.x;
    const double v = 1000.0 + l;
    int row = 0;
    out[row++ * 64 + l] = lane_xor<1>(v);
    out[row++ * 64 + l] = lane_xor<2>(v);
    out[row++ * 64 + l] = lane_xor<4>(v);
    out[row++ * 64 + l] = lane_xor<8>(v);
    out[row++ * 64 + l] = lane_xor<16>(v);
    out[row++ * 64 + l] = lane_xor<32>(v);
    out[row++ * 64 + l] = double(lane_xor<1>(float(v)));
    out[row++ * 64 + l] = double(lane_xor<2>(float(v)));
    out[row++ * 64 + l] = double(lane_xor<4>(float(v)));
    out[row++ * 64 + l] = double(lane_xor<8>(float(v)));
    out[row++ * 64 + l] = double(lane_xor<16>(float(v)));
    out[row++ * 64 + l] = double(lane_xor<32>(float(v)));
    out[row++ * 64 + l] = group_sum<8>(double(l));
    out[row++ * 64 + l] = group_sum<16>(double(l));
    out[row++ * 64 + l] = group_sum<32>(double(l));
    out[row++ * 64 + l] = group_sum<64>(double(l));
    out[row++ * 64 + l] = group_max<16>(float((l * 37) % 64));
    out[row++ * 64 + l] = group_max<64>(float((l * 37) % 64));
}

int main() {
    constexpr int kRows = 18;
    double *d, h[kRows * 64];
    if (hipMalloc(&d, sizeof(h)) != hipSuccess) return 2;
    exchange<<<1, 64>>>(d);
    if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return 2;
    const int masks[6] = {1, 2, 4, 8, 16, 32};
    int bad = 0;
    for (int r = 0; r < 12; ++r)
        for (int l = 0; l < 64; ++l) bad += h[r * 64 + l] != 1000.0 + (l ^ masks[r % 6]);
    const int widths[4] = {8, 16, 32, 64};
    for (int w = 0; w < 4; ++w)
        for (int l = 0; l < 64; ++l) {
            const int base = l / widths[w] * widths[w];
            bad += h[(12 + w) * 64 + l] != double(widths[w]) * base + widths[w] * (widths[w] - 1) / 2.0;
        }
    for (int l = 0; l < 64; ++l) {
        float m16 = 0, m64 = 0;
        for (int i = 0; i < 64; ++i) {
            const float x = float((i * 37) % 64);
            if (i / 16 == l / 16 && x > m16) m16 = x;
            if (x > m64) m64 = x;
        }
        bad += h[16 * 64 + l] != m16;
        bad += h[17 * 64 + l] != m64;
    }
    std::printf("lane_xor check: %d mismatches\n", bad);
    return bad != 0;
}
