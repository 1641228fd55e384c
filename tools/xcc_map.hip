// Which XCD does workgroup w of a dispatch run on?  The class-affine row kernels (csrc/spmm_update.h) rely on the
// round-robin dispatch of MI300-class parts: workgroup w -> XCD w % 8.  This reads HW_REG_XCC_ID in every workgroup
// of a few grids and reports how many follow that rule.  (If the rule ever stops holding, results stay correct --
// only the L2 affinity, i.e. speed, is lost.)   hipcc --offload-arch=gfx950 -O3 -o build/xcc_map tools/xcc_map.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void who(int *xcc) {
    if (threadIdx.x == 0) xcc[blockIdx.x] = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 0xf;  // XCC_ID[3:0]
}

int main() {
    for (int grid : {8, 64, 2048, 100000, 1000003}) {
        for (int block : {64, 256, 1024}) {
            int *d;
            hipMalloc(&d, grid * sizeof(int));
            who<<<grid, block>>>(d);
            std::vector<int> h(grid);
            hipMemcpy(h.data(), d, grid * sizeof(int), hipMemcpyDeviceToHost);
            long ok = 0;
            int hist[16] = {0};
            for (int w = 0; w < grid; ++w) {
                ok += h[w] == w % 8;
                hist[h[w] & 15]++;
            }
            printf("grid %8d x %4d threads: %ld / %d workgroups on XCD (w %% 8); per XCD:", grid, block, ok, grid);
            for (int x = 0; x < 8; ++x) printf(" %d", hist[x]);
            printf("\n");
            hipFree(d);
        }
    }
    return 0;
}
