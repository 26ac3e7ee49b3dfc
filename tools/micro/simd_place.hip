// tools/micro/simd_place.hip — measurement helper (not product code): where do the wavefronts of co-resident workgroups land, and does a second
// sequential FP64 chain on the same CU run beside the first one?  hipcc --offload-arch=gfx950 -O3 simd_place.hip -o simd_place && ./simd_place
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <chrono>

__global__ __launch_bounds__(256) void chain(double* out, int* hw, int iters, int mode) {
    extern __shared__ char smem[];
    const int wave = threadIdx.x >> 6;
    unsigned id = __builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4);        // HW_ID bits [5:4] = SIMD_ID (size-1 = 1)
    unsigned cu = __builtin_amdgcn_s_getreg((3 << 11) | (8 << 6) | 4);        // CU_ID [11:8]
    unsigned se = __builtin_amdgcn_s_getreg((2 << 11) | (13 << 6) | 4);       // SE_ID [15:13]
    unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);      // XCC_ID
    if ((threadIdx.x & 63) == 0) hw[blockIdx.x * 4 + wave] = (int)(id | (cu << 4) | (se << 8) | (xcc << 12));
    int busy = 0;
    if (mode == 0) busy = 0;                       // wave 0 of every workgroup
    else if (mode == 1) busy = blockIdx.x & 3;      // varies with the block index
    else if (mode == 2) busy = (blockIdx.x >> 8) & 3;
    if (wave != busy) return;
    double x = 1.0 + threadIdx.x * 1e-9;
    for (int i = 0; i < iters; i++) x = x * 1.0000001 + 1e-12;               // dependent FP64 FMA-free chain (mul + add)
    if (x == 123.0) smem[0] = 1;
    out[blockIdx.x * 256 + threadIdx.x] = x;
}

int main() {
    double* d_out; int* d_hw;
    const int NB = 1024;
    hipMalloc(&d_out, NB * 256 * 8); hipMalloc(&d_hw, NB * 16);
    const int iters = 2000000;
    for (int lds : {140 * 1024, 70 * 1024, 36 * 1024}) {
        hipFuncSetAttribute((const void*)chain, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        for (int mode = 0; mode < 3; mode++)
            for (int grid : {256, 512, 1024}) {
                hipDeviceSynchronize();
                auto t0 = std::chrono::steady_clock::now();
                hipLaunchKernelGGL(chain, dim3(grid), dim3(256), lds, 0, d_out, d_hw, iters, mode);
                hipDeviceSynchronize();
                double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
                printf("lds %3d KB mode %d grid %4d : %8.2f ms\n", lds / 1024, mode, grid, ms);
            }
    }
    std::vector<int> hw(1024 * 4);
    hipFuncSetAttribute((const void*)chain, hipFuncAttributeMaxDynamicSharedMemorySize, 70 * 1024);
    hipLaunchKernelGGL(chain, dim3(512), dim3(256), 70 * 1024, 0, d_out, d_hw, 1000, 0);
    hipDeviceSynchronize();
    hipMemcpy(hw.data(), d_hw, 512 * 16, hipMemcpyDeviceToHost);
    for (int b : {0, 1, 2, 3, 256, 257, 258, 300, 511}) {
        printf("block %3d:", b);
        for (int w = 0; w < 4; w++) printf("  w%d simd %d cu %d se %d xcc %d", w, hw[b * 4 + w] & 3, (hw[b * 4 + w] >> 4) & 15, (hw[b * 4 + w] >> 8) & 7, (hw[b * 4 + w] >> 12) & 15);
        printf("\n");
    }
    return 0;
}
