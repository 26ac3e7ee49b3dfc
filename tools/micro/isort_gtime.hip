// tools/micro/isort_gtime.hip — developer tool: phase timing of isort::global_tier (build: hipcc -O3 --offload-arch=gfx950 -DISORT_TIMING -o isort_gtime isort_gtime.hip)
// usage: isort_gtime n blocks mode      mode 0: LSD-like keys (most elements in the lowest bins), 1: saw-tooth runs (a plane's voxel keys)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <cmath>
#define ISORT_TIMING 1
#include "../../planarslam_amd/csrc/isort.h"
using namespace planar::isort;
constexpr int T = 1024, SHIFT = 20, NSTAGE = 5888;
__global__ __launch_bounds__(T) void k(uint32_t* arr, int n, int rows_cap, Range* ranges, Block* blocks, int* counts, int* status) {
    extern __shared__ __align__(16) uint8_t lds[];
    __shared__ Range s_init;
    static __device__ HeapJob hj[4096]; static __device__ int hn;
    const HeapSink HS{hj, &hn, 4096};
    if (threadIdx.x == 0) s_init = Range{0, n, depth_limit(n)};
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    global_tier<SHIFT, T>(arr + (size_t)blockIdx.x * n, &s_init, 1, NSTAGE, 64, ranges + (size_t)blockIdx.x * G_FMAX, blocks + (size_t)blockIdx.x * G_FMAX, G_FMAX, counts + 2 * blockIdx.x, lds, rows_cap, HS, status);
    if (threadIdx.x == 0) g_isort_t[(blockIdx.x % ISORT_TBLK) * 16 + 15] += __builtin_readcyclecounter() - t0;
}
int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 195000, NB = argc > 2 ? atoi(argv[2]) : 256, mode = argc > 3 ? atoi(argv[3]) : 0;
    std::vector<uint32_t> h((size_t)n * NB);
    srand(1);
    for (size_t i = 0; i < h.size(); i++) {
        uint32_t key;
        if (mode == 0) { const double u = (rand() + 1.0) / (RAND_MAX + 2.0); const int bin = (rand() % 100 < 88) ? rand() % 6 : std::min(1023, (int)(-60.0 * std::log(u))); key = 1023 - bin; }
        else key = (uint32_t)((((i % n) / 26) % 24) * 7 + (i % n) / 2600 + (rand() % 5 == 0)) % 1000;
        h[i] = (key << SHIFT) | (uint32_t)(i % n);
    }
    uint32_t* d; Range* dr; Block* db; int* dc; int* ds;
    hipMalloc(&d, h.size() * 4); hipMalloc(&dr, sizeof(Range) * G_FMAX * NB); hipMalloc(&db, sizeof(Block) * G_FMAX * NB); hipMalloc(&dc, 8 * NB); hipMalloc(&ds, 4);
    hipMemset(ds, 0, 4);
    const int rows_cap = GlobalLayout<T>::rows_for(n), smem = GlobalLayout<T>::bytes(rows_cap);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    for (int rep = 0; rep < 2; rep++) {
        hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        static long long zz[ISORT_TBLK * 16]; long long z[16] = {0};
        for (auto& q : zz) q = 0;
        hipMemcpyToSymbol(HIP_SYMBOL(g_isort_t), zz, sizeof(zz));
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(NB), dim3(T), smem, 0, d, n, rows_cap, dr, db, dc, ds);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpyFromSymbol(zz, HIP_SYMBOL(g_isort_t), sizeof(zz));
        for (int b = 0; b < ISORT_TBLK; b++) for (int k = 0; k < 16; k++) z[k] += zz[b * 16 + k];
        std::vector<int> c(2 * NB); hipMemcpy(c.data(), dc, 8 * NB, hipMemcpyDeviceToHost);
        int st; hipMemcpy(&st, ds, 4, hipMemcpyDeviceToHost);
        printf("n %d blocks %d mode %d lds %d B: %.3f ms | thread-0 cycles summed over blocks: pivot %lld | pass1 ballots %lld | rank prefix %lld | x* %lld | swaps %lld | whole tier %lld | ranges %d blocks %d status %d\n",
               n, NB, mode, smem, ms, z[10], z[11], z[12], z[13], z[14], z[15], c[0], c[1], st);
    }
    return 0;
}
