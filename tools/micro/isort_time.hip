// tools/micro/isort_time.hip — developer tool: phase timing of isort::lds_tier on one block (build: hipcc -O3 --offload-arch=gfx950 -DISORT_TIMING -o isort_time isort_time.hip)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define ISORT_TIMING 1
#include "../../planarslam_amd/csrc/isort.h"
using namespace planar::isort;
#ifndef T_
#define T_ 1024
#endif
#ifndef E_
#define E_ 23
#endif
constexpr int T = T_, E = E_, SHIFT = 19;
__global__ __launch_bounds__(T) void k(uint32_t* arr, const Range* r, int nr, int n, int* status) {
    extern __shared__ __align__(16) uint8_t lds[];
    static __device__ HeapJob hj[64]; static __device__ int hn; const HeapSink HS{hj, &hn, 64}; lds_tier<SHIFT, T, E>(arr + (size_t)blockIdx.x * n, r, nr, 0, n, lds, HS, status);
}
int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : T * E, nkeys = argc > 2 ? atoi(argv[2]) : 1024, NB = argc > 3 ? atoi(argv[3]) : 1;
    std::vector<uint32_t> h((size_t)n * NB);
    srand(1);
    const int mode = argc > 4 ? atoi(argv[4]) : 0;   // 1: saw-tooth runs (a plane's voxel keys in raster order)
    for (size_t i = 0; i < h.size(); i++) { const uint32_t k = mode ? (uint32_t)((((i % n) / 26) % 24) * 7 + (i % n) / 2600 + (rand() % 5 == 0)) % nkeys : (uint32_t)(rand() % nkeys); h[i] = (k << SHIFT) | (uint32_t)(i % n); }
    uint32_t* d; Range* dr; int* ds;
    hipMalloc(&d, h.size() * 4); hipMalloc(&dr, sizeof(Range)); hipMalloc(&ds, 4);
    const int NR = argc > 5 ? atoi(argv[5]) : 1;
    std::vector<int> bnd{0, n};
    for (int i = 1; i < NR; i++) bnd.push_back(rand() % (n + 1));
    if (NR > 3) { bnd.push_back(bnd[2]); bnd.push_back(std::min(n, bnd[3] + 1)); bnd.push_back(std::min(n, bnd[3] + 17)); }
    std::sort(bnd.begin(), bnd.end());
    std::vector<Range> hr;
    for (size_t i = 0; i + 1 < bnd.size(); i++) { const int m = bnd[i + 1] - bnd[i]; if (m < 2 || (NR > 5 && i % 5 == 4)) continue; int lg = 0; for (int t = m; t > 1; t >>= 1) lg++; hr.push_back(Range{bnd[i], bnd[i + 1], 2 * lg}); }
    hipFree(dr); hipMalloc(&dr, sizeof(Range) * hr.size());
    hipMemcpy(dr, hr.data(), sizeof(Range) * hr.size(), hipMemcpyHostToDevice); hipMemset(ds, 0, 4);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, LdsLayout<T, E>::bytes);
    for (int rep = 0; rep < 3; rep++) {
        hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        static long long zz[ISORT_TBLK * 16]; long long z[16] = {0};
        for (auto& q : zz) q = 0;
        hipMemcpyToSymbol(HIP_SYMBOL(g_isort_t), zz, sizeof(zz));
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        const int smem = LdsLayout<T, E>::bytes; hipLaunchKernelGGL(k, dim3(NB), dim3(T), smem, 0, d, dr, (int)hr.size(), n, ds);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpyFromSymbol(zz, HIP_SYMBOL(g_isort_t), sizeof(zz));
        for (int b = 0; b < ISORT_TBLK; b++) for (int k = 0; k < 16; k++) z[k] += zz[b * 16 + k];
        printf("n %d keys %d blocks %d: %.3f ms | cycles (all blocks, thread 0): stage-in+init %lld | A medians %lld | B flags %lld | scans %lld | D1 %lld | D2 %lld | E swaps %lld | F list %lld | tasks %lld | write-back %lld\n", n, nkeys, NB, ms,
               z[0], z[1], z[2], z[3], z[4], z[5], z[6], z[7], z[8], z[9]);
    }
    std::vector<uint32_t> out(h.size());
    hipMemcpy(out.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int b = 0; b < NB; b++) {
        std::vector<uint32_t> ref(h.begin() + (size_t)b * n, h.begin() + (size_t)(b + 1) * n);
        for (const Range& R : hr) std::sort(ref.begin() + R.f, ref.begin() + R.l, [](uint32_t a, uint32_t b) { return (a >> SHIFT) < (b >> SHIFT); });
        if (!std::equal(ref.begin(), ref.end(), out.begin() + (size_t)b * n)) bad++;
    }
    int st = 0; hipMemcpy(&st, ds, 4, hipMemcpyDeviceToHost);
    printf("%d of %d blocks differ from std::sort (%d ranges), status %d\n", bad, NB, (int)hr.size(), st);
    return 0;
}
