// tools/micro/heap_time.hip — developer tool: time of isort::heap_jobs on one range (build: hipcc -O3 --offload-arch=gfx950 -o heap_time heap_time.hip)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "../../planarslam_amd/csrc/isort.h"
using namespace planar::isort;
constexpr int SHIFT = 19;
__global__ __launch_bounds__(64) void k(uint32_t* arr, const HeapJob* jobs, int nj, int cap) {
    extern __shared__ __align__(16) uint8_t lds[];
    heap_jobs<SHIFT>(arr, jobs, nj, blockIdx.x, gridDim.x, (uint32_t*)lds, cap, 0, 1 << 30);
}
int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 30000, nkeys = argc > 2 ? atoi(argv[2]) : 300, cap = argc > 3 ? atoi(argv[3]) : 36864;
    std::vector<uint32_t> h(n);
    srand(1);
    for (int i = 0; i < n; i++) h[i] = ((uint32_t)((((i / 26) % 24) * 7 + i / 2600 + (rand() % 5 == 0)) % nkeys) << SHIFT) | (uint32_t)i;
    uint32_t* d; HeapJob* dj;
    hipMalloc(&d, (size_t)n * 4); hipMalloc(&dj, sizeof(HeapJob));
    HeapJob J{0, n};
    hipMemcpy(dj, &J, sizeof(J), hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, cap * 4);
    for (int rep = 0; rep < 2; rep++) {
        hipMemcpy(d, h.data(), (size_t)n * 4, hipMemcpyHostToDevice);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), cap * 4, 0, d, dj, 1, cap);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("heap sort of %d elements (%d in LDS): %.2f ms = %.3f us per element\n", n, std::min(n, cap), ms, ms * 1e3 / n);
    }
    std::vector<uint32_t> out(n), ref(h);
    hipMemcpy(out.data(), d, (size_t)n * 4, hipMemcpyDeviceToHost);
    auto comp = [](uint32_t a, uint32_t b) { return (a >> SHIFT) < (b >> SHIFT); };
    std::make_heap(ref.begin(), ref.end(), comp); std::sort_heap(ref.begin(), ref.end(), comp);
    printf("%s std::make_heap + std::sort_heap\n", out == ref ? "==" : "!=");
    return 0;
}
