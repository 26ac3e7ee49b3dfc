// tools/micro/wave_costs.hip — calibration of what a LONE wavefront pays on gfx950 for the building blocks of the single-wavefront kernels
// (dependent LDS reads, ballots + branches, DPP reductions, shuffles, scalar chains, dependent global loads).  One workgroup of 64 threads; cycles per
// operation from s_memtime around loops of N dependent operations.   hipcc --offload-arch=gfx950 -O3 -o wave_costs wave_costs.hip && ./wave_costs
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../planarslam_amd/csrc/wave_ops.h"
using namespace planar;

__global__ __launch_bounds__(64) void k(long long* out, unsigned* gbuf, int gwords, int N) {
    __shared__ unsigned short lds[6144];
    __shared__ unsigned ldsw[2048];
    const int lane = threadIdx.x;
    for (int i = lane; i < 6144; i += 64) lds[i] = (unsigned short)((i * 2654435761u >> 7) % 6144);
    for (int i = lane; i < 2048; i += 64) ldsw[i] = (i * 40503u + 17) & 2047;
    __syncthreads();
    long long t0, t1; int r = 0;
    // (0) dependent LDS u16 reads, random per lane
    { unsigned x = lane * 37 % 6144; t0 = clock64(); for (int i = 0; i < N; i++) x = lds[x]; t1 = clock64(); r += x; out[0] = (t1 - t0) / N; }
    // (1) dependent LDS u32 reads, uniform address
    { unsigned x = 5; t0 = clock64(); for (int i = 0; i < N; i++) x = ldsw[x]; t1 = clock64(); r += x; out[1] = (t1 - t0) / N; }
    // (2) ballot + uniform branch per iteration
    { unsigned x = lane; int c = 0; t0 = clock64(); for (int i = 0; i < N; i++) { if (__ballot((x >> (i & 7)) & 1) & 2) c += 3; else c ^= i; x = x * 5 + 1; } t1 = clock64(); r += c; out[2] = (t1 - t0) / N; }
    // (3) wave_min_f32 chain
    { float v = lane * 1.5f + 3; t0 = clock64(); for (int i = 0; i < N; i++) v = wave_min_f32(v + lane) + 1.0f; t1 = clock64(); r += (int)v; out[3] = (t1 - t0) / N; }
    // (4) wave_scan_add chain
    { int v = lane; t0 = clock64(); for (int i = 0; i < N; i++) v = wave_scan_add(v & 3) + lane; t1 = clock64(); r += v; out[4] = (t1 - t0) / N; }
    // (5) __shfl (ds_bpermute) dependent chain
    { int v = lane; t0 = clock64(); for (int i = 0; i < N; i++) v = __shfl(v, (v + 7) & 63) + 1; t1 = clock64(); r += v; out[5] = (t1 - t0) / N; }
    // (6) v_readlane with uniform index chain
    { int v = lane * 3, s = 1; t0 = clock64(); for (int i = 0; i < N; i++) { s = wave_lane(v, s & 63) + 1; v += s; } t1 = clock64(); r += v; out[6] = (t1 - t0) / N; }
    // (7) SALU dependent chain (4 ops per iteration)
    { int s = wave_uni(N); t0 = clock64(); for (int i = 0; i < N; i++) { s = (s << 1) ^ (s >> 3); s += 7; s &= 0xfffff; s |= 1; } t1 = clock64(); r += s; out[7] = (t1 - t0) / N; }
    // (8) dependent FP64 fma chain (4 per iteration)
    { double a = lane + 1.0; t0 = clock64(); for (int i = 0; i < N; i++) { a = a * 1.0000001 + 0.5; a = a * 0.9999999 + 0.25; a = a * 1.0000002 - 0.5; a = a * 0.9999998 - 0.25; } t1 = clock64(); r += (int)a; out[8] = (t1 - t0) / N; }
    // (9) FP64 division chain
    { double a = lane + 3.0; t0 = clock64(); for (int i = 0; i < N; i++) a = 1000.0 / a + 2.0; t1 = clock64(); r += (int)a; out[9] = (t1 - t0) / N; }
    // (10) FP64 sqrt chain
    { double a = lane + 3.0; t0 = clock64(); for (int i = 0; i < N; i++) a = sqrt(a) + 5.0; t1 = clock64(); r += (int)a; out[10] = (t1 - t0) / N; }
    // (11) dependent global loads, one dword per lane, random lines inside gwords (warm)
    { unsigned x = lane * 977 % gwords; for (int i = 0; i < 64; i++) x = gbuf[x]; t0 = clock64(); for (int i = 0; i < N; i++) x = gbuf[x]; t1 = clock64(); r += x; out[11] = (t1 - t0) / N; }
    // (12) dependent global loads, uniform address (all lanes the same word)
    { unsigned x = 12345 % gwords; t0 = clock64(); for (int i = 0; i < N; i++) x = gbuf[x]; t1 = clock64(); r += x; out[12] = (t1 - t0) / N; }
    // (13) global store then dependent load of the same address by another lane (write -> read turnaround)
    { unsigned x = 0; t0 = clock64(); for (int i = 0; i < N; i++) { gbuf[gwords + ((lane + 1) & 63) + 64 * (i & 7)] = x + i; __threadfence_block(); x = gbuf[gwords + lane + 64 * (i & 7)]; } t1 = clock64(); r += x; out[13] = (t1 - t0) / N; }
    // (14) LDS atomic or (no return) followed by a read
    { unsigned x = lane; t0 = clock64(); for (int i = 0; i < N; i++) { atomicOr(&ldsw[(x + i) & 2047], 1u << (lane & 31)); x = ldsw[(x * 3 + i) & 2047]; } t1 = clock64(); r += x; out[14] = (t1 - t0) / N; }
    // (15) 10 independent VALU int ops per iteration (issue rate)
    { int a = lane, b = 1, c = 2, d = 3, e = 4; t0 = clock64(); for (int i = 0; i < N; i++) { a += i; b ^= i; c += 3; d -= i; e |= i; a ^= 5; b += 7; c ^= 9; d += 11; e ^= 13; } t1 = clock64(); r += a + b + c + d + e; out[15] = (t1 - t0) / N; }
    if (r == 0x7fffffff) out[31] = r;
}
int main() {
    long long* d; unsigned* g; const int gwords = 1 << 20;   // 4 MB
    hipMalloc(&d, 32 * 8); hipMalloc(&g, (gwords + 1024) * 4);
    std::vector<unsigned> h(gwords + 1024);
    for (int i = 0; i < gwords; i++) h[i] = (unsigned)(((unsigned long long)i * 2654435761ull + 12345) % gwords);
    hipMemcpy(g, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    const char* names[16] = {"lds u16 dependent read (random/lane)", "lds u32 dependent read (uniform)", "ballot + branch iteration", "wave_min_f32 + add", "wave_scan_add + ops", "ds_bpermute dependent",
                             "v_readlane (sgpr index) chain", "4 dependent SALU ops", "4 dependent FP64 fma", "FP64 division + add", "FP64 sqrt + add", "global load dependent (random/lane, 4 MB)",
                             "global load dependent (uniform)", "global store -> fence -> load (other lane's word)", "lds atomicOr + dependent read", "10 independent VALU int ops"};
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, g, gwords, 2000);
        hipDeviceSynchronize();
    }
    long long o[32]; hipMemcpy(o, d, sizeof o, hipMemcpyDeviceToHost);
    for (int i = 0; i < 16; i++) printf("%-52s %6lld cycles\n", names[i], o[i]);
    return 0;
}
