// planarslam_amd/csrc/lsd.hip — line extraction for MI355X (gfx950): SURVEY.md §8 rows a10, a11, a12.
//
//   planar_lsd_extract   LineSegment::ExtractLineSegment                       reference src/LSDextractor.cpp:12-39
//                        = cv::line_descriptor::LSDDetector::detect (1 octave) -> cv::LineSegmentDetector(LSD_REFINE_ADV)
//                        + sort by response / keep 40 + cv::line_descriptor::BinaryDescriptor::compute (LBD)
//                        + homogeneous line equations
// The arithmetic follows oracle/lsd_oracle.cpp statement by statement (that file lists what is [assumed] about the
// un-vendored OpenCV sources).  Structure on the GPU:
//   K1 lsd_gauss        8U fixed-point Gaussian (7x7 sigma 0.75 for LSD, 5x5 sigma 1 for LBD), LDS tile, pixel-parallel
//   K2 lsd_grad         0.8x INTER_LINEAR_EXACT resample fused with the 2x2 gradient: level-line angle (float degrees,
//                       exactly what fastAtan2 returned), squared gradient (u32), per-frame max, pixel-parallel
//   K3 lsd_sort_*       per frame: the visiting order = std::sort by 1024-bin gradient norm.  The order libstdc++'s introsort leaves among equal
//                       bins is reproduced by isort.h (level-synchronous parallel Hoare partitions: bitmaps of the scans' stops in the global
//                       tier, whole recursion levels at once in LDS blocks), then the undefined pixels are dropped
//   K4 lsd_detect       ONE WAVEFRONT PER FRAME, the sequential part: region growing in the reference's visiting order
//                       (the level-line angle of a region is updated after every accepted pixel, so acceptance is a
//                       chain), rectangle fit, density refinement, NFA validation.  The wave hides memory latency by
//                       expanding 7 queued pixels x 9 neighbours per round trip and resolving acceptances in lane
//                       order; FP64 moment sums run as three independent chains on three lanes (same order as the
//                       reference => bit-identical), min/max and NFA pixel counts are order-free wave reductions.
//   (K5, the Sobel images of the 5x5-blurred image, is fused into K7 since round 5)
//   K6 lsd_keylines     per frame: KeyLine fields, libstdc++ std::sort emulation on `response`, keep max_lines, equations
//   K7 lbd_describe     one wavefront per kept line: 63 support rows on 63 lanes, band sums in reference order
// HBM traffic is small (a 640x480 frame: 0.3 MB in, 3 KB out, ~5 MB of L2-resident intermediates); K4 is latency bound.
#include "common.h"
#include "wave_ops.h"
#include "lsd_nfa.h"
#include "isort.h"

namespace planar {
namespace lsd {

constexpr double LSD_PI = 3.14159265358979323846;
constexpr double M_3_2_PI = 3 * LSD_PI / 2, M_2__PI = 2 * LSD_PI;
constexpr float NOTDEF_F = -1024.0f;
constexpr double DEG_TO_RADS = LSD_PI / 180;
constexpr double RELATIVE_ERROR_FACTOR = 100.0;
constexpr int N_BINS = 1024;
constexpr int MAX_SEGS = 2048;     // raw LSD segments kept per frame
constexpr int MAX_RECTS = 4096;    // regions that reach the NFA stage, per frame
constexpr int RING = 512;          // recent region points kept in LDS
constexpr int USED_LDS_BITS = 32768;   // `used` flags of the first 32768 defined pixels live in LDS, the rest in global memory

struct Plan {
    int W, H, w, h;                // input and 0.8x sizes
    int taps7[7], taps5[5];        // Q8 Gaussian taps
    double rho, prec, p, log_nt, density_th, log_eps;
    int min_reg_size;
    // per-frame workspace offsets (bytes)
    size_t off_blur7, off_blur5, off_dx, off_dy, off_ang, off_g2, off_pix, off_seed, off_ord, off_ordr, off_gused, off_tmp, off_reg, off_valid, off_sortr, off_sortb, off_heapj, off_segs, off_kl, off_rects, off_res, off_est, frame_bytes;
    // host-evaluated tables (glibc, as the reference library would): log_gamma(x) for integer x, and per halving j of p
    const double* lgamma_tab;   // [w*h + 3]
    double p_log[12], p1_log[12], p_log10[12];
    double gaussCoefL[21], gaussCoefG[63];
};

struct Misc { uint32_t g2max; int n_ord; int n_seg; int n_kl; int status; int n_regions; int n_grown_px; int n_rect; long long t[8]; int sort_counts[2]; int heap_n; int sort_kv; int sort_prefix; int pad_; float imp_T; int imp_done; int imp_redo; int imp_full; };   // sort_counts: ranges, LDS-tier blocks; heap_n: ranges left to the heap-sort fallback; sort_kv: the largest sort key of a pixel with a defined angle, sort_prefix: words with a key <= that

__device__ __forceinline__ float fast_atan2_deg(float y, float x) {
    // cv::fastAtan2: 7th-order odd polynomial, degrees; plain mul/add (no FMA), see oracle/cvprim.cpp
    const float scale = (float)(180 / 3.1415926535897932384626433832795);
    const float p1 = 0.9997878412794807f * scale, p3 = -0.3258083974640975f * scale;
    const float p5 = 0.1555786518463281f * scale, p7 = -0.04432655554792128f * scale;
    const float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + 2.220446049250313e-16f);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + 2.220446049250313e-16f);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

__device__ __forceinline__ int reflect101(int p, int n) {
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = p < 0 ? -p : 2 * n - 2 - p;
    return p;
}

// ---- K1: 8U fixed-point Gaussian, ksize 5 or 7 (cv::GaussianBlur, BORDER_REFLECT_101) ---------------------------
template <int KS>
__global__ __launch_bounds__(256) void lsd_gauss(const uint8_t* __restrict__ src, int pitch, int64_t src_stride, int W, int H,
                                                 const int* __restrict__ taps_g, uint8_t* __restrict__ ws, size_t frame_bytes, size_t off_dst, int B) {
    constexpr int R = KS / 2, TW = 64, TH = 16;
    __shared__ uint8_t s_in[TH + 2 * R][TW + 2 * R];
    __shared__ uint16_t s_h[TH + 2 * R][TW];
    int taps[KS];
    for (int i = 0; i < KS; i++) taps[i] = taps_g[i];
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH, tid = threadIdx.x;
    int b, tile;
    xcd_frame_block(tiles_x * tiles_y, B, b, tile);          // neighbouring tiles share their halos: a frame's tiles on one XCD (common.h)
    const int x0 = (tile % tiles_x) * TW, y0 = (tile / tiles_x) * TH;
    const uint8_t* S = src + (int64_t)b * src_stride;
    for (int i = tid; i < (TH + 2 * R) * (TW + 2 * R); i += 256) {
        const int r = i / (TW + 2 * R), c = i % (TW + 2 * R);
        s_in[r][c] = S[(int64_t)reflect101(y0 + r - R, H) * pitch + reflect101(x0 + c - R, W)];
    }
    __syncthreads();
    for (int i = tid; i < (TH + 2 * R) * TW; i += 256) {
        const int r = i / TW, c = i % TW;
        uint32_t s = 0;
        for (int k = 0; k < KS; k++) s += (uint32_t)taps[k] * s_in[r][c + k];
        s_h[r][c] = (uint16_t)min(s, 65535u);
    }
    __syncthreads();
    uint8_t* D = ws + (size_t)b * frame_bytes + off_dst;
    for (int i = tid; i < TH * TW; i += 256) {
        const int r = i / TW, c = i % TW;
        const int x = x0 + c, y = y0 + r;
        if (x >= W || y >= H) continue;
        uint32_t s = 0;
        for (int k = 0; k < KS; k++) s += (uint32_t)taps[k] * s_h[r + k][c];
        D[(size_t)y * W + x] = (uint8_t)min((s + 32768u) >> 16, 255u);
    }
}

// Both Gaussians (7x7 for the detector, 5x5 for the descriptor) of one 64x16 tile from ONE read of the tile, four pixels per thread: the image as aligned 32-bit
// words, horizontal sums as packed u16 in LDS, four outputs per thread and store.  Same fixed-point arithmetic as lsd_gauss (u16-saturated row sums, (s + 2^15) >> 16).
// Requires W % 4 == 0, pitch % 4 == 0 and 4-byte aligned frames (the host falls back to lsd_gauss otherwise).
__global__ __launch_bounds__(256) void lsd_gauss75(const uint8_t* __restrict__ src, int pitch, int64_t src_stride, int W, int H, const int* __restrict__ taps_g,
                                                   uint8_t* __restrict__ ws, size_t frame_bytes, size_t off7, size_t off5, int B) {
    constexpr int TW = 64, TH = 16, R = 3, NR = TH + 2 * R, NWD = TW / 4 + 2;      // 22 input rows x 18 words (columns x0 - 4 .. x0 + 67)
    __shared__ uint32_t s_in[NR][NWD + 1];
    __shared__ __attribute__((aligned(8))) uint16_t s_h7[NR][TW + 4], s_h5[NR][TW + 4];
    int t7[7], t5[5];
    for (int i = 0; i < 7; i++) t7[i] = taps_g[i];
    for (int i = 0; i < 5; i++) t5[i] = taps_g[8 + i];
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH, tid = threadIdx.x;
    int b, tile;
    xcd_frame_block(tiles_x * tiles_y, B, b, tile);          // neighbouring tiles share their halos: a frame's tiles on one XCD (common.h)
    const int x0 = (tile % tiles_x) * TW, y0 = (tile / tiles_x) * TH;
    const uint8_t* S = src + (int64_t)b * src_stride;
    for (int i = tid; i < NR * NWD; i += 256) {
        const int r = i / NWD, wq = i - r * NWD;
        const uint8_t* row = S + (int64_t)reflect101(y0 + r - R, H) * pitch;
        const int xb = x0 - 4 + 4 * wq;
        uint32_t v;
        if (xb >= 0 && xb + 3 < W) v = *(const uint32_t*)(row + xb);
        else v = (uint32_t)row[reflect101(xb, W)] | ((uint32_t)row[reflect101(xb + 1, W)] << 8) | ((uint32_t)row[reflect101(xb + 2, W)] << 16) | ((uint32_t)row[reflect101(xb + 3, W)] << 24);
        s_in[r][wq] = v;
    }
    __syncthreads();
    for (int i = tid; i < NR * (TW / 4); i += 256) {           // horizontal sums of outputs 4 * q .. 4 * q + 3 of row r: bytes 4 * q + 1 .. 4 * q + 10 of the word row
        const int r = i / (TW / 4), q = i - r * (TW / 4);
        const uint32_t w0 = s_in[r][q], w1 = s_in[r][q + 1], w2 = s_in[r][q + 2];
        uint32_t px[12];
#pragma unroll
        for (int k = 0; k < 4; k++) { px[k] = (w0 >> (8 * k)) & 0xffu; px[4 + k] = (w1 >> (8 * k)) & 0xffu; px[8 + k] = (w2 >> (8 * k)) & 0xffu; }
        uint32_t h7[4], h5[4];
#pragma unroll
        for (int o = 0; o < 4; o++) {                          // output column 4 * q + o = word-row byte 4 + 4 * q + o; taps reach bytes o + 1 .. o + 7 of px
            uint32_t a = 0, c = 0;
#pragma unroll
            for (int k = 0; k < 7; k++) a += (uint32_t)t7[k] * px[o + 1 + k];
#pragma unroll
            for (int k = 0; k < 5; k++) c += (uint32_t)t5[k] * px[o + 2 + k];
            h7[o] = min(a, 65535u); h5[o] = min(c, 65535u);
        }
        *(uint2*)&s_h7[r][4 * q] = make_uint2(h7[0] | (h7[1] << 16), h7[2] | (h7[3] << 16));
        *(uint2*)&s_h5[r][4 * q] = make_uint2(h5[0] | (h5[1] << 16), h5[2] | (h5[3] << 16));
    }
    __syncthreads();
    {
        const int r = tid >> 4, q = tid & 15;                  // 16 rows x 16 groups of four columns
        const int x = x0 + 4 * q, y = y0 + r;
        if (x < W && y < H) {
            uint32_t a[4] = {0, 0, 0, 0}, c[4] = {0, 0, 0, 0};
#pragma unroll
            for (int k = 0; k < 7; k++) {
                const uint2 v = *(const uint2*)&s_h7[r + k][4 * q];
                a[0] += (uint32_t)t7[k] * (v.x & 0xffffu); a[1] += (uint32_t)t7[k] * (v.x >> 16); a[2] += (uint32_t)t7[k] * (v.y & 0xffffu); a[3] += (uint32_t)t7[k] * (v.y >> 16);
            }
#pragma unroll
            for (int k = 0; k < 5; k++) {
                const uint2 v = *(const uint2*)&s_h5[r + 1 + k][4 * q];
                c[0] += (uint32_t)t5[k] * (v.x & 0xffffu); c[1] += (uint32_t)t5[k] * (v.x >> 16); c[2] += (uint32_t)t5[k] * (v.y & 0xffffu); c[3] += (uint32_t)t5[k] * (v.y >> 16);
            }
            uint32_t o7 = 0, o5 = 0;
#pragma unroll
            for (int o = 0; o < 4; o++) { o7 |= min((a[o] + 32768u) >> 16, 255u) << (8 * o); o5 |= min((c[o] + 32768u) >> 16, 255u) << (8 * o); }
            uint8_t* F = ws + (size_t)b * frame_bytes;
            *(uint32_t*)(F + off7 + (size_t)y * W + x) = o7;
            *(uint32_t*)(F + off5 + (size_t)y * W + x) = o5;
        }
    }
}

// ---- K2: INTER_LINEAR_EXACT 0.8x resample + ll_angle gradient ------------------------------------------------------
struct Coef { int ofs, c0, c1; };

__device__ __forceinline__ int scaled_px(const uint8_t* __restrict__ B7, int W, const Coef& cx, const Coef& cy) {
    const uint8_t* r0 = B7 + (size_t)cy.ofs * W + cx.ofs;
    const uint8_t* r1 = r0 + W;
    const uint32_t h0 = min((uint32_t)cx.c0 * r0[0] + (uint32_t)cx.c1 * r0[1], 65535u);
    const uint32_t h1 = min((uint32_t)cx.c0 * r1[0] + (uint32_t)cx.c1 * r1[1], 65535u);
    const uint32_t v = h0 * (uint32_t)cy.c0 + h1 * (uint32_t)cy.c1;
    return (int)min((v + 32768u) >> 16, 255u);
}

// (float)cos / sin of the FP64 angle of a level-line angle in degrees: what a pixel contributes when it SEEDS a region (lsd.cpp region_grow: sumdx = cos(reg_angle), ...).
// Rounds 1-4 stored the pair for every pixel (8 B per pixel written by lsd_grad, two FP64 trigonometric calls each); only a few thousand pixels per frame ever seed.
__device__ __forceinline__ float2 seed_cos_sin(float deg) {
    const double a = (double)deg * DEG_TO_RADS;
    return make_float2((float)cos(a), (float)sin(a));
}

__global__ __launch_bounds__(256) void lsd_grad(const Plan* __restrict__ plan, const Coef* __restrict__ cxs, const Coef* __restrict__ cys,
                                                uint8_t* __restrict__ ws, Misc* __restrict__ miscs) {
    // the 65 x 5 pixels of the 0.8x image this workgroup's 64 x 4 gradients need, each resampled ONCE (rounds 1-4: four times, by the four gradients that share it)
    __shared__ uint8_t s_px[5][68];
    const Plan& P = *plan;
    const int b = blockIdx.z;
    const int x0 = blockIdx.x * 64, y0 = blockIdx.y * 4;
    uint8_t* F = ws + (size_t)b * P.frame_bytes;
    const uint8_t* B7 = F + P.off_blur7;
    for (int i = threadIdx.x; i < 5 * 65; i += 256) {
        const int r = i / 65, c = i - r * 65;
        const int xs = min(x0 + c, P.w - 1), ys = min(y0 + r, P.h - 1);
        s_px[r][c] = (uint8_t)scaled_px(B7, P.W, cxs[xs], cys[ys]);
    }
    __syncthreads();
    const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
    const int x = x0 + lx, y = y0 + ly;
    if (x >= P.w || y >= P.h) return;
    float* ang = (float*)(F + P.off_ang);
    uint32_t* g2a = (uint32_t*)(F + P.off_g2);
    float4* pix4 = (float4*)(F + P.off_pix);      // {angle deg, cosf(angle), sinf(angle), compact index}: one 16-byte gather per neighbour
    Misc* misc = miscs + b;
    const size_t o = (size_t)y * P.w + x;
    if (x >= P.w - 1 || y >= P.h - 1) { ang[o] = NOTDEF_F; g2a[o] = 0; pix4[o] = make_float4(NOTDEF_F, 0.f, 0.f, 0.f); return; }
    const int s00 = s_px[ly][lx], s10 = s_px[ly][lx + 1], s01 = s_px[ly + 1][lx], s11 = s_px[ly + 1][lx + 1];
    const int DA = s11 - s00, BC = s10 - s01;
    const int gx = DA + BC, gy = DA - BC;
    const int g2 = gx * gx + gy * gy;
    const double norm = sqrt(g2 / 4.0);
    g2a[o] = (uint32_t)g2;
    if (norm <= P.rho) { ang[o] = NOTDEF_F; pix4[o] = make_float4(NOTDEF_F, 0.f, 0.f, 0.f); }
    else {
        const float deg = fast_atan2_deg((float)gx, (float)(-gy));
        ang[o] = deg;
        const double a = (double)deg * DEG_TO_RADS;
        const float af = (float)a;
        pix4[o] = make_float4(deg, (float)cos((double)af), (float)sin((double)af), 0.f);   // .w: compact index, filled by lsd_sort_compact
        atomicMax(&misc->g2max, (uint32_t)g2);
    }
}

// ---- K3: the visiting order: descending 1024-bin norm, ties as std::sort leaves them ----------------------------------------------------
template <int NW>   // exclusive scan over a workgroup of NW wavefronts
__device__ inline int block_exscan(int v, int* wsum, int* total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int inc = v;
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
    __syncthreads();
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    int base = 0, tot = 0;
    for (int i = 0; i < NW; i++) { if (i < w) base += wsum[i]; tot += wsum[i]; }
    *total = tot;
    return base + inc - v;
}

__device__ __forceinline__ void wave_sync();
__device__ __forceinline__ void lds_sync();
// lsd.cpp sorts all (w-1)(h-1) gradient pixels by their 1024-bin norm with std::sort(compare_norm: a.norm > b.norm), so pixels of equal bin are
// visited in the order libstdc++'s introsort leaves them.  isort.h computes that arrangement in parallel (tie_order 0): words are
// (1023 - bin) << 20 | pixel, ascending.  Three launches:
//   lsd_sort_global   one workgroup per frame: the words, a bitmap of the pixels with a defined angle, and the partitions of the ranges longer than
//                     an LDS block (bitmaps of the scans' stops in LDS: the array is read once per level, only swapped words are written);
//   lsd_sort_lds      one workgroup per block of <= 23 552 words, up to SORT_R per frame: everything below, in LDS, written back sorted;
//   lsd_sort_compact  one workgroup per frame: drops the undefined pixels (they took part in the partitions), numbers the rest.
// Checked against the real std::sort through the oracle (tests/test_lsd_gpu.py) and on the emulator (tests/test_isort_emul.py).
#ifndef PLANAR_WIDE_T
#define PLANAR_WIDE_T 1024      // (developer build `make narrow`: 256 - the round-6 co-residency experiment)
#endif
constexpr int SORT_T = PLANAR_WIDE_T, SORT_LT = 256, SORT_E = 23, SORT_SHIFT = 20, SORT_R = 64;      // SORT_T: threads of the global tier and the compaction; SORT_LT x SORT_E: an LDS block
constexpr int SORT_HJOBS = 1024, SORT_HCAP = 8192, SORT_HY = 2;       // heap-sort fallback: jobs per frame, words of a job kept in LDS, workgroups per frame
using SortLds = isort::LdsLayout<SORT_LT, SORT_E>;
using SortGl = isort::GlobalLayout<SORT_T>;

__global__ __launch_bounds__(SORT_T) void lsd_sort_global(const Plan* __restrict__ plan, uint8_t* __restrict__ ws, Misc* __restrict__ miscs, int rows_cap, int rows_long) {
    extern __shared__ __align__(16) uint8_t sort_lds[];
    __shared__ isort::Range s_init;
    const Plan& P = *plan;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    uint8_t* F = ws + (size_t)b * P.frame_bytes;
    const uint32_t* g2a = (const uint32_t*)(F + P.off_g2);
    uint32_t* arr = (uint32_t*)(F + P.off_tmp);
    unsigned long long* vb = (unsigned long long*)(F + P.off_valid);
    Misc* misc = miscs + b;
    const int w1 = P.w - 1, h1 = P.h - 1, n = w1 * h1, npix = P.w * P.h;
    const double max_grad = misc->g2max ? sqrt(misc->g2max / 4.0) : -1.0;
    const double bin_coef = (max_grad > 0) ? double(N_BINS - 1) / max_grad : 0;
    const long long ts0 = __builtin_readcyclecounter();
    __shared__ unsigned int s_kv, s_prefix;
    if (tid == 0) { s_kv = 0u; s_prefix = 0u; }
    __syncthreads();
    uint32_t kv = 0;                                        // the largest key of a pixel with a defined angle (they are the only ones lsd_sort_compact keeps)
    for (int p0 = tid - lane; p0 < npix; p0 += SORT_T) {
        const int pix = p0 + lane, y = pix / P.w, x = pix - y * P.w;
        const bool in = pix < npix && x < w1 && y < h1;
        bool valid = false;
        if (in) {
            const double norm = sqrt(g2a[pix] / 4.0);
            const int bin = int(norm * bin_coef);
            arr[pix - y] = ((uint32_t)(N_BINS - 1 - bin) << SORT_SHIFT) | (uint32_t)pix;
            valid = !(norm <= P.rho);                        // = (ang[pix] != NOTDEF): lsd_grad's own test on the same integer, without reading the angle image
            if (valid) kv = max(kv, (uint32_t)(N_BINS - 1 - bin));
        }
        const unsigned long long m = __ballot(valid);
        if (lane == 0) vb[p0 >> 6] = m;
    }
    kv = (uint32_t)planar::wave_max_f64((double)kv);        // (exact: a 10-bit integer)
    if (lane == 0) atomicMax(&s_kv, kv);
    __syncthreads();
    kv = s_kv;
    // The pixels without an angle took part in OpenCV's std::sort, but nothing reads where they end up; they are 85-90 % of the words and have the largest keys.  Ranges
    // that provably hold only keys above kv are left unsorted (isort.h: skip_key) - the wanted words still land exactly where std::sort puts them: the first `prefix` places.
    int cnt = 0;
    for (int p0 = tid - lane; p0 < npix; p0 += SORT_T) {
        const int pix = p0 + lane, y = pix / P.w, x = pix - y * P.w;
        if (pix < npix && x < w1 && y < h1) cnt += ((arr[pix - y] >> SORT_SHIFT) <= kv) ? 1 : 0;
    }
    cnt = planar::wave_sum_i32(cnt);
    if (lane == 0) atomicAdd(&s_prefix, (unsigned int)cnt);
    if (tid == 0) s_init = isort::Range{0, n, isort::depth_limit(n)};
    __threadfence_block();
    __syncthreads();
    if (tid == 0) { misc->sort_kv = (int)kv; misc->sort_prefix = (int)s_prefix; }
    const isort::HeapSink HS{(isort::HeapJob*)(F + P.off_heapj), &misc->heap_n, SORT_HJOBS};
    isort::global_tier<SORT_SHIFT, SORT_T>(arr, &s_init, 1, SortLds::N, 64, (isort::Range*)(F + P.off_sortr), (isort::Block*)(F + P.off_sortb), isort::G_FMAX,
                                           misc->sort_counts, sort_lds, rows_cap, HS, &misc->status, kv,
                                           rows_long, (uint32_t*)(F + P.off_ord), n / 2 + 1);      // (frames beyond ~390 000 words: the long partition's scratch is the visiting-order array, not yet written)
    if (tid == 0) misc->t[5] = __builtin_readcyclecounter() - ts0;
}

// (four wavefronts per SIMD = four workgroups per CU, what their 40 KB of LDS allow: 128 VGPRs with 23 spilled measure 13 % faster than 163 unspilled at three)
__global__ __launch_bounds__(SORT_LT, 4) void lsd_sort_lds(const Plan* __restrict__ plan, uint8_t* __restrict__ ws, Misc* __restrict__ miscs) {
    extern __shared__ __align__(16) uint8_t sort_lds[];
    const Plan& P = *plan;
    const int b = blockIdx.x;
    uint8_t* F = ws + (size_t)b * P.frame_bytes;
    Misc* misc = miscs + b;
    const isort::Range* ranges = (const isort::Range*)(F + P.off_sortr);
    const isort::Block* blocks = (const isort::Block*)(F + P.off_sortb);
    const int nb = misc->sort_counts[1];
    const isort::HeapSink HS{(isort::HeapJob*)(F + P.off_heapj), &misc->heap_n, SORT_HJOBS};
    const long long ts0 = __builtin_readcyclecounter();
    for (int k = blockIdx.y; k < nb; k += gridDim.y) {
        const isort::Block K = blocks[k];
        isort::lds_tier<SORT_SHIFT, SORT_LT, SORT_E>((uint32_t*)(F + P.off_tmp), ranges + K.r0, K.nr, K.f, K.l, sort_lds, HS, &misc->status, (uint32_t)misc->sort_kv);
    }
    if (threadIdx.x == 0 && blockIdx.y == 0) misc->t[6] = __builtin_readcyclecounter() - ts0;
}

// the ranges whose introsort depth budget ran out (none on gradient images so far; the engine is the voxel grid's, planepost.hip): one wavefront per job
__global__ __launch_bounds__(64) void lsd_sort_heap(const Plan* __restrict__ plan, uint8_t* __restrict__ ws, Misc* __restrict__ miscs) {
    extern __shared__ __align__(16) uint8_t sort_lds[];
    const Plan& P = *plan;
    uint8_t* F = ws + (size_t)blockIdx.x * P.frame_bytes;
    const int nj = min(miscs[blockIdx.x].heap_n, SORT_HJOBS);
    if (nj) isort::heap_jobs<SORT_SHIFT>((uint32_t*)(F + P.off_tmp), (const isort::HeapJob*)(F + P.off_heapj), nj, blockIdx.y, gridDim.y, (uint32_t*)sort_lds, SORT_HCAP, 0, 1 << 30);
}

__global__ __launch_bounds__(SORT_T) void lsd_sort_compact(const Plan* __restrict__ plan, uint8_t* __restrict__ ws, Misc* __restrict__ miscs) {
    __shared__ int s_cnt[SORT_T / 64];
    const Plan& P = *plan;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint8_t* F = ws + (size_t)b * P.frame_bytes;
    const uint32_t* arr = (const uint32_t*)(F + P.off_tmp);
    const unsigned long long* vb = (const unsigned long long*)(F + P.off_valid);
    uint32_t* ord = (uint32_t*)(F + P.off_ord);
    uint32_t* ordr = (uint32_t*)(F + P.off_ordr);
    float* pixw = (float*)(F + P.off_pix);
    Misc* misc = miscs + b;
    const int n = min((P.w - 1) * (P.h - 1), misc->sort_prefix);      // the words with a key <= the largest defined pixel's: the sorted front of the array (lsd_sort_global)
    const long long ts0 = __builtin_readcyclecounter();
    constexpr int NW = SORT_T / 64, U = 4;
    const int seg = ((n + NW - 1) / NW + 63) & ~63, w0 = min(n, wave * seg), w1 = min(n, w0 + seg);
    auto is_valid = [&](uint32_t e, bool in) { const uint32_t pix = e & 0xfffffu; return in && ((vb[pix >> 6] >> (pix & 63u)) & 1ull); };
    int cnt = 0;
    for (int i0 = w0; i0 < w1; i0 += 64 * U) {
        uint32_t e[U];
#pragma unroll
        for (int u = 0; u < U; u++) e[u] = arr[min(i0 + 64 * u + lane, w1 - 1)];
#pragma unroll
        for (int u = 0; u < U; u++) cnt += __popcll(__ballot(is_valid(e[u], i0 + 64 * u + lane < w1)));
    }
    if (lane == 0) s_cnt[wave] = cnt;
    __syncthreads();
    int pos = 0, total = 0;
    for (int q = 0; q < NW; q++) { if (q < wave) pos += s_cnt[q]; total += s_cnt[q]; }
    for (int i0 = w0; i0 < w1; i0 += 64 * U) {
        uint32_t e[U];
#pragma unroll
        for (int u = 0; u < U; u++) e[u] = arr[min(i0 + 64 * u + lane, w1 - 1)];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const bool v = is_valid(e[u], i0 + 64 * u + lane < w1);
            const unsigned long long m = __ballot(v);
            if (v) {
                const uint32_t pix = e[u] & 0xfffffu, slot = (uint32_t)(pos + __popcll(m & ((1ull << lane) - 1ull)));
                ord[slot] = pix; ordr[slot] = slot;
                pixw[(size_t)pix * 4 + 3] = __uint_as_float(slot);     // the pixel's compact index: its `used` flag lives there
            }
            pos += __popcll(m);
        }
    }
    if (tid == 0) { misc->n_ord = total; misc->t[7] = __builtin_readcyclecounter() - ts0; }
}

// tie_order 1 (raster order inside a bin; a selectable alternative to the library's order): a stable sort = two 5-bit LSD radix passes.  Every
// wavefront owns a contiguous part of the array and walks it 64 elements at a time (coalesced); the rank of an element among the equal digits of its
// chunk is a popcount over the match mask built from five ballots, the running (digit, wavefront) counters live in LDS.  Undefined pixels are dropped in
// the first pass.
constexpr int SORT_NT = 256, SORT_NW = SORT_NT / 64;
__global__ __launch_bounds__(SORT_NT) void lsd_sort_raster(const Plan* __restrict__ plan, uint8_t* __restrict__ ws, Misc* __restrict__ miscs) {
    __shared__ int wsum[SORT_NW];
    __shared__ int cnt[32 * SORT_NW];
    const Plan& P = *plan;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint8_t* F = ws + (size_t)b * P.frame_bytes;
    const float* ang = (const float*)(F + P.off_ang);
    const uint32_t* g2a = (const uint32_t*)(F + P.off_g2);
    uint32_t* arr = (uint32_t*)(F + P.off_tmp);                 // bin << 20 | pixel, all (w-1)(h-1) gradient pixels
    uint32_t* ord = (uint32_t*)(F + P.off_ord);
    uint32_t* ordr = (uint32_t*)(F + P.off_ordr);
    uint32_t* tmpA = (uint32_t*)(F + P.off_reg);
    Misc* misc = miscs + b;
    const int w1 = P.w - 1, n = w1 * (P.h - 1);
    const double max_grad = misc->g2max ? sqrt(misc->g2max / 4.0) : -1.0;
    const double bin_coef = (max_grad > 0) ? double(N_BINS - 1) / max_grad : 0;
    for (int i = tid; i < n; i += SORT_NT) {
        const int y = i / w1, x = i - y * w1, pix = y * P.w + x;
        arr[i] = ((uint32_t)int(sqrt(g2a[pix] / 4.0) * bin_coef) << 20) | (uint32_t)pix;
    }
    __threadfence_block();
    __syncthreads();
    const long long ts2 = __builtin_readcyclecounter();
    int* wcnt = cnt;                                          // [32][SORT_NW] counters, digit-major = output order
    auto radix_pass = [&](const uint32_t* in, int M, auto digit_of, auto keep, auto emit) -> int {
        const int seg = ((M + SORT_NW - 1) / SORT_NW + 63) & ~63, w0 = min(M, wave * seg), w1 = min(M, w0 + seg);
        for (int t = tid; t < 32 * SORT_NW; t += SORT_NT) wcnt[t] = 0;
        __syncthreads();
        auto match = [&](bool valid, int d) -> unsigned long long {
            unsigned long long same = __ballot(valid);
#pragma unroll
            for (int bit = 0; bit < 5; bit++) { const unsigned long long bm = __ballot(valid && ((d >> bit) & 1)); same &= ((d >> bit) & 1) ? bm : ~bm; }
            return same;
        };
        for (int i0 = w0; i0 < w1; i0 += 64) {
            const int i = i0 + lane;
            const uint32_t e = in[min(i, w1 - 1)];
            const bool valid = i < w1 && keep(e);
            const int d = digit_of(e);
            const unsigned long long same = match(valid, d);
            if (valid && (same & ((1ull << lane) - 1ull)) == 0) wcnt[d * SORT_NW + wave] += __popcll(same);   // leader of its digit group
            lds_sync();
        }
        __syncthreads();
        int total;
        {   // exclusive scan of the 32 x SORT_NW counters in (digit, wavefront) order by the first wavefronts
            const int v = tid < 32 * SORT_NW ? wcnt[tid] : 0;
            const int ex = block_exscan<SORT_NW>(v, wsum, &total);
            if (tid < 32 * SORT_NW) wcnt[tid] = ex;
        }
        __syncthreads();
        for (int i0 = w0; i0 < w1; i0 += 64) {
            const int i = i0 + lane;
            const uint32_t e = in[min(i, w1 - 1)];
            const bool valid = i < w1 && keep(e);
            const int d = digit_of(e);
            const unsigned long long same = match(valid, d);
            const unsigned long long below = same & ((1ull << lane) - 1ull);
            int base = 0;
            if (valid) base = wcnt[d * SORT_NW + wave];
            lds_sync();
            if (valid) {
                emit(e, i, base + __popcll(below));
                if (below == 0) wcnt[d * SORT_NW + wave] = base + __popcll(same);
            }
            lds_sync();
        }
        __syncthreads();
        return total;
    };
    static_assert(32 * SORT_NW <= SORT_NT, "one thread per radix counter");
    float* pixw = (float*)(F + P.off_pix);
    const int N = radix_pass(arr, n,
        [&](uint32_t e) { return ((N_BINS - 1) - (int)(e >> 20)) & 31; },
        [&](uint32_t e) { return ang[e & 0xfffffu] != NOTDEF_F; },
        [&](uint32_t e, int, int slot) {
            const uint32_t pix = e & 0xfffffu;
            tmpA[slot] = ((uint32_t)((N_BINS - 1) - (int)(e >> 20)) << 20) | pix;
            pixw[(size_t)pix * 4 + 3] = __uint_as_float((uint32_t)slot);   // the slot doubles as the pixel's compact index (its `used` flag lives there)
        });
    __threadfence_block();
    __syncthreads();
    radix_pass(tmpA, N,
        [&](uint32_t e) { return (int)((e >> 25) & 31); },
        [&](uint32_t) { return true; },
        [&](uint32_t e, int i, int pos) { ord[pos] = e & 0xfffffu; ordr[pos] = (uint32_t)i; });
    if (tid == 0) { misc->n_ord = N; misc->t[7] = __builtin_readcyclecounter() - ts2; }
}

// ---- K4: the sequential detector, one wavefront per frame -----------------------------------------------------------
struct Rect { double x1, y1, x2, y2, width, x, y, theta, dx, dy, prec, p; int pj; };   // pj: p == P.p / 2^pj

struct Det {
    const float* ang;
    const uint32_t* g2;
    const float4* pix4;
    const Plan* plan;
    uint32_t* reg;      // region points, x | y << 16, in growth order
    uint32_t* tmp;      // scratch (reduce_region_radius)
    int w, h;
    double log_nt;
    // LDS
    uint32_t* used;     // bitmap over compact indices < USED_LDS_BITS (LDS)
    uint32_t* gused;    // bitmap over all compact indices (global; only indices >= USED_LDS_BITS are kept here)
    uint32_t* ring;
    double* stage;      // [64][3]
    int lane;
};

// `used` flags, addressed by the compact index r of a defined pixel.  The global tail is only touched by images with more than
// USED_LDS_BITS defined pixels; there every update is followed by an agent-scope fence and reads are agent-scope atomic loads.
__device__ __forceinline__ bool used_get(const Det& D, uint32_t r) {
    if (r < (uint32_t)USED_LDS_BITS) return (D.used[r >> 5] >> (r & 31)) & 1u;      // plain ds_read: callers order it with lds_sync()
    return (__hip_atomic_load(D.gused + (r >> 5), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> (r & 31)) & 1u;
}
__device__ __forceinline__ void used_set(const Det& D, uint32_t r) {
    if (r < (uint32_t)USED_LDS_BITS) atomicOr(&D.used[r >> 5], 1u << (r & 31));
    else { atomicOr(D.gused + (r >> 5), 1u << (r & 31)); __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent"); }
}
__device__ __forceinline__ void used_clear(const Det& D, uint32_t r) {
    if (r < (uint32_t)USED_LDS_BITS) atomicAnd(&D.used[r >> 5], ~(1u << (r & 31)));
    else { atomicAnd(D.gused + (r >> 5), ~(1u << (r & 31))); __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent"); }
}
__device__ __forceinline__ uint32_t rank_of(const Det& D, int pix) { return __float_as_uint(D.pix4[pix].w); }
__device__ __forceinline__ double dist2(double x1, double y1, double x2, double y2) { return (x2 - x1) * (x2 - x1) + (y2 - y1) * (y2 - y1); }
__device__ __forceinline__ double angle_diff_signed(double a, double b) {
    double diff = a - b;
    while (diff <= -LSD_PI) diff += M_2__PI;
    while (diff > LSD_PI) diff -= M_2__PI;
    return diff;
}
__device__ __forceinline__ bool aligned_rad(double a, double theta, double prec) {
    double n_theta = theta - a;
    if (n_theta < 0) n_theta = -n_theta;
    if (n_theta > M_3_2_PI) {
        n_theta -= M_2__PI;
        if (n_theta < 0) n_theta = -n_theta;
    }
    return n_theta <= prec;
}
__device__ __forceinline__ void wave_sync() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); __builtin_amdgcn_wave_barrier(); }
// LDS traffic of one wavefront is processed in program order, so lanes only need the compiler to keep that order
__device__ __forceinline__ void lds_sync() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); }

// region_grow: returns the region size; reg_angle in/out per the reference (out: final level-line angle).
// Queue entries are expanded 7 at a time (63 lanes = 7 entries x 9 neighbours, in the reference's order).  The neighbour records of the
// NEXT batch (entries already queued behind the current one; any batching gives the same sequence) are requested before the current
// batch is resolved, so their memory latency hides behind the sequential accept loop.  Inside the loop nothing waits on LDS: `used` is read
// once per batch, duplicates of an accepted pixel inside the batch are masked by comparing compact indices in registers.
struct GrowBatch { bool ok; uint32_t nxy; float4 rec4; };
__device__ __forceinline__ GrowBatch grow_fetch(const Det& D, int head, int nb, int reg_n) {
    const int lane = D.lane, w = D.w, h = D.h;
    const int e = lane / 9, k = lane - e * 9;
    GrowBatch g;
    g.ok = false; g.nxy = 0; g.rec4 = make_float4(NOTDEF_F, 0.f, 0.f, 0.f);
    if (lane < 63 && e < nb && k != 4) {
        const int i = head + e;
        // the LDS read is unconditional: if both sources sit behind one branch the compiler selects the ADDRESS and emits a single FLAT
        // load, whose wait also covers every outstanding global load (the next batch's records requested ahead)
        uint32_t pxy = D.ring[i & (RING - 1)];
        asm volatile("" : "+v"(pxy));                                         // (keeps the two loads apart)
        if (reg_n - i > RING) pxy = __builtin_nontemporal_load(D.reg + i);   // (rare) the entry left the ring
        const int xx = (int)(pxy & 0xffff) + (k % 3) - 1, yy = (int)(pxy >> 16) + (k / 3) - 1;
        if (xx >= 0 && xx < w && yy >= 0 && yy < h) { g.ok = true; g.nxy = (uint32_t)xx | ((uint32_t)yy << 16); g.rec4 = D.pix4[yy * w + xx]; }
    }
    return g;
}
// seed_deg / seed_cs: ang[seed_pix] and seedcs[seed_pix] (the caller fetches them for 64 seed candidates at a time).  The region list in
// global memory (D.reg) is written by lane 0 and read back by other lanes of this wavefront: callers that read it issue wave_sync() first.
__device__ int region_grow(const Det& D, int seed_pix, uint32_t seed_rank, float seed_deg, float2 seed_cs, double prec, double& reg_angle) {
    const int lane = D.lane;
    reg_angle = (double)seed_deg * DEG_TO_RADS;
    float sumdx = seed_cs.x, sumdy = seed_cs.y;
    const uint32_t seed_xy = (uint32_t)(seed_pix % D.w) | ((uint32_t)(seed_pix / D.w) << 16);
    if (lane == 0) { D.reg[0] = seed_xy; D.ring[0] = seed_xy; used_set(D, seed_rank); }
    lds_sync();
    int reg_n = 1, head = 0, nb = 1;
    GrowBatch cur = grow_fetch(D, 0, 1, 1);
    // Most regions are a few dozen pixels and their frontier is too short to prefetch from: touch the cache lines of the 16x16 window around
    // the seed now (issued behind the first batch, never waited for before the region is done), so the following batches hit the L2.
    float4 warm = make_float4(0.f, 0.f, 0.f, 0.f);
    if (lane < 48) {
        const int wy = min(max((int)(seed_xy >> 16) - 8 + lane / 3, 0), D.h - 1), wx = min(max((int)(seed_xy & 0xffff) - 8 + (lane % 3) * 8, 0), D.w - 1);
        warm = D.pix4[wy * D.w + wx];
    }
    while (true) {
        const int head_n = head + nb;
        int nb_n = min(7, reg_n - head_n);
        if (reg_n - head_n > RING) nb_n = 0;       // (rare) entries that left the LDS ring are fetched after the wave_sync below
        GrowBatch nxt;
        if (nb_n > 0) nxt = grow_fetch(D, head_n, nb_n, reg_n);
        {
            const float deg = cur.rec4.x, c = cur.rec4.y, sn = cur.rec4.z;
            const uint32_t rk = __float_as_uint(cur.rec4.w);
            bool ok = cur.ok && deg != NOTDEF_F;
            ok = ok && !used_get(D, rk);
            const double a = (double)deg * DEG_TO_RADS;
            int cursor = 0;
            while (true) {
                const bool cand = ok && lane >= cursor && aligned_rad(a, reg_angle, prec);
                const unsigned long long m = __ballot(cand);
                if (!m) break;
                const int f = __builtin_amdgcn_readfirstlane(__ffsll((long long)m) - 1);
                const uint32_t axy = (uint32_t)__builtin_amdgcn_readlane((int)cur.nxy, f);
                const uint32_t ark = (uint32_t)__builtin_amdgcn_readlane((int)rk, f);
                if (lane == 0) { used_set(D, ark); D.reg[reg_n] = axy; D.ring[reg_n & (RING - 1)] = axy; }
                reg_n++;
                sumdx += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c), f));
                sumdy += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sn), f));
                reg_angle = (double)fast_atan2_deg(sumdy, sumdx) * DEG_TO_RADS;
                ok = ok && rk != ark;              // the same pixel as a neighbour of another entry of this batch: now used
                cursor = f + 1;
            }
        }
        lds_sync();                                // ring / used updates of this batch before the next batch reads them
        head = head_n;
        if (head >= reg_n) break;
        if (nb_n > 0) { cur = nxt; nb = nb_n; }
        else {
            if (reg_n - head > RING) wave_sync();  // the batch reads entries that left the LDS ring: make the global copies visible
            nb = min(7, reg_n - head);
            cur = grow_fetch(D, head, nb, reg_n);
        }
    }
    asm volatile("" :: "v"(warm.x));               // keeps the window loads alive
    lds_sync();
    return reg_n;
}

// three sequential FP64 chains (one per lane 0..2) over per-point triples, in region order: `acc[c] (+|-)= v[c]`
// vals() fills the triple of point i; sign[c] = -1 makes chain c a subtraction chain.
template <typename F>
__device__ inline void chains3(const Det& D, int n, double out[3], bool sub2, F vals) {
    const int lane = D.lane;
    double acc = 0;
    for (int base = 0; base < n; base += 64) {
        const int i = base + lane;
        if (i < n) { double v[3]; vals(i, v); D.stage[lane * 3] = v[0]; D.stage[lane * 3 + 1] = v[1]; D.stage[lane * 3 + 2] = v[2]; }
        lds_sync();
        const int cnt = min(64, n - base);
        if (lane < 3) {
            // the additions are a chain, the LDS reads are not: eight values are requested at a time
            const double* st = D.stage + lane;
            const bool neg = lane == 2 && sub2;
            int j = 0;
            for (; j + 8 <= cnt; j += 8) {
                double v[8];
#pragma unroll
                for (int u = 0; u < 8; u++) v[u] = st[(j + u) * 3];
#pragma unroll
                for (int u = 0; u < 8; u++) { if (neg) acc -= v[u]; else acc += v[u]; }
            }
            for (; j < cnt; j++) { if (neg) acc -= st[j * 3]; else acc += st[j * 3]; }
        }
        lds_sync();
    }
    out[0] = planar::wave_lane(acc, 0); out[1] = planar::wave_lane(acc, 1); out[2] = planar::wave_lane(acc, 2);
}

// reductions on the DPP ladder (wave_ops.h): a __shfl_xor butterfly is six LDS-pipe permutes per 32-bit word
__device__ inline double wave_max_d(double v) { return planar::wave_max_f64(v); }
__device__ inline double wave_min_d(double v) { return planar::wave_min_f64(v); }
__device__ inline int wave_sum_i(int v) { return planar::wave_sum_i32(v); }

__device__ __forceinline__ double modgrad_of(const Det& D, uint32_t pxy) { return sqrt(D.g2[(pxy >> 16) * D.w + (pxy & 0xffff)] / 4.0); }

__device__ void region2rect(const Det& D, int n, double reg_angle, double prec, double p, Rect& rec) {
    double s3[3];
    chains3(D, n, s3, false, [&](int i, double* v) {
        const uint32_t e = D.reg[i];
        const double wgt = modgrad_of(D, e);
        v[0] = double(e & 0xffff) * wgt; v[1] = double(e >> 16) * wgt; v[2] = wgt;
    });
    const double x = s3[0] / s3[2], y = s3[1] / s3[2];
    // get_theta
    chains3(D, n, s3, true, [&](int i, double* v) {
        const uint32_t e = D.reg[i];
        const double wgt = modgrad_of(D, e);
        const double dx = double(e & 0xffff) - x, dy = double(e >> 16) - y;
        v[0] = dy * dy * wgt; v[1] = dx * dx * wgt; v[2] = dx * dy * wgt;
    });
    const double Ixx = s3[0], Iyy = s3[1], Ixy = s3[2];
    const double lambda = 0.5 * (Ixx + Iyy - sqrt((Ixx - Iyy) * (Ixx - Iyy) + 4.0 * Ixy * Ixy));
    double theta = (fabs(Ixx) > fabs(Iyy)) ? double(fast_atan2_deg(float(lambda - Ixx), float(Ixy))) : double(fast_atan2_deg(float(Ixy), float(lambda - Iyy)));
    theta *= DEG_TO_RADS;
    if (fabs(angle_diff_signed(theta, reg_angle)) > prec) theta += LSD_PI;
    const double dx = cos(theta), dy = sin(theta);
    double l_min = 0, l_max = 0, w_min = 0, w_max = 0;
    for (int i = D.lane; i < n; i += 64) {
        const uint32_t e = D.reg[i];
        const double regdx = double(e & 0xffff) - x, regdy = double(e >> 16) - y;
        const double l = regdx * dx + regdy * dy;
        const double w = -regdx * dy + regdy * dx;
        l_max = fmax(l_max, l); l_min = fmin(l_min, l); w_max = fmax(w_max, w); w_min = fmin(w_min, w);
    }
    l_max = wave_max_d(l_max); l_min = wave_min_d(l_min); w_max = wave_max_d(w_max); w_min = wave_min_d(w_min);
    rec.x1 = x + l_min * dx; rec.y1 = y + l_min * dy;
    rec.x2 = x + l_max * dx; rec.y2 = y + l_max * dy;
    rec.width = w_max - w_min;
    rec.x = x; rec.y = y; rec.theta = theta; rec.dx = dx; rec.dy = dy; rec.prec = prec; rec.p = p; rec.pj = 0;
    if (rec.width < 1.0) rec.width = 1.0;
}

__device__ bool reduce_region_radius(const Det& D, int& n, double reg_angle, double prec, double p, Rect& rec, double density, double density_th) {
    const int lane = D.lane;
    const uint32_t e0 = D.reg[0];
    const double xc = double(e0 & 0xffff), yc = double(e0 >> 16);
    const double radSq1 = dist2(xc, yc, rec.x1, rec.y1), radSq2 = dist2(xc, yc, rec.x2, rec.y2);
    double radSq = radSq1 > radSq2 ? radSq1 : radSq2;
    while (density < density_th) {
        radSq *= 0.75 * 0.75;
        // the reference's swap-with-last removal == keep near points in place, fill each far slot below the new size with
        // the near points found scanning from the back (LSD)
        int n_near = 0;
        for (int i = lane; i < n; i += 64) {
            const uint32_t e = D.reg[i];
            const bool far = dist2(xc, yc, double(e & 0xffff), double(e >> 16)) > radSq;
            if (far) used_clear(D, rank_of(D, (e >> 16) * D.w + (e & 0xffff)));
            else n_near++;
        }
        n_near = wave_sum_i(n_near);
        int k = 0;
        for (int base = n - 1; base >= n_near; base -= 64) {          // back part, descending positions
            const int pos = base - lane;
            bool near = false;
            uint32_t e = 0;
            if (pos >= n_near) { e = D.reg[pos]; near = !(dist2(xc, yc, double(e & 0xffff), double(e >> 16)) > radSq); }
            const unsigned long long m = __ballot(near);
            if (near) D.tmp[k + __popcll(m & ((1ull << lane) - 1))] = e;
            k += __popcll(m);
        }
        wave_sync();
        k = 0;
        for (int base = 0; base < n_near; base += 64) {               // front part, ascending positions
            const int pos = base + lane;
            bool far = false;
            if (pos < n_near) { const uint32_t e = D.reg[pos]; far = dist2(xc, yc, double(e & 0xffff), double(e >> 16)) > radSq; }
            const unsigned long long m = __ballot(far);
            if (far) D.reg[pos] = D.tmp[k + __popcll(m & ((1ull << lane) - 1))];
            k += __popcll(m);
        }
        wave_sync();
        n = n_near;
        if (n < 2) return false;
        region2rect(D, n, reg_angle, prec, p, rec);
        density = double(n) / (sqrt(dist2(rec.x1, rec.y1, rec.x2, rec.y2)) * rec.width);
    }
    return true;
}

__device__ bool refine(const Det& D, int& n, double& reg_angle, double prec, double p, Rect& rec, double density_th) {
    double density = double(n) / (sqrt(dist2(rec.x1, rec.y1, rec.x2, rec.y2)) * rec.width);
    if (density >= density_th) return true;
    const uint32_t e0 = D.reg[0];
    const double xc = double(e0 & 0xffff), yc = double(e0 >> 16);
    const int seed_pix = (e0 >> 16) * D.w + (e0 & 0xffff);
    const double ang_c = (double)D.ang[seed_pix] * DEG_TO_RADS;
    const double width = rec.width;
    double s3[3];
    int cnt = 0;
    chains3(D, n, s3, false, [&](int i, double* v) {
        const uint32_t e = D.reg[i];
        const int pix = (e >> 16) * D.w + (e & 0xffff);
        used_clear(D, rank_of(D, pix));
        v[0] = 0; v[1] = 0; v[2] = 0;
        if (sqrt(dist2(xc, yc, double(e & 0xffff), double(e >> 16))) < width) {
            const double ang_d = angle_diff_signed((double)D.ang[pix] * DEG_TO_RADS, ang_c);
            v[0] = ang_d; v[1] = ang_d * ang_d; cnt++;
        }
    });
    // adding +0.0 for skipped points leaves the partial sums unchanged bit for bit (x + 0.0 == x, and sums never are -0.0 + ... issue:
    // the first addend 0.0 + v is exact), so the conditional chain equals the reference's
    cnt = wave_sum_i(cnt);
    const double sum = s3[0], s_sum = s3[1];
    const double mean_angle = sum / double(cnt);
    const double tau = 2.0 * sqrt((s_sum - 2.0 * mean_angle * sum) / double(cnt) + mean_angle * mean_angle);
    wave_sync();
    n = region_grow(D, seed_pix, rank_of(D, seed_pix), D.ang[seed_pix], seed_cos_sin(D.ang[seed_pix]), tau, reg_angle);
    wave_sync();
    if (n < 2) return false;
    region2rect(D, n, reg_angle, prec, p, rec);
    density = double(n) / (sqrt(dist2(rec.x1, rec.y1, rec.x2, rec.y2)) * rec.width);
    if (density < density_th) return reduce_region_radius(D, n, reg_angle, prec, p, rec, density, density_th);
    return true;
}

__device__ inline bool double_equal(double a, double b) {
    if (a == b) return true;
    const double abs_diff = fabs(a - b), aa = fabs(a), bb = fabs(b);
    double abs_max = (aa > bb) ? aa : bb;
    if (abs_max < 2.2250738585072014e-308) abs_max = 2.2250738585072014e-308;
    return (abs_diff / abs_max) <= (RELATIVE_ERROR_FACTOR * 2.220446049250313e-16);
}
// `beat`: the caller only asks whether the result exceeds it (rect_improve's "v > log_nfa").  The tail sums non-negative terms onto its first one, so the result is
// at most -log10(first term) - LOG_NT = -log1term / ln 10 - LOG_NT; when that bound is below `beat` (by a margin a million times the rounding of the two
// expressions) the answer is "no" whatever the tail is, and the bound is returned instead of the value: no exp, no 64-step chain, no pow / log10.  A trial that
// could win is evaluated in full, so every value that is kept or compared closely is the reference's.
__device__ double nfa(const Plan& P, int n, int k, double p, int pj, int lane, double beat = -1.7976931348623157e308) {
    const double LOG_NT = P.log_nt;
    if (n == 0 || k == 0) return -LOG_NT;
    if (n == k) return -LOG_NT - double(n) * P.p_log10[pj];
    const double p_term = p / (1 - p);
    // log_gamma(n+1) - log_gamma(k+1) - log_gamma(n-k+1) + k log(p) + (n-k) log(1-p), the transcendental terms from the host tables
    const double* lg = P.lgamma_tab;
    const double log1term = lg[n + 1] - lg[k + 1] - lg[n - k + 1] + double(k) * P.p_log[pj] + double(n - k) * P.p1_log[pj];
    {
        const double bound = -log1term / 2.30258509299404568402 - LOG_NT;
        if (bound < beat - 1e-9 * (1.0 + fabs(beat) + fabs(bound))) return bound;
    }
    double term = exp(log1term);
    if (double_equal(term, 0)) {
        if (k > n * p) return -log1term / 2.30258509299404568402 - LOG_NT;
        else return -LOG_NT;
    }
    return nfa_tail(term, n, k, p_term, LOG_NT, lane);
}

// Counts the pixels of the rectangle (total) and, for each of the np tolerances precs[], those aligned with rec.theta.
template <int NP>
__device__ int rect_counts(const Det& D, const Rect& rec, const double* precs, int* alg_out) {
    const int lane = D.lane;
    const double half_width = rec.width / 2.0;
    const double dyhw = rec.dy * half_width, dxhw = rec.dx * half_width;
    int ox[4], oy[4];
    ox[0] = int(rec.x1 - dyhw); oy[0] = int(rec.y1 + dxhw);
    ox[1] = int(rec.x2 - dyhw); oy[1] = int(rec.y2 + dxhw);
    ox[2] = int(rec.x2 + dyhw); oy[2] = int(rec.y2 - dxhw);
    ox[3] = int(rec.x1 + dyhw); oy[3] = int(rec.y1 - dxhw);
    // std::sort of 4 elements == insertion sort by (x, then y)
    for (int i = 1; i < 4; i++) {
        const int vx = ox[i], vy = oy[i];
        int j = i - 1;
        while (j >= 0 && (vx == ox[j] ? vy < oy[j] : vx < ox[j])) { ox[j + 1] = ox[j]; oy[j + 1] = oy[j]; j--; }
        ox[j + 1] = vx; oy[j + 1] = vy;
    }
    int min_y = 0, max_y = 0;
    for (int i = 1; i < 4; ++i) {
        if (oy[min_y] > oy[i]) min_y = i;
        if (oy[max_y] < oy[i]) max_y = i;
    }
    bool taken[4] = {false, false, false, false};
    taken[min_y] = true;
    int leftmost = -1;
    for (int i = 0; i < 4; ++i) if (!taken[i]) { if (leftmost < 0) leftmost = i; else if (ox[leftmost] > ox[i]) leftmost = i; }
    taken[leftmost] = true;
    int rightmost = -1;
    for (int i = 0; i < 4; ++i) if (!taken[i]) { if (rightmost < 0) rightmost = i; else if (ox[rightmost] < ox[i]) rightmost = i; }
    taken[rightmost] = true;
    int tailp = -1;
    for (int i = 0; i < 4; ++i) if (!taken[i]) { if (tailp < 0) tailp = i; else if (ox[tailp] > ox[i]) tailp = i; }
    // integer divisions and the ox[tailp] (not oy) operands are the library's own
    const long long flstep = (oy[min_y] != oy[leftmost]) ? (ox[min_y] - ox[leftmost]) / (oy[min_y] - oy[leftmost]) : 0;
    const long long slstep = (oy[leftmost] != ox[tailp]) ? (ox[leftmost] - ox[tailp]) / (oy[leftmost] - ox[tailp]) : 0;
    const long long frstep = (oy[min_y] != oy[rightmost]) ? (ox[min_y] - ox[rightmost]) / (oy[min_y] - oy[rightmost]) : 0;
    const long long srstep = (oy[rightmost] != ox[tailp]) ? (ox[rightmost] - ox[tailp]) / (oy[rightmost] - ox[tailp]) : 0;
    // The reference walks the rows with left_x += lstep / right_x += rstep (doubles).  All operands are integers, so the sums are
    // exact and row k of the in-image rows has the closed form below (rows outside the image skip the update in the library).
    const long long x0 = ox[min_y];
    const int ylo = max(oy[min_y], 0), yhi = min(oy[max_y], D.h - 1);
    const int nrows = yhi >= ylo ? yhi - ylo + 1 : 0;
    const int ly = oy[leftmost], ry = oy[rightmost];
    auto row_span = [&](int k, int& xa) -> int {      // clipped [xa, xa + c) of in-image row k; returns c
        const long long nl = min(max((long long)ly - ylo, 0ll), (long long)k), nr = min(max((long long)ry - ylo, 0ll), (long long)k);
        const long long left = x0 + nl * flstep + (k - nl) * slstep, right = x0 + nr * frstep + (k - nr) * srstep;
        const long long a = max(left, 0ll), b = min(right, (long long)D.w - 1);
        xa = (int)a;
        return b >= a ? (int)(b - a + 1) : 0;
    };
    int total = 0, wmax = 0;
    for (int k = lane; k < nrows; k += 64) { int xa; const int c = row_span(k, xa); total += c; wmax = max(wmax, c); }
    total = wave_sum_i(total);
    wmax = (int)planar::wave_max_f64((double)wmax);            // (exact: an int32 is a double)
    int alg[NP];
    for (int q = 0; q < NP; q++) alg[q] = 0;
    auto test = [&](float deg) {
        if (deg == NOTDEF_F) return;
        double n_theta = rec.theta - (double)deg * DEG_TO_RADS;
        if (n_theta < 0) n_theta = -n_theta;
        if (n_theta > M_3_2_PI) {
            n_theta -= M_2__PI;
            if (n_theta < 0) n_theta = -n_theta;
        }
        for (int q = 0; q < NP; q++) if (n_theta <= precs[q]) alg[q]++;
    };
    constexpr int RU = 4;                                    // gathers in flight per lane (unconditional loads of a clamped index)
    if (wmax > 0 && wmax <= 64) {
        int wp = 1; while (wp < wmax) wp <<= 1;
        const int rps = 64 / wp, sub = lane / wp, xo = lane & (wp - 1);
        for (int kb = 0; kb < nrows; kb += rps * RU) {
            int pix[RU]; float deg[RU];
#pragma unroll
            for (int u = 0; u < RU; u++) {
                const int k = kb + u * rps + sub;
                pix[u] = -1;
                if (k < nrows) { int xa; const int c = row_span(k, xa); if (xo < c) pix[u] = (ylo + k) * D.w + xa + xo; }
            }
#pragma unroll
            for (int u = 0; u < RU; u++) deg[u] = D.ang[max(pix[u], 0)];
#pragma unroll
            for (int u = 0; u < RU; u++) if (pix[u] >= 0) test(deg[u]);
        }
    } else if (wmax > 64) {
        for (int k = 0; k < nrows; k++) {
            int xa; const int c = row_span(k, xa);
            const int base = (ylo + k) * D.w + xa;
            for (int x0 = 0; x0 < c; x0 += 64 * RU) {
                float deg[RU];
#pragma unroll
                for (int u = 0; u < RU; u++) deg[u] = D.ang[base + min(x0 + 64 * u + lane, c - 1)];
#pragma unroll
                for (int u = 0; u < RU; u++) if (x0 + 64 * u + lane < c) test(deg[u]);
            }
        }
    }
    for (int q = 0; q < NP; q++) alg_out[q] = wave_sum_i(alg[q]);
    return total;
}

__device__ double rect_nfa(const Det& D, const Rect& rec, double beat) {
    int alg;
    const int total = rect_counts<1>(D, rec, &rec.prec, &alg);
    return nfa(*D.plan, total, alg, rec.p, rec.pj, D.lane, beat);
}

__device__ double rect_improve(const Det& D, Rect& rec, double LOG_EPS) {
    const double delta = 0.5, delta_2 = delta / 2.0;
    const Plan& P = *D.plan;
    // first evaluation + the five "finer precision" trials share the geometry: one pass over the pixels counts all six tolerances
    Rect r = rec;
    double precs[6], ps[6];
    precs[0] = rec.prec; ps[0] = rec.p;
    for (int n = 1; n < 6; ++n) { ps[n] = ps[n - 1] / 2; precs[n] = ps[n] * LSD_PI; }
    int algs[6];
    const int total0 = rect_counts<6>(D, rec, precs, algs);
    double log_nfa = nfa(P, total0, algs[0], ps[0], rec.pj, D.lane);
    if (log_nfa > LOG_EPS) return log_nfa;
    for (int n = 0; n < 5; ++n) {
        r.p /= 2;
        r.prec = r.p * LSD_PI;
        r.pj++;
        const double v = nfa(P, total0, algs[n + 1], r.p, r.pj, D.lane, log_nfa);
        if (v > log_nfa) { log_nfa = v; rec = r; }
    }
    if (log_nfa > LOG_EPS) return log_nfa;
    r = rec;
    for (int n = 0; n < 5; ++n) {
        if ((r.width - delta) >= 0.5) {
            r.width -= delta;
            const double v = rect_nfa(D, r, log_nfa);
            if (v > log_nfa) { rec = r; log_nfa = v; }
        }
    }
    if (log_nfa > LOG_EPS) return log_nfa;
    r = rec;
    for (int n = 0; n < 5; ++n) {
        if ((r.width - delta) >= 0.5) {
            r.x1 += -r.dy * delta_2; r.y1 += r.dx * delta_2;
            r.x2 += -r.dy * delta_2; r.y2 += r.dx * delta_2;
            r.width -= delta;
            const double v = rect_nfa(D, r, log_nfa);
            if (v > log_nfa) { rec = r; log_nfa = v; }
        }
    }
    if (log_nfa > LOG_EPS) return log_nfa;
    r = rec;
    for (int n = 0; n < 5; ++n) {
        if ((r.width - delta) >= 0.5) {
            r.x1 -= -r.dy * delta_2; r.y1 -= r.dx * delta_2;
            r.x2 -= -r.dy * delta_2; r.y2 -= r.dx * delta_2;
            r.width -= delta;
            const double v = rect_nfa(D, r, log_nfa);
            if (v > log_nfa) { rec = r; log_nfa = v; }
        }
    }
    if (log_nfa > LOG_EPS) return log_nfa;
    r = rec;
    if ((r.width - delta) >= 0.5) {      // the width test is the same for all five trials (the width does not change here)
        for (int n = 0; n < 5; ++n) { ps[n] = (n ? ps[n - 1] : r.p) / 2; precs[n] = ps[n] * LSD_PI; }
        const int total = rect_counts<5>(D, r, precs, algs);
        for (int n = 0; n < 5; ++n) {
            r.p /= 2;
            r.prec = r.p * LSD_PI;
            r.pj++;
            const double v = nfa(P, total, algs[n], r.p, r.pj, D.lane, log_nfa);
            if (v > log_nfa) { rec = r; log_nfa = v; }
        }
    }
    return log_nfa;
}

struct Seg { float x1, y1, x2, y2; double width, p, nfa; };

__global__ __launch_bounds__(64) void lsd_detect(const Plan* __restrict__ plan, uint8_t* __restrict__ ws, Misc* __restrict__ miscs) {
    extern __shared__ __align__(16) uint8_t lds_raw[];
    const Plan& P = *plan;
    const int b = blockIdx.x, lane = threadIdx.x;
    uint8_t* F = ws + (size_t)b * P.frame_bytes;
    Misc* misc = miscs + b;
    Det D;
    D.ang = (const float*)(F + P.off_ang); D.g2 = (const uint32_t*)(F + P.off_g2);
    D.pix4 = (const float4*)(F + P.off_pix); D.plan = plan;
    D.reg = (uint32_t*)(F + P.off_reg); D.tmp = (uint32_t*)(F + P.off_tmp);
    D.w = P.w; D.h = P.h; D.log_nt = P.log_nt; D.lane = lane;
    const int used_words = USED_LDS_BITS / 32, gused_words = (P.w * P.h + 31) / 32;
    D.gused = (uint32_t*)(F + P.off_gused);
    uint8_t* q = lds_raw;
    D.stage = (double*)q; q += 64 * 3 * 8;
    D.used = (uint32_t*)q; q += (size_t)used_words * 4;
    D.ring = (uint32_t*)q;
    for (int i = lane; i < used_words; i += 64) D.used[i] = 0;
    if (miscs[b].n_ord > USED_LDS_BITS) for (int i = lane; i < gused_words; i += 64) D.gused[i] = 0;
    wave_sync();
    const uint32_t* ord = (const uint32_t*)(F + P.off_ord);
    const uint32_t* ordr = (const uint32_t*)(F + P.off_ordr);
    Rect* rects = (Rect*)(F + P.off_rects);
    const int n_ord = misc->n_ord;
    int n_rect = 0, n_regions = 0, n_px = 0;
    long long t_grow = 0, t_rect = 0, t_refine = 0, t_nfa = 0;
    const long long t_begin = __builtin_readcyclecounter();
    for (int base = 0; base < n_ord; base += 64) {
        const int pix = base + lane < n_ord ? (int)ord[base + lane] : -1;
        const uint32_t prk = base + lane < n_ord ? ordr[base + lane] : 0u;
        const float sdeg = pix >= 0 ? D.ang[pix] : 0.f;             // seed angle / (cos, sin) of the 64 candidates of this block
        float2 scs = make_float2(0.f, 0.f);                           // (on the fly, 64 candidates at a time, and only for blocks that still hold an unused pixel: see seed_cos_sin)
        if (__ballot(pix >= 0 && !used_get(D, prk)) != 0ull) scs = pix >= 0 && sdeg != NOTDEF_F ? seed_cos_sin(sdeg) : make_float2(0.f, 0.f);
        int cursor = 0;
        while (true) {
            const bool cand = pix >= 0 && lane >= cursor && !used_get(D, prk);
            const unsigned long long m = __ballot(cand);
            if (!m) break;
            const int f = __builtin_amdgcn_readfirstlane(__ffsll((long long)m) - 1);
            cursor = f + 1;
            const int seed = __builtin_amdgcn_readlane(pix, f);
            const uint32_t seed_rank = (uint32_t)__builtin_amdgcn_readlane((int)prk, f);
            const float seed_deg = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sdeg), f));
            const float2 seed_cs = make_float2(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(scs.x), f)),
                                               __int_as_float(__builtin_amdgcn_readlane(__float_as_int(scs.y), f)));
            double reg_angle;
            long long c0 = __builtin_readcyclecounter();
            int n = region_grow(D, seed, seed_rank, seed_deg, seed_cs, P.prec, reg_angle);
            long long c1 = __builtin_readcyclecounter();
            t_grow += c1 - c0;
            n_regions++; n_px += n;
            if (n < P.min_reg_size) continue;
            wave_sync();                                            // region list written by lane 0, read by all lanes below
            Rect rec;
            region2rect(D, n, reg_angle, P.prec, P.p, rec);
            c0 = __builtin_readcyclecounter(); t_rect += c0 - c1;
            const bool keep = refine(D, n, reg_angle, P.prec, P.p, rec, P.density_th);
            c1 = __builtin_readcyclecounter(); t_refine += c1 - c0;
            if (!keep) continue;
            // the NFA stage (rect_improve) reads only the angle image: it runs for all regions in parallel in lsd_improve
            if (lane == 0 && n_rect < MAX_RECTS) rects[n_rect] = rec;
            n_rect++;
        }
    }
    if (lane == 0) {
        misc->n_rect = min(n_rect, MAX_RECTS); misc->n_regions = n_regions; misc->n_grown_px = n_px;
        if (n_rect > MAX_RECTS) misc->status = 1;
        misc->t[0] = __builtin_readcyclecounter() - t_begin; misc->t[1] = t_grow; misc->t[2] = t_rect; misc->t[3] = t_refine; misc->t[4] = t_nfa;
    }
}

// ---- K4b: NFA stage of every region that survived refine(): one wavefront (= one workgroup) per region, all regions of all frames in
// parallel; 128 wavefronts per frame share the frame's regions round-robin
// phase < 0: every region.  The top-lines mode (planar_lsd_set_top_only; lsd_improve_plan below): phase 0 = the regions at least imp_T long, 1 = the others if the
// check after phase 0 could not settle the frame, 2 = the others if the settled frame's kept lines turned out to hold equal responses.
__global__ __launch_bounds__(64) void lsd_improve(const Plan* __restrict__ plan, uint8_t* __restrict__ ws, const Misc* __restrict__ miscs, int phase) {
    const Plan& P = *plan;
    const int b = blockIdx.y, lane = threadIdx.x;
    uint8_t* F = ws + (size_t)b * P.frame_bytes;
    Det D{};
    D.ang = (const float*)(F + P.off_ang); D.plan = plan; D.w = P.w; D.h = P.h; D.log_nt = P.log_nt; D.lane = lane;
    const Rect* rects = (const Rect*)(F + P.off_rects);
    Seg* res = (Seg*)(F + P.off_res);
    const float* est = (const float*)(F + P.off_est);
    const Misc& M = miscs[b];
    const int n = M.n_rect;
    if (phase == 1 && (M.imp_done || M.imp_full)) return;
    if (phase == 2 && !(M.imp_done && M.imp_redo && !M.imp_full)) return;
    const float T = phase >= 0 ? M.imp_T : 0.f;
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
        if (phase == 0 && !(est[i] >= T)) continue;
        if (phase > 0 && est[i] >= T) continue;
        Rect rec = rects[i];
        const double log_nfa = rect_improve(D, rec, P.log_eps);
        rec.x1 += 0.5; rec.y1 += 0.5; rec.x2 += 0.5; rec.y2 += 0.5;
        rec.x1 /= 0.8; rec.y1 /= 0.8; rec.x2 /= 0.8; rec.y2 /= 0.8; rec.width /= 0.8;
        if (lane == 0) res[i] = Seg{float(rec.x1), float(rec.y1), float(rec.x2), float(rec.y2), rec.width, rec.p, log_nfa};
    }
}

// ---- top-lines mode.  LineSegment::ExtractLineSegment keeps the max_lines (40) key lines of largest response (src/LSDextractor.cpp:17-27), and a region's length is fixed before
// rect_improve (its trials move both end points by the same vector).  So: the NFA stage for the IMP_K longest regions first; if more than max_lines of them are accepted and the
// max_lines-th largest response is above anything a shorter region could reach, the kept lines and their order are the full run's - std::sort leaves distinct responses in one
// order whatever else is in the array.  Equal floats among the first max_lines + 1 responses make that order depend on the whole array: then (and when too few are accepted) every
// region goes through the NFA stage after all.  lsd_improve_plan: lengths, the IMP_K-th largest as imp_T, every result preset to "rejected".
constexpr int IMP_K = 96;
__device__ inline planar_keyline make_keyline(const Seg& sg, int cols, int rows, int class_id);
__global__ __launch_bounds__(256) void lsd_improve_plan(const Plan* __restrict__ plan, uint8_t* __restrict__ ws, Misc* __restrict__ miscs) {
    __shared__ int s_cnt[4];
    const Plan& P = *plan;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint8_t* F = ws + (size_t)b * P.frame_bytes;
    const Rect* rects = (const Rect*)(F + P.off_rects);
    Seg* res = (Seg*)(F + P.off_res);
    float* est = (float*)(F + P.off_est);
    Misc* misc = miscs + b;
    const int n = misc->n_rect;
    for (int i = tid; i < n; i += 256) {
        const Rect r = rects[i];
        est[i] = (float)sqrt(dist2(r.x1, r.y1, r.x2, r.y2));
        res[i] = Seg{0.f, 0.f, 0.f, 0.f, 0.0, 0.0, -1.7976931348623157e308};
    }
    __threadfence_block();
    __syncthreads();
    uint32_t tbits = 0;                                     // imp_T = 0: all regions
    if (n > IMP_K && n <= MAX_SEGS) {                       // (more regions than segment slots: all of them, so that an overflow is seen as in the default mode)  the largest t with at least IMP_K lengths >= t (positive floats order as their bit patterns)
        uint32_t lo = 0u, hi = 0x7f800000u;                 // count(lo) >= K, count(hi) < K
        while (hi - lo > 1u) {
            const uint32_t mid = lo + ((hi - lo) >> 1);
            int c = 0;
            for (int i = tid; i < n; i += 256) c += (__float_as_uint(est[i]) >= mid) ? 1 : 0;
            c = planar::wave_sum_i32(c);
            __syncthreads();
            if (lane == 0) s_cnt[wave] = c;
            __syncthreads();
            const int tot = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
            if (tot >= IMP_K) lo = mid; else hi = mid;
        }
        tbits = lo;
    }
    if (tid == 0) { misc->imp_T = __uint_as_float(tbits); misc->imp_done = 0; misc->imp_redo = 0; misc->imp_full = tbits == 0u ? 1 : 0; }
}
// after phase 0: is the frame settled?  One wavefront per frame.
__global__ __launch_bounds__(64) void lsd_improve_check(const Plan* __restrict__ plan, uint8_t* __restrict__ ws, Misc* __restrict__ miscs, int max_lines) {
    constexpr int CAP = 1024;
    __shared__ float s_resp[CAP];
    const Plan& P = *plan;
    const int b = blockIdx.x, lane = threadIdx.x;
    uint8_t* F = ws + (size_t)b * P.frame_bytes;
    const Seg* res = (const Seg*)(F + P.off_res);
    const float* est = (const float*)(F + P.off_est);
    Misc* misc = miscs + b;
    if (misc->imp_full) return;
    const int n = misc->n_rect;
    const float T = misc->imp_T;
    int na = 0;
    for (int base = 0; base < n; base += 64) {
        const int i = base + lane;
        bool ok = false;
        float r = 0.f;
        if (i < n && est[i] >= T) { const Seg sg = res[i]; ok = sg.nfa > P.log_eps; if (ok) r = make_keyline(sg, P.W, P.H, 0).response; }
        const unsigned long long m = __ballot(ok);
        const int pos = na + __popcll(m & ((1ull << lane) - 1ull));
        if (ok && pos < CAP) s_resp[pos] = r;
        na += __popcll(m);
    }
    __syncthreads();
    int done = 0;
    if (na > max_lines && na <= CAP) {
        uint32_t lo = 0u, hi = 0x7f800000u;                 // the max_lines-th largest response: the largest t with at least max_lines responses >= t
        while (hi - lo > 1u) {
            const uint32_t mid = lo + ((hi - lo) >> 1);
            int c = 0;
            for (int i = lane; i < na; i += 64) c += (__float_as_uint(s_resp[i]) >= mid) ? 1 : 0;
            c = planar::wave_sum_i32(c);
            if (c >= max_lines) lo = mid; else hi = mid;
        }
        // what a region shorter than imp_T could reach: its end points move by the same vector in rect_improve, are divided by 0.8, rounded to floats (1e-4 px) and clipped
        // to the image (which only shortens): (T / 0.8 + 1e-3) / max(W, H) bounds its response from above
        const double bound = ((double)T / 0.8 + 1e-3) / (double)max(P.W, P.H);
        done = (double)__uint_as_float(lo) > bound ? 1 : 0;
    }
    if (lane == 0) misc->imp_done = done;
}

// ---- K4c: the accepted segments (NFA above the threshold), in region order -----------------------------------------------------------
__global__ __launch_bounds__(64) void lsd_accept(const Plan* __restrict__ plan, uint8_t* __restrict__ ws, Misc* __restrict__ miscs, int phase) {
    const Plan& P = *plan;
    const int b = blockIdx.x, lane = threadIdx.x;
    if (phase == 1 && !(miscs[b].imp_done && miscs[b].imp_redo && !miscs[b].imp_full)) return;     // (top-lines mode: only the frames that are being redone)
    uint8_t* F = ws + (size_t)b * P.frame_bytes;
    const Seg* res = (const Seg*)(F + P.off_res);
    Seg* segs = (Seg*)(F + P.off_segs);
    const int n = miscs[b].n_rect;
    int n_seg = 0;
    for (int base = 0; base < n; base += 64) {
        Seg sg{};
        const bool in = base + lane < n;
        if (in) sg = res[base + lane];
        const bool ok = in && sg.nfa > P.log_eps;
        const unsigned long long m = __ballot(ok);
        const int pos = n_seg + __popcll(m & ((1ull << lane) - 1ull));
        if (ok && pos < MAX_SEGS) segs[pos] = sg;
        n_seg += __popcll(m);
    }
    if (lane == 0) { miscs[b].n_seg = n_seg; if (n_seg > MAX_SEGS) miscs[b].status = 1; }
}

// ---- K5: Sobel 3x3 (cv::Sobel CV_16S, BORDER_REFLECT_101) of the 5x5-blurred image ------------------------------------
// ---- K6: KeyLines + std::sort by response + keep max_lines + line equations --------------------------------------------
// libstdc++ std::sort (bits/stl_algo.h: __introsort_loop, __unguarded_partition_pivot, __final_insertion_sort,
// heap fallback) on keys[] (descending `response`), carrying idx[]; executed by one lane, data in LDS.
struct SortBuf { float* key; int* idx; };
__device__ inline bool sort_comp(float a, float b) { return a > b; }   // sort_lines_by_response (include/auxiliar.h:43-48)
__device__ inline void sort_swap(const SortBuf& s, int a, int b) {
    const float k = s.key[a]; s.key[a] = s.key[b]; s.key[b] = k;
    const int i = s.idx[a]; s.idx[a] = s.idx[b]; s.idx[b] = i;
}
__device__ void adjust_heap(const SortBuf& s, int first, int hole, int len, float vk, int vi) {
    const int top = hole;
    int child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (sort_comp(s.key[first + child], s.key[first + child - 1])) child--;
        s.key[first + hole] = s.key[first + child]; s.idx[first + hole] = s.idx[first + child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        s.key[first + hole] = s.key[first + child - 1]; s.idx[first + hole] = s.idx[first + child - 1];
        hole = child - 1;
    }
    int parent = (hole - 1) / 2;   // __push_heap
    while (hole > top && sort_comp(s.key[first + parent], vk)) {
        s.key[first + hole] = s.key[first + parent]; s.idx[first + hole] = s.idx[first + parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    s.key[first + hole] = vk; s.idx[first + hole] = vi;
}
__device__ void heap_sort_range(const SortBuf& s, int first, int last) {   // __partial_sort(first, last, last)
    const int len = last - first;
    if (len >= 2) {
        int parent = (len - 2) / 2;
        while (true) {
            adjust_heap(s, first, parent, len, s.key[first + parent], s.idx[first + parent]);
            if (parent == 0) break;
            parent--;
        }
    }
    int l = last;
    while (l - first > 1) {   // __sort_heap -> __pop_heap(first, last-1, last-1)
        --l;
        const float vk = s.key[l]; const int vi = s.idx[l];
        s.key[l] = s.key[first]; s.idx[l] = s.idx[first];
        adjust_heap(s, first, 0, l - first, vk, vi);
    }
}
__device__ void std_sort_desc(const SortBuf& s, int n) {
    if (n <= 0) return;
    // explicit stack replaces the recursion on the right part
    int stack_first[64], stack_last[64], stack_depth[64], sp = 0;
    int lg = 0; { int t = n; while (t > 1) { t >>= 1; lg++; } }
    stack_first[sp] = 0; stack_last[sp] = n; stack_depth[sp] = 2 * lg; sp++;
    while (sp > 0) {
        sp--;
        int first = stack_first[sp], last = stack_last[sp], depth = stack_depth[sp];
        while (last - first > 16) {
            if (depth == 0) { heap_sort_range(s, first, last); break; }
            --depth;
            // __move_median_to_first(first, first+1, mid, last-1)
            const int mid = first + (last - first) / 2;
            const int a = first + 1, bb = mid, c = last - 1;
            if (sort_comp(s.key[a], s.key[bb])) {
                if (sort_comp(s.key[bb], s.key[c])) sort_swap(s, first, bb);
                else if (sort_comp(s.key[a], s.key[c])) sort_swap(s, first, c);
                else sort_swap(s, first, a);
            } else if (sort_comp(s.key[a], s.key[c])) sort_swap(s, first, a);
            else if (sort_comp(s.key[bb], s.key[c])) sort_swap(s, first, c);
            else sort_swap(s, first, bb);
            // __unguarded_partition(first+1, last, first)
            int lo = first + 1, hi = last;
            const float pv = s.key[first];
            while (true) {
                while (sort_comp(s.key[lo], pv)) ++lo;
                --hi;
                while (sort_comp(pv, s.key[hi])) --hi;
                if (!(lo < hi)) break;
                sort_swap(s, lo, hi);
                ++lo;
            }
            const int cut = lo;
            stack_first[sp] = cut; stack_last[sp] = last; stack_depth[sp] = depth; sp++;   // __introsort_loop(cut, last, depth)
            last = cut;
        }
    }
    // __final_insertion_sort
    auto linear_insert = [&](int last) {
        const float vk = s.key[last]; const int vi = s.idx[last];
        int next = last - 1;
        while (sort_comp(vk, s.key[next])) { s.key[last] = s.key[next]; s.idx[last] = s.idx[next]; last = next; --next; }
        s.key[last] = vk; s.idx[last] = vi;
    };
    auto insertion = [&](int first, int last) {
        for (int i = first + 1; i < last; ++i) {
            if (sort_comp(s.key[i], s.key[first])) {
                const float vk = s.key[i]; const int vi = s.idx[i];
                for (int j = i; j > first; --j) { s.key[j] = s.key[j - 1]; s.idx[j] = s.idx[j - 1]; }
                s.key[first] = vk; s.idx[first] = vi;
            } else linear_insert(i);
        }
    };
    if (n > 16) { insertion(0, 16); for (int i = 16; i < n; ++i) linear_insert(i); }
    else insertion(0, n);
}

__device__ inline planar_keyline make_keyline(const Seg& sg, int cols, int rows, int class_id) {
    float e[4] = {sg.x1, sg.y1, sg.x2, sg.y2};
    // LSDDetector::checkLineExtremes
    if (e[0] < 0) e[0] = 0;
    if (e[0] >= cols) e[0] = (float)cols - 1.0f;
    if (e[2] < 0) e[2] = 0;
    if (e[2] >= cols) e[2] = (float)cols - 1.0f;
    if (e[1] < 0) e[1] = 0;
    if (e[1] >= rows) e[1] = (float)rows - 1.0f;
    if (e[3] < 0) e[3] = 0;
    if (e[3] >= rows) e[3] = (float)rows - 1.0f;
    planar_keyline kl;
    const float octaveScale = 1.0f;
    kl.start_x = e[0] * octaveScale; kl.start_y = e[1] * octaveScale; kl.end_x = e[2] * octaveScale; kl.end_y = e[3] * octaveScale;
    kl.s_oct_x = e[0]; kl.s_oct_y = e[1]; kl.e_oct_x = e[2]; kl.e_oct_y = e[3];
    const double d0 = (double)(e[0] - e[2]), d1 = (double)(e[1] - e[3]);
    kl.line_length = (float)sqrt(d0 * d0 + d1 * d1);
    const int ax = (int)rintf(e[0]), ay = (int)rintf(e[1]), bx = (int)rintf(e[2]), by = (int)rintf(e[3]);
    kl.num_pixels = max(abs(bx - ax), abs(by - ay)) + 1;
    kl.angle = (float)atan2((double)(kl.end_y - kl.start_y), (double)(kl.end_x - kl.start_x));
    kl.class_id = class_id;
    kl.octave = 0;
    kl.size = (kl.end_x - kl.start_x) * (kl.end_y - kl.start_y);
    kl.response = kl.line_length / (float)max(cols, rows);
    kl.pt_x = (kl.end_x + kl.start_x) / 2;
    kl.pt_y = (kl.end_y + kl.start_y) / 2;
    return kl;
}

__global__ __launch_bounds__(64) void lsd_keylines(const Plan* __restrict__ plan, uint8_t* __restrict__ ws, Misc* __restrict__ miscs, int max_lines,
                                                   planar_keyline* __restrict__ out_kl, double* __restrict__ out_eq, int32_t* __restrict__ n_out, int phase) {
    // 8 KB of LDS so that this kernel can start while peac_segment still holds most of every CU's LDS; frames with more raw
    // segments than that sort in global scratch
    constexpr int LDS_SORT = 1024;
    __shared__ float key_l[LDS_SORT];
    __shared__ int idx_l[LDS_SORT];
    const Plan& P = *plan;
    const int b = blockIdx.x, lane = threadIdx.x;
    uint8_t* F = ws + (size_t)b * P.frame_bytes;
    Misc* misc = miscs + b;
    if (phase == 1 && !(misc->imp_done && misc->imp_redo && !misc->imp_full)) return;             // (top-lines mode: only the frames that are being redone)
    const bool partial = (phase == 0 || phase == 3) && misc->imp_done && !misc->imp_full;          // only the longest regions went through the NFA stage (3: the test mode that redoes every such frame)
    const Seg* segs = (const Seg*)(F + P.off_segs);
    const int n = min(misc->n_seg, MAX_SEGS);
    const int nk = min(n, max_lines);
    planar_keyline* K = out_kl + (size_t)b * max_lines;
    planar_keyline* Kws = (planar_keyline*)(F + P.off_kl);
    auto body = [&](float* key, int* idx) {
        for (int i = lane; i < n; i += 64) { key[i] = make_keyline(segs[i], P.W, P.H, i).response; idx[i] = i; }
        __threadfence_block();
        __syncthreads();
        if (n > max_lines) {
            if (lane == 0) { SortBuf s{key, idx}; std_sort_desc(s, n); }
            __threadfence_block();
            __syncthreads();
            if (partial) {                       // equal responses among the kept lines (or at their border): std::sort's arrangement of them depends on the segments that were left out
                bool tie = false;
                for (int i = lane; i < min(n - 1, max_lines); i += 64) tie = tie || key[i] == key[i + 1];
                if ((__ballot(tie) != 0ull || phase == 3) && lane == 0) misc->imp_redo = 1;
            }
        }
        for (int i = lane; i < nk; i += 64) {
            const planar_keyline kl = make_keyline(segs[idx[i]], P.W, P.H, n > max_lines ? i : idx[i]);
            K[i] = kl; Kws[i] = kl;
            const double sp0 = kl.start_x, sp1 = kl.start_y, ep0 = kl.end_x, ep1 = kl.end_y;
            const double l0 = sp1 * 1.0 - 1.0 * ep1, l1 = 1.0 * ep0 - sp0 * 1.0, l2 = sp0 * ep1 - sp1 * ep0;
            const double nrm = sqrt(l0 * l0 + l1 * l1 + l2 * l2);
            double* E = out_eq + ((size_t)b * max_lines + i) * 3;
            E[0] = l0 / nrm; E[1] = l1 / nrm; E[2] = l2 / nrm;
        }
    };
    if (n <= LDS_SORT) body(key_l, idx_l);
    else body((float*)(F + P.off_tmp), (int*)(F + P.off_tmp) + MAX_SEGS);
    // a frame whose workspace overflowed (status 1) or whose sort is not reproducible (2, 3) delivers ZERO lines - the count the device consumers read stays >= 0 - and
    // reports its status through planar_lsd_check
    if (lane == 0) { n_out[b] = misc->status ? 0 : nk; misc->n_kl = nk; }
}

// ---- K7: LBD (BinaryDescriptor::computeLBD + binaryConversion), one wavefront per kept line ----------------------------
constexpr int NUM_OF_BANDS = 9, WIDTH_OF_BAND = 7, LSP_H = 63;
__constant__ int LBD_COMB[32][2] = {{0, 1}, {0, 2}, {0, 3}, {0, 4}, {0, 5}, {0, 6}, {1, 2}, {1, 3}, {1, 4}, {1, 5}, {1, 6}, {2, 3}, {2, 4}, {2, 5}, {2, 6}, {2, 7},
                                    {2, 8}, {3, 4}, {3, 5}, {3, 6}, {3, 7}, {3, 8}, {4, 5}, {4, 6}, {4, 7}, {4, 8}, {5, 6}, {5, 7}, {5, 8}, {6, 7}, {6, 8}, {7, 8}};

__global__ __launch_bounds__(64) void lbd_describe(const Plan* __restrict__ plan, uint8_t* __restrict__ ws, const Misc* __restrict__ miscs, int max_lines,
                                                   uint8_t* __restrict__ out_desc, int B) {
    __shared__ float row[LSP_H][8];    // pgdL, ngdL, pgdL2, ngdL2, pgdO, ngdO, pgdO2, ngdO2 row sums (after the global weight)
    __shared__ float des[NUM_OF_BANDS * 8];
    const Plan& P = *plan;
    int b, li;
    xcd_frame_block(max_lines, B, b, li);           // a frame's support regions overlap: its lines on one XCD (common.h)
    const int lane = threadIdx.x;
    uint8_t* F = ws + (size_t)b * P.frame_bytes;
    const Misc* misc = miscs + b;
    if (li >= misc->n_kl) return;
    const planar_keyline L = ((const planar_keyline*)(F + P.off_kl))[li];
    // the Sobel derivatives (16S, 3x3, BORDER_REFLECT_101) of the 5x5-blurred image are taken where they are read: eight byte reads of a 0.3 MB image per sample
    // instead of two 2-byte reads of two 0.6 MB images that a separate kernel had to write first (rounds 1-4: lbd_sobel, 1.9 GB + lbd_describe 6.1 GB per 1024 frames)
    const uint8_t* S5 = F + P.off_blur5;
    const short realWidth = (short)P.W, imageWidth = (short)(P.W - 1), imageHeight = (short)(P.H - 1);
    const short lengthOfLSP = (short)L.num_pixels;
    const short halfWidth = (lengthOfLSP - 1) / 2, halfHeight = (LSP_H - 1) / 2;
    const float lineMiddlePointX = (float)(0.5 * (L.s_oct_x + L.e_oct_x));
    const float lineMiddlePointY = (float)(0.5 * (L.s_oct_y + L.e_oct_y));
    float dL[2], dO[2];
    dL[0] = (float)cos((double)L.angle); dL[1] = (float)sin((double)L.angle);
    dO[0] = -dL[1]; dO[1] = dL[0];
    if (lane < LSP_H) {
        float sCorX0 = -dL[0] * halfWidth + dL[1] * halfHeight + lineMiddlePointX;
        float sCorY0 = -dL[1] * halfWidth - dL[0] * halfHeight + lineMiddlePointY;
        for (int h = 0; h < lane; h++) { sCorX0 -= dL[1]; sCorY0 += dL[0]; }
        float sCorX = sCorX0, sCorY = sCorY0;
        float pgdLRowSum = 0, ngdLRowSum = 0, pgdORowSum = 0, ngdORowSum = 0;
        for (short wID = 0; wID < lengthOfLSP; wID++) {
            short tempCor = (short)roundf(sCorX);
            const short xCor = (tempCor < 0) ? 0 : (tempCor > imageWidth) ? imageWidth : tempCor;
            tempCor = (short)roundf(sCorY);
            const short yCor = (tempCor < 0) ? 0 : (tempCor > imageHeight) ? imageHeight : tempCor;
            const uint8_t* r0 = S5 + (size_t)reflect101(yCor - 1, P.H) * realWidth;
            const uint8_t* r1 = S5 + (size_t)yCor * realWidth;
            const uint8_t* r2 = S5 + (size_t)reflect101(yCor + 1, P.H) * realWidth;
            const int xm = reflect101(xCor - 1, P.W), xp = reflect101(xCor + 1, P.W);
            const int a0 = r0[xm], a1 = r0[xCor], a2 = r0[xp], b0 = r1[xm], b2 = r1[xp], c0 = r2[xm], c1 = r2[xCor], c2 = r2[xp];
            const short dx = (short)((a2 - a0) + 2 * (b2 - b0) + (c2 - c0)), dy = (short)((c0 - a0) + 2 * (c1 - a1) + (c2 - a2));
            const float gDL = dx * dL[0] + dy * dL[1];
            const float gDO = dx * dO[0] + dy * dO[1];
            if (gDL > 0) pgdLRowSum += gDL; else ngdLRowSum -= gDL;
            if (gDO > 0) pgdORowSum += gDO; else ngdORowSum -= gDO;
            sCorX += dL[0];
            sCorY += dL[1];
        }
        const float coef = (float)P.gaussCoefG[lane];
        pgdLRowSum = coef * pgdLRowSum; ngdLRowSum = coef * ngdLRowSum;
        pgdORowSum = coef * pgdORowSum; ngdORowSum = coef * ngdORowSum;
        row[lane][0] = pgdLRowSum; row[lane][1] = ngdLRowSum; row[lane][2] = pgdLRowSum * pgdLRowSum; row[lane][3] = ngdLRowSum * ngdLRowSum;
        row[lane][4] = pgdORowSum; row[lane][5] = ngdORowSum; row[lane][6] = pgdORowSum * pgdORowSum; row[lane][7] = ngdORowSum * ngdORowSum;
    }
    __syncthreads();
    if (lane < NUM_OF_BANDS) {
        // rows reach band `lane` in hID order: band-1 rows (as their "below" band), own rows, band+1 rows (as their "above" band)
        float sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const int h0 = max(0, (lane - 1) * WIDTH_OF_BAND), h1 = min(LSP_H, (lane + 2) * WIDTH_OF_BAND);
        for (int hID = h0; hID < h1; hID++) {
            const int own = hID / WIDTH_OF_BAND;
            float c;
            if (own == lane) c = (float)P.gaussCoefL[hID % WIDTH_OF_BAND + WIDTH_OF_BAND];
            else if (own == lane + 1) c = (float)P.gaussCoefL[hID % WIDTH_OF_BAND + 2 * WIDTH_OF_BAND];
            else c = (float)P.gaussCoefL[hID % WIDTH_OF_BAND];
            sum[0] += c * row[hID][0]; sum[1] += c * row[hID][1]; sum[2] += c * c * row[hID][2]; sum[3] += c * c * row[hID][3];
            sum[4] += c * row[hID][4]; sum[5] += c * row[hID][5]; sum[6] += c * c * row[hID][6]; sum[7] += c * c * row[hID][7];
        }
        const float invN2 = (float)(1.0 / (WIDTH_OF_BAND * 2.0)), invN3 = (float)(1.0 / (WIDTH_OF_BAND * 3.0));
        const float invN = (lane == 0 || lane == NUM_OF_BANDS - 1) ? invN2 : invN3;
        float* d = des + lane * 8;
        float temp = sum[0] * invN;
        d[0] = temp; d[4] = sqrtf(sum[2] * invN - temp * temp);
        temp = sum[1] * invN;
        d[1] = temp; d[5] = sqrtf(sum[3] * invN - temp * temp);
        temp = sum[4] * invN;
        d[2] = temp; d[6] = sqrtf(sum[6] * invN - temp * temp);
        temp = sum[5] * invN;
        d[3] = temp; d[7] = sqrtf(sum[7] * invN - temp * temp);
    }
    __syncthreads();
    if (lane == 0) {
        float tempM = 0, tempS = 0;
        for (int i = 0; i < NUM_OF_BANDS; i++) {
            const float* d = des + 8 * i;
            tempM += d[0] * d[0]; tempM += d[1] * d[1]; tempM += d[2] * d[2]; tempM += d[3] * d[3];
            tempS += d[4] * d[4]; tempS += d[5] * d[5]; tempS += d[6] * d[6]; tempS += d[7] * d[7];
        }
        tempM = 1 / sqrtf(tempM);
        tempS = 1 / sqrtf(tempS);
        for (int i = 0; i < NUM_OF_BANDS; i++) {
            float* d = des + 8 * i;
            d[0] *= tempM; d[1] *= tempM; d[2] *= tempM; d[3] *= tempM;
            d[4] *= tempS; d[5] *= tempS; d[6] *= tempS; d[7] *= tempS;
        }
        for (int i = 0; i < NUM_OF_BANDS * 8; i++) if ((double)des[i] > 0.4) des[i] = (float)0.4;
        float temp = 0;
        for (int i = 0; i < NUM_OF_BANDS * 8; i++) temp += des[i] * des[i];
        temp = 1 / sqrtf(temp);
        for (int i = 0; i < NUM_OF_BANDS * 8; i++) des[i] = des[i] * temp;
    }
    __syncthreads();
    if (lane < 32) {
        const float *f1 = &des[8 * LBD_COMB[lane][0]], *f2 = &des[8 * LBD_COMB[lane][1]];
        uint8_t result = 0;
        for (int i = 0; i < 8; i++) if (f1[i] > f2[i]) result += (uint8_t)(128 >> i);
        out_desc[((size_t)b * max_lines + li) * 32 + lane] = result;
    }
}

#ifdef PLANAR_TEST_HOOKS
// TEST BUILD ONLY (make paranoid -> libplanar_hip_paranoid.so): the device std::sort emulation of lsd_keylines on raw keys (tests/test_lsd_gpu.py checks it against libstdc++)
__global__ __launch_bounds__(64) void debug_sort_kernel(float* key, int* idx, int n) {
    if (threadIdx.x == 0) { SortBuf s{key, idx}; std_sort_desc(s, n); }
}
#endif

}  // namespace lsd
}  // namespace planar

using namespace planar;

struct planar_lsd {
    planar_ctx* ctx = nullptr;
    int W = 0, H = 0, max_batch = 0;
    lsd::Plan plan{};
    int detect_smem = 0;
    DevBuf d_plan, d_cx, d_cy, d_taps, d_ws, d_misc, d_lgamma;
    DevBuf d_in, d_kl, d_desc, d_eq, d_n;   // staging for the host-pointer entry point
    int stage_lines = 0;
    int pre_B = 0;
    int tie_order = 0;   // 0: libstdc++ std::sort order inside a gradient bin (what the reference library produces), 1: raster order
    int top_only = 0;    // planar_lsd_set_top_only: the NFA stage only for the regions that can end among the max_lines kept key lines
    int sort_smem_g = 0, sort_smem_l = 0, sort_rows = 0, sort_rows_long = 0;
    // planar_lsd_set_profiling: HIP events around the launches of a recorded call; slots: preprocessing (two blurs, gradient, Sobel), lsd_sort, lsd_detect,
    // the rest (improve, accept, KeyLines, LBD)
    bool profiling = false;
    std::vector<std::vector<hipEvent_t>> ev_sets;
    std::vector<char> ev_complete;       // the detect half of the set was recorded too (planar_lsd_detect_dev followed planar_lsd_preprocess_dev)
    size_t ev_used = 0;
    std::vector<hipEvent_t>* ev_cur = nullptr;
    ~planar_lsd() { for (auto& v : ev_sets) for (hipEvent_t e : v) (void)hipEventDestroy(e); }
};

// host mirrors of the oracle's coefficient tables (same expressions, same libm)
static void host_taps_q8(int ksize, double sigma, int* taps) {
    const double scale2X = -0.5 / (sigma * sigma);
    std::vector<double> v(ksize);
    double sum = 0;
    for (int i = 0; i < ksize; i++) { const double x = i - (ksize - 1) * 0.5; v[i] = std::exp(scale2X * x * x); sum += v[i]; }
    sum = 1. / sum;
    for (int i = 0; i < ksize; i++) taps[i] = (int)std::nearbyint(v[i] * sum * 256.0);
}
static bool host_exact_coefs(double inv_scale, int srcsize, int dstsize, std::vector<lsd::Coef>& out) {
    const double scale = 1.0 / inv_scale;
    out.assign(dstsize, lsd::Coef{0, 0, 0});
    if (srcsize < 2) return false;
    for (int val = 0; val < dstsize; val++) {
        const double fval = scale * ((double)val + 0.5) - 0.5;
        const int ival = (int)std::floor(fval);
        if (ival < 0) out[val] = lsd::Coef{0, 256, 0};                               // left of minofst: the first pixel (src[0] << 8)
        else if (ival >= srcsize - 1) out[val] = lsd::Coef{srcsize - 2, 0, 256};     // from maxofst on: the last pixel (src[n-1] << 8)
        else {
            out[val].ofs = ival;
            out[val].c1 = (int)std::nearbyint((fval - (double)ival) * 256.0);
            out[val].c0 = 256 - out[val].c1;
        }
    }
    return true;
}

extern "C" {

int planar_lsd_max_segments(void) { return lsd::MAX_SEGS; }

int planar_lsd_create(planar_ctx* ctx, int width, int height, int max_batch, planar_lsd** out) {
    PLANAR_REQUIRE(ctx && out, PLANAR_EINVAL, "null argument");
    *out = nullptr;
    PLANAR_REQUIRE(width >= 32 && height >= 32 && width <= 4096 && height <= 1280, PLANAR_EINVAL, "image size out of range (32..4096 x 32..1280)");
    PLANAR_REQUIRE(max_batch >= 1, PLANAR_EINVAL, "max_batch must be >= 1");
    PLANAR_HIP_CHECK(hipSetDevice(ctx->device));
    planar_lsd* o = new (std::nothrow) planar_lsd();
    PLANAR_REQUIRE(o != nullptr, PLANAR_ENOMEM, "host allocation failed");
    o->ctx = ctx; o->W = width; o->H = height; o->max_batch = max_batch;
    lsd::Plan& P = o->plan;
    // createLineSegmentDetector(LSD_REFINE_ADV) defaults, evaluated as LineSegmentDetectorImpl::flsd does
    const double SCALE = 0.8, SIGMA_SCALE = 0.6, QUANT = 2.0, ANG_TH = 22.5;
    P.W = width; P.H = height;
    P.w = (int)std::nearbyint(width * SCALE); P.h = (int)std::nearbyint(height * SCALE);
    P.prec = lsd::LSD_PI * ANG_TH / 180; P.p = ANG_TH / 180; P.rho = QUANT / std::sin(P.prec);
    P.density_th = 0.7; P.log_eps = 0;
    const double sigma = SIGMA_SCALE / SCALE;
    const unsigned hk = (unsigned)(std::ceil(sigma * std::sqrt(2 * 3.0 * std::log(10.0))));
    if (1 + 2 * (int)hk != 7) { delete o; set_error("planar_lsd_create: unexpected LSD Gaussian size"); return PLANAR_EINVAL; }
    host_taps_q8(7, sigma, P.taps7);
    host_taps_q8(5, 1.0, P.taps5);
    P.log_nt = 5 * (std::log10(double(P.w)) + std::log10(double(P.h))) / 2 + std::log10(11.0);
    P.min_reg_size = (int)(size_t)(-P.log_nt / std::log10(P.p));
    {   // BinaryDescriptor::BinaryDescriptor weights (integer divisions are the library's)
        double u = (7 * 3 - 1) / 2, sg = (7 * 2 + 1) / 2, inv = -1 / (2 * sg * sg);
        for (int i = 0; i < 21; i++) { const double dis = i - u; P.gaussCoefL[i] = std::exp(dis * dis * inv); }
        u = (9 * 7 - 1) / 2; sg = (9 * 7) / 2; inv = -1 / (2 * sg * sg);
        for (int i = 0; i < 63; i++) { const double dis = i - u; P.gaussCoefG[i] = std::exp(dis * dis * inv); }
    }
    std::vector<lsd::Coef> cx, cy;
    if (!host_exact_coefs(SCALE, width, P.w, cx) || !host_exact_coefs(SCALE, height, P.h, cy)) {
        delete o; set_error("planar_lsd_create: image too small to resample"); return PLANAR_EINVAL;
    }
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o2 = off; off = align_up(off + bytes, (size_t)256); return o2; };
    const size_t NPf = (size_t)width * height, NPs = (size_t)P.w * P.h;
    P.off_blur7 = carve(NPf); P.off_blur5 = carve(NPf); P.off_dx = 0; P.off_dy = 0;   // (no Sobel images since round 5)
    
    P.off_ang = carve(NPs * 4); P.off_g2 = carve(NPs * 4); P.off_pix = carve(NPs * 16); P.off_seed = 0;   // (no per-pixel seed table since round 5)
     P.off_ordr = carve(NPs * 4); P.off_gused = carve((NPs + 31) / 32 * 4 + 256); P.off_ord = carve(NPs * 4); P.off_tmp = carve(NPs * 4); P.off_reg = carve(NPs * 4 + 64);
    P.off_valid = carve((NPs + 63) / 64 * 8 + 8); P.off_sortr = carve((size_t)isort::G_FMAX * sizeof(isort::Range)); P.off_sortb = carve((size_t)isort::G_FMAX * sizeof(isort::Block)); P.off_heapj = carve((size_t)lsd::SORT_HJOBS * sizeof(isort::HeapJob));
    P.off_segs = carve((size_t)lsd::MAX_SEGS * sizeof(lsd::Seg)); P.off_kl = carve((size_t)lsd::MAX_SEGS * sizeof(planar_keyline));
    P.off_rects = carve((size_t)lsd::MAX_RECTS * sizeof(lsd::Rect)); P.off_res = carve((size_t)lsd::MAX_RECTS * sizeof(lsd::Seg)); P.off_est = carve((size_t)lsd::MAX_RECTS * 4);
    P.frame_bytes = off;
    o->detect_smem = 64 * 3 * 8 + lsd::USED_LDS_BITS / 8 + lsd::RING * 4 + 16;
    if (o->detect_smem > 150 * 1024 || NPs > (1u << 20)) { delete o; set_error("planar_lsd_create: image too large for the LDS-resident used map"); return PLANAR_EINVAL; }
    // log_gamma(x) (lsd.cpp: Windschitl for x > 15, Lanczos otherwise) at every integer argument nfa() can see, and log(p), log(1-p),
    // log10(p) for p = P.p / 2^j: evaluated here with the host libm, exactly as the reference library evaluates them
    std::vector<double> lgam(NPs + 3, 0.0);
    for (size_t i = 1; i < lgam.size(); i++) {
        const double x = (double)i;
        if (x > 15.0) lgam[i] = 0.918938533204673 + (x - 0.5) * std::log(x) - x + 0.5 * x * std::log(x * std::sinh(1 / x) + 1 / (810.0 * std::pow(x, 6.0)));
        else {
            static const double q[7] = {75122.6331530, 80916.6278952, 36308.2951477, 8687.24529705, 1168.92649479, 83.8676043424, 2.50662827511};
            double a = (x + 0.5) * std::log(x + 5.5) - (x + 5.5), bsum = 0;
            for (int n = 0; n < 7; ++n) { a -= std::log(x + double(n)); bsum += q[n] * std::pow(x, double(n)); }
            lgam[i] = a + std::log(bsum);
        }
    }
    { double pp = P.p; for (int j = 0; j < 12; j++) { P.p_log[j] = std::log(pp); P.p1_log[j] = std::log(1.0 - pp); P.p_log10[j] = std::log10(pp); pp /= 2; } }
    int rc = o->d_lgamma.alloc(lgam.size() * 8);
    if (rc) { delete o; return rc; }
    P.lgamma_tab = o->d_lgamma.as<double>();
    if (hipMemcpy(o->d_lgamma.p, lgam.data(), lgam.size() * 8, hipMemcpyHostToDevice) != hipSuccess) { delete o; set_error("planar_lsd_create: table upload failed"); return PLANAR_EDEVICE; }
    rc = o->d_plan.alloc(sizeof(lsd::Plan));
    if (!rc) rc = o->d_cx.alloc(cx.size() * sizeof(lsd::Coef));
    if (!rc) rc = o->d_cy.alloc(cy.size() * sizeof(lsd::Coef));
    if (!rc) rc = o->d_taps.alloc(16 * 4);
    if (!rc) rc = o->d_ws.alloc(P.frame_bytes * (size_t)max_batch);
    if (!rc) rc = o->d_misc.alloc(sizeof(lsd::Misc) * (size_t)max_batch);
    if (rc) { delete o; return rc; }
    int taps[16] = {0};
    for (int i = 0; i < 7; i++) taps[i] = P.taps7[i];
    for (int i = 0; i < 5; i++) taps[8 + i] = P.taps5[i];
    hipError_t e = hipMemcpy(o->d_plan.p, &P, sizeof(P), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(o->d_cx.p, cx.data(), cx.size() * sizeof(lsd::Coef), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(o->d_cy.p, cy.data(), cy.size() * sizeof(lsd::Coef), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(o->d_taps.p, taps, sizeof(taps), hipMemcpyHostToDevice);
    // LDS of the sort kernels: the global tier keeps two stop bitmaps + two rank arrays over the longest range (all (w-1)(h-1) pixels), the LDS tier a block
    // (a working image of more than ~390 000 pixels - 1280x720 - does not fit the bitmaps: its long ranges go through isort::wg_partition_long, round 6)
    if (!lsd::SortGl::plan((P.w - 1) * (P.h - 1), o->sort_rows, o->sort_rows_long)) { delete o; set_error("planar_lsd_create: image too large for the sort's LDS-resident rank prefixes (about 1.1 M working pixels)"); return PLANAR_EINVAL; }
    o->sort_smem_g = lsd::SortGl::bytes(o->sort_rows);
    o->sort_smem_l = lsd::SortLds::bytes;
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)lsd::lsd_detect, hipFuncAttributeMaxDynamicSharedMemorySize, o->detect_smem);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)lsd::lsd_sort_global, hipFuncAttributeMaxDynamicSharedMemorySize, o->sort_smem_g);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)lsd::lsd_sort_lds, hipFuncAttributeMaxDynamicSharedMemorySize, o->sort_smem_l);
    if (e != hipSuccess) { delete o; set_error("planar_lsd_create: %s", hipGetErrorString(e)); return PLANAR_EDEVICE; }
    *out = o;
    return PLANAR_OK;
}

void planar_lsd_destroy(planar_lsd* o) { delete o; }

int planar_lsd_preprocess_dev(planar_lsd* o, const uint8_t* d_gray, int B, int pitch, int64_t frame_stride) {
    PLANAR_REQUIRE(o && d_gray, PLANAR_EINVAL, "null argument");
    PLANAR_REQUIRE(B >= 1 && B <= o->max_batch, PLANAR_EINVAL, "B out of range");
    PLANAR_REQUIRE(B <= 65535, PLANAR_EINVAL, "at most 65535 frames per call (the frame index is a grid's y / z coordinate in several launches)");
    PLANAR_REQUIRE(pitch >= o->W, PLANAR_EINVAL, "bad pitch");
    hipStream_t st = o->ctx->stream;
    const lsd::Plan& P = o->plan;
    const lsd::Plan* dP = o->d_plan.as<lsd::Plan>();
    uint8_t* ws = o->d_ws.as<uint8_t>();
    lsd::Misc* dm = o->d_misc.as<lsd::Misc>();
    PLANAR_HIP_CHECK(hipMemsetAsync(dm, 0, sizeof(lsd::Misc) * (size_t)B, st));
    o->ev_cur = nullptr;
    if (o->profiling) {
        if (o->ev_used == o->ev_sets.size()) {
            std::vector<hipEvent_t> v(5);
            for (hipEvent_t& e : v) PLANAR_HIP_CHECK(hipEventCreate(&e));
            o->ev_sets.push_back(v);
            o->ev_complete.push_back(0);
        }
        o->ev_complete[o->ev_used] = 0;
        o->ev_cur = &o->ev_sets[o->ev_used++];
        (void)hipEventRecord((*o->ev_cur)[0], st);
    }
    const dim3 gfull((unsigned)(((P.W + 63) / 64) * ((P.H + 15) / 16) * B));
    if ((P.W & 3) == 0 && (pitch & 3) == 0 && (frame_stride & 3) == 0 && ((uintptr_t)d_gray & 3) == 0 && (P.off_blur7 & 3) == 0 && (P.off_blur5 & 3) == 0 && (P.frame_bytes & 3) == 0)
        hipLaunchKernelGGL(lsd::lsd_gauss75, gfull, dim3(256), 0, st, d_gray, pitch, frame_stride, P.W, P.H, o->d_taps.as<int>(), ws, P.frame_bytes, P.off_blur7, P.off_blur5, B);
    else {
        hipLaunchKernelGGL(lsd::lsd_gauss<7>, gfull, dim3(256), 0, st, d_gray, pitch, frame_stride, P.W, P.H, o->d_taps.as<int>(), ws, P.frame_bytes, P.off_blur7, B);
        hipLaunchKernelGGL(lsd::lsd_gauss<5>, gfull, dim3(256), 0, st, d_gray, pitch, frame_stride, P.W, P.H, o->d_taps.as<int>() + 8, ws, P.frame_bytes, P.off_blur5, B);
    }
    hipLaunchKernelGGL(lsd::lsd_grad, dim3((P.w + 63) / 64, (P.h + 3) / 4, B), dim3(256), 0, st, dP, o->d_cx.as<lsd::Coef>(), o->d_cy.as<lsd::Coef>(), ws, dm);
    if (o->ev_cur) (void)hipEventRecord((*o->ev_cur)[1], st);
    if (o->tie_order != 0) hipLaunchKernelGGL(lsd::lsd_sort_raster, dim3(B), dim3(lsd::SORT_NT), 0, st, dP, ws, dm);
    else {
        hipLaunchKernelGGL(lsd::lsd_sort_global, dim3(B), dim3(lsd::SORT_T), o->sort_smem_g, st, dP, ws, dm, o->sort_rows, o->sort_rows_long);
        hipLaunchKernelGGL(lsd::lsd_sort_lds, dim3(B, lsd::SORT_R), dim3(lsd::SORT_LT), o->sort_smem_l, st, dP, ws, dm);
        hipLaunchKernelGGL(lsd::lsd_sort_heap, dim3(B, lsd::SORT_HY), dim3(64), lsd::SORT_HCAP * 4, st, dP, ws, dm);
        hipLaunchKernelGGL(lsd::lsd_sort_compact, dim3(B), dim3(lsd::SORT_T), 0, st, dP, ws, dm);
    }
    if (o->ev_cur) (void)hipEventRecord((*o->ev_cur)[2], st);
    PLANAR_HIP_CHECK(hipGetLastError());
    o->pre_B = B;
    return PLANAR_OK;
}

int planar_lsd_detect_dev(planar_lsd* o, int B, int max_lines, planar_keyline* d_keylines, uint8_t* d_ldesc, double* d_line_eq, int32_t* d_n_lines) {
    PLANAR_REQUIRE(o && d_keylines && d_ldesc && d_line_eq && d_n_lines, PLANAR_EINVAL, "null argument");
    PLANAR_REQUIRE(B >= 1 && B == o->pre_B, PLANAR_ESTATE, "planar_lsd_preprocess_dev must have been enqueued for the same B");
    PLANAR_REQUIRE(max_lines >= 1 && max_lines <= lsd::MAX_SEGS, PLANAR_EINVAL, "bad max_lines");
    hipStream_t st = o->ctx->stream;
    const lsd::Plan* dP = o->d_plan.as<lsd::Plan>();
    uint8_t* ws = o->d_ws.as<uint8_t>();
    lsd::Misc* dm = o->d_misc.as<lsd::Misc>();
    hipLaunchKernelGGL(lsd::lsd_detect, dim3(B), dim3(64), o->detect_smem, o->ctx->seq_begin(), dP, ws, dm);      // (the context's side stream when it has one)
    o->ctx->seq_end();
    if (o->ev_cur) (void)hipEventRecord((*o->ev_cur)[3], st);
    if (!o->top_only) {
        hipLaunchKernelGGL(lsd::lsd_improve, dim3(128, B), dim3(64), 0, st, dP, ws, dm, -1);   // 512 / 2048 wavefronts per frame measure the same
        hipLaunchKernelGGL(lsd::lsd_accept, dim3(B), dim3(64), 0, st, dP, ws, dm, 0);
        hipLaunchKernelGGL(lsd::lsd_keylines, dim3(B), dim3(64), 0, st, dP, ws, dm, max_lines, d_keylines, d_line_eq, d_n_lines, -1);
    } else {
        // the longest regions first; the others only for the frames the first pass cannot settle (the later launches return at once for the settled frames)
        hipLaunchKernelGGL(lsd::lsd_improve_plan, dim3(B), dim3(256), 0, st, dP, ws, dm);
        hipLaunchKernelGGL(lsd::lsd_improve, dim3(32, B), dim3(64), 0, st, dP, ws, dm, 0);
        hipLaunchKernelGGL(lsd::lsd_improve_check, dim3(B), dim3(64), 0, st, dP, ws, dm, max_lines);
        hipLaunchKernelGGL(lsd::lsd_improve, dim3(128, B), dim3(64), 0, st, dP, ws, dm, 1);
        hipLaunchKernelGGL(lsd::lsd_accept, dim3(B), dim3(64), 0, st, dP, ws, dm, 0);
        hipLaunchKernelGGL(lsd::lsd_keylines, dim3(B), dim3(64), 0, st, dP, ws, dm, max_lines, d_keylines, d_line_eq, d_n_lines, o->top_only == 2 ? 3 : 0);
        hipLaunchKernelGGL(lsd::lsd_improve, dim3(128, B), dim3(64), 0, st, dP, ws, dm, 2);
        hipLaunchKernelGGL(lsd::lsd_accept, dim3(B), dim3(64), 0, st, dP, ws, dm, 1);
        hipLaunchKernelGGL(lsd::lsd_keylines, dim3(B), dim3(64), 0, st, dP, ws, dm, max_lines, d_keylines, d_line_eq, d_n_lines, 1);
    }
    hipLaunchKernelGGL(lsd::lbd_describe, dim3(max_lines * B), dim3(64), 0, st, dP, ws, dm, max_lines, d_ldesc, B);
    if (o->ev_cur) { (void)hipEventRecord((*o->ev_cur)[4], st); o->ev_complete[o->ev_used - 1] = 1; o->ev_cur = nullptr; }
    PLANAR_HIP_CHECK(hipGetLastError());
    return PLANAR_OK;
}

// Per-launch HIP-event timing (bench.py's roofline leg), as planar_peac_set_profiling: slots preprocessing, lsd_sort, lsd_detect, the rest
int planar_lsd_set_profiling(planar_lsd* o, int enable) {
    PLANAR_REQUIRE(o != nullptr, PLANAR_EINVAL, "lsd is null");
    PLANAR_HIP_CHECK(hipStreamSynchronize(o->ctx->stream));
    o->profiling = enable != 0;
    o->ev_used = 0; o->ev_cur = nullptr;
    return PLANAR_OK;
}
int planar_lsd_get_profile(planar_lsd* o, double* total_ms /* [4] */, int64_t* calls) {
    PLANAR_REQUIRE(o && total_ms && calls, PLANAR_EINVAL, "null argument");
    PLANAR_HIP_CHECK(hipStreamSynchronize(o->ctx->stream));
    for (int i = 0; i < 4; i++) total_ms[i] = 0;
    int64_t counted = 0;
    for (size_t c = 0; c < o->ev_used; c++) {
        if (!o->ev_complete[c]) continue;            // a preprocess-only call: events [3], [4] were never recorded
        for (int i = 0; i < 4; i++) {
            float ms = 0;
            PLANAR_HIP_CHECK(hipEventElapsedTime(&ms, o->ev_sets[c][i], o->ev_sets[c][i + 1]));
            total_ms[i] += ms;
        }
        counted++;
    }
    *calls = counted;
    o->ev_used = 0;
    return PLANAR_OK;
}

int planar_lsd_extract_dev(planar_lsd* o, const uint8_t* d_gray, int B, int pitch, int64_t frame_stride, int max_lines, planar_keyline* d_keylines,
                           uint8_t* d_ldesc, double* d_line_eq, int32_t* d_n_lines) {
    PLANAR_REQUIRE(o && d_gray && d_keylines && d_ldesc && d_line_eq && d_n_lines, PLANAR_EINVAL, "null argument");
    int rc = planar_lsd_preprocess_dev(o, d_gray, B, pitch, frame_stride);
    if (rc) return rc;
    return planar_lsd_detect_dev(o, B, max_lines, d_keylines, d_ldesc, d_line_eq, d_n_lines);
}

int planar_lsd_extract(planar_lsd* o, const uint8_t* gray, int B, int pitch, int64_t frame_stride, int max_lines, planar_keyline* keylines,
                       uint8_t* ldesc, double* line_eq, int32_t* n_lines) {
    PLANAR_REQUIRE(o && gray && keylines && ldesc && line_eq && n_lines, PLANAR_EINVAL, "null argument");
    PLANAR_REQUIRE(B >= 1 && B <= o->max_batch, PLANAR_EINVAL, "B out of range");
    PLANAR_REQUIRE(pitch >= o->W && frame_stride >= (int64_t)pitch * o->H, PLANAR_EINVAL, "bad pitch / frame stride");
    PLANAR_REQUIRE(max_lines >= 1 && max_lines <= lsd::MAX_SEGS, PLANAR_EINVAL, "bad max_lines");
    PLANAR_HIP_CHECK(hipSetDevice(o->ctx->device));
    hipStream_t st = o->ctx->stream;
    const size_t in_bytes = (size_t)frame_stride * B, nl = (size_t)o->max_batch * max_lines;
    if (o->d_in.bytes < in_bytes) { int rc = o->d_in.alloc(in_bytes); if (rc) return rc; }
    if (o->stage_lines < max_lines) {
        int rc = o->d_kl.alloc(nl * sizeof(planar_keyline));
        if (!rc) rc = o->d_desc.alloc(nl * 32);
        if (!rc) rc = o->d_eq.alloc(nl * 24);
        if (!rc) rc = o->d_n.alloc((size_t)o->max_batch * 4);
        if (rc) return rc;
        o->stage_lines = max_lines;
    }
    PLANAR_HIP_CHECK(hipMemcpyAsync(o->d_in.p, gray, in_bytes, hipMemcpyHostToDevice, st));
    int rc = planar_lsd_extract_dev(o, o->d_in.as<uint8_t>(), B, pitch, frame_stride, max_lines, o->d_kl.as<planar_keyline>(), o->d_desc.as<uint8_t>(),
                                    o->d_eq.as<double>(), o->d_n.as<int32_t>());
    if (rc) return rc;
    const size_t n = (size_t)B * max_lines;
    PLANAR_HIP_CHECK(hipMemcpyAsync(keylines, o->d_kl.p, n * sizeof(planar_keyline), hipMemcpyDeviceToHost, st));
    PLANAR_HIP_CHECK(hipMemcpyAsync(ldesc, o->d_desc.p, n * 32, hipMemcpyDeviceToHost, st));
    PLANAR_HIP_CHECK(hipMemcpyAsync(line_eq, o->d_eq.p, n * 24, hipMemcpyDeviceToHost, st));
    PLANAR_HIP_CHECK(hipMemcpyAsync(n_lines, o->d_n.p, (size_t)B * 4, hipMemcpyDeviceToHost, st));
    PLANAR_HIP_CHECK(hipStreamSynchronize(st));
    return planar_lsd_check(o, B);
}

// Per-frame status of the last detect call (synchronises): PLANAR_ECAPACITY if a frame's workspace overflowed (more regions / rectangles / segments than it holds: code 1)
// or its std::sort order could not be reproduced (2, 3).  Such a frame delivered zero lines.
int planar_lsd_check(planar_lsd* o, int B) {
    PLANAR_REQUIRE(o && B >= 1 && B <= o->max_batch, PLANAR_EINVAL, "bad argument");
    PLANAR_HIP_CHECK(hipSetDevice(o->ctx->device));
    PLANAR_HIP_CHECK(hipStreamSynchronize(o->ctx->stream));
    std::vector<lsd::Misc> m(B);
    PLANAR_HIP_CHECK(hipMemcpy(m.data(), o->d_misc.p, (size_t)B * sizeof(lsd::Misc), hipMemcpyDeviceToHost));
    for (int b = 0; b < B; b++)
        if (m[b].status != 0) {
            set_error("planar_lsd: frame %d (zero lines delivered): %s (code %d)", b, m[b].status == 1 ? "more regions / segments than the workspace holds"
                      : "the std::sort order of the gradient pixels is not reproducible (introsort depth limit / sort workspace)", m[b].status);
            return PLANAR_ECAPACITY;
        }
    return PLANAR_OK;
}

/* diagnostics: stage 0 = level-line angle (float degrees, -1024 = undefined) [w*h]; 1 = squared gradient u32 [w*h];
 * 2 = pixel visiting order int32 [n] (returns n); 3 = raw segments, 40 bytes each {x1,y1,x2,y2 float; width,p,nfa double}
 * (returns the count; at most planar_lsd_max_segments() are stored); 4 = {n_regions} int32[1] */
int planar_lsd_read_stage(planar_lsd* o, int frame, int stage, void* out, int64_t out_bytes) {
    PLANAR_REQUIRE(o && out, PLANAR_EINVAL, "null argument");
    PLANAR_REQUIRE(frame >= 0 && frame < o->max_batch, PLANAR_EINVAL, "frame out of range");
    PLANAR_HIP_CHECK(hipSetDevice(o->ctx->device));
    PLANAR_HIP_CHECK(hipStreamSynchronize(o->ctx->stream));
    const lsd::Plan& P = o->plan;
    const uint8_t* F = o->d_ws.as<uint8_t>() + (size_t)frame * P.frame_bytes;
    lsd::Misc m;
    PLANAR_HIP_CHECK(hipMemcpy(&m, o->d_misc.as<lsd::Misc>() + frame, sizeof(m), hipMemcpyDeviceToHost));
    const size_t NPs = (size_t)P.w * P.h;
    size_t bytes = 0, off = 0;
    int ret = 0;
    switch (stage) {
        case 0: bytes = NPs * 4; off = P.off_ang; break;
        case 1: bytes = NPs * 4; off = P.off_g2; break;
        case 2: bytes = (size_t)m.n_ord * 4; off = P.off_ord; ret = m.n_ord; break;
        case 3: bytes = (size_t)std::min(m.n_seg, lsd::MAX_SEGS) * sizeof(lsd::Seg); off = P.off_segs; ret = m.n_seg; break;
        case 4: { PLANAR_REQUIRE(out_bytes >= 4, PLANAR_EINVAL, "buffer too small"); *(int32_t*)out = m.n_regions; return PLANAR_OK; }
        case 5: {   // shader-clock cycles: total detect, region_grow, region2rect, refine, rect_improve ; then n_ord, grown pixels
            PLANAR_REQUIRE(out_bytes >= 56, PLANAR_EINVAL, "buffer too small");
            long long* o64 = (long long*)out;
            for (int i = 0; i < 5; i++) o64[i] = m.t[i];
            o64[5] = m.n_ord; o64[6] = m.n_grown_px;
            if (out_bytes >= 80) { o64[7] = m.t[5]; o64[8] = m.t[6]; o64[9] = m.t[7]; }   // sort: keys + global tier, LDS tier (workgroup y = 0), compaction
            return PLANAR_OK;
        }
        case 6: {   // top-lines mode (planar_lsd_set_top_only) of the last call: settled by the longest regions alone, redone for equal responses, all regions from the start, regions
            PLANAR_REQUIRE(out_bytes >= 32, PLANAR_EINVAL, "buffer too small");
            long long* o64 = (long long*)out;
            o64[0] = m.imp_done; o64[1] = m.imp_redo; o64[2] = m.imp_full; o64[3] = m.n_rect;
            return PLANAR_OK;
        }
        default: set_error("planar_lsd_read_stage: unknown stage"); return PLANAR_EINVAL;
    }
    PLANAR_REQUIRE((int64_t)bytes <= out_bytes, PLANAR_EINVAL, "buffer too small");
    if (bytes) PLANAR_HIP_CHECK(hipMemcpy(out, F + off, bytes, hipMemcpyDeviceToHost));
    return ret;
}

int planar_lsd_set_top_only(planar_lsd* o, int enable) {
    PLANAR_REQUIRE(o != nullptr, PLANAR_EINVAL, "null argument");
    o->top_only = enable == 2 ? 2 : (enable != 0);          // (2: as 1, and every frame the longest regions settled is redone with all regions - exercises that path in the tests)
    return PLANAR_OK;
}
int planar_lsd_set_tie_order(planar_lsd* o, int tie_order) {
    PLANAR_REQUIRE(o && (tie_order == 0 || tie_order == 1), PLANAR_EINVAL, "tie_order must be 0 (libstdc++ std::sort order) or 1 (raster order)");
    o->tie_order = tie_order;
    return PLANAR_OK;
}

int planar_lsd_scaled_size(planar_lsd* o, int* w, int* h) {
    PLANAR_REQUIRE(o && w && h, PLANAR_EINVAL, "null argument");
    *w = o->plan.w; *h = o->plan.h;
    return PLANAR_OK;
}

#ifdef PLANAR_TEST_HOOKS
/* test hook (test build only): the device emulation of libstdc++ std::sort(first, last, greater-by-key) used for sort_lines_by_response;
 * keys [n] float (in/out), perm [n] int32 (out: original index of each sorted element) */
int planar_debug_std_sort_desc(planar_ctx* ctx, float* keys, int32_t* perm, int n) {
    PLANAR_REQUIRE(ctx && keys && perm && n >= 0 && n <= (1 << 20), PLANAR_EINVAL, "bad argument");
    PLANAR_HIP_CHECK(hipSetDevice(ctx->device));
    std::vector<int32_t> idx(n);
    for (int i = 0; i < n; i++) idx[i] = i;
    DevBuf dk, di;
    int rc = dk.alloc((size_t)n * 4);
    if (!rc) rc = di.alloc((size_t)n * 4);
    if (rc) return rc;
    PLANAR_HIP_CHECK(hipMemcpyAsync(dk.p, keys, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    PLANAR_HIP_CHECK(hipMemcpyAsync(di.p, idx.data(), (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(lsd::debug_sort_kernel, dim3(1), dim3(64), 0, ctx->stream, dk.as<float>(), di.as<int>(), n);
    PLANAR_HIP_CHECK(hipGetLastError());
    PLANAR_HIP_CHECK(hipMemcpyAsync(keys, dk.p, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    PLANAR_HIP_CHECK(hipMemcpyAsync(perm, di.p, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    PLANAR_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return PLANAR_OK;
}
#endif

}  // extern "C"
