// planarslam_amd/csrc/planepost.hip — plane post-processing for MI355X (gfx950).
//
// Replaces the head of Frame::ComputePlanes (reference src/Frame.cc:652-692) with Frame::MaxPointDistanceFromPlane (:755-812): for every plane the
// detector extracted, the member pixels' camera points (float) -> pcl::VoxelGrid(0.1 m) centroids (mvPlanePoints) -> coefficient (n, -n.c) ->
// all centroids within Plane.DistanceThreshold or the plane is dropped -> pcl::SACSegmentation (plane model, RANSAC, optimised coefficients) refits
// the coefficient (mvPlaneCoefficients).  Also Map::FlagMatchedPlanePoints (src/Map.cc:366-393) and the cloud merge of
// MapPlane::UpdateCoefficientsAndPoints (src/MapPlane.cc:335-352).  PCL is un-vendored: the arithmetic follows the published PCL 1.7-1.9 sources as
// oracle/planepost_oracle.cpp restates them (PARITY UNPINNED there, with the assumed Eigen evaluation orders listed).
//
// plane_clouds_kernel: one workgroup of 512 lanes per FRAME.
//   B  voxel sums every lane walks runs of 8 consecutive pixels (labels read as 2 x int4) with the PCL voxel coordinates floor(x / leaf) (float arithmetic
//                 as published); runs of equal (plane, voxel) are summed in registers and flushed to a 16 K-slot open-addressing table in the frame's
//                 workspace (64-bit CAS on the key, then 4 fire-and-forget atomics): about one flush per 8 pixels.  Sums are 64-bit fixed point
//                 (2^-36 m): integer addition is associative, so the centroids do not depend on the order the hardware retires the atomics in.
//                 PCL sums floats in std::sort's (unstable) order; a centroid here is the correctly rounded mean, within the float-summation
//                 error of the reference's (a few 1e-6 m), and reproducible.  PCL's bounding-box pass (getMinMax3D) is not needed: see voxel_key.
//   C  order      occupied slots -> LDS list of (plane, i2, i1, i0, slot), compacted -> bitonic sort = PCL's output order (ascending voxel index).
//   D  centroids  -> workspace, per-plane [first, last).
//   E  refit      one WAVEFRONT per plane: distance gate (ballot), RANSAC with PCL's deterministic sampler (mt19937 seeded 12345: the stream is the
//                 same for every plane, so it is a table made at create time), counts by ballot + popcount, the covariance sums as nine float
//                 chains (lane t owns accumulator t, all lanes walk the points together), eigen33 in float.
//   F  compaction of the kept planes into the output arrays.
#include "common.h"
#include "isort.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>

namespace planar {
namespace planepost {

#ifndef PLANAR_PP_NT
#define PLANAR_PP_NT 512
#endif
constexpr int NT = PLANAR_PP_NT;     // threads of the voxel / tail kernels (a developer build may narrow them: make items256)
constexpr int MAXP = 128;                 // planar_peac_max_planes()
constexpr int NRNG = 32768;               // sampler values kept (a refit that needs more reports PLANAR_ECAPACITY)
constexpr int VOX_BIAS = 8192;            // voxel coordinates floor(x / leaf) are kept in 14 bits each
constexpr unsigned long long EMPTY = ~0ull;

struct Geo {
    int dev_skip_heap = 0;                          // (always 0; was a timing switch of the round-4 experiments: leaving the fallback jobs out)
    int pl_first = 0, pl_count = 1 << 20;         // plane window (planar_plane_clouds_set_plane_window): only the detector planes [pl_first, pl_first + pl_count) are processed
    int W, H, max_points, pl_stride, tcap, mini;   // tcap = 2 * max_points key slots (frame workspace); mini: entries of a wavefront's tile table (LDS)
    float fx, fy, cx, cy, factor, leaf;
    double dist_th, log_probability, rfx, rfy;        // rfx, rfy = 1 / fx, 1 / fy (doubles)
    size_t ws_stride, off_cnt, off_cent, off_key, off_rank, off_vstart, off_pl, off_init, off_meta, off_ranges, off_blocks, off_heapj, off_items, off_dsort, off_gpos;
    int rows_long = 0, gpos_half = 0;               // frames whose largest possible plane does not fit the global tier's stop bitmaps (isort::wg_partition_long): prefix rows, scratch half
};

struct Pt { float x, y, z; };
// PlaneDetection::readDepthImage (src/PlaneExtractor.cpp:45-52) in double, narrowed as Frame.cc:659-661 does: x = (float)(((double)px - cx) * z / fx).
// The double division is replaced by a multiplication with 1 / fx where that provably gives the same FLOAT: the product is within 2.5 double ulps of
// the correctly rounded quotient, so the two can only round to different floats when a float rounding boundary (low 29 mantissa bits = 0x10000000)
// lies that close; those rare values take the division.
__device__ __forceinline__ double mul_rcp(double n, double rd, bool& near) {
    const double t = n * rd;
    const int low = (int)((unsigned)__double_as_longlong(t) & 0x1fffffffu) - 0x10000000;
    near = low >= -8 && low <= 8;
    return t;
}
__device__ __forceinline__ Pt cam_point(const Geo& G, unsigned short d, int px, int py) {
    const double z = (double)d * (double)G.factor;
    const double nx = ((double)px - (double)G.cx) * z, ny = ((double)py - (double)G.cy) * z;
    bool near_x, near_y;
    double tx = mul_rcp(nx, G.rfx, near_x), ty = mul_rcp(ny, G.rfy, near_y);
    if (__ballot(near_x || near_y) != 0ull) {         // a wave-level branch (about one wavefront in 10^5 takes it): the divisions stay out of the common path
        if (near_x) tx = nx / (double)G.fx;
        if (near_y) ty = ny / (double)G.fy;
    }
    return {(float)tx, (float)ty, (float)z};
}

// PCL's voxel index is idx = (i0 - min_b0) + (i1 - min_b1) * div_b0 + (i2 - min_b2) * div_b0 * div_b1 with i = floor(x * inv_leaf) (the float subtraction
// floor(..) - (float)min_b is exact): a voxel is identified by (i0, i1, i2) alone, and ascending idx is the lexicographic order of (i2, i1, i0).  The
// bounding box (getMinMax3D) is therefore not needed: the key carries the three coordinates, biased, 14 bits each.  false: out of the 14-bit range.
__device__ __forceinline__ bool voxel_key(float x, float y, float z, float inv, unsigned plane, unsigned long long& key) {
    const int i0 = (int)floorf(x * inv) + VOX_BIAS, i1 = (int)floorf(y * inv) + VOX_BIAS, i2 = (int)floorf(z * inv) + VOX_BIAS;
    key = ((unsigned long long)plane << 42) | ((unsigned long long)(unsigned)i2 << 28) | ((unsigned long long)(unsigned)i1 << 14) | (unsigned long long)(unsigned)i0;
    return ((unsigned)i0 | (unsigned)i1 | (unsigned)i2) < 2u * VOX_BIAS;
}

__device__ __forceinline__ float red4(float a0, float a1, float a2, float a3) { return (a0 + a2) + (a1 + a3); }
__device__ __forceinline__ float plane_dot(const float m[4], float x, float y, float z) { return red4(m[0] * x, m[1] * y, m[2] * z, m[3] * 1.0f); }

__device__ __forceinline__ void roots2(float b, float c, float roots[3]) {
    roots[0] = 0.f;
    float d = (float)((double)(b * b) - 4.0 * (double)c);
    if (d < 0.0f) d = 0.0f;
    const float sd = sqrtf(d);
    roots[2] = 0.5f * (b + sd);
    roots[1] = 0.5f * (b - sd);
}
__device__ __forceinline__ void roots3(const float m[9], float roots[3]) {
    const float c0 = m[0] * m[4] * m[8] + 2.f * m[1] * m[2] * m[5] - m[0] * m[5] * m[5] - m[4] * m[2] * m[2] - m[8] * m[1] * m[1];
    const float c1 = m[0] * m[4] - m[1] * m[1] + m[0] * m[8] - m[2] * m[2] + m[4] * m[8] - m[5] * m[5];
    const float c2 = m[0] + m[4] + m[8];
    if (fabsf(c0) < 1.1920929e-07f) { roots2(c2, c1, roots); return; }
    const float s_inv3 = (float)(1.0 / 3.0), s_sqrt3 = sqrtf(3.0f);
    const float c2_over_3 = c2 * s_inv3;
    float a_over_3 = (c1 - c2 * c2_over_3) * s_inv3;
    if (a_over_3 > 0.f) a_over_3 = 0.f;
    const float half_b = 0.5f * (c0 + c2_over_3 * (2.f * c2_over_3 * c2_over_3 - c1));
    float q = half_b * half_b + a_over_3 * a_over_3 * a_over_3;
    if (q > 0.f) q = 0.f;
    const float rho = sqrtf(-a_over_3);
    const float theta = atan2f(sqrtf(-q), half_b) * s_inv3;
    const float cos_theta = cosf(theta), sin_theta = sinf(theta);
    roots[0] = c2_over_3 + 2.f * rho * cos_theta;
    roots[1] = c2_over_3 - rho * (cos_theta + s_sqrt3 * sin_theta);
    roots[2] = c2_over_3 - rho * (cos_theta - s_sqrt3 * sin_theta);
    float t;
    if (roots[0] >= roots[1]) { t = roots[0]; roots[0] = roots[1]; roots[1] = t; }
    if (roots[1] >= roots[2]) {
        t = roots[1]; roots[1] = roots[2]; roots[2] = t;
        if (roots[0] >= roots[1]) { t = roots[0]; roots[0] = roots[1]; roots[1] = t; }
    }
    if (roots[0] <= 0) roots2(c2, c1, roots);
}
// pcl::eigen33(mat, eigenvalue, eigenvector): eigenvector of the smallest eigenvalue
__device__ __forceinline__ void eigen33_smallest(const float cov[9], float vec[3]) {
    float scale = 0.f;
#pragma unroll
    for (int k = 0; k < 9; k++) scale = fmaxf(scale, fabsf(cov[k]));
    if (scale <= 1.17549435e-38f) scale = 1.0f;
    float m[9];
#pragma unroll
    for (int k = 0; k < 9; k++) m[k] = cov[k] / scale;
    float roots[3];
    roots3(m, roots);
    m[0] -= roots[0]; m[4] -= roots[0]; m[8] -= roots[0];
    float v1[3], v2[3], v3[3];
    v1[0] = m[1] * m[5] - m[2] * m[4]; v1[1] = m[2] * m[3] - m[0] * m[5]; v1[2] = m[0] * m[4] - m[1] * m[3];      // row0 x row1
    v2[0] = m[1] * m[8] - m[2] * m[7]; v2[1] = m[2] * m[6] - m[0] * m[8]; v2[2] = m[0] * m[7] - m[1] * m[6];      // row0 x row2
    v3[0] = m[4] * m[8] - m[5] * m[7]; v3[1] = m[5] * m[6] - m[3] * m[8]; v3[2] = m[3] * m[7] - m[4] * m[6];      // row1 x row2
    const float l1 = v1[0] * v1[0] + (v1[1] * v1[1] + v1[2] * v1[2]), l2 = v2[0] * v2[0] + (v2[1] * v2[1] + v2[2] * v2[2]),
                l3 = v3[0] * v3[0] + (v3[1] * v3[1] + v3[2] * v3[2]);
    if (l1 >= l2 && l1 >= l3) { const float s = sqrtf(l1); vec[0] = v1[0] / s; vec[1] = v1[1] / s; vec[2] = v1[2] / s; }
    else if (l2 >= l1 && l2 >= l3) { const float s = sqrtf(l2); vec[0] = v2[0] / s; vec[1] = v2[1] / s; vec[2] = v2[2] / s; }
    else { const float s = sqrtf(l3); vec[0] = v3[0] / s; vec[1] = v3[1] / s; vec[2] = v3[2] / s; }
}

// pcl::SACSegmentation::segment as Frame::MaxPointDistanceFromPlane configures it, by one wavefront.  pts: n voxel centroids (global), shuf: n u16 in LDS.
// Returns 0 ok, 2 no inliers, 4 sampler table exhausted; info (or NULL): [12] as oracle/planepost_oracle.cpp documents.
__device__ __forceinline__ int sac_plane(const Geo& G, const float* __restrict__ pts, int n, unsigned short* shuf, const int* __restrict__ rng, float coef[4], int* info) {
    const int lane = threadIdx.x & 63;
    auto wfence = [&]() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); };
    for (int i = lane; i < n; i += 64) shuf[i] = (unsigned short)i;
    wfence();
    const double threshold = G.dist_th, one_over_indices = 1.0 / (double)n;
    int iterations = 0, best = -2147483647, draws = 0, bs0 = -1, bs1 = -1, bs2 = -1;
    double k = 1.0;
    unsigned skipped = 0;
    float bm[4] = {0.f, 0.f, 0.f, 0.f};
    bool have = false, exhausted = false;
    auto count_within = [&](const float m[4]) {
        int c = 0;
        for (int i0 = 0; i0 < n; i0 += 64) {
            const int i = i0 + lane;
            bool in = false;
            if (i < n) in = (double)fabsf(plane_dot(m, pts[i * 3], pts[i * 3 + 1], pts[i * 3 + 2])) < threshold;
            c += __popcll(__ballot(in));
        }
        return c;
    };
    while ((double)iterations < k && skipped < 500u) {
        int s0 = 0, s1 = 0, s2 = 0;
        bool got = false;
        if (n >= 3) {
            for (unsigned t = 0; t < 1000u && !got; t++) {
                if (draws + 3 > NRNG) { exhausted = true; break; }
#pragma unroll
                for (int i = 0; i < 3; i++) {
                    const int j = i + (int)((unsigned)rng[draws + i] % (unsigned)(n - i));
                    const unsigned short a = shuf[i], b = shuf[j];
                    if (lane == 0) { shuf[i] = b; shuf[j] = a; }
                    wfence();
                }
                draws += 3;
                s0 = shuf[0]; s1 = shuf[1]; s2 = shuf[2];
                float r[3];
#pragma unroll
                for (int c = 0; c < 3; c++) r[c] = (pts[s1 * 3 + c] - pts[s0 * 3 + c]) / (pts[s2 * 3 + c] - pts[s0 * 3 + c]);
                got = (r[0] != r[1]) || (r[2] != r[1]);
            }
        }
        if (!got) break;
        float a[3], b[3], r[3], m[4];
#pragma unroll
        for (int c = 0; c < 3; c++) { a[c] = pts[s1 * 3 + c] - pts[s0 * 3 + c]; b[c] = pts[s2 * 3 + c] - pts[s0 * 3 + c]; r[c] = a[c] / b[c]; }
        if ((r[0] == r[1]) && (r[2] == r[1])) { ++skipped; continue; }
        m[0] = a[1] * b[2] - a[2] * b[1];
        m[1] = a[2] * b[0] - a[0] * b[2];
        m[2] = a[0] * b[1] - a[1] * b[0];
        m[3] = 0.f;
        const float nrm = sqrtf(red4(m[0] * m[0], m[1] * m[1], m[2] * m[2], 0.f));
#pragma unroll
        for (int c = 0; c < 4; c++) m[c] = m[c] / nrm;
        m[3] = -1 * red4(m[0] * pts[s0 * 3], m[1] * pts[s0 * 3 + 1], m[2] * pts[s0 * 3 + 2], m[3] * 1.0f);
        const int cnt = count_within(m);
        if (cnt > best) {
            best = cnt; have = true;
#pragma unroll
            for (int c = 0; c < 4; c++) bm[c] = m[c];
            bs0 = s0; bs1 = s1; bs2 = s2;
            const double w = (double)best * one_over_indices;
            double p_no_outliers = 1.0 - pow(w, 3.0);
            p_no_outliers = fmax(2.220446049250313e-16, p_no_outliers);
            p_no_outliers = fmin(1.0 - 2.220446049250313e-16, p_no_outliers);
            k = G.log_probability / log(p_no_outliers);
        }
        ++iterations;
        if (iterations > 50) break;
    }
    if (info && lane == 0) {
        info[0] = iterations; info[1] = have ? best : 0; info[2] = bs0; info[3] = bs1; info[4] = bs2; info[5] = 0; info[6] = 0; info[7] = draws;
#pragma unroll
        for (int c = 0; c < 4; c++) info[8 + c] = __float_as_int(bm[c]);
    }
    if (exhausted) return 4;
    if (!have) return 2;
    // selectWithinDistance + computeMeanAndCovarianceMatrix: nine float chains over the inliers in index order, lane t owns accumulator t
    float acc = 0.f;
    int n_inl = 0;
    for (int i0 = 0; i0 < n; i0 += 64) {            // 64 points per coalesced read; the inliers among them are then walked in index order
        const int i = i0 + lane;
        float x = 0.f, y = 0.f, z = 0.f;
        bool in = false;
        if (i < n) { x = pts[i * 3]; y = pts[i * 3 + 1]; z = pts[i * 3 + 2]; in = (double)fabsf(plane_dot(bm, x, y, z)) < threshold; }
        unsigned long long mask = __ballot(in);
        n_inl += __popcll(mask);
        while (mask) {
            const int j = __builtin_ctzll(mask);
            mask &= mask - 1;
            const float qx = __shfl(x, j), qy = __shfl(y, j), qz = __shfl(z, j);
            const float u = lane < 3 ? qx : (lane < 5 ? qy : (lane == 5 ? qz : (lane == 6 ? qx : (lane == 7 ? qy : qz))));
            const float v = (lane == 0) ? qx : ((lane == 1 || lane == 3) ? qy : ((lane == 2 || lane == 4 || lane == 5) ? qz : 1.0f));
            acc += lane < 6 ? u * v : u;
        }
    }
    float refined[4];
    if (n_inl < 4) {
#pragma unroll
        for (int c = 0; c < 4; c++) refined[c] = bm[c];
    } else {
        acc = acc / (float)n_inl;
        float a[9];
#pragma unroll
        for (int t = 0; t < 9; t++) a[t] = __shfl(acc, t);
        float cov[9];
        cov[0] = a[0] - a[6] * a[6]; cov[1] = a[1] - a[6] * a[7]; cov[2] = a[2] - a[6] * a[8];
        cov[4] = a[3] - a[7] * a[7]; cov[5] = a[4] - a[7] * a[8]; cov[8] = a[5] - a[8] * a[8];
        cov[3] = cov[1]; cov[6] = cov[2]; cov[7] = cov[5];
        float v[3];
        eigen33_smallest(cov, v);
        refined[0] = v[0]; refined[1] = v[1]; refined[2] = v[2]; refined[3] = 0.f;
        refined[3] = -1 * red4(refined[0] * a[6], refined[1] * a[7], refined[2] * a[8], refined[3] * 1.0f);
    }
    const int nref = count_within(refined);
    if (info && lane == 0) { info[5] = n_inl; info[6] = nref; }
#pragma unroll
    for (int c = 0; c < 4; c++) coef[c] = refined[c];
    return nref != 0 ? 0 : 2;
}

// Frame::MaxPointDistanceFromPlane, one wavefront: plane in/out; returns the state (0 kept, 1 distance, 2 no inliers, 4 sampler exhausted)
__device__ __forceinline__ int max_point_distance(const Geo& G, float plane[4], const float* __restrict__ pts, int n, unsigned short* shuf, const int* __restrict__ rng, int* info) {
    const int lane = threadIdx.x & 63;
    bool far = false;
    for (int i0 = 0; i0 < n; i0 += 64) {
        const int i = i0 + lane;
        bool f = false;
        if (i < n) f = (double)fabsf(plane[0] * pts[i * 3] + plane[1] * pts[i * 3 + 1] + plane[2] * pts[i * 3 + 2] + plane[3]) > G.dist_th;
        far = far || (__ballot(f) != 0ull);
    }
    if (info && lane == 0) for (int c = 0; c < 12; c++) info[c] = c >= 2 && c <= 4 ? -1 : 0;
    if (far) return 1;
    float c[4];
    const int st = sac_plane(G, pts, n, shuf, rng, c, info);
    if (st) return st;
    const float oldVal = plane[3], newVal = c[3];
    const bool flip = (newVal < 0 && oldVal > 0) || (newVal > 0 && oldVal < 0);
#pragma unroll
    for (int t = 0; t < 4; t++) plane[t] = flip ? -c[t] : c[t];
    return 0;
}

__device__ __forceinline__ unsigned hash64(unsigned long long k) {
    k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33;
    return (unsigned)k;
}

// LDS bitonic sort of n2 (power of two) keys
__device__ __forceinline__ void bitonic(unsigned long long* a, int n2) {
    for (int k = 2; k <= n2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < n2 / 2; t += NT) {
                const int lo = ((t / j) * 2 * j) + (t % j), hi = lo + j;
                const bool up = (lo & k) == 0;
                const unsigned long long x = a[lo], y = a[hi];
                if ((x > y) == up) { a[lo] = y; a[hi] = x; }
            }
            __syncthreads();
        }
}

// ---------------------------------------------------------------------------------------------------------------------------------------------------------
// Frame::ComputePlanes head, five launches per batch (one workgroup per frame unless noted):
//   plane_voxels_kernel   the occupied voxels of every plane (key table + per-voxel point counts), sorted = PCL's output order; the start of every voxel in the
//                         frame's item array; one sort range per plane
//   plane_items_kernel    the item array: for every plane its member pixels in RASTER order (the order Frame.cc:655-668 pushes them into the cloud), each as
//                         (voxel << 19 | pixel): a stable partition of the label image by plane (ballot ranks per 64-pixel row, per-wavefront counters)
//   plane_sort_global / plane_sort_lds   (isort.h) every plane's items arranged as std::sort(index_vector) of VoxelGrid::applyFilter leaves them: sorted by
//                         voxel, points of one voxel in libstdc++'s introsort order - the order PCL adds them up in
//   plane_tail_kernel     the items' depths gathered in sorted order (streaming), then thread per voxel: the FLOAT sums in that order, centroid = sum / (float)count; then the distance gate and the RANSAC refit (one
//                         wavefront per plane) and the compaction of the kept planes
// ---------------------------------------------------------------------------------------------------------------------------------------------------------
#ifndef PLANAR_WIDE_T
#define PLANAR_WIDE_T 1024      // (developer build `make narrow`: 256 - the round-6 co-residency experiment)
#endif
constexpr int PS_T = PLANAR_WIDE_T, PS_LT = 256, PS_E = 23, PS_SHIFT = 19, PS_R = 96;      // PS_T: threads of the item / global-tier kernels; PS_LT x PS_E: an LDS block (40 KB: four per CU, and room for the other streams' workgroups)
constexpr int PS_HJOBS = 1024;                                     // heap-sort fallback: jobs per frame
// ... run in three launches by size, so that the many short ranges do not each hold a CU's LDS: (longest range, words of LDS per wavefront, wavefronts
// per workgroup, workgroups per frame); a range longer than the last class's LDS keeps the top of its heap there and the rest in place
struct HeapClass { int max_len, cap, waves, wgs; };
constexpr HeapClass PS_HC[3] = {{2048, 2048, 4, 1}, {8192, 8192, 1, 2}, {1 << 30, 8192, 1, 4}};   // (a pop costs ~0.6 us with the heap in LDS and ~1 us with only its top there: instruction latency
// of a lone wavefront either way - but a workgroup that wants a whole CU's LDS waits for a CU to drain while the other streams keep them busy, so the long ranges keep 32 KB)
using PsLds = isort::LdsLayout<PS_LT, PS_E>;
using PsGl = isort::GlobalLayout<PS_T>;
// plane_items_kernel comes in two shapes: PS_T threads per frame (small batches: a frame wants all the wavefronts it can get) and 256 (throughput: 56 registers, one
// wavefront per SIMD and 2 KB of LDS fit beside the four clustering wavefronts of a CU - 11 818 -> 11 917 frames/s, round 6)
constexpr int PS_IT_SMALL = 256, PS_IT_BATCH = 256;     // batches of at least PS_IT_BATCH frames take the 256-thread shape
constexpr int ERR_SORT = 5;

struct Meta { int n_init, counts[2], err, M, npl, sort_status, heap_n, heap_n_global; };   // heap_n_global: the fallback jobs the global tier left (the long ones), known before the LDS tier runs

__global__ __launch_bounds__(NT) void plane_voxels_kernel(Geo G, const unsigned short* __restrict__ depth_all, int pitch_px, long frame_stride_px,
                                                          const int* __restrict__ labels_all, const int* __restrict__ n_planes, unsigned char* ws_all, long long* timing) {
    extern __shared__ unsigned long long s_list[];        // the wavefronts' tile tables (B); then the list of occupied slots (C, D: max_points keys)
    __shared__ int s_first[MAXP], s_last[MAXP];
    __shared__ int s_n, s_err, s_kept, s_scan[NT / 64];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int HW = G.W * G.H;
    const unsigned short* D = depth_all + (size_t)b * frame_stride_px;
    const int* lab = labels_all + (size_t)b * HW;
    unsigned char* ws = ws_all + (size_t)b * G.ws_stride;
    unsigned* tcnt = (unsigned*)(ws + G.off_cnt);
    unsigned long long* gkey = (unsigned long long*)(ws + G.off_key);   // the frame's key table: global memory (L2), touched once per (tile, voxel)
    unsigned short* srank = (unsigned short*)(ws + G.off_rank);
    int* vstart = (int*)(ws + G.off_vstart);
    int* pfl = (int*)(ws + G.off_pl);
    isort::Range* init = (isort::Range*)(ws + G.off_init);
    Meta* meta = (Meta*)(ws + G.off_meta);
    int npl = n_planes[b];
    if (npl > MAXP) npl = MAXP;
    if (npl > G.pl_stride) npl = G.pl_stride;
    const float inv = 1.0f / G.leaf;
    const int TC = G.tcap;
    long long* tmark = timing ? timing + (size_t)b * 16 : nullptr;     // 100 MHz ticks at the phase boundaries (profiling aid)
    auto mark = [&](int q) { if (tmark && tid == 0) tmark[q] = wall_clock64(); };
    mark(0);

    for (int i = tid; i < TC; i += NT) { gkey[i] = EMPTY; tcnt[i] = 0u; }
    for (int i = tid; i < MAXP; i += NT) { s_first[i] = 0; s_last[i] = 0; }
    if (tid == 0) { s_n = 0; s_err = 0; s_kept = 0; }
    __threadfence();
    __syncthreads();

    mark(1);
    // ---- B: the voxels and their point counts.  A wavefront takes a tile of 64 columns x ROWS rows and every lane walks DOWN its column: the 64 labels / depths
    //      of a row are one coalesced read, and a lane counts its run of equal (plane, voxel) in a register.  A finished run goes to the wavefront's own 128-entry
    //      LDS table (CAS on the key, one LDS atomic); at the end of the tile the table's entries - one per voxel the tile touched - go to the frame's table: a CAS
    //      on the frame's key table (global memory, L2) for the slot, then one fire-and-forget global atomic on the slot's count.  A run that finds the small table
    //      crowded goes to the frame's table directly. ----
    {
        constexpr int ROWS = 60, UNR = 4;
        const int MINI = G.mini;                                    // 128 entries per wavefront
        unsigned long long* mk = s_list + (size_t)wave * MINI;
        unsigned* mc = (unsigned*)(s_list + (size_t)(NT / 64) * MINI) + (size_t)wave * MINI;
        auto wfence = [&]() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); };
        auto to_frame = [&](unsigned long long key, unsigned cnt) {
            unsigned h = hash64(key) & (unsigned)(TC - 1);
            for (int probe = 0; probe < TC; probe++) {
                const unsigned long long k = atomicCAS(&gkey[h], EMPTY, key);
                if (k == EMPTY) { if (atomicAdd(&s_n, 1) >= G.max_points) s_err = 3; }
                if (k == EMPTY || k == key) { atomicAdd(&tcnt[h], cnt); return; }
                h = (h + 1) & (unsigned)(TC - 1);
            }
            s_err = 3;
        };
        const int strips = (G.W + 63) / 64, tiles = strips * ((G.H + ROWS - 1) / ROWS);
        for (int tile = wave; tile < tiles; tile += NT / 64) {
            const int px = (tile % strips) * 64 + lane, y0 = (tile / strips) * ROWS, y1 = min(y0 + ROWS, G.H);
            const bool col = px < G.W;
            for (int e = lane; e < MINI; e += 64) { mk[e] = EMPTY; mc[e] = 0u; }
            wfence();
            // the run being counted (cur) and the last finished one (pend).  Finished runs are parked: the insertion code below runs - for every lane
            // that has something parked, together - only when some lane finishes a second run, i.e. every ten rows or so instead of at every row
            unsigned long long cur = EMPTY, pend = EMPTY;
            unsigned cnt = 0, pcnt = 0;
            auto flush_parked = [&]() {
                if (pend != EMPTY) {
                    unsigned h = (hash64(pend) >> 7) & (unsigned)(MINI - 1);
                    bool done = false;
                    for (int probe = 0; probe < 8 && !done; probe++) {
                        const unsigned long long k = atomicCAS(&mk[h], EMPTY, pend);
                        if (k == EMPTY || k == pend) { atomicAdd(&mc[h], pcnt); done = true; }
                        h = (h + 1) & (unsigned)(MINI - 1);
                    }
                    if (!done) to_frame(pend, pcnt);
                    pend = EMPTY;
                }
            };
            int l4[UNR], ln[UNR];
            unsigned short d4[UNR], dn[UNR];
            const int pxc = min(px, G.W - 1);
            auto load = [&](int yb, int* l, unsigned short* d) {        // rows / columns outside the tile read a clamped address and are masked
#pragma unroll
                for (int q = 0; q < UNR; q++) {
                    const int yy = min(yb + q, G.H - 1);
                    const int lv = lab[(size_t)yy * G.W + pxc];
                    d[q] = D[(size_t)yy * pitch_px + pxc];
                    l[q] = (col && yb + q < y1) ? lv : -1;
                }
            };
            load(y0, l4, d4);
            for (int yb = y0; yb < y1; yb += UNR) {
                load(yb + UNR, ln, dn);                       // the next rows are in flight while these are counted
#pragma unroll
                for (int q = 0; q < UNR; q++) {
                    const int l = l4[q];
                    const bool on = l >= G.pl_first && l < npl && l - G.pl_first < G.pl_count;
                    unsigned long long key = cur;
                    bool fin = false;
                    const Pt p = cam_point(G, d4[q], px, yb + q);
                    if (on) {
                        if (!voxel_key(p.x, p.y, p.z, inv, (unsigned)l, key)) { s_err = 3; key = cur; }
                        fin = key != cur;
                    }
                    if (__ballot(fin && cur != EMPTY && pend != EMPTY) != 0ull) flush_parked();
                    if (fin) {
                        if (cur != EMPTY) { pend = cur; pcnt = cnt; }
                        cur = key; cnt = 0;
                    }
                    if (on && key == cur && cur != EMPTY) cnt++;
                }
#pragma unroll
                for (int q = 0; q < UNR; q++) { l4[q] = ln[q]; d4[q] = dn[q]; }
            }
            flush_parked();
            pend = cur; pcnt = cnt;
            flush_parked();
            wfence();
            for (int e = lane; e < MINI; e += 64)
                if (mk[e] != EMPTY) to_frame(mk[e], mc[e]);
            wfence();
        }
    }
    __threadfence();
    __syncthreads();
    const int err = s_err;
    const int M = err ? 0 : s_n;
    mark(2);

    // ---- C: the key table, tagged with its slot numbers, sorted in place = PCL's output order (empty slots sort to the end) ----
    if (!err) {
        int n2 = 64;
        while (n2 < M) n2 <<= 1;                                     // M <= max_points: the list fits the LDS block
        if (tid == 0) s_kept = 0;
        for (int i = M + tid; i < n2; i += NT) s_list[i] = EMPTY;
        __syncthreads();
        for (int i = tid; i < TC; i += NT) {
            const unsigned long long k = __hip_atomic_load(&gkey[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (k != EMPTY) s_list[atomicAdd(&s_kept, 1)] = (k << 14) | (unsigned long long)i;
        }
        __syncthreads();
        bitonic(s_list, n2);
        // ---- D: every voxel's place in the sorted order (its sort key), per-plane voxel ranges, the start of every voxel in the frame's item array ----
        const int per = (M + NT - 1) / NT, r0 = min(M, tid * per), r1 = min(M, r0 + per);
        int mine = 0;
        for (int r = r0; r < r1; r++) {
            const unsigned long long e = s_list[r];
            const int s = (int)(e & 0x3fffull), p = (int)(e >> 56);
            srank[s] = (unsigned short)r;
            mine += (int)__hip_atomic_load(&tcnt[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (r == 0 || (int)(s_list[r - 1] >> 56) != p) s_first[p] = r;
            if (r == M - 1 || (int)(s_list[r + 1] >> 56) != p) s_last[p] = r + 1;
        }
        int total;
        int run = isort::block_exscan<NT, int>(mine, s_scan, &total);
        for (int r = r0; r < r1; r++) {
            vstart[r] = run;
            run += (int)__hip_atomic_load(&tcnt[(int)(s_list[r] & 0x3fffull)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (tid == 0) vstart[M] = total;
    }
    __threadfence();
    __syncthreads();
    for (int p = tid; p < MAXP; p += NT) { pfl[p] = s_first[p]; pfl[MAXP + p] = s_last[p]; }
    if (tid == 0) {
        int k = 0;
        for (int p = 0; p < npl && !err; p++) {
            if (s_last[p] <= s_first[p]) continue;
            const int f = vstart[s_first[p]], l = vstart[s_last[p]];
            init[k++] = isort::Range{f, l, isort::depth_limit(l - f)};
        }
        meta->n_init = k; meta->counts[0] = 0; meta->counts[1] = 0; meta->err = err; meta->M = M; meta->npl = npl; meta->sort_status = 0; meta->heap_n = 0; meta->heap_n_global = 0;
    }
    mark(3);
}

// The item array.  Workgroup of IT threads per frame; wavefront w owns the pixels [w * S, (w + 1) * S) in raster order.
template <int SH, int IT>     // SH: bits of the pixel index in an item (19: frames of up to 2^19 pixels, 8192 voxels; 20: up to 2^20 pixels, 4096 voxels)
__global__ __launch_bounds__(IT) void plane_items_kernel(Geo G, const unsigned short* __restrict__ depth_all, int pitch_px, long frame_stride_px,
                                                           const int* __restrict__ labels_all, unsigned char* ws_all) {
    constexpr int NW = IT / 64;
    __shared__ unsigned s_cnt[NW][MAXP];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int HW = G.W * G.H;
    unsigned char* ws = ws_all + (size_t)b * G.ws_stride;
    const Meta* meta = (const Meta*)(ws + G.off_meta);
    if (meta->err || meta->n_init == 0) return;
    const unsigned short* D = depth_all + (size_t)b * frame_stride_px;
    const int* lab = labels_all + (size_t)b * HW;
    const unsigned long long* gkey = (const unsigned long long*)(ws + G.off_key);
    const unsigned short* srank = (const unsigned short*)(ws + G.off_rank);
    const int* vstart = (const int*)(ws + G.off_vstart);
    const int* pfl = (const int*)(ws + G.off_pl);
    uint32_t* items = (uint32_t*)(ws + G.off_items);
    const int npl = meta->npl, TC = G.tcap;
    const float inv = 1.0f / G.leaf;
    for (int i = tid; i < NW * MAXP; i += IT) (&s_cnt[0][0])[i] = 0u;
    __syncthreads();
    const int S = ((HW + NW - 1) / NW + 63) & ~63, w0 = min(HW, wave * S), w1 = min(HW, w0 + S);
    auto wfence = [&]() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); };
    constexpr int U = 4;
    for (int i0 = w0; i0 < w1; i0 += 64 * U) {            // pass 1: members per (wavefront, plane)
        int l[U];
#pragma unroll
        for (int u = 0; u < U; u++) l[u] = lab[min(i0 + 64 * u + lane, w1 - 1)];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const bool on = i0 + 64 * u + lane < w1 && l[u] >= G.pl_first && l[u] < npl && l[u] - G.pl_first < G.pl_count;
            unsigned long long rem = __ballot(on);
            while (rem) {
                const int L = __builtin_amdgcn_readlane(l[u], __builtin_ctzll(rem));
                const unsigned long long m = __ballot(on && l[u] == L);
                if (lane == 0) s_cnt[wave][L] += (unsigned)__popcll(m);
                rem &= ~m;
            }
        }
    }
    __syncthreads();
    if (tid < npl) {                                       // exclusive scan over the wavefronts, from the plane's start in the item array
        const int first = pfl[tid], last = pfl[MAXP + tid];
        unsigned run = last > first ? (unsigned)vstart[first] : 0u;
        for (int w = 0; w < NW; w++) { const unsigned t = s_cnt[w][tid]; s_cnt[w][tid] = run; run += t; }
    }
    __syncthreads();
    for (int i0 = w0; i0 < w1; i0 += 64 * U) {            // pass 2: every member pixel's item at its rank among the plane's pixels
        int l[U];
        unsigned short d[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int pix = min(i0 + 64 * u + lane, w1 - 1), py = pix / G.W, px = pix - py * G.W;
            l[u] = lab[pix];
            d[u] = D[(size_t)py * pitch_px + px];
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int pix = i0 + 64 * u + lane, pc = min(pix, w1 - 1), py = pc / G.W, px = pc - py * G.W;
            const bool on = pix < w1 && l[u] >= G.pl_first && l[u] < npl && l[u] - G.pl_first < G.pl_count;
            const Pt p = cam_point(G, d[u], px, py);
            uint32_t item = 0;
            if (on) {
                unsigned long long key;
                voxel_key(p.x, p.y, p.z, inv, (unsigned)l[u], key);
                unsigned h = hash64(key) & (unsigned)(TC - 1);
                for (int probe = 0; probe < TC && gkey[h] != key; probe++) h = (h + 1) & (unsigned)(TC - 1);
                item = ((uint32_t)srank[h] << SH) | (uint32_t)pix;
            }
            unsigned long long rem = __ballot(on);
            while (rem) {
                const int L = __builtin_amdgcn_readlane(l[u], __builtin_ctzll(rem));
                const bool mine = on && l[u] == L;
                const unsigned long long m = __ballot(mine);
                const unsigned base = s_cnt[wave][L];
                wfence();
                if (mine) items[base + (unsigned)__popcll(m & ((1ull << lane) - 1ull))] = item;
                if (lane == 0) s_cnt[wave][L] = base + (unsigned)__popcll(m);
                wfence();
                rem &= ~m;
            }
        }
    }
}

template <int SH>
__global__ __launch_bounds__(PS_T) void plane_sort_global(Geo G, unsigned char* ws_all, int rows_cap) {
    extern __shared__ __align__(16) uint8_t sort_lds[];
    unsigned char* ws = ws_all + (size_t)blockIdx.x * G.ws_stride;
    Meta* meta = (Meta*)(ws + G.off_meta);
    if (meta->err || meta->n_init == 0) return;
    const isort::HeapSink HS{(isort::HeapJob*)(ws + G.off_heapj), &meta->heap_n, PS_HJOBS};
    isort::global_tier<SH, PS_T>((uint32_t*)(ws + G.off_items), (const isort::Range*)(ws + G.off_init), meta->n_init, PsLds::N, 64, (isort::Range*)(ws + G.off_ranges),
                                 (isort::Block*)(ws + G.off_blocks), isort::G_FMAX, meta->counts, sort_lds, rows_cap, HS, &meta->sort_status, 0xffffffffu,
                                 G.rows_long, G.rows_long ? (uint32_t*)(ws + G.off_gpos) : nullptr, G.gpos_half);
    __syncthreads();
    if (threadIdx.x == 0) meta->heap_n_global = min(meta->heap_n, PS_HJOBS);
}

// Workgroups blockIdx.y < PS_EARLY do not take LDS-tier blocks: their first wavefront heap-sorts the fallback jobs the GLOBAL tier left (ranges longer than an
// LDS block - the 100 000-element ones are here; disjoint from everything the LDS tier touches), so the longest sequential job of the chain starts with the
// LDS tier instead of after it, on the same stream (a side stream per handle cost more than it hid: the pipeline's streams already outnumber the hardware queues).
constexpr int PS_EARLY = 4;
// (four wavefronts per SIMD = four workgroups per CU, what their 40 KB of LDS allow: 128 VGPRs with 23 spilled measure 13 % faster than 163 unspilled at three)
template <int SH>
__global__ __launch_bounds__(PS_LT, 4) void plane_sort_lds(Geo G, unsigned char* ws_all) {
    extern __shared__ __align__(16) uint8_t sort_lds[];
    unsigned char* ws = ws_all + (size_t)blockIdx.x * G.ws_stride;
    Meta* meta = (Meta*)(ws + G.off_meta);
    if (meta->err) return;
    if (blockIdx.y < PS_EARLY) {
        static_assert(PsLds::bytes >= PS_HC[2].cap * 4, "the early heap jobs keep the top PS_HC[2].cap words of a range in this workgroup's LDS");
        const int ng = meta->heap_n_global;
        if (threadIdx.x < 64 && ng > 0 && !G.dev_skip_heap)
            isort::heap_jobs<SH>((uint32_t*)(ws + G.off_items), (const isort::HeapJob*)(ws + G.off_heapj), ng, blockIdx.y, PS_EARLY, (uint32_t*)sort_lds, PS_HC[2].cap, 0, 1 << 30);
        return;
    }
    const isort::Range* ranges = (const isort::Range*)(ws + G.off_ranges);
    const isort::Block* blocks = (const isort::Block*)(ws + G.off_blocks);
    const int nb = meta->counts[1];
    const isort::HeapSink HS{(isort::HeapJob*)(ws + G.off_heapj), &meta->heap_n, PS_HJOBS};
    for (int k = blockIdx.y - PS_EARLY; k < nb; k += gridDim.y - PS_EARLY) {
        const isort::Block K = blocks[k];
        isort::lds_tier<SH, PS_LT, PS_E>((uint32_t*)(ws + G.off_items), ranges + K.r0, K.nr, K.f, K.l, sort_lds, HS, &meta->sort_status);
    }
}

// part 1: the jobs of the LDS tier (part 0, the global tier's, ran inside plane_sort_lds)
template <int SH>
__global__ __launch_bounds__(256) void plane_sort_heap(Geo G, unsigned char* ws_all, int part, int min_len, int max_len, int cap) {
    extern __shared__ __align__(16) uint8_t sort_lds[];
    unsigned char* ws = ws_all + (size_t)blockIdx.x * G.ws_stride;
    const Meta* meta = (const Meta*)(ws + G.off_meta);
    const int ng = meta->heap_n_global, lo = part ? ng : 0, hi = meta->err ? 0 : (part ? min(meta->heap_n, PS_HJOBS) : ng), wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if (hi > lo) isort::heap_jobs<SH>((uint32_t*)(ws + G.off_items), (const isort::HeapJob*)(ws + G.off_heapj) + lo, hi - lo, blockIdx.y * nw + wave, gridDim.y * nw,
                                           (uint32_t*)sort_lds + (size_t)wave * cap, cap, min_len, max_len);
}

// PlaneDetection::readDepthImage for one thread (no wave-level branch: callers are in divergent loops)
__device__ __forceinline__ Pt cam_point_thread(const Geo& G, unsigned short d, int px, int py) {
    const double z = (double)d * (double)G.factor;
    const double nx = ((double)px - (double)G.cx) * z, ny = ((double)py - (double)G.cy) * z;
    bool near_x, near_y;
    double tx = mul_rcp(nx, G.rfx, near_x), ty = mul_rcp(ny, G.rfy, near_y);
    if (near_x) tx = nx / (double)G.fx;
    if (near_y) ty = ny / (double)G.fy;
    return {(float)tx, (float)ty, (float)z};
}

template <int SH>
__global__ __launch_bounds__(NT, 4) void plane_tail_kernel(Geo G, const unsigned short* __restrict__ depth_all, int pitch_px, long frame_stride_px,
                                                        const double* __restrict__ planes_all, int planes_stride, const int* __restrict__ rng, unsigned char* ws_all,
                                                        int* n_out, float* coef_out, int* src_out, int* off_out, float* pts_out, int* status, int* state_out,
                                                        int* nvox_out, int* info_out, long long* timing) {
    extern __shared__ unsigned long long s_list[];        // the refit's shuffle array (u16 per voxel)
    __shared__ int s_first[MAXP], s_last[MAXP], s_state[MAXP], s_k[MAXP], s_o[MAXP + 1];
    __shared__ float s_coef[MAXP][4];
    __shared__ int s_err, s_kept;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned short* D = depth_all + (size_t)b * frame_stride_px;
    const double* planes = planes_all + (size_t)b * planes_stride * 8;
    unsigned char* ws = ws_all + (size_t)b * G.ws_stride;
    float* cent = (float*)(ws + G.off_cent);
    const int* vstart = (const int*)(ws + G.off_vstart);
    const int* pfl = (const int*)(ws + G.off_pl);
    const uint32_t* items = (const uint32_t*)(ws + G.off_items);
    const Meta* meta = (const Meta*)(ws + G.off_meta);
    const int npl = meta->npl;
    int err = meta->err ? meta->err : (meta->sort_status ? ERR_SORT : 0);
    const int M = err ? 0 : meta->M;
    long long* tmark = timing ? timing + (size_t)b * 16 : nullptr;
    auto mark = [&](int q) { if (tmark && tid == 0) tmark[q] = wall_clock64(); };
    for (int i = tid; i < MAXP; i += NT) { s_first[i] = pfl[i]; s_last[i] = pfl[MAXP + i]; s_state[i] = (i >= G.pl_first && i - G.pl_first < G.pl_count) ? 0 : 3; }   // 3: outside the plane window
    if (tid == 0) { s_err = 0; s_kept = 0; }
    // ---- the voxel centroids: a voxel's points added up as floats in the sorted order, then divided by the count (VoxelGrid::applyFilter).
    //      Pass 1, all lanes streaming over the sorted items: item -> its pixel's depth, written next to the item (dsort[i]): consecutive items belong to one voxel,
    //      i.e. to one image patch, so the 2-byte reads of an instruction share a few cache lines.  Pass 2, a lane per voxel: the float chains, every lane walking
    //      its own CONTIGUOUS run of (item, depth) - round 4 fetched the depth from the image inside this loop, a line per lane and step from 64 patches at once:
    //      10 GB of fetches per 1024 frames for a 0.6 MB image ----
    {
        unsigned short* dsort = (unsigned short*)(ws + G.off_dsort);
        const int total = M > 0 ? vstart[M] : 0;
        for (int i = tid; i < total; i += NT) {
            const int pix = (int)(items[i] & ((1u << SH) - 1u)), yy = pix / G.W;
            dsort[i] = D[(size_t)yy * pitch_px + (pix - yy * G.W)];
        }
        __threadfence_block();
        __syncthreads();
        for (int r = tid; r < M; r += NT) {
            const int i0 = vstart[r], i1 = vstart[r + 1];
            float cx = 0.f, cy = 0.f, cz = 0.f;
            constexpr int U = 16;       // loads in flight per lane: the chain is latency-bound (a lane streams its own run; 4 in flight: 4.4 ms alone per 1024 frames)
            for (int i = i0; i < i1; i += U) {
                uint32_t it[U];
                unsigned short d[U];
#pragma unroll
                for (int u = 0; u < U; u++) { it[u] = items[min(i + u, i1 - 1)]; d[u] = dsort[min(i + u, i1 - 1)]; }
#pragma unroll
                for (int u = 0; u < U; u++)
                    if (i + u < i1) {
                        const int pix = (int)(it[u] & ((1u << SH) - 1u)), yy = pix / G.W;
                        const Pt p = cam_point_thread(G, d[u], pix - yy * G.W, yy);
                        cx += p.x; cy += p.y; cz += p.z;
                    }
            }
            const float cnt = (float)(i1 - i0);
            cent[(size_t)r * 3] = cx / cnt; cent[(size_t)r * 3 + 1] = cy / cnt; cent[(size_t)r * 3 + 2] = cz / cnt;
        }
    }
    __threadfence();
    __syncthreads();

    mark(3);
    // ---- E: distance gate + RANSAC refit, one wavefront per plane ----
    unsigned short* shuf = (unsigned short*)s_list;
    if (!err)
    for (int p = wave; p < npl; p += NT / 64) {
        if (s_state[p] != 0) continue;                  // outside the plane window
        const double* P = planes + (size_t)p * 8;
        const double nx = P[1], ny = P[2], nz = P[3];
        float c[4] = {(float)nx, (float)ny, (float)nz, (float)-(nx * P[4] + ny * P[5] + nz * P[6])};
        const int first = s_first[p], n = s_last[p] - first;
        int* info = info_out ? info_out + ((size_t)b * G.pl_stride + p) * 12 : nullptr;
        const int st = max_point_distance(G, c, cent + (size_t)first * 3, n, shuf + first, rng, info);
        if (lane == 0) {
            s_state[p] = st;
#pragma unroll
            for (int t = 0; t < 4; t++) s_coef[p][t] = c[t];
            if (st == 4) s_err = 4;
        }
    }
    __syncthreads();
    if (!err) err = s_err;
    mark(4);

    // ---- F: the kept planes, in detector order ----
    if (tid == 0) {
        int k = 0, o = 0;
        s_o[0] = 0;
        for (int p = 0; p < npl && !err; p++) {
            s_k[p] = -1;
            if (s_state[p] != 0) continue;
            s_k[p] = k;
            o += s_last[p] - s_first[p];
            k++;
            s_o[k] = o;
        }
        s_kept = k;
        n_out[b] = k;
        status[b] = err;
    }
    __syncthreads();
    const int kept = s_kept;
    for (int p = tid; p < npl; p += NT) {
        if (state_out) state_out[(size_t)b * G.pl_stride + p] = err ? -2 : s_state[p];
        if (nvox_out) nvox_out[(size_t)b * G.pl_stride + p] = s_last[p] - s_first[p];
        if (err || s_k[p] < 0) continue;
        const int k = s_k[p];
#pragma unroll
        for (int t = 0; t < 4; t++) coef_out[((size_t)b * G.pl_stride + k) * 4 + t] = s_coef[p][t];
        src_out[(size_t)b * G.pl_stride + k] = p;
    }
    for (int k = tid; k <= kept; k += NT) off_out[(size_t)b * (G.pl_stride + 1) + k] = s_o[k];
    for (int p = 0; p < npl && !err; p++) {
        if (s_k[p] < 0) continue;
        const int first = s_first[p], n = s_last[p] - first, o = s_o[s_k[p]];
        for (int i = tid; i < n * 3; i += NT) pts_out[((size_t)b * G.max_points + o) * 3 + i] = cent[(size_t)first * 3 + i];
    }
    mark(5);
    if (tmark && tid == 0) { tmark[6] = M; tmark[7] = npl; }
}

// Standalone refit of given clouds (Frame::MaxPointDistanceFromPlane): one wavefront per cloud
__global__ __launch_bounds__(64) void refit_kernel(Geo G, int n_clouds, const float* __restrict__ pts, const int* __restrict__ off, const int* __restrict__ rng,
                                                   float* plane, int* state, int* info) {
    extern __shared__ unsigned long long s_list[];
    const int q = blockIdx.x;
    if (q >= n_clouds) return;
    float c[4];
    for (int t = 0; t < 4; t++) c[t] = plane[q * 4 + t];
    const int n = off[q + 1] - off[q];
    const int st = max_point_distance(G, c, pts + (size_t)off[q] * 3, n, (unsigned short*)s_list, rng, info ? info + q * 12 : nullptr);
    if ((threadIdx.x & 63) == 0) {
        state[q] = st;
        if (st == 0) for (int t = 0; t < 4; t++) plane[q * 4 + t] = c[t];
    }
}

// Map::FlagMatchedPlanePoints: thread = map point, loop over the frame's matched planes
__global__ void flag_points_kernel(int B, const float* __restrict__ Tcw, const float* __restrict__ coef, const unsigned char* __restrict__ matched,
                                   const int* __restrict__ n_planes, int pl_stride, const float* __restrict__ xw, int n_points, int points_shared,
                                   unsigned char* flags, int* n_matches) {
    const int b = blockIdx.y, j = blockIdx.x * blockDim.x + threadIdx.x;
    const float* T = Tcw + (size_t)b * 16;
    int nm = 0;
    bool hit = false;
    if (j < n_points) {
        const float* pW = xw + ((size_t)(points_shared ? 0 : b) * n_points + j) * 3;
        const float X = pW[0], Y = pW[1], Z = pW[2];
        for (int i = 0; i < n_planes[b]; i++) {
            if (!matched[(size_t)b * pl_stride + i]) continue;
            const float* c = coef + ((size_t)b * pl_stride + i) * 4;
            float pM[4];
#pragma unroll
            for (int r = 0; r < 4; r++) {      // cv::transpose(mTcw) * coef: cv::gemm's float small-matrix path
                const float t = T[r] * c[0] + T[4 + r] * c[1] + T[8 + r] * c[2] + T[12 + r] * c[3];
                pM[r] = (float)((double)t * 1.0);
            }
            const double dis = (double)fabsf(pM[0] * X + pM[1] * Y + pM[2] * Z + pM[3]);
            if (dis < 0.5) { hit = true; nm++; }
        }
        if (hit) flags[(size_t)b * n_points + j] = 1;
    }
    // nMatches of the frame
    for (int o = 32; o > 0; o >>= 1) nm += __shfl_down(nm, o);
    if ((threadIdx.x & 63) == 0 && nm && n_matches) atomicAdd(&n_matches[b], nm);
}

// pcl::transformPointCloud with a double 4x4 (src/MapPlane.cc:340) fused with the append of the map plane's points
__global__ void merge_gather_kernel(const double* __restrict__ T, const float* __restrict__ fp, int nf, const float* __restrict__ mp, int nm, float* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nf) {
        const double x = fp[i * 3], y = fp[i * 3 + 1], z = fp[i * 3 + 2];
#pragma unroll
        for (int r = 0; r < 3; r++) out[i * 3 + r] = (float)(T[r * 4 + 0] * x + T[r * 4 + 1] * y + T[r * 4 + 2] * z + T[r * 4 + 3]);
    } else if (i < nf + nm) {
#pragma unroll
        for (int r = 0; r < 3; r++) out[i * 3 + r] = mp[(i - nf) * 3 + r];
    }
}

// pcl::VoxelGrid of one free-standing cloud (the map-side merge), workspace of frame 0: cloud_voxels_kernel (key table in LDS, voxel order, every point's item
// voxel << 19 | index, in index order) -> plane_sort_global / plane_sort_lds (one range: the whole cloud) -> cloud_sums_kernel (the float sums in std::sort's order)
__global__ __launch_bounds__(NT) void cloud_voxels_kernel(Geo G, const float* __restrict__ pts, int n, unsigned char* ws) {
    extern __shared__ unsigned long long s_list[];
    __shared__ int s_n, s_err, s_scan[NT / 64];
    const int tid = threadIdx.x, TC = G.tcap;
    unsigned* tcnt = (unsigned*)(ws + G.off_cnt);
    unsigned short* srank = (unsigned short*)(ws + G.off_rank);
    int* vstart = (int*)(ws + G.off_vstart);
    uint32_t* items = (uint32_t*)(ws + G.off_items);
    isort::Range* init = (isort::Range*)(ws + G.off_init);
    Meta* meta = (Meta*)(ws + G.off_meta);
    const float inv = 1.0f / G.leaf;
    for (int i = tid; i < TC; i += NT) { s_list[i] = EMPTY; tcnt[i] = 0u; }
    if (tid == 0) { s_n = 0; s_err = 0; }
    __threadfence();
    __syncthreads();
    for (int i = tid; i < n; i += NT) {
        const float x = pts[i * 3], y = pts[i * 3 + 1], z = pts[i * 3 + 2];
        unsigned long long key;
        if (!voxel_key(x, y, z, inv, 0u, key)) { s_err = 3; continue; }
        unsigned h = hash64(key) & (unsigned)(TC - 1);
        bool done = false;
        for (int probe = 0; probe < TC && !done; probe++) {
            const unsigned long long k = atomicCAS(&s_list[h], EMPTY, key);
            if (k == EMPTY) { if (atomicAdd(&s_n, 1) >= G.max_points) s_err = 3; }
            if (k == EMPTY || k == key) { atomicAdd(&tcnt[h], 1u); items[i] = h; done = true; }
            else h = (h + 1) & (unsigned)(TC - 1);
        }
        if (!done) s_err = 3;
    }
    __threadfence();
    __syncthreads();
    const int err = s_err, M = err ? 0 : s_n;
    if (!err) {
        for (int i = tid; i < TC; i += NT) { const unsigned long long k = s_list[i]; if (k != EMPTY) s_list[i] = (k << 14) | (unsigned long long)i; }
        __syncthreads();
        bitonic(s_list, TC);
        const int per = (M + NT - 1) / NT, r0 = min(M, tid * per), r1 = min(M, r0 + per);
        int mine = 0;
        for (int r = r0; r < r1; r++) {
            const int s = (int)(s_list[r] & 0x3fffull);
            srank[s] = (unsigned short)r;
            mine += (int)__hip_atomic_load(&tcnt[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        int total;
        int run = isort::block_exscan<NT, int>(mine, s_scan, &total);
        for (int r = r0; r < r1; r++) { vstart[r] = run; run += (int)__hip_atomic_load(&tcnt[(int)(s_list[r] & 0x3fffull)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        if (tid == 0) vstart[M] = total;
        __threadfence();
        __syncthreads();
        for (int i = tid; i < n; i += NT) items[i] = ((uint32_t)srank[items[i]] << PS_SHIFT) | (uint32_t)i;
    }
    if (tid == 0) {
        init[0] = isort::Range{0, n, isort::depth_limit(n)};
        meta->n_init = (!err && n > 0) ? 1 : 0; meta->counts[0] = 0; meta->counts[1] = 0; meta->err = err; meta->M = M; meta->npl = 1; meta->sort_status = 0; meta->heap_n = 0; meta->heap_n_global = 0;
    }
}

__global__ __launch_bounds__(NT) void cloud_sums_kernel(Geo G, const float* __restrict__ pts, unsigned char* ws, float* out, int* n_out, int* status) {
    const int tid = threadIdx.x;
    const int* vstart = (const int*)(ws + G.off_vstart);
    const uint32_t* items = (const uint32_t*)(ws + G.off_items);
    const Meta* meta = (const Meta*)(ws + G.off_meta);
    const int err = meta->err ? meta->err : (meta->sort_status ? ERR_SORT : 0), M = err ? 0 : meta->M;
    for (int r = tid; r < M; r += NT) {
        const int i0 = vstart[r], i1 = vstart[r + 1];
        float cx = 0.f, cy = 0.f, cz = 0.f;
        for (int i = i0; i < i1; i++) { const int j = (int)(items[i] & ((1u << PS_SHIFT) - 1u)); cx += pts[j * 3]; cy += pts[j * 3 + 1]; cz += pts[j * 3 + 2]; }
        const float cnt = (float)(i1 - i0);
        out[(size_t)r * 3] = cx / cnt; out[(size_t)r * 3 + 1] = cy / cnt; out[(size_t)r * 3 + 2] = cz / cnt;
    }
    if (tid == 0) { *n_out = M; *status = err; }
}

// boost::mt19937 seeded 12345 (SampleConsensusModel with random = false) through uniform_int<>(0, INT_MAX): engine() / 2
static void sampler_table(std::vector<int>& tab) {
    std::vector<uint32_t> s(624);
    s[0] = 12345u;
    for (int i = 1; i < 624; i++) s[i] = 1812433253u * (s[i - 1] ^ (s[i - 1] >> 30)) + (uint32_t)i;
    int at = 624;
    tab.resize(NRNG);
    for (int q = 0; q < NRNG; q++) {
        if (at >= 624) {
            for (int i = 0; i < 624; i++) {
                const uint32_t y = (s[i] & 0x80000000u) | (s[(i + 1) % 624] & 0x7fffffffu);
                s[i] = s[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
            }
            at = 0;
        }
        uint32_t y = s[at++];
        y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
        tab[q] = (int)(y >> 1);
    }
}

}  // namespace planepost
}  // namespace planar

struct planar_plane_clouds {
    planar_ctx* ctx = nullptr;
    int max_batch = 0;
    planar::planepost::Geo G{};
    size_t smem = 0, smem_tail = 0, smem_cloud = 0, smem_sort_g = 0, smem_sort_l = 0;
    int sort_rows = 0, shift = 19;
    planar::DevBuf ws, rng, dbg;
    bool timing = false;
    std::vector<int32_t> last_status;           // per-frame codes of the last host-pointer compute call
    // planar_plane_clouds_set_profiling: HIP events around the launches of a recorded call; slots: plane_voxels, plane_items, plane_sort_global, plane_sort_lds,
    // plane_sort_heap (three launches), plane_tail
    bool profiling = false;
    std::vector<std::vector<hipEvent_t>> ev_sets;
    size_t ev_used = 0;
    ~planar_plane_clouds() { for (auto& v : ev_sets) for (hipEvent_t e : v) (void)hipEventDestroy(e); }
};

using namespace planar;

extern "C" {

int planar_plane_clouds_create(planar_ctx* ctx, int width, int height, int max_batch, int max_points, planar_plane_clouds** out) {
    PLANAR_REQUIRE(ctx && out, PLANAR_EINVAL, "null argument");
    PLANAR_REQUIRE(width >= 16 && height >= 16 && width <= 4096 && height <= 4096 && max_batch >= 1, PLANAR_EINVAL, "bad size");
    PLANAR_REQUIRE(max_points >= 64 && max_points <= 8192 && (max_points & (max_points - 1)) == 0, PLANAR_EINVAL, "max_points must be a power of two in [64, 8192]");
    // a sort word is voxel << shift | pixel: 19 bits of pixel (8192 voxels) up to 2^19 pixels, 20 bits (4096 voxels) up to 2^20 - 1280x720 (round 6)
    PLANAR_REQUIRE((long long)width * height <= (1ll << 20), PLANAR_EINVAL, "at most 2^20 pixels per frame (a sort word is voxel << 20 | pixel)");
    PLANAR_REQUIRE((long long)width * height <= (1ll << planepost::PS_SHIFT) || max_points <= 4096, PLANAR_EINVAL, "frames of more than 2^19 pixels take max_points <= 4096 (12 bits of voxel beside 20 bits of pixel)");
    PLANAR_HIP_CHECK(hipSetDevice(ctx->device));
    planar_plane_clouds* p = new planar_plane_clouds;
    p->ctx = ctx; p->max_batch = max_batch;
    planepost::Geo& G = p->G;
    G.W = width; G.H = height; G.max_points = max_points; G.pl_stride = planepost::MAXP;
    G.leaf = 0.1f; G.dist_th = 0.05;
    G.log_probability = std::log(1.0 - 0.99);
    G.tcap = 2 * max_points;
    {   // per-frame workspace
        size_t off = 0;
        auto carve = [&](size_t bytes) { const size_t o = off; off = align_up(off + bytes, (size_t)256); return o; };
        G.off_cnt = carve((size_t)G.tcap * 4);                      // points per voxel slot
        G.off_key = carve((size_t)G.tcap * 8);                      // the frame's voxel key table
        G.off_rank = carve((size_t)G.tcap * 2);                     // slot -> place in PCL's output order
        G.off_cent = carve((size_t)max_points * 12);                // centroids in that order
        G.off_vstart = carve((size_t)(max_points + 1) * 4);         // first item of every voxel
        G.off_pl = carve((size_t)2 * planepost::MAXP * 4);          // per plane: first / last voxel
        G.off_init = carve((size_t)planepost::MAXP * sizeof(isort::Range));
        G.off_meta = carve(sizeof(planepost::Meta));
        G.off_ranges = carve((size_t)isort::G_FMAX * sizeof(isort::Range));
        G.off_blocks = carve((size_t)isort::G_FMAX * sizeof(isort::Block));
        G.off_heapj = carve((size_t)planepost::PS_HJOBS * sizeof(isort::HeapJob));
        G.off_items = carve(std::max((size_t)width * height, (size_t)65536) * 4);   // (voxel << 19 | pixel) per member pixel; the map-side merge sorts up to 65536 points here
        G.off_dsort = carve((size_t)width * height * 2);            // the items' depth values, in sorted order (plane_tail_kernel)
        // the global tier of the sort: stop bitmaps for a range as long as the frame where that fits the LDS, else rank prefixes only + a scratch array for the swap partners
        if (!planepost::PsGl::plan(std::max(width * height, 65536), p->sort_rows, G.rows_long)) { delete p; set_error("plane_clouds: frame too large for the sort's LDS-resident rank prefixes"); return PLANAR_EINVAL; }
        G.gpos_half = G.rows_long ? width * height / 2 + 2 : 0;
        G.off_gpos = carve(G.rows_long ? (size_t)G.gpos_half * 2 * 4 : 0);
        G.ws_stride = off;
    }
    p->shift = (long long)width * height <= (1ll << planepost::PS_SHIFT) ? planepost::PS_SHIFT : 20;
    G.mini = 128;
    // LDS of plane_voxels_kernel: eight tile tables of 128 entries x 12 B while the pixels are counted, the list of occupied slots (max_points x 8 B)
    // while they are sorted: 32 KB at the default 4096 voxels per frame, 64 KB at 8192.  The frame's key table itself (2 * max_points slots) lives in
    // the workspace.  plane_tail_kernel: the refit's shuffle array (2 B per voxel).  cloud_voxels_kernel (one workgroup, the map side) keeps its key table in LDS.
    p->smem = std::max((size_t)(planepost::NT / 64) * G.mini * 12, (size_t)max_points * 8);
    p->smem_tail = (size_t)max_points * 2;
    p->smem_cloud = (size_t)G.tcap * 8;
    p->smem_sort_g = (size_t)planepost::PsGl::bytes(p->sort_rows);
    p->smem_sort_l = (size_t)planepost::PsLds::bytes;
    int rc = p->ws.alloc(G.ws_stride * (size_t)max_batch);
    if (!rc) rc = p->rng.alloc((size_t)planepost::NRNG * 4);
    if (rc) { delete p; return rc; }
    std::vector<int> tab;
    planepost::sampler_table(tab);
    if (hipMemcpy(p->rng.p, tab.data(), tab.size() * 4, hipMemcpyHostToDevice) != hipSuccess) { set_error("plane_clouds: sampler table upload failed"); delete p; return PLANAR_EDEVICE; }
    {
        hipError_t e = hipSuccess;
        if (p->smem > 40 * 1024) e = hipFuncSetAttribute((const void*)planepost::plane_voxels_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p->smem);
        if (e == hipSuccess && p->smem_cloud > 40 * 1024) e = hipFuncSetAttribute((const void*)planepost::cloud_voxels_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p->smem_cloud);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)planepost::plane_sort_global<19>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p->smem_sort_g);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)planepost::plane_sort_lds<19>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p->smem_sort_l);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)planepost::plane_sort_heap<19>, hipFuncAttributeMaxDynamicSharedMemorySize, planepost::PS_HC[2].cap * 4);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)planepost::plane_sort_global<20>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p->smem_sort_g);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)planepost::plane_sort_lds<20>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p->smem_sort_l);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)planepost::plane_sort_heap<20>, hipFuncAttributeMaxDynamicSharedMemorySize, planepost::PS_HC[2].cap * 4);
        if (e != hipSuccess) { (void)hipGetLastError(); set_error("plane_clouds: %zu bytes of LDS per workgroup are not available", std::max(std::max(p->smem, p->smem_cloud), p->smem_sort_g)); delete p; return PLANAR_EINVAL; }
    }
    *out = p;
    return PLANAR_OK;
}

void planar_plane_clouds_destroy(planar_plane_clouds* p) { delete p; }

// The following compute calls only process the detector planes [first, first + count) of every frame (count < 0: all planes again).  For callers whose frames can
// hold more voxels than max_points: pcl::VoxelGrid has no cap (src/Frame.cc:674-679), the frame's voxel table here has, and PLANAR_ECAPACITY (code 3) from a compute
// call means "this frame's planes together have more than max_points voxels" - the same frame goes through plane by plane (include/planar_adapters.hpp does that).
int planar_plane_clouds_set_plane_window(planar_plane_clouds* p, int first, int count) {
    PLANAR_REQUIRE(p && first >= 0, PLANAR_EINVAL, "bad argument");
    p->G.pl_first = count < 0 ? 0 : first;
    p->G.pl_count = count < 0 ? (1 << 20) : count;
    return PLANAR_OK;
}

// Profiling aid: per-frame phase timestamps of the last call, out[B][8] (100 MHz ticks: [0] entry, [1] table cleared, [2] voxel sums, [3] sorted + centroids,
// [4] refit, [5] end; [6] voxels, [7] planes).  Enabling allocates the buffer.
int planar_plane_clouds_set_timing(planar_plane_clouds* p, int enable) {
    PLANAR_REQUIRE(p != nullptr, PLANAR_EINVAL, "null argument");
    if (enable && !p->dbg.p) { int rc = p->dbg.alloc((size_t)p->max_batch * 128); if (rc) return rc; }
    p->timing = enable != 0;
    return PLANAR_OK;
}
int planar_plane_clouds_read_timing(planar_plane_clouds* p, int B, int64_t* out) {
    PLANAR_REQUIRE(p && out && p->dbg.p && B >= 1 && B <= p->max_batch, PLANAR_EINVAL, "timing not enabled / bad B");
    PLANAR_HIP_CHECK(hipStreamSynchronize(p->ctx->stream));
    PLANAR_HIP_CHECK(hipMemcpy(out, p->dbg.p, (size_t)B * 128, hipMemcpyDeviceToHost));
    return PLANAR_OK;
}

// Per-launch HIP-event timing (bench.py's roofline leg), as planar_peac_set_profiling: slots plane_voxels, plane_items, plane_sort_global, plane_sort_lds,
// plane_sort_heap (its three launches together), plane_tail
int planar_plane_clouds_set_profiling(planar_plane_clouds* p, int enable) {
    PLANAR_REQUIRE(p != nullptr, PLANAR_EINVAL, "null argument");
    PLANAR_HIP_CHECK(hipStreamSynchronize(p->ctx->stream));
    p->profiling = enable != 0;
    p->ev_used = 0;
    return PLANAR_OK;
}
int planar_plane_clouds_get_profile(planar_plane_clouds* p, double* total_ms /* [6] */, int64_t* calls) {
    PLANAR_REQUIRE(p && total_ms && calls, PLANAR_EINVAL, "null argument");
    PLANAR_HIP_CHECK(hipStreamSynchronize(p->ctx->stream));
    for (int i = 0; i < 6; i++) total_ms[i] = 0;
    for (size_t c = 0; c < p->ev_used; c++)
        for (int i = 0; i < 6; i++) {
            float ms = 0;
            PLANAR_HIP_CHECK(hipEventElapsedTime(&ms, p->ev_sets[c][i], p->ev_sets[c][i + 1]));
            total_ms[i] += ms;
        }
    *calls = (int64_t)p->ev_used;
    p->ev_used = 0;
    return PLANAR_OK;
}

// diagnostics: per frame of the last call {ranges left to libstdc++'s heap-sort fallback, their elements, the longest, LDS-tier blocks}
int planar_plane_clouds_sort_stats(planar_plane_clouds* p, int B, int64_t* out /* [B][4] */) {
    PLANAR_REQUIRE(p && out && B >= 1 && B <= p->max_batch, PLANAR_EINVAL, "bad argument");
    PLANAR_HIP_CHECK(hipSetDevice(p->ctx->device));
    PLANAR_HIP_CHECK(hipStreamSynchronize(p->ctx->stream));
    std::vector<isort::HeapJob> jobs(planepost::PS_HJOBS);
    for (int b = 0; b < B; b++) {
        const unsigned char* ws = p->ws.as<unsigned char>() + (size_t)b * p->G.ws_stride;
        planepost::Meta m;
        PLANAR_HIP_CHECK(hipMemcpy(&m, ws + p->G.off_meta, sizeof(m), hipMemcpyDeviceToHost));
        const int nj = std::min(m.heap_n, planepost::PS_HJOBS);
        if (nj) PLANAR_HIP_CHECK(hipMemcpy(jobs.data(), ws + p->G.off_heapj, (size_t)nj * sizeof(isort::HeapJob), hipMemcpyDeviceToHost));
        int64_t el = 0, mx = 0;
        for (int j = 0; j < nj; j++) { el += jobs[j].l - jobs[j].f; mx = std::max<int64_t>(mx, jobs[j].l - jobs[j].f); }
        out[b * 4] = m.heap_n; out[b * 4 + 1] = el; out[b * 4 + 2] = mx; out[b * 4 + 3] = m.counts[1];
    }
    return PLANAR_OK;
}

int planar_plane_clouds_stride(const planar_plane_clouds* p, int* pl_stride, int* max_points) {
    PLANAR_REQUIRE(p != nullptr, PLANAR_EINVAL, "null argument");
    if (pl_stride) *pl_stride = p->G.pl_stride;
    if (max_points) *max_points = p->G.max_points;
    return PLANAR_OK;
}

int planar_plane_clouds_compute_dev(planar_plane_clouds* p, const uint16_t* d_depth, int B, int pitch_px, int64_t frame_stride_px, float fx, float fy, float cx,
                                    float cy, float depth_factor, const int32_t* d_labels, const double* d_planes, const int32_t* d_n_planes, double dist_th,
                                    float leaf, int32_t* d_n_out, float* d_coef, int32_t* d_src, int32_t* d_pt_off, float* d_points, int32_t* d_status,
                                    int32_t* d_state, int32_t* d_nvox, int32_t* d_info) {
    PLANAR_REQUIRE(p && d_depth && d_labels && d_planes && d_n_planes && d_n_out && d_coef && d_src && d_pt_off && d_points && d_status, PLANAR_EINVAL, "null argument");
    PLANAR_REQUIRE(B >= 1 && B <= p->max_batch, PLANAR_EINVAL, "B must be in [1, max_batch]");
    PLANAR_REQUIRE(pitch_px >= p->G.W && frame_stride_px >= (int64_t)pitch_px * p->G.H, PLANAR_EINVAL, "pitch/frame_stride too small");
    PLANAR_REQUIRE(leaf > 0.f && dist_th >= 0.0, PLANAR_EINVAL, "leaf / dist_th");
    planepost::Geo G = p->G;
    G.fx = fx; G.fy = fy; G.cx = cx; G.cy = cy; G.factor = depth_factor; G.leaf = leaf; G.dist_th = dist_th;
    G.rfx = 1.0 / (double)fx; G.rfy = 1.0 / (double)fy;
    hipStream_t st = p->ctx->stream;
    unsigned char* ws = p->ws.as<unsigned char>();
    long long* tm = p->timing ? p->dbg.as<long long>() : nullptr;
    std::vector<hipEvent_t>* evs = nullptr;
    if (p->profiling) {
        if (p->ev_used == p->ev_sets.size()) {
            std::vector<hipEvent_t> v(7);
            for (hipEvent_t& e : v) PLANAR_HIP_CHECK(hipEventCreate(&e));
            p->ev_sets.push_back(v);
        }
        evs = &p->ev_sets[p->ev_used++];
    }
    int li = 0;
    auto mark = [&]() { if (evs) (void)hipEventRecord((*evs)[li], st); li++; };
    mark();
    hipLaunchKernelGGL(planepost::plane_voxels_kernel, dim3(B), dim3(planepost::NT), p->smem, st, G, d_depth, pitch_px, (long)frame_stride_px, d_labels, d_n_planes, ws, tm);
    mark();
    // (the item word's pixel field: 19 bits, or 20 for frames of more than 2^19 pixels - two instantiations of the five kernels that read it)
#define PLANAR_BY_SHIFT(K, ...) do { if (p->shift == 20) hipLaunchKernelGGL(planepost::K<20>, __VA_ARGS__); else hipLaunchKernelGGL(planepost::K<19>, __VA_ARGS__); } while (0)
    if (B >= planepost::PS_IT_BATCH) {
        if (p->shift == 20) hipLaunchKernelGGL((planepost::plane_items_kernel<20, planepost::PS_IT_SMALL>), dim3(B), dim3(planepost::PS_IT_SMALL), 0, st, G, d_depth, pitch_px, (long)frame_stride_px, d_labels, ws);
        else hipLaunchKernelGGL((planepost::plane_items_kernel<19, planepost::PS_IT_SMALL>), dim3(B), dim3(planepost::PS_IT_SMALL), 0, st, G, d_depth, pitch_px, (long)frame_stride_px, d_labels, ws);
    } else {
        if (p->shift == 20) hipLaunchKernelGGL((planepost::plane_items_kernel<20, planepost::PS_T>), dim3(B), dim3(planepost::PS_T), 0, st, G, d_depth, pitch_px, (long)frame_stride_px, d_labels, ws);
        else hipLaunchKernelGGL((planepost::plane_items_kernel<19, planepost::PS_T>), dim3(B), dim3(planepost::PS_T), 0, st, G, d_depth, pitch_px, (long)frame_stride_px, d_labels, ws);
    }
    mark();
    PLANAR_BY_SHIFT(plane_sort_global, dim3(B), dim3(planepost::PS_T), p->smem_sort_g, st, G, ws, p->sort_rows);
    mark();
    auto heap_launch = [&](hipStream_t q, int part, int c) {
        const planepost::HeapClass& H = planepost::PS_HC[c];
        PLANAR_BY_SHIFT(plane_sort_heap, dim3(B, H.wgs), dim3(64 * H.waves), (size_t)H.cap * 4 * H.waves, q, G, ws, part, c ? planepost::PS_HC[c - 1].max_len + 1 : 0, H.max_len, H.cap);
    };
    PLANAR_BY_SHIFT(plane_sort_lds, dim3(B, planepost::PS_EARLY + planepost::PS_R), dim3(planepost::PS_LT), p->smem_sort_l, st, G, ws);   // + the global tier's fallback jobs
    mark();
    heap_launch(st, 1, 0); heap_launch(st, 1, 1);            // the LDS tier's jobs (at most a block long)
    mark();
    PLANAR_BY_SHIFT(plane_tail_kernel, dim3(B), dim3(planepost::NT), p->smem_tail, st, G, d_depth, pitch_px, (long)frame_stride_px, d_planes,
                    planar_peac_max_planes(), p->rng.as<int>(), ws, d_n_out, d_coef, d_src, d_pt_off, d_points, d_status, d_state, d_nvox, d_info, tm);
#undef PLANAR_BY_SHIFT
    mark();
    PLANAR_HIP_CHECK(hipGetLastError());
    return PLANAR_OK;
}

int planar_plane_clouds_compute(planar_plane_clouds* p, const uint16_t* depth, int B, int pitch_px, int64_t frame_stride_px, float fx, float fy, float cx, float cy,
                                float depth_factor, const int32_t* labels, const double* planes, const int32_t* n_planes, double dist_th, float leaf, int32_t* n_out,
                                float* coef, int32_t* src, int32_t* pt_off, float* points, int32_t* state, int32_t* nvox, int32_t* info) {
    PLANAR_REQUIRE(p && depth && labels && planes && n_planes && n_out && coef && src && pt_off && points, PLANAR_EINVAL, "null argument");
    PLANAR_REQUIRE(B >= 1 && B <= p->max_batch, PLANAR_EINVAL, "B must be in [1, max_batch]");
    PLANAR_REQUIRE(pitch_px >= p->G.W && frame_stride_px >= (int64_t)pitch_px * p->G.H, PLANAR_EINVAL, "pitch/frame_stride too small");
    PLANAR_HIP_CHECK(hipSetDevice(p->ctx->device));
    const int PS = p->G.pl_stride, MP = p->G.max_points, HW = p->G.W * p->G.H;
    Stager s;
    const int i_depth = s.in(depth, ((size_t)frame_stride_px * (B - 1) + (size_t)pitch_px * p->G.H) * 2), i_lab = s.in(labels, (size_t)B * HW * 4),
              i_pl = s.in(planes, (size_t)B * planar_peac_max_planes() * 64), i_np = s.in(n_planes, (size_t)B * 4);
    const int o_n = s.out(n_out, (size_t)B * 4), o_coef = s.out(coef, (size_t)B * PS * 16), o_src = s.out(src, (size_t)B * PS * 4),
              o_off = s.out(pt_off, (size_t)B * (PS + 1) * 4), o_pts = s.out(points, (size_t)B * MP * 12);
    std::vector<int32_t> h_status(B);
    const int o_st = s.out(h_status.data(), (size_t)B * 4);
    const int o_state = state ? s.out(state, (size_t)B * PS * 4) : -1, o_nvox = nvox ? s.out(nvox, (size_t)B * PS * 4) : -1,
              o_info = info ? s.out(info, (size_t)B * PS * 48) : -1;
    hipStream_t st = p->ctx->stream;
    int rc = s.upload(st);
    if (rc) return rc;
    // outputs not written for dropped planes are defined (zero)
    PLANAR_HIP_CHECK(hipMemsetAsync(s.dev<uint8_t>(o_n), 0, s.total - s.items[o_n].off, st));
    if ((rc = planar_plane_clouds_compute_dev(p, s.dev<uint16_t>(i_depth), B, pitch_px, frame_stride_px, fx, fy, cx, cy, depth_factor, s.dev<int32_t>(i_lab), s.dev<double>(i_pl),
                                              s.dev<int32_t>(i_np), dist_th, leaf, s.dev<int32_t>(o_n), s.dev<float>(o_coef), s.dev<int32_t>(o_src), s.dev<int32_t>(o_off),
                                              s.dev<float>(o_pts), s.dev<int32_t>(o_st), state ? s.dev<int32_t>(o_state) : nullptr, nvox ? s.dev<int32_t>(o_nvox) : nullptr,
                                              info ? s.dev<int32_t>(o_info) : nullptr)))
        return rc;
    if ((rc = s.download(st))) return rc;
    p->last_status = h_status;                  // planar_plane_clouds_last_status: the per-frame codes of this call, for callers that branch on them
    for (int b = 0; b < B; b++)
        if (h_status[b]) { set_error("plane_clouds: frame %d exceeded a capacity (code %d: 3 = voxels / index range, 4 = sampler table, 5 = the std::sort order of a plane's points is not reproducible: introsort depth limit / sort workspace)", b, h_status[b]); return PLANAR_ECAPACITY; }
    return PLANAR_OK;
}

int planar_plane_clouds_last_status(planar_plane_clouds* p, int B, int32_t* out) {
    PLANAR_REQUIRE(p && out && B >= 1, PLANAR_EINVAL, "bad argument");
    PLANAR_REQUIRE((size_t)B <= p->last_status.size(), PLANAR_EINVAL, "the last planar_plane_clouds_compute call had fewer frames");
    for (int b = 0; b < B; b++) out[b] = p->last_status[b];
    return PLANAR_OK;
}

int planar_plane_refit(planar_plane_clouds* p, int n_clouds, const float* points, const int32_t* pt_off, double dist_th, float* planes, int32_t* state, int32_t* info) {
    PLANAR_REQUIRE(p && points && pt_off && planes && state, PLANAR_EINVAL, "null argument");
    PLANAR_REQUIRE(n_clouds >= 1, PLANAR_EINVAL, "n_clouds");
    PLANAR_HIP_CHECK(hipSetDevice(p->ctx->device));
    int max_n = 0;
    for (int q = 0; q < n_clouds; q++) { PLANAR_REQUIRE(pt_off[q + 1] >= pt_off[q], PLANAR_EINVAL, "pt_off must be non-decreasing"); max_n = std::max(max_n, pt_off[q + 1] - pt_off[q]); }
    PLANAR_REQUIRE(max_n <= 32768 && pt_off[0] == 0, PLANAR_EINVAL, "a cloud may hold at most 32768 points");
    Stager s;
    const int i_pts = s.in(points, (size_t)pt_off[n_clouds] * 12), i_off = s.in(pt_off, (size_t)(n_clouds + 1) * 4), io_pl = s.inout(planes, (size_t)n_clouds * 16),
              o_st = s.out(state, (size_t)n_clouds * 4), o_info = info ? s.out(info, (size_t)n_clouds * 48) : -1;
    hipStream_t st = p->ctx->stream;
    int rc = s.upload(st);
    if (rc) return rc;
    planepost::Geo G = p->G;
    G.dist_th = dist_th;
    const size_t smem = align_up((size_t)std::max(max_n, 1) * 2, (size_t)16);
    if (smem > 40 * 1024) PLANAR_HIP_CHECK(hipFuncSetAttribute((const void*)planepost::refit_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipLaunchKernelGGL(planepost::refit_kernel, dim3(n_clouds), dim3(64), smem, st, G, n_clouds, s.dev<float>(i_pts), s.dev<int>(i_off), p->rng.as<int>(), s.dev<float>(io_pl),
                       s.dev<int>(o_st), info ? s.dev<int>(o_info) : nullptr);
    PLANAR_HIP_CHECK(hipGetLastError());
    return s.download(st);
}

int planar_flag_matched_plane_points_dev(planar_ctx* ctx, int B, const float* d_Tcw, const float* d_coef, const uint8_t* d_matched, const int32_t* d_n_planes,
                                         int pl_stride, const float* d_xw, int n_points, int points_shared, uint8_t* d_flags, int32_t* d_n_matches) {
    PLANAR_REQUIRE(ctx && d_Tcw && d_coef && d_matched && d_n_planes && d_xw && d_flags, PLANAR_EINVAL, "null argument");
    PLANAR_REQUIRE(B >= 1 && n_points >= 0 && pl_stride >= 1, PLANAR_EINVAL, "bad size");
    if (d_n_matches) PLANAR_HIP_CHECK(hipMemsetAsync(d_n_matches, 0, (size_t)B * 4, ctx->stream));
    if (n_points == 0) return PLANAR_OK;
    hipLaunchKernelGGL(planepost::flag_points_kernel, dim3((n_points + 255) / 256, B), dim3(256), 0, ctx->stream, B, d_Tcw, d_coef, d_matched, d_n_planes, pl_stride, d_xw,
                       n_points, points_shared, d_flags, d_n_matches);
    PLANAR_HIP_CHECK(hipGetLastError());
    return PLANAR_OK;
}

int planar_flag_matched_plane_points(planar_ctx* ctx, int B, const float* Tcw, const float* coef, const uint8_t* matched, const int32_t* n_planes, int pl_stride,
                                     const float* xw, int n_points, int points_shared, uint8_t* flags, int32_t* n_matches) {
    PLANAR_REQUIRE(ctx && Tcw && coef && matched && n_planes && xw && flags, PLANAR_EINVAL, "null argument");
    PLANAR_REQUIRE(B >= 1 && n_points >= 0 && pl_stride >= 1, PLANAR_EINVAL, "bad size");
    PLANAR_HIP_CHECK(hipSetDevice(ctx->device));
    Stager s;
    const int i_T = s.in(Tcw, (size_t)B * 64), i_c = s.in(coef, (size_t)B * pl_stride * 16), i_m = s.in(matched, (size_t)B * pl_stride), i_n = s.in(n_planes, (size_t)B * 4),
              i_x = s.in(xw, (size_t)(points_shared ? 1 : B) * n_points * 12), io_f = s.inout(flags, (size_t)B * n_points), o_nm = n_matches ? s.out(n_matches, (size_t)B * 4) : -1;
    int rc = s.upload(ctx->stream);
    if (rc) return rc;
    if ((rc = planar_flag_matched_plane_points_dev(ctx, B, s.dev<float>(i_T), s.dev<float>(i_c), s.dev<uint8_t>(i_m), s.dev<int32_t>(i_n), pl_stride, s.dev<float>(i_x), n_points,
                                                   points_shared, s.dev<uint8_t>(io_f), n_matches ? s.dev<int32_t>(o_nm) : nullptr)))
        return rc;
    return s.download(ctx->stream);
}

int planar_merge_plane_points(planar_plane_clouds* p, const double* Twc, const float* frame_points, int n_frame, const float* map_points, int n_map, float leaf,
                              float* out_points, int out_cap, int32_t* n_out) {
    PLANAR_REQUIRE(p && Twc && n_out && out_points && (frame_points || !n_frame) && (map_points || !n_map), PLANAR_EINVAL, "null argument");
    PLANAR_REQUIRE(n_frame >= 0 && n_map >= 0 && n_frame + n_map <= 65536 && leaf > 0.f, PLANAR_EINVAL, "bad size (at most 65536 points)");
    PLANAR_HIP_CHECK(hipSetDevice(p->ctx->device));
    const int n = n_frame + n_map;
    Stager s;
    const int i_T = s.in(Twc, 128), i_f = s.in(frame_points, (size_t)n_frame * 12), i_m = s.in(map_points, (size_t)n_map * 12), t_all = s.add(nullptr, nullptr, (size_t)std::max(n, 1) * 12),
              t_out = s.add(nullptr, nullptr, (size_t)p->G.max_points * 12);
    int32_t h[2] = {0, 0};
    const int o_h = s.out(h, 8);
    hipStream_t st = p->ctx->stream;
    int rc = s.upload(st);
    if (rc) return rc;
    planepost::Geo G = p->G;
    G.leaf = leaf;
    if (n) hipLaunchKernelGGL(planepost::merge_gather_kernel, dim3((n + 255) / 256), dim3(256), 0, st, s.dev<double>(i_T), s.dev<float>(i_f), n_frame, s.dev<float>(i_m), n_map, s.dev<float>(t_all));
    unsigned char* ws = p->ws.as<unsigned char>();
    hipLaunchKernelGGL(planepost::cloud_voxels_kernel, dim3(1), dim3(planepost::NT), p->smem_cloud, st, G, s.dev<float>(t_all), n, ws);
    hipLaunchKernelGGL(planepost::plane_sort_global<planepost::PS_SHIFT>, dim3(1), dim3(planepost::PS_T), p->smem_sort_g, st, G, ws, p->sort_rows);
    hipLaunchKernelGGL(planepost::plane_sort_lds<planepost::PS_SHIFT>, dim3(1, planepost::PS_EARLY + planepost::PS_R), dim3(planepost::PS_LT), p->smem_sort_l, st, G, ws);
    for (int c = 0; c < 2; c++) {                            // the LDS tier's fallback jobs (the global tier's ran inside plane_sort_lds)
        const planepost::HeapClass& H = planepost::PS_HC[c];
        hipLaunchKernelGGL(planepost::plane_sort_heap<planepost::PS_SHIFT>, dim3(1, H.wgs), dim3(64 * H.waves), (size_t)H.cap * 4 * H.waves, st, G, ws, 1, c ? planepost::PS_HC[c - 1].max_len + 1 : 0, H.max_len, H.cap);
    }
    hipLaunchKernelGGL(planepost::cloud_sums_kernel, dim3(1), dim3(planepost::NT), 0, st, G, s.dev<float>(t_all), ws, s.dev<float>(t_out), s.dev<int>(o_h), s.dev<int>(o_h) + 1);
    PLANAR_HIP_CHECK(hipGetLastError());
    if ((rc = s.download(st))) return rc;
    if (h[1]) { set_error("merge_plane_points: more than %d voxels, voxel index overflow, or a std::sort order that is not reproducible (code %d)", p->G.max_points, h[1]); return PLANAR_ECAPACITY; }
    PLANAR_REQUIRE(h[0] <= out_cap, PLANAR_ECAPACITY, "out_cap too small");
    *n_out = h[0];
    if (h[0]) PLANAR_HIP_CHECK(hipMemcpy(out_points, s.dev<float>(t_out), (size_t)h[0] * 12, hipMemcpyDeviceToHost));
    return PLANAR_OK;
}

}  // extern "C"
