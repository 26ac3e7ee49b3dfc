// planarslam_amd/csrc/match.hip — brute-force Hamming matchers for MI355X (gfx950).
//
//   planar_hamming_knn            cv::BFMatcher(NORM_HAMMING).match / knnMatch(k=2)
//                                 (reference call sites src/ORBmatcher.cc:1346-1347, src/LSDmatcher.cpp:249-254)
//   planar_match_orb_points       ORBmatcher::MatchORBPoints            (src/ORBmatcher.cc:1332-1394)
//   planar_lsd_search_by_descriptor LSDmatcher::SearchByDescriptor       (src/LSDmatcher.cpp:242-279)
//
// One thread per query descriptor (held in 4 x u64 registers); the train set streams through LDS in
// tiles of 256 descriptors and is read back as wave-uniform (broadcast) 16-byte loads, so a tile costs
// 2 ds_read_b128 + 4 v_xor/popcount pairs per (query, train) pair.  Integer throughput bound
// (10^6 pairs = 4x10^6 64-bit popcounts per 1000x1000 frame pair), not HBM bound.
#include "common.h"

#include <algorithm>

namespace planar {
namespace match {

constexpr int NT = 256;

template <int K>
__global__ __launch_bounds__(NT) void hamming_knn_kernel(const uint8_t* __restrict__ q, const int32_t* __restrict__ nq, int q_stride,
                                                         const uint8_t* __restrict__ t, const int32_t* __restrict__ nt, int t_stride,
                                                         int32_t* __restrict__ idx, int32_t* __restrict__ dist) {
    __shared__ ulonglong2 tile[NT * 2];
    const int b = blockIdx.y, tid = threadIdx.x;
    const int qi = blockIdx.x * NT + tid;
    const int NQ = nq[b], NTr = nt[b];
    if (blockIdx.x * NT >= NQ) return;
    const bool active = qi < NQ;
    ulonglong2 q0 = {0, 0}, q1 = {0, 0};
    if (active) {
        const ulonglong2* qp = (const ulonglong2*)(q + ((size_t)b * q_stride + qi) * 32);
        q0 = qp[0]; q1 = qp[1];
    }
    int bi0 = -1, bi1 = -1, bd0 = 0x7fffffff, bd1 = 0x7fffffff;
    const uint8_t* tb = t + (size_t)b * t_stride * 32;
    for (int j0 = 0; j0 < NTr; j0 += NT) {
        __syncthreads();
        if (j0 + tid < NTr) {
            const ulonglong2* tp = (const ulonglong2*)(tb + (size_t)(j0 + tid) * 32);
            tile[2 * tid] = tp[0]; tile[2 * tid + 1] = tp[1];
        }
        __syncthreads();
        const int n = min(NT, NTr - j0);
        for (int j = 0; j < n; j++) {
            const ulonglong2 a = tile[2 * j], c = tile[2 * j + 1];
            const int d = __popcll(q0.x ^ a.x) + __popcll(q0.y ^ a.y) + __popcll(q1.x ^ c.x) + __popcll(q1.y ^ c.y);
            if (d < bd0) { bd1 = bd0; bi1 = bi0; bd0 = d; bi0 = j0 + j; }
            else if (K == 2 && d < bd1) { bd1 = d; bi1 = j0 + j; }
        }
    }
    if (active) {
        const size_t o = ((size_t)b * q_stride + qi) * K;
        idx[o] = bi0; dist[o] = bd0;
        if (K == 2) { idx[o + 1] = bi1; dist[o + 1] = bd1; }
    }
}

// ORBmatcher::MatchORBPoints post-step (src/ORBmatcher.cc:1350-1392), one workgroup per frame pair.
__global__ __launch_bounds__(NT) void match_orb_points_kernel(const int32_t* __restrict__ idx, const int32_t* __restrict__ dist,
                                                              const int32_t* __restrict__ n_cur, const int32_t* __restrict__ n_last,
                                                              int cur_stride, int last_stride, const uint8_t* __restrict__ last_has_mp,
                                                              const uint8_t* __restrict__ last_outlier, int32_t* __restrict__ cur_match,
                                                              int32_t* __restrict__ npair) {
    __shared__ int s_red[4];
    __shared__ int s_base;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int NC = n_cur[b], NL = n_last[b];
    if (NL == 0 || NC == 0) { if (tid == 0) npair[b] = 0; return; }
    const int32_t* I = idx + (size_t)b * cur_stride;
    const int32_t* D = dist + (size_t)b * cur_stride;
    int m = 1000;   // double min_dist = 1000 (:1350)
    for (int i = tid; i < NC; i += NT) m = min(m, D[i]);
    for (int o = 32; o > 0; o >>= 1) m = min(m, __shfl_xor(m, o));
    if (lane == 0) s_red[wave] = m;
    if (tid == 0) s_base = 0;
    __syncthreads();
    m = min(min(s_red[0], s_red[1]), min(s_red[2], s_red[3]));
    const double thr = fmax(2.0 * (double)m, 15.0);
    const uint8_t* has = last_has_mp + (size_t)b * last_stride;
    const uint8_t* outl = last_outlier + (size_t)b * last_stride;
    int32_t* out = cur_match + (size_t)b * cur_stride;
    for (int i0 = 0; i0 < NC; i0 += NT) {
        const int i = i0 + tid;
        const bool good = i < NC && (double)D[i] < thr;
        const unsigned long long bal = __ballot(good);
        __syncthreads();
        if (lane == 0) s_red[wave] = __popcll(bal);
        __syncthreads();
        int rank = s_base + __popcll(bal & ((1ull << lane) - 1ull));
        for (int w = 0; w < wave; w++) rank += s_red[w];
        if (good) {
            const int tr = I[i];
            // quirk (:1385): outlier flag indexed by the good-match counter
            if (has[tr] && !(rank < NL && outl[rank])) out[i] = tr;
        }
        __syncthreads();
        if (tid == 0) s_base += s_red[0] + s_red[1] + s_red[2] + s_red[3];
    }
    __syncthreads();
    if (tid == 0) npair[b] = s_base;
}

// LSDmatcher::SearchByDescriptor post-step (src/LSDmatcher.cpp:262-277): the reference's sequential
// overwrite "last passing query wins" == max over passing query indices.
__global__ __launch_bounds__(NT) void lsd_assign_kernel(const int32_t* __restrict__ idx, const int32_t* __restrict__ dist,
                                                        const int32_t* __restrict__ n_kf, const int32_t* __restrict__ n_cur,
                                                        int kf_stride, int cur_stride, const uint8_t* __restrict__ kf_has_ml,
                                                        int32_t* __restrict__ cur_match, int32_t* __restrict__ nmatches) {
    __shared__ int s_cnt;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int NK = n_kf[b], NC = n_cur[b];
    int32_t* out = cur_match + (size_t)b * cur_stride;
    for (int i = tid; i < NC; i += NT) out[i] = -1;
    if (tid == 0) s_cnt = 0;
    __syncthreads();
    if (NC >= 2) {
        for (int i = tid; i < NK; i += NT) {
            const size_t o = ((size_t)b * kf_stride + i) * 2;
            const double r = (double)((float)dist[o] / (float)dist[o + 1]);
            if (r < (double)(1.0f / 1.5f) && kf_has_ml[(size_t)b * kf_stride + i]) {
                atomicMax(&out[idx[o]], i);
                atomicAdd(&s_cnt, 1);
            }
        }
    }
    __syncthreads();
    if (tid == 0) nmatches[b] = s_cnt;
}

static int launch_knn(planar_ctx* ctx, const uint8_t* q, const int32_t* nq, int q_stride, const uint8_t* t, const int32_t* nt,
                      int t_stride, int B, int k, int32_t* idx, int32_t* dist) {
    dim3 grid((q_stride + NT - 1) / NT, B);
    if (k == 1) hipLaunchKernelGGL(hamming_knn_kernel<1>, grid, dim3(NT), 0, ctx->stream, q, nq, q_stride, t, nt, t_stride, idx, dist);
    else hipLaunchKernelGGL(hamming_knn_kernel<2>, grid, dim3(NT), 0, ctx->stream, q, nq, q_stride, t, nt, t_stride, idx, dist);
    PLANAR_HIP_CHECK(hipGetLastError());
    return PLANAR_OK;
}

// MapPoint::ComputeDistinctiveDescriptors (reference src/MapPoint.cc:259-324) for a batch of map points: one wavefront per point.  The point's n descriptors are
// staged in LDS; lane i owns rows i, i + 64, ...: the median of row i (sorted[(n - 1) / 2], the row includes the zero of the diagonal) is the smallest v with
// #{j : d(i, j) <= v} > (n - 1) / 2, found by bisection over v in [0, 256] (nine passes over the row, distances recomputed: 8 popcounts each); the winner is the
// smallest (median, index).  best: index within the point's list, -1 without observations, -2 if it has more than max_obs.
__global__ __launch_bounds__(64) void distinctive_kernel(const unsigned char* __restrict__ desc, const int* __restrict__ off, int max_obs, int* __restrict__ best,
                                                         int* __restrict__ median) {
    extern __shared__ unsigned s_desc[];                          // [n][8] words
    const int p = blockIdx.x, lane = threadIdx.x;
    const int o0 = off[p], n = off[p + 1] - o0;
    if (n <= 0 || n > max_obs) { if (lane == 0) { best[p] = n <= 0 ? -1 : -2; if (median) median[p] = 0; } return; }
    const unsigned* src = (const unsigned*)(desc + (size_t)o0 * 32);
    for (int t = lane; t < n * 8; t += 64) s_desc[t] = src[t];
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();
    const int r = (n - 1) >> 1;                                   // int(0.5 * (n - 1))
    unsigned key = 0xffffffffu;                                   // median << 11 | row, smallest wins
    for (int i = lane; i < n; i += 64) {
        unsigned a[8];
#pragma unroll
        for (int w = 0; w < 8; w++) a[w] = s_desc[i * 8 + w];
        int lo = 0, hi = 256;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            int c = 0;
            for (int j = 0; j < n; j++) {
                int d = 0;
#pragma unroll
                for (int w = 0; w < 8; w++) d += __popc(a[w] ^ s_desc[j * 8 + w]);
                c += d <= mid;
            }
            if (c > r) hi = mid; else lo = mid + 1;
        }
        key = min(key, ((unsigned)lo << 11) | (unsigned)i);
    }
    for (int o = 32; o >= 1; o >>= 1) key = min(key, (unsigned)__shfl_xor((int)key, o, 64));
    if (lane == 0) { best[p] = (int)(key & 2047u); if (median) median[p] = (int)(key >> 11); }
}

}  // namespace match
}  // namespace planar

using namespace planar;

extern "C" {

int planar_hamming_knn_dev(planar_ctx* ctx, const uint8_t* d_q, const int32_t* d_nq, int q_stride, const uint8_t* d_t,
                           const int32_t* d_nt, int t_stride, int B, int k, int32_t* d_idx, int32_t* d_dist) {
    PLANAR_REQUIRE(ctx && d_q && d_nq && d_t && d_nt && d_idx && d_dist, PLANAR_EINVAL, "null argument");
    PLANAR_REQUIRE(B >= 1 && q_stride >= 1 && t_stride >= 1 && (k == 1 || k == 2), PLANAR_EINVAL, "bad sizes (k must be 1 or 2)");
    return match::launch_knn(ctx, d_q, d_nq, q_stride, d_t, d_nt, t_stride, B, k, d_idx, d_dist);
}

int planar_hamming_knn(planar_ctx* ctx, const uint8_t* q, const int32_t* nq, int q_stride, const uint8_t* t, const int32_t* nt,
                       int t_stride, int B, int k, int32_t* idx, int32_t* dist) {
    PLANAR_REQUIRE(ctx && q && nq && t && nt && idx && dist, PLANAR_EINVAL, "null argument");
    PLANAR_REQUIRE(B >= 1 && q_stride >= 1 && t_stride >= 1 && (k == 1 || k == 2), PLANAR_EINVAL, "bad sizes (k must be 1 or 2)");
    PLANAR_HIP_CHECK(hipSetDevice(ctx->device));
    Stager s;
    const int iq = s.in(q, (size_t)B * q_stride * 32), inq = s.in(nq, (size_t)B * 4), it = s.in(t, (size_t)B * t_stride * 32), int_ = s.in(nt, (size_t)B * 4);
    const int ii = s.out(idx, (size_t)B * q_stride * k * 4), id = s.out(dist, (size_t)B * q_stride * k * 4);
    int rc = s.upload(ctx->stream);
    if (rc) return rc;
    PLANAR_HIP_CHECK(hipMemsetAsync(s.dev<int32_t>(ii), 0xff, (size_t)B * q_stride * k * 4, ctx->stream));
    PLANAR_HIP_CHECK(hipMemsetAsync(s.dev<int32_t>(id), 0, (size_t)B * q_stride * k * 4, ctx->stream));
    rc = match::launch_knn(ctx, s.dev<uint8_t>(iq), s.dev<int32_t>(inq), q_stride, s.dev<uint8_t>(it), s.dev<int32_t>(int_), t_stride, B, k,
                           s.dev<int32_t>(ii), s.dev<int32_t>(id));
    if (rc) return rc;
    return s.download(ctx->stream);
}

int planar_match_orb_points_dev(planar_ctx* ctx, const uint8_t* d_cur, const int32_t* d_n_cur, int cur_stride, const uint8_t* d_last,
                                const int32_t* d_n_last, int last_stride, const uint8_t* d_last_has_mp, const uint8_t* d_last_outlier,
                                int B, int32_t* d_cur_match, int32_t* d_npair) {
    PLANAR_REQUIRE(ctx && d_cur && d_n_cur && d_last && d_n_last && d_last_has_mp && d_last_outlier && d_cur_match && d_npair, PLANAR_EINVAL, "null argument");
    PLANAR_REQUIRE(B >= 1 && cur_stride >= 1 && last_stride >= 1, PLANAR_EINVAL, "bad sizes");
    const size_t n = (size_t)B * cur_stride;
    int rc = ctx->ensure_scratch(2 * n * 4);
    if (rc) return rc;
    int32_t* idx = ctx->scratch.as<int32_t>();
    int32_t* dist = idx + n;
    rc = match::launch_knn(ctx, d_cur, d_n_cur, cur_stride, d_last, d_n_last, last_stride, B, 1, idx, dist);
    if (rc) return rc;
    hipLaunchKernelGGL(match::match_orb_points_kernel, dim3(B), dim3(match::NT), 0, ctx->stream, idx, dist, d_n_cur, d_n_last, cur_stride,
                       last_stride, d_last_has_mp, d_last_outlier, d_cur_match, d_npair);
    PLANAR_HIP_CHECK(hipGetLastError());
    return PLANAR_OK;
}

int planar_match_orb_points(planar_ctx* ctx, const uint8_t* cur, const int32_t* n_cur, int cur_stride, const uint8_t* last,
                            const int32_t* n_last, int last_stride, const uint8_t* last_has_mp, const uint8_t* last_outlier, int B,
                            int32_t* cur_match, int32_t* npair) {
    PLANAR_REQUIRE(ctx && cur && n_cur && last && n_last && last_has_mp && last_outlier && cur_match && npair, PLANAR_EINVAL, "null argument");
    PLANAR_REQUIRE(B >= 1 && cur_stride >= 1 && last_stride >= 1, PLANAR_EINVAL, "bad sizes");
    PLANAR_HIP_CHECK(hipSetDevice(ctx->device));
    Stager s;
    const int a = s.in(cur, (size_t)B * cur_stride * 32), b = s.in(n_cur, (size_t)B * 4), c = s.in(last, (size_t)B * last_stride * 32),
              d = s.in(n_last, (size_t)B * 4), e = s.in(last_has_mp, (size_t)B * last_stride), f = s.in(last_outlier, (size_t)B * last_stride);
    const int g = s.inout(cur_match, (size_t)B * cur_stride * 4), h = s.out(npair, (size_t)B * 4);
    int rc = s.upload(ctx->stream);
    if (rc) return rc;
    rc = planar_match_orb_points_dev(ctx, s.dev<uint8_t>(a), s.dev<int32_t>(b), cur_stride, s.dev<uint8_t>(c), s.dev<int32_t>(d), last_stride,
                                     s.dev<uint8_t>(e), s.dev<uint8_t>(f), B, s.dev<int32_t>(g), s.dev<int32_t>(h));
    if (rc) return rc;
    return s.download(ctx->stream);
}

int planar_lsd_search_by_descriptor_dev(planar_ctx* ctx, const uint8_t* d_kf, const int32_t* d_n_kf, int kf_stride, const uint8_t* d_cur,
                                        const int32_t* d_n_cur, int cur_stride, const uint8_t* d_kf_has_ml, int B, int32_t* d_cur_match,
                                        int32_t* d_nmatches) {
    PLANAR_REQUIRE(ctx && d_kf && d_n_kf && d_cur && d_n_cur && d_kf_has_ml && d_cur_match && d_nmatches, PLANAR_EINVAL, "null argument");
    PLANAR_REQUIRE(B >= 1 && kf_stride >= 1 && cur_stride >= 1, PLANAR_EINVAL, "bad sizes");
    const size_t n = (size_t)B * kf_stride * 2;
    int rc = ctx->ensure_scratch(2 * n * 4);
    if (rc) return rc;
    int32_t* idx = ctx->scratch.as<int32_t>();
    int32_t* dist = idx + n;
    rc = match::launch_knn(ctx, d_kf, d_n_kf, kf_stride, d_cur, d_n_cur, cur_stride, B, 2, idx, dist);
    if (rc) return rc;
    hipLaunchKernelGGL(match::lsd_assign_kernel, dim3(B), dim3(match::NT), 0, ctx->stream, idx, dist, d_n_kf, d_n_cur, kf_stride, cur_stride,
                       d_kf_has_ml, d_cur_match, d_nmatches);
    PLANAR_HIP_CHECK(hipGetLastError());
    return PLANAR_OK;
}

int planar_lsd_search_by_descriptor(planar_ctx* ctx, const uint8_t* kf, const int32_t* n_kf, int kf_stride, const uint8_t* cur,
                                    const int32_t* n_cur, int cur_stride, const uint8_t* kf_has_ml, int B, int32_t* cur_match,
                                    int32_t* nmatches) {
    PLANAR_REQUIRE(ctx && kf && n_kf && cur && n_cur && kf_has_ml && cur_match && nmatches, PLANAR_EINVAL, "null argument");
    PLANAR_REQUIRE(B >= 1 && kf_stride >= 1 && cur_stride >= 1, PLANAR_EINVAL, "bad sizes");
    PLANAR_HIP_CHECK(hipSetDevice(ctx->device));
    Stager s;
    const int a = s.in(kf, (size_t)B * kf_stride * 32), b = s.in(n_kf, (size_t)B * 4), c = s.in(cur, (size_t)B * cur_stride * 32),
              d = s.in(n_cur, (size_t)B * 4), e = s.in(kf_has_ml, (size_t)B * kf_stride);
    const int g = s.out(cur_match, (size_t)B * cur_stride * 4), h = s.out(nmatches, (size_t)B * 4);
    int rc = s.upload(ctx->stream);
    if (rc) return rc;
    rc = planar_lsd_search_by_descriptor_dev(ctx, s.dev<uint8_t>(a), s.dev<int32_t>(b), kf_stride, s.dev<uint8_t>(c), s.dev<int32_t>(d),
                                             cur_stride, s.dev<uint8_t>(e), B, s.dev<int32_t>(g), s.dev<int32_t>(h));
    if (rc) return rc;
    return s.download(ctx->stream);
}

int planar_distinctive_descriptors_dev(planar_ctx* ctx, int n_points, const uint8_t* d_desc, const int32_t* d_off, int max_obs, int32_t* d_best, int32_t* d_median) {
    PLANAR_REQUIRE(ctx && d_desc && d_off && d_best, PLANAR_EINVAL, "null argument");
    PLANAR_REQUIRE(n_points >= 1 && max_obs >= 1 && max_obs <= 2047, PLANAR_EINVAL, "n_points >= 1, 1 <= max_obs <= 2047");
    const size_t smem = (size_t)max_obs * 32;
    if (smem > 40 * 1024) PLANAR_HIP_CHECK(hipFuncSetAttribute((const void*)match::distinctive_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipLaunchKernelGGL(match::distinctive_kernel, dim3(n_points), dim3(64), smem, ctx->stream, d_desc, d_off, max_obs, d_best, d_median);
    PLANAR_HIP_CHECK(hipGetLastError());
    return PLANAR_OK;
}

int planar_distinctive_descriptors(planar_ctx* ctx, int n_points, const uint8_t* desc, const int32_t* off, int32_t* best, int32_t* median) {
    PLANAR_REQUIRE(ctx && desc && off && best, PLANAR_EINVAL, "null argument");
    PLANAR_REQUIRE(n_points >= 1 && off[0] == 0, PLANAR_EINVAL, "n_points >= 1, off[0] == 0");
    int max_obs = 1;
    for (int p = 0; p < n_points; p++) { PLANAR_REQUIRE(off[p + 1] >= off[p], PLANAR_EINVAL, "off must be non-decreasing"); max_obs = std::max(max_obs, off[p + 1] - off[p]); }
    PLANAR_REQUIRE(max_obs <= 2047, PLANAR_EINVAL, "a map point may have at most 2047 observations");
    PLANAR_HIP_CHECK(hipSetDevice(ctx->device));
    Stager s;
    const int i_d = s.in(desc, (size_t)off[n_points] * 32), i_o = s.in(off, (size_t)(n_points + 1) * 4), o_b = s.out(best, (size_t)n_points * 4),
              o_m = median ? s.out(median, (size_t)n_points * 4) : -1;
    int rc = s.upload(ctx->stream);
    if (rc) return rc;
    if ((rc = planar_distinctive_descriptors_dev(ctx, n_points, s.dev<uint8_t>(i_d), s.dev<int32_t>(i_o), max_obs, s.dev<int32_t>(o_b), median ? s.dev<int32_t>(o_m) : nullptr))) return rc;
    return s.download(ctx->stream);
}

}  // extern "C"
