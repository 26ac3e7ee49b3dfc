// planarslam_amd/csrc/guided.hip — guided matchers for MI355X (gfx950): SURVEY.md §8 rows a20, a21, a22, a24, a25.
//
//   planar_search_by_projection_frame   ORBmatcher::SearchByProjection(Frame&, const Frame&, th, bMono)  src/ORBmatcher.cc:1396-1535
//   planar_search_by_projection_map     ORBmatcher::SearchByProjection(Frame&, vector<MapPoint*>&, th)   src/ORBmatcher.cc:46-130
//   planar_search_by_bow                ORBmatcher::SearchByBoW(KeyFrame*, Frame&, ...)                  src/ORBmatcher.cc:160-292
//   planar_lsd_search_by_projection     LSDmatcher::SearchByProjection + Frame::GetLinesInArea           src/LSDmatcher.cpp:141-211, src/Frame.cc:491-524
//   planar_plane_search_by_coefficients PlaneMatcher::SearchMapByCoefficients                           src/PlaneMatcher.cpp:10-79
//   planar_fuse_search                  ORBmatcher::Fuse(KeyFrame*, vpMapPoints, th), the search half   src/ORBmatcher.cc:829-951
//   planar_lsd_fuse_search              LSDmatcher::Fuse(KeyFrame*, vpMapLines, th), the search half    src/LSDmatcher.cpp:884-991
//
// The reference resolves probes one after another and every assignment changes what later probes may take
// ("mvpMapPoints[i2]->Observations() > 0"), so the result depends on probe order.  The kernels keep that order
// but split each probe into an order-free part and an order-bound part:
//   * order-free, all 256 threads of the frame's workgroup: projection, 64x48 grid window walk
//     (Frame::GetFeaturesInArea order: column-major cells, ascending keypoint index inside a cell), level /
//     window / stereo gates and the 256-bit Hamming distance of every surviving candidate.  Candidates of a
//     chunk of up to 256 consecutive probes are packed (dist | octave | index) into an LDS list;
//   * order-bound, one wavefront: for each probe in order, lanes read its candidate list, drop the ones whose
//     "blocked" bit is set NOW, take best / second best by a wave-wide min over (dist, list position) — the
//     stable order the reference's `<` comparisons induce — and update the blocked bits.
// The chunk size adapts to the LDS list capacity, so there is no overflow path.  One workgroup per frame
// (pair); the batch dimension fills the GPU.  Integer / float32 work, bit-exact with the oracle.
#include "common.h"

namespace planar {
namespace guided {

constexpr int NT = 256;
constexpr int NCELL = PLANAR_GRID_COLS * PLANAR_GRID_ROWS;
constexpr int MAXN = PLANAR_MAX_FRAME_KEYS;
constexpr int CAND_CAP = 8192;       // candidates of one chunk of probes; with it the workgroup needs 61 KB of LDS (two per CU)
constexpr int TH_HIGH = 100, TH_LOW = 50, HISTO_LENGTH = 30;   // src/ORBmatcher.cc:38-40

enum { MODE_FRAME = 0, MODE_MAP = 1, MODE_BOW = 2 };

struct Lds {
    uint32_t cand[CAND_CAP];       // dist << 16 | octave << 12 | index ; doubles as scratch while the grid is built
    uint16_t cell_start[NCELL + 1];
    uint16_t items[MAXN];
    uint32_t blocked[MAXN / 32];
    int pid[NT];
    int poff[NT + 1];
    uint16_t ev_idx[MAXN];
    uint8_t ev_bin[MAXN];
    int hist[HISTO_LENGTH];
    int keep[3];
    int n_ev, nmatches, m_fit, wsum[NT / 64];
};

__device__ inline uint32_t wave_min_u32(uint32_t v) {
    for (int o = 32; o >= 1; o >>= 1) v = min(v, (uint32_t)__shfl_xor((int)v, o, 64));
    return v;
}

// exclusive scan of one int per thread over the workgroup; returns the exclusive prefix, total in *total
__device__ inline int block_exscan(int v, int* wsum, int* total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int inc = v;
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    __syncthreads();
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    int base = 0, tot = 0;
    for (int i = 0; i < NT / 64; i++) { if (i < w) base += wsum[i]; tot += wsum[i]; }
    *total = tot;
    return base + inc - v;
}

__device__ inline int hamming256(const uint32_t* a, const uint8_t* b) {
    const uint4* p = (const uint4*)b;
    const uint4 x = p[0], y = p[1];
    return __popc(a[0] ^ x.x) + __popc(a[1] ^ x.y) + __popc(a[2] ^ x.z) + __popc(a[3] ^ x.w) + __popc(a[4] ^ y.x) + __popc(a[5] ^ y.y) +
           __popc(a[6] ^ y.z) + __popc(a[7] ^ y.w);
}

__device__ inline void load_desc(uint32_t* a, const uint8_t* p) {
    const uint4* q = (const uint4*)p;
    const uint4 x = q[0], y = q[1];
    a[0] = x.x; a[1] = x.y; a[2] = x.z; a[3] = x.w; a[4] = y.x; a[5] = y.y; a[6] = y.z; a[7] = y.w;
}

// Frame::AssignFeaturesToGrid (src/Frame.cc:155-166, PosInGrid :526-535) into cell_start / items.
template <typename L>
__device__ void build_grid(L& s, const planar_frame_view& f, const planar_keypoint* keys, int N) {
    const int tid = threadIdx.x;
    uint32_t* cnt = s.cand;            // [NCELL] counters, then cursors
    for (int c = tid; c < NCELL; c += NT) cnt[c] = 0;
    __syncthreads();
    for (int i = tid; i < N; i += NT) {
        const int px = (int)roundf((keys[i].x - f.min_x) * f.grid_w_inv);
        const int py = (int)roundf((keys[i].y - f.min_y) * f.grid_h_inv);
        if (px < 0 || px >= PLANAR_GRID_COLS || py < 0 || py >= PLANAR_GRID_ROWS) continue;
        atomicAdd(&cnt[px * PLANAR_GRID_ROWS + py], 1u);
    }
    __syncthreads();
    constexpr int PER = NCELL / NT;    // 12 consecutive cells per thread
    int local = 0;
    for (int k = 0; k < PER; k++) local += (int)cnt[tid * PER + k];
    int total;
    int run = block_exscan(local, s.wsum, &total);
    for (int k = 0; k < PER; k++) {
        const int c = tid * PER + k, n = (int)cnt[c];
        s.cell_start[c] = (uint16_t)run;
        cnt[c] = (uint32_t)run;        // cursor
        run += n;
    }
    if (tid == NT - 1) s.cell_start[NCELL] = (uint16_t)run;
    __syncthreads();
    for (int i = tid; i < N; i += NT) {
        const int px = (int)roundf((keys[i].x - f.min_x) * f.grid_w_inv);
        const int py = (int)roundf((keys[i].y - f.min_y) * f.grid_h_inv);
        if (px < 0 || px >= PLANAR_GRID_COLS || py < 0 || py >= PLANAR_GRID_ROWS) continue;
        const uint32_t pos = atomicAdd(&cnt[px * PLANAR_GRID_ROWS + py], 1u);
        s.items[pos] = (uint16_t)i;
    }
    __syncthreads();
    // push_back order inside a cell is ascending keypoint index: insertion-sort each (tiny) cell list
    for (int k = 0; k < PER; k++) {
        const int c = tid * PER + k;
        const int a = s.cell_start[c], e = s.cell_start[c + 1];
        for (int i = a + 1; i < e; i++) {
            const uint16_t v = s.items[i];
            int j = i - 1;
            while (j >= a && s.items[j] > v) { s.items[j + 1] = s.items[j]; j--; }
            s.items[j + 1] = v;
        }
    }
    __syncthreads();
}

// Frame::GetFeaturesInArea (src/Frame.cc:440-489) + the per-candidate gates of the two SearchByProjection
// loops that do not depend on the assignment state.  emit(idx, octave) is called in the reference's order.
template <typename Emit>
__device__ inline void walk_window(const Lds& s, const planar_frame_view& f, const planar_keypoint* keys, const float* uR, float x, float y,
                                   float r, int minLevel, int maxLevel, float ur, Emit emit) {
    const int nMinCellX = max(0, (int)floorf((x - f.min_x - r) * f.grid_w_inv));
    if (nMinCellX >= PLANAR_GRID_COLS) return;
    const int nMaxCellX = min(PLANAR_GRID_COLS - 1, (int)ceilf((x - f.min_x + r) * f.grid_w_inv));
    if (nMaxCellX < 0) return;
    const int nMinCellY = max(0, (int)floorf((y - f.min_y - r) * f.grid_h_inv));
    if (nMinCellY >= PLANAR_GRID_ROWS) return;
    const int nMaxCellY = min(PLANAR_GRID_ROWS - 1, (int)ceilf((y - f.min_y + r) * f.grid_h_inv));
    if (nMaxCellY < 0) return;
    const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
    for (int ix = nMinCellX; ix <= nMaxCellX; ix++) {
        if (nMinCellY > nMaxCellY) break;
        // cells (ix, nMinCellY..nMaxCellY) are contiguous in the column-major cell order
        const int a = s.cell_start[ix * PLANAR_GRID_ROWS + nMinCellY], e = s.cell_start[ix * PLANAR_GRID_ROWS + nMaxCellY + 1];
        for (int k = a; k < e; k++) {
            const int idx = s.items[k];
            const planar_keypoint kp = keys[idx];
            if (bCheckLevels) {
                if (kp.octave < minLevel) continue;
                if (maxLevel >= 0 && kp.octave > maxLevel) continue;
            }
            const float distx = kp.x - x, disty = kp.y - y;
            if (!(fabsf(distx) < r && fabsf(disty) < r)) continue;
            const float u2 = uR[idx];
            if (u2 > 0) {
                const float er = fabsf(ur - u2);
                if (er > r) continue;
            }
            emit(idx, kp.octave);
        }
    }
}

// cv::gemm float32 small-matrix path (see oracle/guided_oracle.cpp header)
__device__ inline float gemm3_row(float a0, float a1, float a2, const float* x, float c) {
    const float t = a0 * x[0] + a1 * x[1] + a2 * x[2];
    return (float)((double)t * 1.0 + (double)c * 1.0);
}

struct Args {
    planar_frame_view f;
    planar_last_frame_view last;
    planar_map_probes mp;
    float th, nn_ratio;
    int mono, check_orientation;
    int32_t* match;
    int32_t* nmatches;
};

// ORBmatcher::ComputeThreeMaxima (src/ORBmatcher.cc:1666-1708) on bin counts
__device__ inline void three_maxima(const int* h, int& ind1, int& ind2, int& ind3) {
    int max1 = 0, max2 = 0, max3 = 0;
    ind1 = ind2 = ind3 = -1;
    for (int i = 0; i < HISTO_LENGTH; i++) {
        const int sz = h[i];
        if (sz > max1) { max3 = max2; max2 = max1; max1 = sz; ind3 = ind2; ind2 = ind1; ind1 = i; }
        else if (sz > max2) { max3 = max2; max2 = sz; ind3 = ind2; ind2 = i; }
        else if (sz > max3) { max3 = sz; ind3 = i; }
    }
    if ((float)max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
    else if ((float)max3 < 0.1f * (float)max1) ind3 = -1;
}

__device__ inline int rot_bin(float a_from, float a_to) {
    const float factor = 1.0f / HISTO_LENGTH;
    float rot = a_from - a_to;
    if (rot < 0.0f) rot += 360.0f;
    int bin = (int)roundf(rot * factor);
    if (bin == HISTO_LENGTH) bin = 0;
    return bin;
}

// Order-bound part: wavefront 0 resolves probes [0, m) of the current chunk in order.
//   MODE_FRAME: best only, TH_HIGH;  MODE_MAP: best + second with the same-level ratio test;  MODE_BOW: TH_LOW + ratio.
template <int MODE>
__device__ void resolve_chunk(Lds& s, int m, const Args& a, int32_t* match, int b, const float* from_angle, const float* to_angle_f,
                              const planar_keypoint* keys, const uint8_t* observed) {
    const int lane = threadIdx.x;
    volatile uint32_t* blk = s.blocked;
    for (int q = 0; q < m; q++) {
        const int id = s.pid[q];
        const int off = s.poff[q], cnt = s.poff[q + 1] - off;
        if (id < 0 || cnt == 0) continue;
        uint32_t k1 = 0xffffffffu;
        for (int base = 0; base < cnt; base += 64) {
            const int k = base + lane;
            uint32_t key = 0xffffffffu;
            if (k < cnt) {
                const uint32_t e = s.cand[off + k];
                const int idx = e & 0xfff;
                if (!((blk[idx >> 5] >> (idx & 31)) & 1u)) key = ((e >> 16) << 16) | (uint32_t)k;
            }
            k1 = min(k1, key);
        }
        k1 = wave_min_u32(k1);
        if (k1 == 0xffffffffu) continue;
        const int bestDist = (int)(k1 >> 16), bestK = (int)(k1 & 0xffff);
        const uint32_t e1 = s.cand[off + bestK];
        const int bestIdx = e1 & 0xfff, bestLevel = (e1 >> 12) & 0xf;
        int bestDist2 = 256, bestLevel2 = -1;
        if (MODE != MODE_FRAME) {
            uint32_t k2 = 0xffffffffu;
            for (int base = 0; base < cnt; base += 64) {
                const int k = base + lane;
                uint32_t key = 0xffffffffu;
                if (k < cnt && k != bestK) {
                    const uint32_t e = s.cand[off + k];
                    const int idx = e & 0xfff;
                    if (!((blk[idx >> 5] >> (idx & 31)) & 1u)) key = ((e >> 16) << 16) | (uint32_t)k;
                }
                k2 = min(k2, key);
            }
            k2 = wave_min_u32(k2);
            if (k2 != 0xffffffffu) {
                bestDist2 = (int)(k2 >> 16);
                bestLevel2 = (s.cand[off + (k2 & 0xffff)] >> 12) & 0xf;
            }
        }
        bool take;
        if (MODE == MODE_FRAME) take = bestDist <= TH_HIGH;
        else if (MODE == MODE_MAP) take = bestDist <= TH_HIGH && !(bestLevel == bestLevel2 && (float)bestDist > a.nn_ratio * (float)bestDist2);
        else take = bestDist <= TH_LOW && (float)bestDist < a.nn_ratio * (float)bestDist2;
        if (!take) continue;
        if (lane == 0) {
            match[bestIdx] = id;
            const bool now_blocked = MODE == MODE_BOW ? true : (observed[id] != 0);
            const uint32_t w = blk[bestIdx >> 5], bit = 1u << (bestIdx & 31);
            blk[bestIdx >> 5] = now_blocked ? (w | bit) : (w & ~bit);
            s.nmatches++;
            if (MODE != MODE_MAP && a.check_orientation) {
                const float to = MODE == MODE_FRAME ? keys[bestIdx].angle : to_angle_f[bestIdx];
                const int n = s.n_ev++;
                s.ev_idx[n] = (uint16_t)bestIdx;
                s.ev_bin[n] = (uint8_t)rot_bin(from_angle[id], to);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// rotation-consistency post-step shared by MODE_FRAME and MODE_BOW
__device__ void rotation_filter(Lds& s, int32_t* match) {
    const int tid = threadIdx.x;
    if (tid < HISTO_LENGTH) s.hist[tid] = 0;
    __syncthreads();
    const int n = s.n_ev;
    for (int i = tid; i < n; i += NT) atomicAdd(&s.hist[s.ev_bin[i]], 1);
    __syncthreads();
    if (tid == 0) {
        int i1, i2, i3;
        three_maxima(s.hist, i1, i2, i3);
        s.keep[0] = i1; s.keep[1] = i2; s.keep[2] = i3;
        int removed = 0;
        for (int i = 0; i < HISTO_LENGTH; i++)
            if (i != i1 && i != i2 && i != i3) removed += s.hist[i];
        s.nmatches -= removed;
    }
    __syncthreads();
    const int k1 = s.keep[0], k2 = s.keep[1], k3 = s.keep[2];
    for (int i = tid; i < n; i += NT) {
        const int bin = s.ev_bin[i];
        if (bin != k1 && bin != k2 && bin != k3) match[s.ev_idx[i]] = -1;
    }
}

template <int MODE>
__global__ __launch_bounds__(NT) void projection_kernel(Args a) {
    extern __shared__ __align__(16) uint8_t lds_raw[];
    Lds& s = *(Lds*)lds_raw;
    const int b = blockIdx.x, tid = threadIdx.x;
    const planar_frame_view& f = a.f;
    const int N = f.n[b];
    const planar_keypoint* keys = f.keys_un + (size_t)b * f.stride;
    const float* uR = f.u_right + (size_t)b * f.stride;
    const uint8_t* desc = f.desc + (size_t)b * f.stride * 32;
    int32_t* const match_b = a.match + (size_t)b * f.stride;     // (the by-value argument struct is never written: a modified copy would live in scratch)

    for (int w = tid; w < MAXN / 32; w += NT) s.blocked[w] = 0;
    if (tid == 0) { s.n_ev = 0; s.nmatches = 0; }
    build_grid(s, f, keys, N);
    if (f.blocked) {
        const uint8_t* bl = f.blocked + (size_t)b * f.stride;
        for (int i = tid; i < N; i += NT)
            if (bl[i]) atomicOr(&s.blocked[i >> 5], 1u << (i & 31));
    }

    // per-frame constants of the frame-to-frame variant (src/ORBmatcher.cc:1408-1420)
    float Rcw[9], tcw[3];
    bool bForward = false, bBackward = false;
    size_t po;
    int NP;
    const uint8_t *probe_desc, *observed;
    if (MODE == MODE_FRAME) {
        const float* Tc = f.Tcw + (size_t)b * 16;
        const float* Tl = a.last.Tcw + (size_t)b * 16;
        for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) Rcw[3 * r + c] = Tc[4 * r + c]; tcw[r] = Tc[4 * r + 3]; }
        float twc[3];
        for (int i = 0; i < 3; i++) {
            double sum = 0;
            for (int k = 0; k < 3; k++) sum += (double)Rcw[3 * k + i] * (double)tcw[k];
            twc[i] = (float)(sum * -1.0);
        }
        const float tlc2 = gemm3_row(Tl[8], Tl[9], Tl[10], twc, Tl[11]);
        bForward = tlc2 > f.b && !a.mono;
        bBackward = -tlc2 > f.b && !a.mono;
        po = (size_t)b * a.last.stride;
        NP = a.last.n[b];
        probe_desc = a.last.mp_desc + po * 32;
        observed = a.last.mp_observed + po;
    } else {
        po = (size_t)b * a.mp.stride;
        NP = a.mp.n[b];
        probe_desc = a.mp.desc + po * 32;
        observed = a.mp.observed + po;
    }
    const bool bFactor = a.th != 1.0f;
    __syncthreads();

    for (int base = 0; base < NP;) {
        // ---- order-free: parameters of probe base + tid
        const int p = base + tid;
        bool valid = false;
        float u = 0, v = 0, r = 0, ur = 0;
        int minL = -1, maxL = -1;
        if (p < NP) {
            if (MODE == MODE_FRAME) {
                if (a.last.usable[po + p]) {
                    const float* xw = a.last.xw + (po + p) * 3;
                    const float xc = gemm3_row(Rcw[0], Rcw[1], Rcw[2], xw, tcw[0]);
                    const float yc = gemm3_row(Rcw[3], Rcw[4], Rcw[5], xw, tcw[1]);
                    const float zc = gemm3_row(Rcw[6], Rcw[7], Rcw[8], xw, tcw[2]);
                    const float invzc = (float)(1.0 / (double)zc);
                    if (!(invzc < 0)) {
                        u = f.fx * xc * invzc + f.cx;
                        v = f.fy * yc * invzc + f.cy;
                        if (!(u < f.min_x || u > f.max_x) && !(v < f.min_y || v > f.max_y)) {
                            const int oct = a.last.octave[po + p];
                            r = a.th * f.scale_factors[oct];
                            if (bForward) { minL = oct; maxL = -1; }
                            else if (bBackward) { minL = 0; maxL = oct; }
                            else { minL = oct - 1; maxL = oct + 1; }
                            ur = u - f.bf * invzc;
                            valid = true;
                        }
                    }
                }
            } else {
                if (a.mp.in_view[po + p]) {
                    const int lvl = a.mp.level[po + p];
                    float rr = (double)a.mp.view_cos[po + p] > 0.998 ? 2.5f : 4.0f;
                    if (bFactor) rr *= a.th;
                    r = rr * f.scale_factors[lvl];
                    u = a.mp.proj_x[po + p]; v = a.mp.proj_y[po + p]; ur = a.mp.proj_xr[po + p];
                    minL = lvl - 1; maxL = lvl;
                    valid = true;
                }
            }
        }
        int cnt = 0;
        if (valid) walk_window(s, f, keys, uR, u, v, r, minL, maxL, ur, [&](int, int) { cnt++; });
        int total;
        const int off = block_exscan(cnt, s.wsum, &total);
        if (tid == 0) s.m_fit = 0;
        __syncthreads();
        const bool fits = p < NP && off + cnt <= CAND_CAP;
        if (fits) atomicAdd(&s.m_fit, 1);   // prefix property: fits is monotone in tid
        s.pid[tid] = valid ? p : -1;
        s.poff[tid] = off;
        if (tid == NT - 1) s.poff[NT] = total;
        __syncthreads();
        const int m = s.m_fit;
        if (fits && valid && cnt > 0) {
            uint32_t d[8];
            load_desc(d, probe_desc + (size_t)p * 32);
            int k = off;
            walk_window(s, f, keys, uR, u, v, r, minL, maxL, ur, [&](int idx, int oct) {
                const int dist = hamming256(d, desc + (size_t)idx * 32);
                s.cand[k++] = ((uint32_t)dist << 16) | ((uint32_t)(oct & 0xf) << 12) | (uint32_t)idx;
            });
        }
        __syncthreads();
        // ---- order-bound
        if (tid < 64) resolve_chunk<MODE>(s, m, a, match_b, b, MODE == MODE_FRAME ? a.last.angle + po : nullptr, nullptr, keys, observed);
        __syncthreads();
        base += m;
    }
    if (MODE == MODE_FRAME && a.check_orientation) rotation_filter(s, match_b);
    __syncthreads();
    if (tid == 0) a.nmatches[b] = s.nmatches;
}

// ---- a22: SearchByBoW -----------------------------------------------------------------------------
struct BowArgs {
    const int32_t *n_kf, *kf_node, *n_f, *f_node;
    const uint8_t *kf_usable, *kf_desc, *f_desc;
    const float *kf_angle, *f_angle;
    int kf_stride, f_stride;
};

struct BowLds {
    Lds base;
    unsigned long long kkey[MAXN], fkey[MAXN];   // node << 12 | feature index, ascending
};

__device__ void bitonic_sort_u64(unsigned long long* key, int n_pow2) {
    for (int k = 2; k <= n_pow2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            __syncthreads();
            for (int i = threadIdx.x; i < n_pow2; i += NT) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const unsigned long long x = key[i], y = key[ixj];
                    const bool up = (i & k) == 0;
                    if ((x > y) == up) { key[i] = y; key[ixj] = x; }
                }
            }
        }
    __syncthreads();
}

__global__ __launch_bounds__(NT) void bow_kernel(BowArgs g, Args a) {
    extern __shared__ __align__(16) uint8_t lds_raw[];
    BowLds& L = *(BowLds*)lds_raw;
    Lds& s = L.base;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int NK = g.n_kf[b], NF = g.n_f[b];
    const size_t ko = (size_t)b * g.kf_stride, fo = (size_t)b * g.f_stride;
    int32_t* const match_b = a.match + fo;
    for (int w = tid; w < MAXN / 32; w += NT) s.blocked[w] = 0;
    if (tid == 0) { s.n_ev = 0; s.nmatches = 0; }
    for (int i = tid; i < NF; i += NT) match_b[i] = -1;   // vpMapPointMatches = vector<MapPoint*>(F.N, NULL)
    int pk = 1; while (pk < NK) pk <<= 1;
    int pf = 1; while (pf < NF) pf <<= 1;
    for (int i = tid; i < pk; i += NT) {
        const int node = i < NK ? g.kf_node[ko + i] : -1;
        L.kkey[i] = node >= 0 ? ((unsigned long long)node << 12) | (unsigned)i : ~0ull;
    }
    for (int i = tid; i < pf; i += NT) {
        const int node = i < NF ? g.f_node[fo + i] : -1;
        L.fkey[i] = node >= 0 ? ((unsigned long long)node << 12) | (unsigned)i : ~0ull;
    }
    bitonic_sort_u64(L.kkey, pk);
    bitonic_sort_u64(L.fkey, pf);

    for (int base = 0; base < NK;) {
        const int p = base + tid;
        bool valid = false;
        int lo = 0, hi = 0, kf = -1;
        if (p < NK && L.kkey[p] != ~0ull) {
            kf = (int)(L.kkey[p] & 0xfff);
            if (g.kf_usable[ko + kf]) {
                const unsigned long long node = L.kkey[p] >> 12;
                // [lo, hi) = features of F in the same vocabulary node (ascending feature index)
                // padding / node-less entries are ~0 (node field 2^52-1) and sort last, so plain bounds work
                int x = 0, y = NF;
                while (x < y) { const int mid = (x + y) >> 1; if ((L.fkey[mid] >> 12) < node) x = mid + 1; else y = mid; }
                lo = x; y = NF;
                while (x < y) { const int mid = (x + y) >> 1; if ((L.fkey[mid] >> 12) <= node) x = mid + 1; else y = mid; }
                hi = x;
                valid = true;
            }
        }
        const int cnt = valid ? hi - lo : 0;
        int total;
        const int off = block_exscan(cnt, s.wsum, &total);
        if (tid == 0) s.m_fit = 0;
        __syncthreads();
        const bool fits = p < NK && off + cnt <= CAND_CAP;
        if (fits) atomicAdd(&s.m_fit, 1);
        s.pid[tid] = valid ? kf : -1;
        s.poff[tid] = off;
        if (tid == NT - 1) s.poff[NT] = total;
        __syncthreads();
        const int m = s.m_fit;
        if (fits && cnt > 0) {
            uint32_t d[8];
            load_desc(d, g.kf_desc + (ko + kf) * 32);
            for (int k = 0; k < cnt; k++) {
                const int idx = (int)(L.fkey[lo + k] & 0xfff);
                const int dist = hamming256(d, g.f_desc + (fo + idx) * 32);
                s.cand[off + k] = ((uint32_t)dist << 16) | (uint32_t)idx;
            }
        }
        __syncthreads();
        if (tid < 64) resolve_chunk<MODE_BOW>(s, m, a, match_b, b, g.kf_angle + ko, g.f_angle + fo, nullptr, nullptr);
        __syncthreads();
        base += m;
    }
    if (a.check_orientation) rotation_filter(s, match_b);
    __syncthreads();
    if (tid == 0) a.nmatches[b] = s.nmatches;
}

// ---- a24: LSDmatcher::SearchByProjection, one wavefront per frame ----------------------------------
constexpr int MAX_LINES = 1024;

struct LineArgs {
    const int32_t *n_lines, *n_ml, *ml_level;
    const planar_keyline* keylines;
    const uint8_t *ldesc, *blocked, *ml_in_view, *ml_desc, *ml_observed;
    const float *ml_proj, *ml_view_cos;
    int line_stride, ml_stride;
    float scale_factors[PLANAR_MAX_LEVELS];
    float th, nn_ratio;
    int32_t *match, *nmatches;
};

__global__ __launch_bounds__(64) void lsd_projection_kernel(LineArgs a) {
    __shared__ uint32_t blocked[MAX_LINES / 32];
    const int b = blockIdx.x, lane = threadIdx.x;
    const int NLn = a.n_lines[b], NM = a.n_ml[b];
    const size_t lo = (size_t)b * a.line_stride, mo = (size_t)b * a.ml_stride;
    const planar_keyline* kl = a.keylines + lo;
    const uint8_t* ldesc = a.ldesc + lo * 32;
    int32_t* match = a.match + lo;
    volatile uint32_t* blk = blocked;
    for (int w = lane; w < MAX_LINES / 32; w += 64) blocked[w] = 0;
    __builtin_amdgcn_wave_barrier();
    if (a.blocked)
        for (int i = lane; i < NLn; i += 64)
            if (a.blocked[lo + i]) atomicOr(&blocked[i >> 5], 1u << (i & 31));
    __builtin_amdgcn_wave_barrier();
    const bool bFactor = a.th != 1.0f;
    int nmatches = 0;
    for (int j = 0; j < NM; j++) {
        if (!a.ml_in_view[mo + j]) continue;
        const int lvl = a.ml_level[mo + j];
        if (lvl < 0 || lvl >= PLANAR_MAX_LEVELS) continue;   // F.mvScaleFactors[lvl] would be out of bounds in the reference (UB): skipped here
        float r = (double)a.ml_view_cos[mo + j] > 0.998 ? 5.0f : 8.0f;   // LSDmatcher::RadiusByViewingCos, src/LSDmatcher.cpp:369-375
        if (bFactor) r *= a.th;
        const float* pr = a.ml_proj + (mo + j) * 4;
        const float x1 = pr[0], y1 = pr[1], x2 = pr[2], y2 = pr[3];
        const float rr = r * a.scale_factors[lvl];
        const int minLevel = lvl - 1, maxLevel = lvl;
        const bool bCheckLevels = (minLevel > 0) || (maxLevel > 0);
        uint32_t d[8];
        load_desc(d, a.ml_desc + (mo + j) * 32);
        uint32_t k1 = 0xffffffffu, k2 = 0xffffffffu;   // dist << 16 | line index
        // pass 1: best ; pass 2: second best (stable order == ascending line index)
        for (int pass = 0; pass < 2; pass++) {
            uint32_t kmin = 0xffffffffu;
            for (int base = 0; base < NLn; base += 64) {
                const int i = base + lane;
                uint32_t key = 0xffffffffu;
                if (i < NLn && !(pass == 1 && i == (int)(k1 & 0xffff))) {
                    const planar_keyline k = kl[i];
                    const double mx = 0.5 * (double)(x1 + x2) - (double)k.pt_x, my = 0.5 * (double)(y1 + y2) - (double)k.pt_y;
                    const float distance = (float)(mx * mx + my * my);
                    bool ok = !(distance > rr * rr);
                    const float slope = (y1 - y2) / (x1 - x2) - k.angle;
                    if ((double)slope > (double)rr * 0.01) ok = false;
                    if (bCheckLevels) {
                        if (k.octave < minLevel) ok = false;
                        if (maxLevel >= 0 && k.octave > maxLevel) ok = false;
                    }
                    if (ok && !((blk[i >> 5] >> (i & 31)) & 1u)) key = ((uint32_t)hamming256(d, ldesc + (size_t)i * 32) << 16) | (uint32_t)i;
                }
                kmin = min(kmin, key);
            }
            kmin = wave_min_u32(kmin);
            if (pass == 0) { k1 = kmin; if (k1 == 0xffffffffu) break; } else k2 = kmin;
        }
        if (k1 == 0xffffffffu) continue;
        const int bestDist = (int)(k1 >> 16), bestIdx = (int)(k1 & 0xffff);
        const int bestLevel = kl[bestIdx].octave;
        int bestDist2 = 256, bestLevel2 = -1;
        if (k2 != 0xffffffffu) { bestDist2 = (int)(k2 >> 16); bestLevel2 = kl[k2 & 0xffff].octave; }
        if (bestDist <= TH_HIGH) {
            if (bestLevel == bestLevel2 && (float)bestDist > a.nn_ratio * (float)bestDist2) continue;
            if (lane == 0) {
                match[bestIdx] = j;
                const uint32_t w = blk[bestIdx >> 5], bit = 1u << (bestIdx & 31);
                blk[bestIdx >> 5] = a.ml_observed[mo + j] ? (w | bit) : (w & ~bit);
            }
            nmatches++;
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (lane == 0) a.nmatches[b] = nmatches;
}

// ---- a25: PlaneMatcher::SearchMapByCoefficients, one wavefront per (frame, frame plane) -------------
__global__ __launch_bounds__(64) void plane_match_kernel(const int32_t* n_planes, int pl_stride, const float* pl_coef, const float* Tcw,
                                                         int map_shared, const int32_t* n_mp, int mp_stride, const uint8_t* mp_valid,
                                                         const float* mp_coef, const int32_t* mp_npts, int pts_stride, const float* mp_pts,
                                                         float dTh, float aTh, float verTh, float parTh, int32_t* match, int32_t* ver,
                                                         int32_t* par, int32_t* nmatches) {
    const int b = blockIdx.y, i = blockIdx.x, lane = threadIdx.x;
    if (i >= n_planes[b]) return;
    const int m = map_shared ? 0 : b;
    const size_t po = (size_t)b * pl_stride + i, mo = (size_t)m * mp_stride;
    const float* T = Tcw + (size_t)b * 16;
    const float* c = pl_coef + po * 4;
    float pM[4];
    for (int r = 0; r < 4; r++) {   // Frame::ComputePlaneWorldCoeff: transpose(mTcw) * coef
        const float t = T[r] * c[0] + T[4 + r] * c[1] + T[8 + r] * c[2] + T[12 + r] * c[3];
        pM[r] = (float)((double)t * 1.0);
    }
    float ldTh = dTh, lverTh = verTh, lparTh = parTh;
    bool found = false;
    int im = -1, iv = -1, ip = -1;
    const int NM = n_mp[m];
    for (int j = 0; j < NM; j++) {
        if (!mp_valid[mo + j]) continue;
        const float* pW = mp_coef + (mo + j) * 4;
        const float angle = pM[0] * pW[0] + pM[1] * pW[1] + pM[2] * pW[2];
        if (angle > aTh || angle < -aTh) {
            // PointDistanceFromPlane: min over the boundary cloud (order-free), lanes over points
            const float* pts = mp_pts + (mo + j) * (size_t)pts_stride * 3;
            const int np = mp_npts[mo + j];
            float res = 100.0f;
            for (int k = lane; k < np; k += 64) {
                const float dis = fabsf(pM[0] * pts[3 * k] + pM[1] * pts[3 * k + 1] + pM[2] * pts[3 * k + 2] + pM[3]);
                res = fminf(res, dis);
            }
            for (int o = 32; o >= 1; o >>= 1) res = fminf(res, __shfl_xor(res, o, 64));
            if ((double)res < (double)ldTh) { ldTh = res; im = j; found = true; continue; }
        }
        if (angle < lverTh && angle > -lverTh) { lverTh = fabsf(angle); iv = j; continue; }
        if (angle > lparTh || angle < -lparTh) { lparTh = fabsf(angle); ip = j; }
    }
    if (lane == 0) {
        if (im >= 0) match[po] = im;
        if (iv >= 0) ver[po] = iv;
        if (ip >= 0) par[po] = ip;
        if (found) atomicAdd(&nmatches[b], 1);
    }
}

// ---- Frame::isInFrustum for points (src/Frame.cc:312-367) and lines (:369-438): one thread per map point / map line ------------
struct FrustumPose { float Rcw[9], tcw[3], Ow[3]; };
__device__ inline FrustumPose frustum_pose(const float* T) {
    FrustumPose p;
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) p.Rcw[3 * r + c] = T[4 * r + c]; p.tcw[r] = T[4 * r + 3]; }
    for (int i = 0; i < 3; i++) {   // Frame::UpdatePoseMatrices: mOw = -mRcw.t()*mtcw (general gemm path, double accumulation)
        double s = 0;
        for (int k = 0; k < 3; k++) s += (double)p.Rcw[3 * k + i] * (double)p.tcw[k];
        p.Ow[i] = (float)(s * -1.0);
    }
    return p;
}
__device__ inline float norm3(const float* v) { return (float)sqrt((double)v[0] * v[0] + (double)v[1] * v[1] + (double)v[2] * v[2]); }
__device__ inline double dot3(const float* a, const float* b) { return (double)a[0] * b[0] + (double)a[1] * b[1] + (double)a[2] * b[2]; }

__global__ __launch_bounds__(256) void frustum_points_kernel(planar_frame_view F, float lsf, int n_levels, const int32_t* __restrict__ n, int stride,
                                                             const uint8_t* __restrict__ valid, const float* __restrict__ xw,
                                                             const float* __restrict__ normal, const float* __restrict__ min_dist,
                                                             const float* __restrict__ max_dist, float limit, uint8_t* __restrict__ in_view,
                                                             float* __restrict__ proj_x, float* __restrict__ proj_y, float* __restrict__ proj_xr,
                                                             int32_t* __restrict__ level, float* __restrict__ view_cos) {
    const int b = blockIdx.y, j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n[b]) return;
    const size_t o = (size_t)b * stride + j;
    in_view[o] = 0;
    if (!valid[o]) return;
    const FrustumPose P = frustum_pose(F.Tcw + (size_t)b * 16);
    const float X[3] = {xw[3 * o], xw[3 * o + 1], xw[3 * o + 2]};
    const float PcX = gemm3_row(P.Rcw[0], P.Rcw[1], P.Rcw[2], X, P.tcw[0]), PcY = gemm3_row(P.Rcw[3], P.Rcw[4], P.Rcw[5], X, P.tcw[1]);
    const float PcZ = gemm3_row(P.Rcw[6], P.Rcw[7], P.Rcw[8], X, P.tcw[2]);
    if (PcZ < 0.0f) return;
    const float invz = 1.0f / PcZ;
    const float u = F.fx * PcX * invz + F.cx, v = F.fy * PcY * invz + F.cy;
    if (u < F.min_x || u > F.max_x) return;
    if (v < F.min_y || v > F.max_y) return;
    const float maxDistance = 1.2f * max_dist[o], minDistance = 0.8f * min_dist[o];
    const float PO[3] = {X[0] - P.Ow[0], X[1] - P.Ow[1], X[2] - P.Ow[2]};
    const float dist = norm3(PO);
    if (dist < minDistance || dist > maxDistance) return;
    const float Pn[3] = {normal[3 * o], normal[3 * o + 1], normal[3 * o + 2]};
    const float viewCos = (float)(dot3(PO, Pn) / (double)dist);
    if (viewCos < limit) return;
    const float ratio = max_dist[o] / dist;   // MapPoint::PredictScale (src/MapPoint.cc:419-434)
    int nScale = (int)ceilf((float)log((double)ratio) / lsf);
    if (nScale < 0) nScale = 0; else if (nScale >= n_levels) nScale = n_levels - 1;
    in_view[o] = 1; proj_x[o] = u; proj_xr[o] = u - F.bf * invz; proj_y[o] = v; level[o] = nScale; view_cos[o] = viewCos;
}

__global__ __launch_bounds__(256) void frustum_lines_kernel(planar_frame_view F, float lsf, const int32_t* __restrict__ n, int stride,
                                                            const uint8_t* __restrict__ valid, const double* __restrict__ xw6,
                                                            const double* __restrict__ normal, const float* __restrict__ min_dist,
                                                            const float* __restrict__ max_dist, float limit, uint8_t* __restrict__ in_view,
                                                            float* __restrict__ proj, int32_t* __restrict__ level, float* __restrict__ view_cos) {
    const int b = blockIdx.y, j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n[b]) return;
    const size_t o = (size_t)b * stride + j;
    in_view[o] = 0;
    if (!valid[o]) return;
    const FrustumPose P = frustum_pose(F.Tcw + (size_t)b * 16);
    float SP[3], EP[3];
    for (int k = 0; k < 3; k++) { SP[k] = (float)xw6[6 * o + k]; EP[k] = (float)xw6[6 * o + 3 + k]; }
    const float SPcX = gemm3_row(P.Rcw[0], P.Rcw[1], P.Rcw[2], SP, P.tcw[0]), SPcY = gemm3_row(P.Rcw[3], P.Rcw[4], P.Rcw[5], SP, P.tcw[1]);
    const float SPcZ = gemm3_row(P.Rcw[6], P.Rcw[7], P.Rcw[8], SP, P.tcw[2]);
    const float EPcX = gemm3_row(P.Rcw[0], P.Rcw[1], P.Rcw[2], EP, P.tcw[0]), EPcY = gemm3_row(P.Rcw[3], P.Rcw[4], P.Rcw[5], EP, P.tcw[1]);
    const float EPcZ = gemm3_row(P.Rcw[6], P.Rcw[7], P.Rcw[8], EP, P.tcw[2]);
    if (SPcZ < 0.0f || EPcZ < 0.0f) return;
    const float invz1 = 1.0f / SPcZ;
    const float u1 = F.fx * SPcX * invz1 + F.cx, v1 = F.fy * SPcY * invz1 + F.cy;
    if (u1 < F.min_x || u1 > F.max_x) return;
    if (v1 < F.min_y || v1 > F.max_y) return;
    const float invz2 = 1.0f / EPcZ;
    const float u2 = F.fx * EPcX * invz2 + F.cx, v2 = F.fy * EPcY * invz2 + F.cy;
    if (u2 < F.min_x || u2 > F.max_x) return;
    if (v2 < F.min_y || v2 > F.max_y) return;
    const float maxDistance = 1.2f * max_dist[o], minDistance = 0.8f * min_dist[o];
    float OM[3];
    for (int k = 0; k < 3; k++) OM[k] = (float)((double)(SP[k] + EP[k]) * 0.5) - P.Ow[k];
    const float dist = norm3(OM);
    if (dist < minDistance || dist > maxDistance) return;
    const float pn[3] = {(float)normal[3 * o], (float)normal[3 * o + 1], (float)normal[3 * o + 2]};
    const float viewCos = (float)(dot3(OM, pn) / (double)dist);
    if (viewCos < limit) return;
    const float ratio = max_dist[o] / dist;   // MapLine::PredictScale (src/MapLine.cpp:381-390): not clamped
    in_view[o] = 1;
    proj[4 * o] = u1; proj[4 * o + 1] = v1; proj[4 * o + 2] = u2; proj[4 * o + 3] = v2;
    level[o] = (int)ceilf((float)log((double)ratio) / lsf);
    view_cos[o] = viewCos;
}

// ---- ORBmatcher::Fuse(KeyFrame*, vpMapPoints, th), search half (src/ORBmatcher.cc:829-951): one workgroup per key frame, one thread per map point.
// No probe order to keep: a point's gates read only its own state on entry (the map edits of :953-974 are the caller's).
struct FuseLds {
    uint32_t cand[NCELL];          // build_grid's counters / cursors
    uint16_t cell_start[NCELL + 1];
    uint16_t items[MAXN];
    int wsum[NT / 64];
    int n_fused;
};
struct FuseArgs {
    planar_frame_view f;
    float inv_sigma2[PLANAR_MAX_LEVELS];
    float lsf, th;
    int n_levels, stride, shared;
    const int32_t* n;
    const uint8_t *usable, *desc;
    const float *xw, *normal, *min_dist, *max_dist;
    int32_t *fuse_idx, *fuse_dist, *n_fused;
};
__device__ inline float fuse_norm3(const float* v) { return (float)sqrt((double)v[0] * v[0] + (double)v[1] * v[1] + (double)v[2] * v[2]); }   // cv::norm: double accumulation

__global__ __launch_bounds__(NT) void fuse_kernel(FuseArgs a) {
    __shared__ FuseLds s;
    const int b = blockIdx.x, tid = threadIdx.x;
    const planar_frame_view& f = a.f;
    const int N = f.n[b];
    const planar_keypoint* keys = f.keys_un + (size_t)b * f.stride;
    const float* uR = f.u_right + (size_t)b * f.stride;
    const uint8_t* kdesc = f.desc + (size_t)b * f.stride * 32;
    if (tid == 0) s.n_fused = 0;
    build_grid(s, f, keys, N);                                   // KeyFrame::mGrid is the frame's (src/KeyFrame.cc:56-63)
    // GetRotation / GetTranslation / GetCameraCenter (src/KeyFrame.cc:79-93, 107-130): Ow = -Rwc * tcw, general gemm path (double accumulation)
    const float* T = f.Tcw + (size_t)b * 16;
    float Rcw[9], tcw[3], Ow[3];
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) Rcw[3 * r + c] = T[4 * r + c]; tcw[r] = T[4 * r + 3]; }
    for (int i = 0; i < 3; i++) {
        double sum = 0;
        for (int k = 0; k < 3; k++) sum += (double)Rcw[3 * k + i] * (double)tcw[k];
        Ow[i] = (float)(sum * -1.0);
    }
    const size_t po = a.shared ? 0 : (size_t)b * a.stride;
    const size_t oo = (size_t)b * a.stride;
    const int NP = a.n[a.shared ? 0 : b];
    int fused = 0;
    for (int j = tid; j < a.stride; j += NT) {
        int bestDist = 256, bestIdx = -1;
        if (j < NP && a.usable[po + j]) {                                              // rows beyond n[b] read -1 / 256
            const float* X = a.xw + (po + j) * 3;
            const float xc = gemm3_row(Rcw[0], Rcw[1], Rcw[2], X, tcw[0]), yc = gemm3_row(Rcw[3], Rcw[4], Rcw[5], X, tcw[1]);
            const float zc = gemm3_row(Rcw[6], Rcw[7], Rcw[8], X, tcw[2]);
            if (!(zc < 0.0f)) {                                                        // :858
                const float invz = 1.0f / zc;
                const float x = xc * invz, y = yc * invz;
                const float u = f.fx * x + f.cx, v = f.fy * y + f.cy;
                if (u >= f.min_x && u < f.max_x && v >= f.min_y && v < f.max_y) {       // KeyFrame::IsInImage
                    const float ur = u - f.bf * invz;
                    const float maxDistance = 1.2f * a.max_dist[po + j], minDistance = 0.8f * a.min_dist[po + j];
                    const float PO[3] = {X[0] - Ow[0], X[1] - Ow[1], X[2] - Ow[2]};
                    const float dist3D = fuse_norm3(PO);
                    const float* Pn = a.normal + (po + j) * 3;
                    const double dotp = (double)PO[0] * Pn[0] + (double)PO[1] * Pn[1] + (double)PO[2] * Pn[2];
                    if (!(dist3D < minDistance || dist3D > maxDistance) && !(dotp < 0.5 * (double)dist3D)) {     // :878, :884
                        const float ratio = a.max_dist[po + j] / dist3D;                // MapPoint::PredictScale (src/MapPoint.cc:402-417)
                        int lvl = (int)ceilf((float)log((double)ratio) / a.lsf);
                        if (lvl < 0) lvl = 0; else if (lvl >= a.n_levels) lvl = a.n_levels - 1;
                        const float radius = a.th * f.scale_factors[lvl];
                        uint32_t d[8];
                        load_desc(d, a.desc + (po + j) * 32);
                        // KeyFrame::GetFeaturesInArea (src/KeyFrame.cc:639-678): the frame's window walk without level bounds
                        const int nMinCellX = max(0, (int)floorf((u - f.min_x - radius) * f.grid_w_inv));
                        const int nMaxCellX = min(PLANAR_GRID_COLS - 1, (int)ceilf((u - f.min_x + radius) * f.grid_w_inv));
                        const int nMinCellY = max(0, (int)floorf((v - f.min_y - radius) * f.grid_h_inv));
                        const int nMaxCellY = min(PLANAR_GRID_ROWS - 1, (int)ceilf((v - f.min_y + radius) * f.grid_h_inv));
                        if (nMinCellX < PLANAR_GRID_COLS && nMaxCellX >= 0 && nMinCellY < PLANAR_GRID_ROWS && nMaxCellY >= 0 && nMinCellY <= nMaxCellY)
                            for (int ix = nMinCellX; ix <= nMaxCellX; ix++) {
                                const int c0 = s.cell_start[ix * PLANAR_GRID_ROWS + nMinCellY], c1 = s.cell_start[ix * PLANAR_GRID_ROWS + nMaxCellY + 1];
                                for (int k = c0; k < c1; k++) {
                                    const int idx = s.items[k];
                                    const planar_keypoint kp = keys[idx];
                                    if (!(fabsf(kp.x - u) < radius && fabsf(kp.y - v) < radius)) continue;
                                    const int kl = kp.octave;
                                    if (kl < lvl - 1 || kl > lvl) continue;                                     // :912
                                    const float kr = uR[idx];
                                    const float ex = u - kp.x, ey = v - kp.y;
                                    if (kr >= 0) {
                                        const float er = ur - kr;
                                        const float e2 = ex * ex + ey * ey + er * er;
                                        if ((double)(e2 * a.inv_sigma2[kl]) > 7.8) continue;                    // :927
                                    } else {
                                        const float e2 = ex * ex + ey * ey;
                                        if ((double)(e2 * a.inv_sigma2[kl]) > 5.99) continue;                   // :938
                                    }
                                    const int dist = hamming256(d, kdesc + (size_t)idx * 32);
                                    if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
                                }
                            }
                    }
                }
            }
        }
        const bool hit = bestDist <= TH_LOW;                                            // :953
        a.fuse_idx[oo + j] = hit ? bestIdx : -1;
        if (a.fuse_dist) a.fuse_dist[oo + j] = bestDist;
        fused += hit ? 1 : 0;
    }
    if (fused) atomicAdd(&s.n_fused, fused);
    __syncthreads();
    if (tid == 0) a.n_fused[b] = s.n_fused;
}

// ---- LSDmatcher::Fuse(KeyFrame*, vpMapLines, th), search half (src/LSDmatcher.cpp:884-991): one wavefront per key frame, one lane per map line; every
//      lane scans the key frame's key lines in index order (KeyFrame::GetLinesInArea is a linear scan, src/KeyFrame.cc:680-712).
struct LineFuseArgs {
    planar_frame_view f;
    float lsf, th;
    int n_levels, line_stride, ml_stride, shared;
    const int32_t *n_lines, *n_ml;
    const planar_keyline* keylines;
    const uint8_t *ldesc, *usable, *ml_desc;
    const double *xw6, *normal;
    const float *min_dist, *max_dist;
    int32_t *fuse_idx, *fuse_dist, *n_fused;
};

__global__ __launch_bounds__(64) void lsd_fuse_kernel(LineFuseArgs a) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const planar_frame_view& f = a.f;
    const int NLn = a.n_lines[b];
    const size_t lo = (size_t)b * a.line_stride, mo = a.shared ? 0 : (size_t)b * a.ml_stride, oo = (size_t)b * a.ml_stride;
    const int NM = a.n_ml[a.shared ? 0 : b];
    const planar_keyline* kl = a.keylines + lo;
    const uint8_t* ldesc = a.ldesc + lo * 32;
    const float* T = f.Tcw + (size_t)b * 16;
    float Rcw[9], tcw[3], Ow[3];
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) Rcw[3 * r + c] = T[4 * r + c]; tcw[r] = T[4 * r + 3]; }
    for (int i = 0; i < 3; i++) {
        double sum = 0;
        for (int k = 0; k < 3; k++) sum += (double)Rcw[3 * k + i] * (double)tcw[k];
        Ow[i] = (float)(sum * -1.0);
    }
    int fused = 0;
    for (int j = lane; j < a.ml_stride; j += 64) {
        int bestDist = 0x7fffffff, bestIdx = -1;
        bool go = j < NM && a.usable[mo + j] != 0;
        float u1 = 0, v1 = 0, u2 = 0, v2 = 0, radius = 0;
        int lvl = 0;
        if (go) {
            float SP[3], EP[3];
            for (int k = 0; k < 3; k++) { SP[k] = (float)a.xw6[6 * (mo + j) + k]; EP[k] = (float)a.xw6[6 * (mo + j) + 3 + k]; }
            const float SPcX = gemm3_row(Rcw[0], Rcw[1], Rcw[2], SP, tcw[0]), SPcY = gemm3_row(Rcw[3], Rcw[4], Rcw[5], SP, tcw[1]);
            const float SPcZ = gemm3_row(Rcw[6], Rcw[7], Rcw[8], SP, tcw[2]);
            const float EPcX = gemm3_row(Rcw[0], Rcw[1], Rcw[2], EP, tcw[0]), EPcY = gemm3_row(Rcw[3], Rcw[4], Rcw[5], EP, tcw[1]);
            const float EPcZ = gemm3_row(Rcw[6], Rcw[7], Rcw[8], EP, tcw[2]);
            go = !(SPcZ < 0.0f || EPcZ < 0.0f);
            const float invz1 = 1.0f / SPcZ, invz2 = 1.0f / EPcZ;
            u1 = f.fx * SPcX * invz1 + f.cx; v1 = f.fy * SPcY * invz1 + f.cy;
            u2 = f.fx * EPcX * invz2 + f.cx; v2 = f.fy * EPcY * invz2 + f.cy;
            if (u1 < f.min_x || u1 > f.max_x || v1 < f.min_y || v1 > f.max_y) go = false;
            if (u2 < f.min_x || u2 > f.max_x || v2 < f.min_y || v2 > f.max_y) go = false;
            const float maxDistance = 1.2f * a.max_dist[mo + j], minDistance = 0.8f * a.min_dist[mo + j];
            float OM[3];
            for (int k = 0; k < 3; k++) OM[k] = (float)((double)(SP[k] + EP[k]) * 0.5) - Ow[k];
            const float dist = fuse_norm3(OM);
            if (dist < minDistance || dist > maxDistance) go = false;
            const float pn[3] = {(float)a.normal[3 * (mo + j)], (float)a.normal[3 * (mo + j) + 1], (float)a.normal[3 * (mo + j) + 2]};
            const double dotp = (double)OM[0] * pn[0] + (double)OM[1] * pn[1] + (double)OM[2] * pn[2];
            if (dotp < 0.5 * (double)dist) go = false;
            const float ratio = a.max_dist[mo + j] / dist;                   // MapLine::PredictScale: not clamped
            lvl = (int)ceilf((float)log((double)ratio) / a.lsf);
            if (lvl < 0 || lvl >= a.n_levels) go = false;                    // mvScaleFactors[lvl] out of bounds in the reference (UB): skipped
            if (go) radius = a.th * f.scale_factors[lvl];
        }
        if (go) {
            uint32_t d[8];
            load_desc(d, a.ml_desc + (mo + j) * 32);
            for (int i = 0; i < NLn; i++) {
                const planar_keyline k = kl[i];
                const double mx = 0.5 * (double)(u1 + u2) - (double)k.pt_x, my = 0.5 * (double)(v1 + v2) - (double)k.pt_y;
                const float distance = (float)(mx * mx + my * my);
                if (distance > radius * radius) continue;
                const float slope = (v1 - v2) / (u1 - u2) - k.angle;
                if ((double)slope > (double)radius * 0.01) continue;
                if (k.octave < lvl - 1 || k.octave > lvl) continue;          // :968
                const int dist = hamming256(d, ldesc + (size_t)i * 32);
                if (dist < bestDist) { bestDist = dist; bestIdx = i; }
            }
        }
        const bool hit = bestDist <= TH_LOW;
        a.fuse_idx[oo + j] = hit ? bestIdx : -1;
        if (a.fuse_dist) a.fuse_dist[oo + j] = bestDist;
        fused += hit ? 1 : 0;
    }
    for (int o = 32; o > 0; o >>= 1) fused += __shfl_xor(fused, o, 64);
    if (lane == 0) a.n_fused[b] = fused;
}

static int check_view(const planar_frame_view* f) {
    PLANAR_REQUIRE(f->B >= 1 && f->stride >= 1 && f->stride <= MAXN, PLANAR_EINVAL, "frame view: B >= 1 and 1 <= stride <= PLANAR_MAX_FRAME_KEYS required");
    PLANAR_REQUIRE(f->n && f->keys_un && f->u_right && f->desc, PLANAR_EINVAL, "frame view: null array");
    return PLANAR_OK;
}

template <int MODE>
static int launch_projection(planar_ctx* ctx, const Args& a) {
    static bool attr_set = false;
    if (!attr_set) {
        PLANAR_HIP_CHECK(hipFuncSetAttribute((const void*)projection_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(Lds)));
        attr_set = true;
    }
    hipLaunchKernelGGL(projection_kernel<MODE>, dim3(a.f.B), dim3(NT), sizeof(Lds), ctx->stream, a);
    PLANAR_HIP_CHECK(hipGetLastError());
    return PLANAR_OK;
}

}  // namespace guided
}  // namespace planar

using namespace planar;
using guided::Args;

extern "C" {

int planar_search_by_projection_frame_dev(planar_ctx* ctx, const planar_frame_view* cur, const planar_last_frame_view* last, float th,
                                          int mono, int check_orientation, int32_t* d_cur_match, int32_t* d_nmatches) {
    PLANAR_REQUIRE(ctx && cur && last && d_cur_match && d_nmatches, PLANAR_EINVAL, "null argument");
    int rc = guided::check_view(cur);
    if (rc) return rc;
    PLANAR_REQUIRE(cur->Tcw && last->n && last->Tcw && last->usable && last->xw && last->octave && last->angle && last->mp_desc && last->mp_observed,
                   PLANAR_EINVAL, "null array in view");
    PLANAR_REQUIRE(last->stride >= 1 && last->stride <= guided::MAXN, PLANAR_EINVAL, "last-frame stride out of range");
    Args a{};
    a.f = *cur; a.last = *last; a.th = th; a.mono = mono; a.check_orientation = check_orientation; a.nn_ratio = 0;
    a.match = d_cur_match; a.nmatches = d_nmatches;
    return guided::launch_projection<guided::MODE_FRAME>(ctx, a);
}

int planar_search_by_projection_map_dev(planar_ctx* ctx, const planar_frame_view* frame, const planar_map_probes* probes, float th,
                                        float nn_ratio, int32_t* d_match, int32_t* d_nmatches) {
    PLANAR_REQUIRE(ctx && frame && probes && d_match && d_nmatches, PLANAR_EINVAL, "null argument");
    int rc = guided::check_view(frame);
    if (rc) return rc;
    PLANAR_REQUIRE(probes->n && probes->in_view && probes->proj_x && probes->proj_y && probes->proj_xr && probes->level && probes->view_cos &&
                       probes->desc && probes->observed, PLANAR_EINVAL, "null array in probes");
    PLANAR_REQUIRE(probes->stride >= 1, PLANAR_EINVAL, "probe stride out of range");
    Args a{};
    a.f = *frame; a.mp = *probes; a.th = th; a.nn_ratio = nn_ratio; a.match = d_match; a.nmatches = d_nmatches;
    return guided::launch_projection<guided::MODE_MAP>(ctx, a);
}

static int stage_view(Stager& s, const planar_frame_view* f, bool with_pose, int* ix) {
    const size_t n = (size_t)f->B * f->stride;
    ix[0] = s.in(f->n, (size_t)f->B * 4);
    ix[1] = s.in(f->keys_un, n * sizeof(planar_keypoint));
    ix[2] = s.in(f->u_right, n * 4);
    ix[3] = s.in(f->desc, n * 32);
    ix[4] = f->blocked ? s.in(f->blocked, n) : -1;
    ix[5] = with_pose ? s.in(f->Tcw, (size_t)f->B * 64) : -1;
    return 0;
}
static void patch_view(const Stager& s, planar_frame_view* d, const int* ix) {
    d->n = s.dev<int32_t>(ix[0]); d->keys_un = s.dev<planar_keypoint>(ix[1]); d->u_right = s.dev<float>(ix[2]); d->desc = s.dev<uint8_t>(ix[3]);
    d->blocked = ix[4] >= 0 ? s.dev<uint8_t>(ix[4]) : nullptr;
    d->Tcw = ix[5] >= 0 ? s.dev<float>(ix[5]) : nullptr;
}

int planar_search_by_projection_frame(planar_ctx* ctx, const planar_frame_view* cur, const planar_last_frame_view* last, float th, int mono,
                                      int check_orientation, int32_t* cur_match, int32_t* nmatches) {
    PLANAR_REQUIRE(ctx && cur && last && cur_match && nmatches, PLANAR_EINVAL, "null argument");
    int rc = guided::check_view(cur);
    if (rc) return rc;
    PLANAR_REQUIRE(cur->Tcw && last->n && last->Tcw && last->usable && last->xw && last->octave && last->angle && last->mp_desc && last->mp_observed,
                   PLANAR_EINVAL, "null array in view");
    PLANAR_REQUIRE(last->stride >= 1 && last->stride <= guided::MAXN, PLANAR_EINVAL, "last-frame stride out of range");
    PLANAR_HIP_CHECK(hipSetDevice(ctx->device));
    Stager s;
    int ix[6];
    stage_view(s, cur, true, ix);
    const int B = cur->B;
    const size_t nl = (size_t)B * last->stride;
    const int l0 = s.in(last->n, (size_t)B * 4), l1 = s.in(last->Tcw, (size_t)B * 64), l2 = s.in(last->usable, nl), l3 = s.in(last->xw, nl * 12),
              l4 = s.in(last->octave, nl * 4), l5 = s.in(last->angle, nl * 4), l6 = s.in(last->mp_desc, nl * 32), l7 = s.in(last->mp_observed, nl);
    const int om = s.inout(cur_match, (size_t)B * cur->stride * 4), on = s.out(nmatches, (size_t)B * 4);
    rc = s.upload(ctx->stream);
    if (rc) return rc;
    planar_frame_view dc = *cur;
    patch_view(s, &dc, ix);
    planar_last_frame_view dl = *last;
    dl.n = s.dev<int32_t>(l0); dl.Tcw = s.dev<float>(l1); dl.usable = s.dev<uint8_t>(l2); dl.xw = s.dev<float>(l3); dl.octave = s.dev<int32_t>(l4);
    dl.angle = s.dev<float>(l5); dl.mp_desc = s.dev<uint8_t>(l6); dl.mp_observed = s.dev<uint8_t>(l7);
    rc = planar_search_by_projection_frame_dev(ctx, &dc, &dl, th, mono, check_orientation, s.dev<int32_t>(om), s.dev<int32_t>(on));
    if (rc) return rc;
    return s.download(ctx->stream);
}

int planar_search_by_projection_map(planar_ctx* ctx, const planar_frame_view* frame, const planar_map_probes* probes, float th, float nn_ratio,
                                    int32_t* match, int32_t* nmatches) {
    PLANAR_REQUIRE(ctx && frame && probes && match && nmatches, PLANAR_EINVAL, "null argument");
    int rc = guided::check_view(frame);
    if (rc) return rc;
    PLANAR_REQUIRE(probes->n && probes->in_view && probes->proj_x && probes->proj_y && probes->proj_xr && probes->level && probes->view_cos &&
                       probes->desc && probes->observed, PLANAR_EINVAL, "null array in probes");
    PLANAR_REQUIRE(probes->stride >= 1, PLANAR_EINVAL, "probe stride out of range");
    PLANAR_HIP_CHECK(hipSetDevice(ctx->device));
    Stager s;
    int ix[6];
    stage_view(s, frame, false, ix);
    const int B = frame->B;
    const size_t np = (size_t)B * probes->stride;
    const int p0 = s.in(probes->n, (size_t)B * 4), p1 = s.in(probes->in_view, np), p2 = s.in(probes->proj_x, np * 4), p3 = s.in(probes->proj_y, np * 4),
              p4 = s.in(probes->proj_xr, np * 4), p5 = s.in(probes->level, np * 4), p6 = s.in(probes->view_cos, np * 4), p7 = s.in(probes->desc, np * 32),
              p8 = s.in(probes->observed, np);
    const int om = s.inout(match, (size_t)B * frame->stride * 4), on = s.out(nmatches, (size_t)B * 4);
    rc = s.upload(ctx->stream);
    if (rc) return rc;
    planar_frame_view df = *frame;
    patch_view(s, &df, ix);
    planar_map_probes dp = *probes;
    dp.n = s.dev<int32_t>(p0); dp.in_view = s.dev<uint8_t>(p1); dp.proj_x = s.dev<float>(p2); dp.proj_y = s.dev<float>(p3); dp.proj_xr = s.dev<float>(p4);
    dp.level = s.dev<int32_t>(p5); dp.view_cos = s.dev<float>(p6); dp.desc = s.dev<uint8_t>(p7); dp.observed = s.dev<uint8_t>(p8);
    rc = planar_search_by_projection_map_dev(ctx, &df, &dp, th, nn_ratio, s.dev<int32_t>(om), s.dev<int32_t>(on));
    if (rc) return rc;
    return s.download(ctx->stream);
}

int planar_fuse_search_dev(planar_ctx* ctx, const planar_frame_view* kf, const float* inv_level_sigma2, float log_scale_factor, int n_levels, const int32_t* d_n,
                           int stride, int points_shared, const uint8_t* d_usable, const float* d_xw, const float* d_normal, const float* d_min_dist,
                           const float* d_max_dist, const uint8_t* d_desc, float th, int32_t* d_fuse_idx, int32_t* d_fuse_dist, int32_t* d_n_fused) {
    PLANAR_REQUIRE(ctx && kf && inv_level_sigma2 && d_n && d_usable && d_xw && d_normal && d_min_dist && d_max_dist && d_desc && d_fuse_idx && d_n_fused,
                   PLANAR_EINVAL, "null argument");
    int rc = guided::check_view(kf);
    if (rc) return rc;
    PLANAR_REQUIRE(kf->Tcw != nullptr, PLANAR_EINVAL, "key-frame view: Tcw required");
    PLANAR_REQUIRE(stride >= 1 && n_levels >= 1 && n_levels <= PLANAR_MAX_LEVELS, PLANAR_EINVAL, "stride >= 1 and 1 <= n_levels <= PLANAR_MAX_LEVELS required");
    PLANAR_HIP_CHECK(hipSetDevice(ctx->device));
    guided::FuseArgs a{};
    a.f = *kf;
    for (int l = 0; l < n_levels; l++) a.inv_sigma2[l] = inv_level_sigma2[l];
    a.lsf = log_scale_factor; a.th = th; a.n_levels = n_levels; a.stride = stride; a.shared = points_shared ? 1 : 0;
    a.n = d_n; a.usable = d_usable; a.desc = d_desc; a.xw = d_xw; a.normal = d_normal; a.min_dist = d_min_dist; a.max_dist = d_max_dist;
    a.fuse_idx = d_fuse_idx; a.fuse_dist = d_fuse_dist; a.n_fused = d_n_fused;
    hipLaunchKernelGGL(guided::fuse_kernel, dim3(kf->B), dim3(guided::NT), 0, ctx->stream, a);
    PLANAR_HIP_CHECK(hipGetLastError());
    return PLANAR_OK;
}

int planar_fuse_search(planar_ctx* ctx, const planar_frame_view* kf, const float* inv_level_sigma2, float log_scale_factor, int n_levels, const int32_t* n,
                       int stride, int points_shared, const uint8_t* usable, const float* xw, const float* normal, const float* min_dist,
                       const float* max_dist, const uint8_t* desc, float th, int32_t* fuse_idx, int32_t* fuse_dist, int32_t* n_fused) {
    PLANAR_REQUIRE(ctx && kf && inv_level_sigma2 && n && usable && xw && normal && min_dist && max_dist && desc && fuse_idx && n_fused, PLANAR_EINVAL, "null argument");
    int rc = guided::check_view(kf);
    if (rc) return rc;
    PLANAR_REQUIRE(kf->Tcw != nullptr && stride >= 1, PLANAR_EINVAL, "key-frame view: Tcw required, stride >= 1");
    PLANAR_HIP_CHECK(hipSetDevice(ctx->device));
    Stager s;
    int ix[6];
    stage_view(s, kf, true, ix);
    const int B = kf->B;
    const size_t np = (size_t)(points_shared ? 1 : B) * stride, no = (size_t)B * stride;
    const int p0 = s.in(n, (size_t)(points_shared ? 1 : B) * 4), p1 = s.in(usable, np), p2 = s.in(xw, np * 12), p3 = s.in(normal, np * 12), p4 = s.in(min_dist, np * 4),
              p5 = s.in(max_dist, np * 4), p6 = s.in(desc, np * 32);
    const int o0 = s.out(fuse_idx, no * 4), o1 = fuse_dist ? s.out(fuse_dist, no * 4) : -1, o2 = s.out(n_fused, (size_t)B * 4);
    rc = s.upload(ctx->stream);
    if (rc) return rc;
    planar_frame_view d = *kf;
    patch_view(s, &d, ix);
    rc = planar_fuse_search_dev(ctx, &d, inv_level_sigma2, log_scale_factor, n_levels, s.dev<int32_t>(p0), stride, points_shared, s.dev<uint8_t>(p1), s.dev<float>(p2),
                                s.dev<float>(p3), s.dev<float>(p4), s.dev<float>(p5), s.dev<uint8_t>(p6), th, s.dev<int32_t>(o0),
                                o1 >= 0 ? s.dev<int32_t>(o1) : nullptr, s.dev<int32_t>(o2));
    if (rc) return rc;
    return s.download(ctx->stream);
}

int planar_lsd_fuse_search_dev(planar_ctx* ctx, const planar_frame_view* kf, float log_scale_factor, int n_levels, const int32_t* d_n_lines, int line_stride,
                               const planar_keyline* d_keylines, const uint8_t* d_ldesc, const int32_t* d_n_ml, int ml_stride, int lines_shared,
                               const uint8_t* d_usable, const double* d_xw6, const double* d_normal, const float* d_min_dist, const float* d_max_dist,
                               const uint8_t* d_ml_desc, float th, int32_t* d_fuse_idx, int32_t* d_fuse_dist, int32_t* d_n_fused) {
    PLANAR_REQUIRE(ctx && kf && kf->Tcw && d_n_lines && d_keylines && d_ldesc && d_n_ml && d_usable && d_xw6 && d_normal && d_min_dist && d_max_dist && d_ml_desc &&
                       d_fuse_idx && d_n_fused, PLANAR_EINVAL, "null argument");
    PLANAR_REQUIRE(kf->B >= 1 && line_stride >= 1 && ml_stride >= 1 && n_levels >= 1 && n_levels <= PLANAR_MAX_LEVELS, PLANAR_EINVAL,
                   "B, strides >= 1 and 1 <= n_levels <= PLANAR_MAX_LEVELS required");
    PLANAR_HIP_CHECK(hipSetDevice(ctx->device));
    guided::LineFuseArgs a{};
    a.f = *kf; a.lsf = log_scale_factor; a.th = th; a.n_levels = n_levels; a.line_stride = line_stride; a.ml_stride = ml_stride; a.shared = lines_shared ? 1 : 0;
    a.n_lines = d_n_lines; a.n_ml = d_n_ml; a.keylines = d_keylines; a.ldesc = d_ldesc; a.usable = d_usable; a.ml_desc = d_ml_desc; a.xw6 = d_xw6; a.normal = d_normal;
    a.min_dist = d_min_dist; a.max_dist = d_max_dist; a.fuse_idx = d_fuse_idx; a.fuse_dist = d_fuse_dist; a.n_fused = d_n_fused;
    hipLaunchKernelGGL(guided::lsd_fuse_kernel, dim3(kf->B), dim3(64), 0, ctx->stream, a);
    PLANAR_HIP_CHECK(hipGetLastError());
    return PLANAR_OK;
}

int planar_lsd_fuse_search(planar_ctx* ctx, const planar_frame_view* kf, float log_scale_factor, int n_levels, const int32_t* n_lines, int line_stride,
                           const planar_keyline* keylines, const uint8_t* ldesc, const int32_t* n_ml, int ml_stride, int lines_shared, const uint8_t* usable,
                           const double* xw6, const double* normal, const float* min_dist, const float* max_dist, const uint8_t* ml_desc, float th,
                           int32_t* fuse_idx, int32_t* fuse_dist, int32_t* n_fused) {
    PLANAR_REQUIRE(ctx && kf && kf->Tcw && n_lines && keylines && ldesc && n_ml && usable && xw6 && normal && min_dist && max_dist && ml_desc && fuse_idx && n_fused,
                   PLANAR_EINVAL, "null argument");
    PLANAR_REQUIRE(kf->B >= 1 && line_stride >= 1 && ml_stride >= 1, PLANAR_EINVAL, "B, strides >= 1 required");
    PLANAR_HIP_CHECK(hipSetDevice(ctx->device));
    Stager s;
    const int B = kf->B;
    const size_t nl = (size_t)B * line_stride, nm = (size_t)(lines_shared ? 1 : B) * ml_stride, no = (size_t)B * ml_stride;
    const int t = s.in(kf->Tcw, (size_t)B * 64), l0 = s.in(n_lines, (size_t)B * 4), l1 = s.in(keylines, nl * sizeof(planar_keyline)), l2 = s.in(ldesc, nl * 32);
    const int m0 = s.in(n_ml, (size_t)(lines_shared ? 1 : B) * 4), m1 = s.in(usable, nm), m2 = s.in(xw6, nm * 48), m3 = s.in(normal, nm * 24), m4 = s.in(min_dist, nm * 4),
              m5 = s.in(max_dist, nm * 4), m6 = s.in(ml_desc, nm * 32);
    const int o0 = s.out(fuse_idx, no * 4), o1 = fuse_dist ? s.out(fuse_dist, no * 4) : -1, o2 = s.out(n_fused, (size_t)B * 4);
    int rc = s.upload(ctx->stream);
    if (rc) return rc;
    planar_frame_view d = *kf;
    d.Tcw = s.dev<float>(t);
    rc = planar_lsd_fuse_search_dev(ctx, &d, log_scale_factor, n_levels, s.dev<int32_t>(l0), line_stride, s.dev<planar_keyline>(l1), s.dev<uint8_t>(l2), s.dev<int32_t>(m0),
                                    ml_stride, lines_shared, s.dev<uint8_t>(m1), s.dev<double>(m2), s.dev<double>(m3), s.dev<float>(m4), s.dev<float>(m5),
                                    s.dev<uint8_t>(m6), th, s.dev<int32_t>(o0), o1 >= 0 ? s.dev<int32_t>(o1) : nullptr, s.dev<int32_t>(o2));
    if (rc) return rc;
    return s.download(ctx->stream);
}

int planar_is_in_frustum_points_dev(planar_ctx* ctx, const planar_frame_view* f, float log_scale_factor, int n_levels, const int32_t* d_n, int stride,
                                    const uint8_t* d_valid, const float* d_xw, const float* d_normal, const float* d_min_dist, const float* d_max_dist,
                                    float viewing_cos_limit, uint8_t* d_in_view, float* d_proj_x, float* d_proj_y, float* d_proj_xr, int32_t* d_level,
                                    float* d_view_cos) {
    PLANAR_REQUIRE(ctx && f && f->Tcw && d_n && d_valid && d_xw && d_normal && d_min_dist && d_max_dist && d_in_view && d_proj_x && d_proj_y && d_proj_xr &&
                       d_level && d_view_cos, PLANAR_EINVAL, "null argument");
    PLANAR_REQUIRE(f->B >= 1 && stride >= 1 && n_levels >= 1 && n_levels <= PLANAR_MAX_LEVELS, PLANAR_EINVAL, "bad sizes");
    hipLaunchKernelGGL(guided::frustum_points_kernel, dim3((stride + 255) / 256, f->B), dim3(256), 0, ctx->stream, *f, log_scale_factor, n_levels, d_n, stride,
                       d_valid, d_xw, d_normal, d_min_dist, d_max_dist, viewing_cos_limit, d_in_view, d_proj_x, d_proj_y, d_proj_xr, d_level, d_view_cos);
    PLANAR_HIP_CHECK(hipGetLastError());
    return PLANAR_OK;
}

int planar_is_in_frustum_points(planar_ctx* ctx, const planar_frame_view* f, float log_scale_factor, int n_levels, const int32_t* n, int stride,
                                const uint8_t* valid, const float* xw, const float* normal, const float* min_dist, const float* max_dist,
                                float viewing_cos_limit, uint8_t* in_view, float* proj_x, float* proj_y, float* proj_xr, int32_t* level, float* view_cos) {
    PLANAR_REQUIRE(ctx && f && f->Tcw && n && valid && xw && normal && min_dist && max_dist && in_view && proj_x && proj_y && proj_xr && level && view_cos,
                   PLANAR_EINVAL, "null argument");
    PLANAR_REQUIRE(f->B >= 1 && stride >= 1, PLANAR_EINVAL, "bad sizes");
    PLANAR_HIP_CHECK(hipSetDevice(ctx->device));
    Stager s;
    const size_t N = (size_t)f->B * stride;
    const int t = s.in(f->Tcw, (size_t)f->B * 64), a0 = s.in(n, (size_t)f->B * 4), a1 = s.in(valid, N), a2 = s.in(xw, N * 12), a3 = s.in(normal, N * 12),
              a4 = s.in(min_dist, N * 4), a5 = s.in(max_dist, N * 4);
    const int o0 = s.inout(in_view, N), o1 = s.inout(proj_x, N * 4), o2 = s.inout(proj_y, N * 4), o3 = s.inout(proj_xr, N * 4), o4 = s.inout(level, N * 4),
              o5 = s.inout(view_cos, N * 4);
    int rc = s.upload(ctx->stream);
    if (rc) return rc;
    planar_frame_view d = *f;
    d.Tcw = s.dev<float>(t);
    rc = planar_is_in_frustum_points_dev(ctx, &d, log_scale_factor, n_levels, s.dev<int32_t>(a0), stride, s.dev<uint8_t>(a1), s.dev<float>(a2), s.dev<float>(a3),
                                         s.dev<float>(a4), s.dev<float>(a5), viewing_cos_limit, s.dev<uint8_t>(o0), s.dev<float>(o1), s.dev<float>(o2),
                                         s.dev<float>(o3), s.dev<int32_t>(o4), s.dev<float>(o5));
    if (rc) return rc;
    return s.download(ctx->stream);
}

int planar_is_in_frustum_lines_dev(planar_ctx* ctx, const planar_frame_view* f, float log_scale_factor, const int32_t* d_n, int stride, const uint8_t* d_valid,
                                   const double* d_xw6, const double* d_normal, const float* d_min_dist, const float* d_max_dist, float viewing_cos_limit,
                                   uint8_t* d_in_view, float* d_proj, int32_t* d_level, float* d_view_cos) {
    PLANAR_REQUIRE(ctx && f && f->Tcw && d_n && d_valid && d_xw6 && d_normal && d_min_dist && d_max_dist && d_in_view && d_proj && d_level && d_view_cos,
                   PLANAR_EINVAL, "null argument");
    PLANAR_REQUIRE(f->B >= 1 && stride >= 1, PLANAR_EINVAL, "bad sizes");
    hipLaunchKernelGGL(guided::frustum_lines_kernel, dim3((stride + 255) / 256, f->B), dim3(256), 0, ctx->stream, *f, log_scale_factor, d_n, stride, d_valid, d_xw6,
                       d_normal, d_min_dist, d_max_dist, viewing_cos_limit, d_in_view, d_proj, d_level, d_view_cos);
    PLANAR_HIP_CHECK(hipGetLastError());
    return PLANAR_OK;
}

int planar_is_in_frustum_lines(planar_ctx* ctx, const planar_frame_view* f, float log_scale_factor, const int32_t* n, int stride, const uint8_t* valid,
                               const double* xw6, const double* normal, const float* min_dist, const float* max_dist, float viewing_cos_limit, uint8_t* in_view,
                               float* proj, int32_t* level, float* view_cos) {
    PLANAR_REQUIRE(ctx && f && f->Tcw && n && valid && xw6 && normal && min_dist && max_dist && in_view && proj && level && view_cos, PLANAR_EINVAL,
                   "null argument");
    PLANAR_REQUIRE(f->B >= 1 && stride >= 1, PLANAR_EINVAL, "bad sizes");
    PLANAR_HIP_CHECK(hipSetDevice(ctx->device));
    Stager s;
    const size_t N = (size_t)f->B * stride;
    const int t = s.in(f->Tcw, (size_t)f->B * 64), a0 = s.in(n, (size_t)f->B * 4), a1 = s.in(valid, N), a2 = s.in(xw6, N * 48), a3 = s.in(normal, N * 24),
              a4 = s.in(min_dist, N * 4), a5 = s.in(max_dist, N * 4);
    const int o0 = s.inout(in_view, N), o1 = s.inout(proj, N * 16), o2 = s.inout(level, N * 4), o3 = s.inout(view_cos, N * 4);
    int rc = s.upload(ctx->stream);
    if (rc) return rc;
    planar_frame_view d = *f;
    d.Tcw = s.dev<float>(t);
    rc = planar_is_in_frustum_lines_dev(ctx, &d, log_scale_factor, s.dev<int32_t>(a0), stride, s.dev<uint8_t>(a1), s.dev<double>(a2), s.dev<double>(a3),
                                        s.dev<float>(a4), s.dev<float>(a5), viewing_cos_limit, s.dev<uint8_t>(o0), s.dev<float>(o1), s.dev<int32_t>(o2),
                                        s.dev<float>(o3));
    if (rc) return rc;
    return s.download(ctx->stream);
}

int planar_search_by_bow_dev(planar_ctx* ctx, int B, const int32_t* d_n_kf, int kf_stride, const int32_t* d_kf_node, const uint8_t* d_kf_usable,
                             const float* d_kf_angle, const uint8_t* d_kf_desc, const int32_t* d_n_f, int f_stride, const int32_t* d_f_node,
                             const float* d_f_angle, const uint8_t* d_f_desc, float nn_ratio, int check_orientation, int32_t* d_match,
                             int32_t* d_nmatches) {
    PLANAR_REQUIRE(ctx && d_n_kf && d_kf_node && d_kf_usable && d_kf_angle && d_kf_desc && d_n_f && d_f_node && d_f_angle && d_f_desc && d_match && d_nmatches,
                   PLANAR_EINVAL, "null argument");
    PLANAR_REQUIRE(B >= 1 && kf_stride >= 1 && kf_stride <= guided::MAXN && f_stride >= 1 && f_stride <= guided::MAXN, PLANAR_EINVAL,
                   "1 <= stride <= PLANAR_MAX_FRAME_KEYS required");
    static bool attr_set = false;
    if (!attr_set) {
        PLANAR_HIP_CHECK(hipFuncSetAttribute((const void*)guided::bow_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(guided::BowLds)));
        attr_set = true;
    }
    guided::BowArgs g{d_n_kf, d_kf_node, d_n_f, d_f_node, d_kf_usable, d_kf_desc, d_f_desc, d_kf_angle, d_f_angle, kf_stride, f_stride};
    Args a{};
    a.nn_ratio = nn_ratio; a.check_orientation = check_orientation; a.match = d_match; a.nmatches = d_nmatches;
    hipLaunchKernelGGL(guided::bow_kernel, dim3(B), dim3(guided::NT), sizeof(guided::BowLds), ctx->stream, g, a);
    PLANAR_HIP_CHECK(hipGetLastError());
    return PLANAR_OK;
}

int planar_search_by_bow(planar_ctx* ctx, int B, const int32_t* n_kf, int kf_stride, const int32_t* kf_node, const uint8_t* kf_usable,
                         const float* kf_angle, const uint8_t* kf_desc, const int32_t* n_f, int f_stride, const int32_t* f_node,
                         const float* f_angle, const uint8_t* f_desc, float nn_ratio, int check_orientation, int32_t* match, int32_t* nmatches) {
    PLANAR_REQUIRE(ctx && n_kf && kf_node && kf_usable && kf_angle && kf_desc && n_f && f_node && f_angle && f_desc && match && nmatches, PLANAR_EINVAL,
                   "null argument");
    PLANAR_REQUIRE(B >= 1 && kf_stride >= 1 && f_stride >= 1, PLANAR_EINVAL, "bad sizes");
    PLANAR_HIP_CHECK(hipSetDevice(ctx->device));
    Stager s;
    const size_t nk = (size_t)B * kf_stride, nf = (size_t)B * f_stride;
    const int a0 = s.in(n_kf, (size_t)B * 4), a1 = s.in(kf_node, nk * 4), a2 = s.in(kf_usable, nk), a3 = s.in(kf_angle, nk * 4), a4 = s.in(kf_desc, nk * 32),
              b0 = s.in(n_f, (size_t)B * 4), b1 = s.in(f_node, nf * 4), b2 = s.in(f_angle, nf * 4), b3 = s.in(f_desc, nf * 32);
    const int om = s.inout(match, nf * 4), on = s.out(nmatches, (size_t)B * 4);   // rows >= n_f[b] keep their value
    int rc = s.upload(ctx->stream);
    if (rc) return rc;
    rc = planar_search_by_bow_dev(ctx, B, s.dev<int32_t>(a0), kf_stride, s.dev<int32_t>(a1), s.dev<uint8_t>(a2), s.dev<float>(a3), s.dev<uint8_t>(a4),
                                  s.dev<int32_t>(b0), f_stride, s.dev<int32_t>(b1), s.dev<float>(b2), s.dev<uint8_t>(b3), nn_ratio, check_orientation,
                                  s.dev<int32_t>(om), s.dev<int32_t>(on));
    if (rc) return rc;
    return s.download(ctx->stream);
}

int planar_lsd_search_by_projection_dev(planar_ctx* ctx, int B, const int32_t* d_n_lines, int line_stride, const planar_keyline* d_keylines,
                                        const uint8_t* d_ldesc, const uint8_t* d_blocked, const int32_t* d_n_ml, int ml_stride,
                                        const uint8_t* d_ml_in_view, const float* d_ml_proj, const int32_t* d_ml_level, const float* d_ml_view_cos,
                                        const uint8_t* d_ml_desc, const uint8_t* d_ml_observed, const float* scale_factors, int n_levels, float th,
                                        float nn_ratio, int32_t* d_match, int32_t* d_nmatches) {
    PLANAR_REQUIRE(ctx && d_n_lines && d_keylines && d_ldesc && d_n_ml && d_ml_in_view && d_ml_proj && d_ml_level && d_ml_view_cos && d_ml_desc &&
                       d_ml_observed && scale_factors && d_match && d_nmatches, PLANAR_EINVAL, "null argument");
    PLANAR_REQUIRE(B >= 1 && line_stride >= 1 && line_stride <= guided::MAX_LINES && ml_stride >= 1, PLANAR_EINVAL, "bad sizes (line_stride <= 1024)");
    PLANAR_REQUIRE(n_levels >= 1 && n_levels <= PLANAR_MAX_LEVELS, PLANAR_EINVAL, "n_levels out of range");
    guided::LineArgs a{};
    a.n_lines = d_n_lines; a.n_ml = d_n_ml; a.ml_level = d_ml_level; a.keylines = d_keylines; a.ldesc = d_ldesc; a.blocked = d_blocked;
    a.ml_in_view = d_ml_in_view; a.ml_desc = d_ml_desc; a.ml_observed = d_ml_observed; a.ml_proj = d_ml_proj; a.ml_view_cos = d_ml_view_cos;
    a.line_stride = line_stride; a.ml_stride = ml_stride; a.th = th; a.nn_ratio = nn_ratio; a.match = d_match; a.nmatches = d_nmatches;
    for (int i = 0; i < n_levels; i++) a.scale_factors[i] = scale_factors[i];
    hipLaunchKernelGGL(guided::lsd_projection_kernel, dim3(B), dim3(64), 0, ctx->stream, a);
    PLANAR_HIP_CHECK(hipGetLastError());
    return PLANAR_OK;
}

int planar_lsd_search_by_projection(planar_ctx* ctx, int B, const int32_t* n_lines, int line_stride, const planar_keyline* keylines,
                                    const uint8_t* ldesc, const uint8_t* blocked, const int32_t* n_ml, int ml_stride, const uint8_t* ml_in_view,
                                    const float* ml_proj, const int32_t* ml_level, const float* ml_view_cos, const uint8_t* ml_desc,
                                    const uint8_t* ml_observed, const float* scale_factors, int n_levels, float th, float nn_ratio, int32_t* match,
                                    int32_t* nmatches) {
    PLANAR_REQUIRE(ctx && n_lines && keylines && ldesc && n_ml && ml_in_view && ml_proj && ml_level && ml_view_cos && ml_desc && ml_observed &&
                       scale_factors && match && nmatches, PLANAR_EINVAL, "null argument");
    PLANAR_REQUIRE(B >= 1 && line_stride >= 1 && ml_stride >= 1, PLANAR_EINVAL, "bad sizes");
    PLANAR_HIP_CHECK(hipSetDevice(ctx->device));
    Stager s;
    const size_t nl = (size_t)B * line_stride, nm = (size_t)B * ml_stride;
    const int a0 = s.in(n_lines, (size_t)B * 4), a1 = s.in(keylines, nl * sizeof(planar_keyline)), a2 = s.in(ldesc, nl * 32),
              a3 = blocked ? s.in(blocked, nl) : -1, b0 = s.in(n_ml, (size_t)B * 4), b1 = s.in(ml_in_view, nm), b2 = s.in(ml_proj, nm * 16),
              b3 = s.in(ml_level, nm * 4), b4 = s.in(ml_view_cos, nm * 4), b5 = s.in(ml_desc, nm * 32), b6 = s.in(ml_observed, nm);
    const int om = s.inout(match, nl * 4), on = s.out(nmatches, (size_t)B * 4);
    int rc = s.upload(ctx->stream);
    if (rc) return rc;
    rc = planar_lsd_search_by_projection_dev(ctx, B, s.dev<int32_t>(a0), line_stride, s.dev<planar_keyline>(a1), s.dev<uint8_t>(a2),
                                             a3 >= 0 ? s.dev<uint8_t>(a3) : nullptr, s.dev<int32_t>(b0), ml_stride, s.dev<uint8_t>(b1), s.dev<float>(b2),
                                             s.dev<int32_t>(b3), s.dev<float>(b4), s.dev<uint8_t>(b5), s.dev<uint8_t>(b6), scale_factors, n_levels, th,
                                             nn_ratio, s.dev<int32_t>(om), s.dev<int32_t>(on));
    if (rc) return rc;
    return s.download(ctx->stream);
}

int planar_plane_search_by_coefficients_dev(planar_ctx* ctx, int B, const int32_t* d_n_planes, int pl_stride, const float* d_pl_coef,
                                            const float* d_Tcw, int map_shared, const int32_t* d_n_mp, int mp_stride, const uint8_t* d_mp_valid,
                                            const float* d_mp_coef, const int32_t* d_mp_npts, int pts_stride, const float* d_mp_pts, const float* th,
                                            int32_t* d_match, int32_t* d_ver, int32_t* d_par, int32_t* d_nmatches) {
    PLANAR_REQUIRE(ctx && d_n_planes && d_pl_coef && d_Tcw && d_n_mp && d_mp_valid && d_mp_coef && d_mp_npts && d_mp_pts && th && d_match && d_ver &&
                       d_par && d_nmatches, PLANAR_EINVAL, "null argument");
    PLANAR_REQUIRE(B >= 1 && pl_stride >= 1 && mp_stride >= 1 && pts_stride >= 1, PLANAR_EINVAL, "bad sizes");
    PLANAR_HIP_CHECK(hipMemsetAsync(d_nmatches, 0, (size_t)B * 4, ctx->stream));
    hipLaunchKernelGGL(guided::plane_match_kernel, dim3(pl_stride, B), dim3(64), 0, ctx->stream, d_n_planes, pl_stride, d_pl_coef, d_Tcw, map_shared,
                       d_n_mp, mp_stride, d_mp_valid, d_mp_coef, d_mp_npts, pts_stride, d_mp_pts, th[0], th[1], th[2], th[3], d_match, d_ver, d_par,
                       d_nmatches);
    PLANAR_HIP_CHECK(hipGetLastError());
    return PLANAR_OK;
}

int planar_plane_search_by_coefficients(planar_ctx* ctx, int B, const int32_t* n_planes, int pl_stride, const float* pl_coef, const float* Tcw,
                                        int map_shared, const int32_t* n_mp, int mp_stride, const uint8_t* mp_valid, const float* mp_coef,
                                        const int32_t* mp_npts, int pts_stride, const float* mp_pts, const float* th, int32_t* match, int32_t* ver,
                                        int32_t* par, int32_t* nmatches) {
    PLANAR_REQUIRE(ctx && n_planes && pl_coef && Tcw && n_mp && mp_valid && mp_coef && mp_npts && mp_pts && th && match && ver && par && nmatches,
                   PLANAR_EINVAL, "null argument");
    PLANAR_REQUIRE(B >= 1 && pl_stride >= 1 && mp_stride >= 1 && pts_stride >= 1, PLANAR_EINVAL, "bad sizes");
    PLANAR_HIP_CHECK(hipSetDevice(ctx->device));
    Stager s;
    const int MB = map_shared ? 1 : B;
    const size_t np = (size_t)B * pl_stride, nm = (size_t)MB * mp_stride;
    const int a0 = s.in(n_planes, (size_t)B * 4), a1 = s.in(pl_coef, np * 16), a2 = s.in(Tcw, (size_t)B * 64), b0 = s.in(n_mp, (size_t)MB * 4),
              b1 = s.in(mp_valid, nm), b2 = s.in(mp_coef, nm * 16), b3 = s.in(mp_npts, nm * 4), b4 = s.in(mp_pts, nm * pts_stride * 12);
    const int o0 = s.inout(match, np * 4), o1 = s.inout(ver, np * 4), o2 = s.inout(par, np * 4), on = s.out(nmatches, (size_t)B * 4);
    int rc = s.upload(ctx->stream);
    if (rc) return rc;
    rc = planar_plane_search_by_coefficients_dev(ctx, B, s.dev<int32_t>(a0), pl_stride, s.dev<float>(a1), s.dev<float>(a2), map_shared, s.dev<int32_t>(b0),
                                                 mp_stride, s.dev<uint8_t>(b1), s.dev<float>(b2), s.dev<int32_t>(b3), pts_stride, s.dev<float>(b4), th,
                                                 s.dev<int32_t>(o0), s.dev<int32_t>(o1), s.dev<int32_t>(o2), s.dev<int32_t>(on));
    if (rc) return rc;
    return s.download(ctx->stream);
}

}  // extern "C"
