// planarslam_amd/csrc/orb.hip — batched ORB extractor for MI355X (gfx950, wave64).
//
// Replaces Planar_SLAM::ORBextractor::operator() (reference src/ORBextractor.cc:1043-1105)
// for a batch of B independent frames.  Bit-exact against the CPU oracle (oracle/orb_oracle.cpp),
// which is itself pinned against the real reference translation unit.
//
// Pipeline (all on one stream; every kernel covers the whole batch, blockIdx.y = frame):
//   K1 orb_resize      x(nlevels-1)  fixed-point bilinear chain, level l from level l-1      (:1107-1131)
//   K2 orb_fast_cells  one workgroup per FAST cell: LDS tile, threshold-free 9/16 score,
//                      window-local NMS, ini/min threshold fallback, ordered compaction        (:771-827)
//   K3 orb_sort        one workgroup per (frame, level): gather cells, quadtree path code per
//                      candidate, LSD radix sort by path code (wave ballots for stable ranks)
//   K4 orb_octree      one wave per (frame, level): the reference's node-list algorithm on
//                      contiguous ranges of the sorted array, then per-node arg-max           (:539-763)
//   K5 orb_blur        7x7 sigma-2 separable fixed-point Gaussian, LDS tiles                   (:1085-1086)
//   K6 orb_describe    16 lanes per keypoint: IC_Angle + steered 256-bit BRIEF                 (:77-147)
//
// Why a sort: DivideNode splits at midpoints that depend only on the node's bounds, so the
// quadtree cell of a key at every depth is a pure function of its (x,y): its "path code"
// (2 bits per depth).  After sorting by path code every node of the reference's list, at any
// depth, is a contiguous range, a split is three boundary searches, and no key ever moves.
// The sequential part (list order, largest-first expansion, N-stop) is then ~N tiny steps.
#include <algorithm>
#include <cmath>

#include "common.h"

namespace planar {
namespace orb {

constexpr int PATCH_SIZE = 31;
constexpr int HALF_PATCH = 15;
constexpr int EDGE_THRESHOLD = 19;
constexpr int MAX_LEVELS = 16;
constexpr int CELL_W = 30;
constexpr int RADIX_BITS = 4;

struct LevelDev {
    int w, h, pitch;          // level image (borderless)
    int64_t off;              // byte offset inside one frame's pyramid block
    int minBX, minBY, maxBX, maxBY;
    int nCols, nRows, wCell, hCell;
    int cell_begin, ncells;   // range in the cell table
    int cand_off, cand_cap;   // range in one frame's candidate-slot array (u32 each)
    int nfeat;                // mnFeaturesPerLevel[level]
    int depth;                // quadtree depth D (2 bits per depth in the path code)
    int nIni;                 // initial nodes (:543)
    float hX;                 // (:545)
    int code_bits;            // 2*D + bits(nIni-1), rounded up to RADIX_BITS
    int kept_off, kept_cap;   // range in one frame's kept-key array
    float scale;              // mvScaleFactor[level]
    float patch;              // (float)(int)(31*scale)
    int rs_off;               // offset of this level's resize tables (x then y), in short4 units
};

struct CellDev {
    short level;
    short x0, y0;             // FAST window origin (level coords): iniX+3, iniY+3
    short ww, wh;             // window size: (maxX-3)-(iniX+3), ...
    int slot_off, slot_cap;   // inside one frame's candidate-slot array
};

struct PlanDev {
    int nlevels, ini_th, min_th;
    int ncells_total;
    int64_t pyr_stride;       // bytes per frame in the pyramid / blurred buffers
    int cand_stride;          // u32 per frame in the candidate-slot array
    int kept_stride;          // u32 per frame in the kept array
    int kp_cap;               // keypoints per frame in the output arrays
    int umax[HALF_PATCH + 1];
    LevelDev lv[MAX_LEVELS];
};

__constant__ signed char c_pattern[1024] = {
#include "brief_pattern.inc"
};

// ------------------------------------------------------------------------------------------
// K1: cv::resize(INTER_LINEAR) 8UC1, one level.  Tables are built on the host exactly as
// OpenCV does (double/float coordinate math, saturate_cast<short> coefficients); the kernel
// is the integer part: HResizeLinear (x2048) + the 8-bit VResizeLinear rounding.
// tabx[dx] = {sx0, sx1, a0, a1}; taby[dy] = {sy0, sy1, b0, b1}.
// Each thread produces 4 horizontally adjacent pixels and stores one dword.
__global__ __launch_bounds__(256) void orb_resize(const PlanDev* __restrict__ plan, const short4* __restrict__ tabs,
                                                  uint8_t* __restrict__ pyr, int level) {
    const LevelDev& L = plan->lv[level];
    const LevelDev& S = plan->lv[level - 1];
    const int qw = L.pitch >> 2;
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= qw * L.h) return;
    const int dy = q / qw, dx0 = (q - dy * qw) << 2;
    uint8_t* frame = pyr + (int64_t)blockIdx.y * plan->pyr_stride;
    const uint8_t* src = frame + S.off;
    const short4* tx = tabs + L.rs_off;
    const short4 ty = tx[L.pitch + dy];
    const uint8_t* r0 = src + (int64_t)ty.x * S.pitch;
    const uint8_t* r1 = src + (int64_t)ty.y * S.pitch;
    uint32_t packed = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int dx = dx0 + i;
        uint32_t v = 0;
        if (dx < L.w) {
            const short4 t = tx[dx];
            const int h0 = r0[t.x] * t.z + r0[t.y] * t.w;
            const int h1 = r1[t.x] * t.z + r1[t.y] * t.w;
            v = (uint32_t)(((((int)ty.z * (h0 >> 4)) >> 16) + (((int)ty.w * (h1 >> 4)) >> 16) + 2) >> 2) & 0xffu;
        }
        packed |= v << (8 * i);
    }
    *(uint32_t*)(frame + L.off + (int64_t)dy * L.pitch + dx0) = packed;
}

// Level 0: copy the caller's frames into the pyramid block (pitch conversion).
__global__ __launch_bounds__(256) void orb_copy_level0(const PlanDev* __restrict__ plan, const uint8_t* __restrict__ gray,
                                                       int pitch, int64_t frame_stride, uint8_t* __restrict__ pyr) {
    const LevelDev& L = plan->lv[0];
    const int qw = L.pitch >> 2;
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= qw * L.h) return;
    const int y = q / qw, x0 = (q - y * qw) << 2;
    const uint8_t* s = gray + (int64_t)blockIdx.y * frame_stride + (int64_t)y * pitch;
    uint32_t packed = 0;
#pragma unroll
    for (int i = 0; i < 4; i++)
        if (x0 + i < L.w) packed |= (uint32_t)s[x0 + i] << (8 * i);
    *(uint32_t*)(pyr + (int64_t)blockIdx.y * plan->pyr_stride + L.off + (int64_t)y * L.pitch + x0) = packed;
}

// ------------------------------------------------------------------------------------------
// K2: FAST-9/16 per cell.
//
// cv::FAST(roi, t, nms=true) keeps pixel p iff p is a 9/16 corner at t and score(p) > score(q)
// for its 8 neighbours q, where score(q) counts as 0 if q is not a corner at t or lies outside
// the band [3, n-3) of the ROI.  score(p) = (largest t' for which p is still a corner) and does
// not depend on t, and "corner at t" <=> score >= t, so with S = the threshold-free score map
//   survivors(t) = { p in window : S(p) >= t  and  S(p) > S(q) for all neighbours q in window }.
// The reference calls FAST at iniTh and, if that returns nothing, again at minTh (:809-817):
//   cell output = survivors(minTh) filtered to S >= iniTh if any survivor has S >= iniTh.
// Windows of adjacent cells tile the level (cell stride wCell, ROI = wCell+6, band = 3), so
// NMS never looks across a cell seam: neighbours outside the window score 0.
__device__ __forceinline__ bool has9(uint32_t m16) {
    uint32_t m = m16 | (m16 << 16);
    uint32_t a = m & (m >> 1);
    a &= a >> 2;
    a &= a >> 4;
    a &= m >> 8;
    return (a & 0xffffu) != 0;
}

// Three stages of cv::FAST's test for one pixel, each a separate DENSE pass over the survivors of the one before (orb_fast_cells): with all three in one function a
// wavefront executes the long stages whenever ONE of its 64 pixels needs them - on textured images that is always (rounds 1-4: 229 lane-instructions per pixel).
// fast_quick: a run of 9 of the 16 circle pixels contains one pixel of every opposite pair, so two pairs decide most non-corners with four reads (cv::FAST does the same)
__device__ __forceinline__ bool fast_quick(const uint8_t* c, int stride, int min_th) {
    const int v = c[0];
    const int a0 = v - c[3 * stride], a8 = v - c[-3 * stride], a4 = v - c[3], a12 = v - c[-3];
    const bool dark = (a0 > min_th || a8 > min_th) && (a4 > min_th || a12 > min_th);
    const bool bright = (a0 < -min_th || a8 < -min_th) && (a4 < -min_th || a12 < -min_th);
    return dark || bright;
}
__device__ __forceinline__ void fast_ring(const uint8_t* c, int stride, int d[16]) {
    const int v = c[0];
    d[0] = v - c[3 * stride];       d[1] = v - c[3 * stride + 1];   d[2] = v - c[2 * stride + 2];   d[3] = v - c[stride + 3];
    d[4] = v - c[3];                d[5] = v - c[-stride + 3];      d[6] = v - c[-2 * stride + 2];  d[7] = v - c[-3 * stride + 1];
    d[8] = v - c[-3 * stride];      d[9] = v - c[-3 * stride - 1];  d[10] = v - c[-2 * stride - 2]; d[11] = v - c[-stride - 3];
    d[12] = v - c[-3];              d[13] = v - c[stride - 3];      d[14] = v - c[2 * stride - 2];  d[15] = v - c[3 * stride - 1];
}
// fast_is_corner: nine contiguous circle pixels darker / brighter than the centre by more than min_th
__device__ __forceinline__ bool fast_is_corner(const uint8_t* c, int stride, int min_th) {
    int d[16];
    fast_ring(c, stride, d);
    uint32_t dark = 0, bright = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        dark |= (uint32_t)(d[k] > min_th) << k;
        bright |= (uint32_t)(d[k] < -min_th) << k;
    }
    return has9(dark) || has9(bright);
}
// fast_corner_score: cornerScore<16> of a pixel that IS a corner: max over the 16 nine-pixel arcs of min(d) (dark) / min(-d) (bright), minus 1
__device__ __forceinline__ int fast_corner_score(const uint8_t* c, int stride) {
    int d[16];
    fast_ring(c, stride, d);
    int lo2[16], hi2[16], lo4[16], hi4[16];
#pragma unroll
    for (int k = 0; k < 16; k++) { lo2[k] = min(d[k], d[(k + 1) & 15]); hi2[k] = max(d[k], d[(k + 1) & 15]); }
#pragma unroll
    for (int k = 0; k < 16; k++) { lo4[k] = min(lo2[k], lo2[(k + 2) & 15]); hi4[k] = max(hi2[k], hi2[(k + 2) & 15]); }
    int best_dark = -256, best_bright = -256;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const int lo9 = min(min(lo4[k], lo4[(k + 4) & 15]), d[(k + 8) & 15]);
        const int hi9 = max(max(hi4[k], hi4[(k + 4) & 15]), d[(k + 8) & 15]);
        best_dark = max(best_dark, lo9);
        best_bright = max(best_bright, -hi9);
    }
    return max(best_dark, best_bright) - 1;
}

__global__ __launch_bounds__(256) void orb_fast_cells(const PlanDev* __restrict__ plan, const CellDev* __restrict__ cells,
                                                      const uint8_t* __restrict__ pyr, uint32_t* __restrict__ cand,
                                                      int* __restrict__ cell_count, int* __restrict__ dropped, int B) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    __shared__ int s_any_ini, s_cnt;
    int frame, cell;
    xcd_frame_block(plan->ncells_total, B, frame, cell);      // neighbouring cells share their 3-pixel halos: a frame's cells on one XCD (common.h)
    const CellDev C = cells[cell];
    const LevelDev& L = plan->lv[C.level];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ww = C.ww, wh = C.wh;
    int* out_count = cell_count + (int64_t)frame * plan->ncells_total + cell;
    if (ww <= 0 || wh <= 0) { if (tid == 0) *out_count = 0; return; }
    const int tw = ww + 6, th = wh + 6;            // pixel tile with 3-px halo
    // the tile is fetched as aligned 32-bit words (rows of the pyramid start on 16-byte boundaries): it begins `mis` bytes left of the window's halo
    const int gx0 = C.x0 - 3, mis = gx0 & 3, nd = (mis + tw + 3) >> 2;
    const int tstride = nd * 4;
    const int sw = ww + 2, sh = wh + 2;            // score tile with 1-px zero border
    uint8_t* tile = smem;
    uint8_t* score = smem + ((tstride * th + 15) & ~15);
    const uint8_t* img = pyr + (int64_t)frame * plan->pyr_stride + L.off;
    {
        const int q = tid & 15;                    // (nd <= 16: cells are at most 54 pixels wide, checked at plan creation)
        if (q < nd)
            for (int y = tid >> 4; y < th; y += 16)
                *(uint32_t*)(tile + y * tstride + 4 * q) = *(const uint32_t*)(img + (int64_t)(C.y0 - 3 + y) * L.pitch + (gx0 - mis) + 4 * q);
    }
    for (int i = tid; i < sw * sh; i += 256) score[i] = 0;
    if (tid == 0) { s_any_ini = 0; s_cnt = 0; }
    __syncthreads();
    const int min_th = plan->min_th, ini_th = plan->ini_th;
    const int npx = ww * wh;
    const float inv_ww = 1.0f / (float)ww;
    auto row_of = [&](int p_) { return (int)(((float)p_ + 0.5f) * inv_ww); };      // p / ww for p < 2^12, ww <= 64: the product is off by < 2^-11, the nearest integers are 0.5 / ww away
    // the score map in three dense passes: quick test of every pixel -> list; the nine-contiguous test of the list -> list; the exact score of those
    unsigned short* list = (unsigned short*)(score + ((sw * sh + 15) & ~15));      // [npx]
    auto compact = [&](bool keep, int value, int& count) {                          // appends `value` of the lanes with `keep` (order irrelevant: scores go to their pixel's place)
        const unsigned long long m = __ballot(keep);
        int base = 0;
        if (lane == 0 && m) base = atomicAdd(&s_cnt, __popcll(m));
        base = __builtin_amdgcn_readfirstlane(base);
        if (keep) list[base + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)value;
        (void)count;
    };
    int dummy = 0;
    for (int p0 = 0; p0 < npx; p0 += 256) {
        const int p = p0 + tid;
        bool k1 = false;
        if (p < npx) { const int y = row_of(p), x = p - y * ww; k1 = fast_quick(tile + (y + 3) * tstride + (x + 3 + mis), tstride, min_th); }
        compact(k1, p, dummy);
    }
    __syncthreads();
    const int n1 = s_cnt;
    __syncthreads();
    if (tid == 0) s_cnt = 0;
    __syncthreads();
    // (the second list is built in place: entry i is read by the thread that may overwrite a slot j <= i only after every thread of this pass has read its own)
    for (int i0 = 0; i0 < n1; i0 += 256) {
        const int i = i0 + tid;
        int p = 0;
        bool k2 = false;
        if (i < n1) { p = list[i]; const int y = row_of(p), x = p - y * ww; k2 = fast_is_corner(tile + (y + 3) * tstride + (x + 3 + mis), tstride, min_th); }
        __syncthreads();
        compact(k2, p, dummy);
        __syncthreads();
    }
    const int n2 = s_cnt;
    for (int i = tid; i < n2; i += 256) {
        const int p = list[i];
        const int y = row_of(p), x = p - y * ww;
        score[(y + 1) * sw + (x + 1)] = (uint8_t)fast_corner_score(tile + (y + 3) * tstride + (x + 3 + mis), tstride);   // [min_th, 254]
    }
    __syncthreads();
    // Non-maximum suppression and emission over the CORNERS only (the list of the third pass; every other pixel scores 0 and can neither survive nor win):
    // pass A: the corners that are strict 8-neighbour maxima, and whether one of them reaches iniTh
    __shared__ unsigned s_keep[128];                     // a bit per window pixel (npx < 4096)
    __shared__ unsigned short s_kpre[128];
    for (int i = tid; i < 128; i += 256) s_keep[i] = 0u;
    __syncthreads();
    int any = 0;
    bool ismax[4] = {false, false, false, false};      // (n2 <= slot_cap would need <= 4 rounds of 256 only for cells of > 2048 pixels; larger lists loop again below)
    for (int i = tid, k = 0; i < n2; i += 256, k++) {
        const int p = list[i];
        const int y = row_of(p), x = p - y * ww;
        const uint8_t* sc = score + (y + 1) * sw + (x + 1);
        const int s = sc[0];
        const bool mx = s > sc[-1] && s > sc[1] && s > sc[-sw - 1] && s > sc[-sw] && s > sc[-sw + 1] && s > sc[sw - 1] && s > sc[sw] && s > sc[sw + 1];
        if (k < 4) ismax[k] = mx;
        if (mx && s >= ini_th) any = 1;
    }
    if (__any(any) && lane == 0) s_any_ini = 1;
    __syncthreads();
    const int th_cell = s_any_ini ? ini_th : min_th;
    // pass B: the survivors' bits, prefix counts of the bitmap's words, and every survivor to its rank: row-major over the window == cv::FAST's emission order
    for (int i = tid, k = 0; i < n2; i += 256, k++) {
        const int p = list[i];
        const int y = row_of(p), x = p - y * ww;
        const uint8_t* sc = score + (y + 1) * sw + (x + 1);
        const int s = sc[0];
        const bool mx = k < 4 ? ismax[k] : (s > sc[-1] && s > sc[1] && s > sc[-sw - 1] && s > sc[-sw] && s > sc[-sw + 1] && s > sc[sw - 1] && s > sc[sw] && s > sc[sw + 1]);
        if (mx && s >= th_cell) atomicOr(&s_keep[p >> 5], 1u << (p & 31));
    }
    __syncthreads();
    if (wave == 0) {                                    // 128 words: two per lane
        const int c0 = __popc(s_keep[2 * lane]), c1 = __popc(s_keep[2 * lane + 1]);
        int incl = c0 + c1;
        for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o); if (lane >= o) incl += v; }
        s_kpre[2 * lane] = (unsigned short)(incl - c0 - c1); s_kpre[2 * lane + 1] = (unsigned short)(incl - c1);
        if (lane == 63) s_cnt = incl;
    }
    __syncthreads();
    uint32_t* slots = cand + (int64_t)frame * plan->cand_stride + C.slot_off;
    const int relx = C.x0 - L.minBX, rely = C.y0 - L.minBY;
    for (int i = tid; i < n2; i += 256) {
        const int p = list[i];
        const unsigned bits = s_keep[p >> 5];
        if ((bits >> (p & 31)) & 1u) {
            const int pos = (int)s_kpre[p >> 5] + __popc(bits & ((1u << (p & 31)) - 1u));
            const int y = row_of(p), x = p - y * ww;
            if (pos < C.slot_cap) slots[pos] = (uint32_t)(x + relx) | ((uint32_t)(y + rely) << 12) | ((uint32_t)score[(y + 1) * sw + (x + 1)] << 24);
        }
    }
    const int base = s_cnt;
    if (tid == 0) {
        *out_count = min(base, C.slot_cap);
        // slot_cap = ceil(w/2) * ceil(h/2) bounds the strict 8-neighbour maxima of a cell, so this never fires; counted (planar_orb_check), not assumed
        if (base > C.slot_cap) atomicAdd(dropped, base - C.slot_cap);
    }
}

// ------------------------------------------------------------------------------------------
// K3: gather + path code + radix sort, one workgroup per (frame, level).
// element = (code << 32) | key, key = x | y<<12 | score<<24 (x,y relative to minBorder).
__device__ __forceinline__ uint32_t path_code(const LevelDev& L, int x, int y) {
    int ini = 0, X0 = 0, X1 = L.maxBX - L.minBX;
    if (L.nIni > 1) {
        ini = (int)((float)x / L.hX);                       // :569
        X0 = (int)(L.hX * (float)ini);
        X1 = (int)(L.hX * (float)(ini + 1));
    }
    int Y0 = 0, Y1 = L.maxBY - L.minBY;
    uint32_t code = (uint32_t)ini;
    for (int d = 0; d < L.depth; d++) {
        const int xm = X0 + ((X1 - X0 + 1) >> 1);           // UL.x + ceil((UR.x-UL.x)/2)  (:483)
        const int ym = Y0 + ((Y1 - Y0 + 1) >> 1);
        const uint32_t q = (x >= xm ? 1u : 0u) | (y >= ym ? 2u : 0u);   // n1,n2,n3,n4 (:513-526)
        code = (code << 2) | q;
        if (x >= xm) X0 = xm; else X1 = xm;
        if (y >= ym) Y0 = ym; else Y1 = ym;
    }
    return code;
}

__global__ __launch_bounds__(256) void orb_sort(const PlanDev* __restrict__ plan, const CellDev* __restrict__ cells,
                                                const uint32_t* __restrict__ cand, const int* __restrict__ cell_count,
                                                uint64_t* __restrict__ sortA, uint64_t* __restrict__ sortB,
                                                int* __restrict__ level_count) {
    __shared__ int s_scan[4];
    __shared__ int s_base;
    __shared__ int s_digit_base[16];
    __shared__ int s_wave_digit[4][16];
    __shared__ int s_hist[16];
    const int level = blockIdx.x, frame = blockIdx.y;
    const LevelDev& L = plan->lv[level];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint64_t* A = sortA + (int64_t)frame * plan->cand_stride + L.cand_off;
    uint64_t* Bf = sortB + (int64_t)frame * plan->cand_stride + L.cand_off;
    const uint32_t* slots = cand + (int64_t)frame * plan->cand_stride;
    const int* counts = cell_count + (int64_t)frame * plan->ncells_total + L.cell_begin;

    // 1. gather cells in order: exclusive scan of counts in chunks of 256 cells, copy + code
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (int c0 = 0; c0 < L.ncells; c0 += 256) {
        const int c = c0 + tid;
        const int cnt = c < L.ncells ? counts[c] : 0;
        int incl = cnt;   // inclusive wave scan
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o); if (lane >= o) incl += t; }
        if (lane == 63) s_scan[wave] = incl;
        __syncthreads();
        int start = s_base + incl - cnt;
        for (int w = 0; w < wave; w++) start += s_scan[w];
        if (cnt > 0) {
            const CellDev C = cells[L.cell_begin + c];
            for (int i = 0; i < cnt; i++) {
                const uint32_t key = slots[C.slot_off + i];
                const uint32_t code = path_code(L, (int)(key & 0xfffu), (int)((key >> 12) & 0xfffu));
                A[start + i] = ((uint64_t)code << 32) | key;
            }
        }
        __syncthreads();
        if (tid == 255) s_base = start + cnt;
        __syncthreads();
    }
    const int K = s_base;
    if (tid == 0) level_count[frame * MAX_LEVELS + level] = K;
    __syncthreads();

    // 2. LSD radix sort, RADIX_BITS per pass, stable: ranks inside a wave by ballot matching.
    uint64_t* src = A;
    uint64_t* dst = Bf;
    for (int shift = 0; shift < L.code_bits; shift += RADIX_BITS) {
        if (tid < 16) s_hist[tid] = 0;
        __syncthreads();
        for (int i = tid; i < K; i += 256) atomicAdd(&s_hist[(int)((src[i] >> (32 + shift)) & 15u)], 1);
        __syncthreads();
        if (tid == 0) { int run = 0; for (int d = 0; d < 16; d++) { s_digit_base[d] = run; run += s_hist[d]; } }
        __syncthreads();
        if (lane < 16) s_wave_digit[wave][lane] = 0;
        __syncthreads();
        for (int i0 = 0; i0 < K; i0 += 256) {
            const int i = i0 + tid;
            const bool valid = i < K;
            const uint64_t e = valid ? src[i] : 0;
            const int dg = (int)((e >> (32 + shift)) & 15u);
            unsigned long long peers = __ballot(valid);          // lanes of this wave holding the same digit
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const unsigned long long bm = __ballot(valid && ((dg >> b) & 1));
                peers &= ((dg >> b) & 1) ? bm : ~bm;
            }
            const int rank = __popcll(peers & ((1ull << lane) - 1ull));
            if (valid && rank == 0) s_wave_digit[wave][dg] = __popcll(peers);
            __syncthreads();
            if (valid) {
                int pos = s_digit_base[dg] + rank;
                for (int w = 0; w < wave; w++) pos += s_wave_digit[w][dg];
                dst[pos] = e;
            }
            __syncthreads();
            if (tid < 16) s_digit_base[tid] += s_wave_digit[0][tid] + s_wave_digit[1][tid] + s_wave_digit[2][tid] + s_wave_digit[3][tid];
            __syncthreads();
            if (lane < 16) s_wave_digit[wave][lane] = 0;         // own row; rewritten by this wave only
        }
        __syncthreads();
        uint64_t* t = src; src = dst; dst = t;
    }
    // result is in `src`; the host knows the pass parity (code_bits / RADIX_BITS).
}

// ------------------------------------------------------------------------------------------
// K4: DistributeOctTree on the sorted array, one wave per (frame, level).
// A node is (lo, hi, depth): keys [lo,hi) share the first `depth` path digits (after the
// initial-node index).  LDS holds the node list (doubly linked, slots recycled) and the
// "to expand" lists.  All 64 lanes run the same control flow; lanes cooperate on the
// boundary searches and on the per-node arg-max.
__device__ __forceinline__ int digit_at(const uint64_t* __restrict__ S, int idx, int shift) {
    return (int)((S[idx] >> (32 + shift)) & 3u);
}

// first index in [lo,hi) whose digit >= q (digits are non-decreasing inside a node)
__device__ int wave_lower_bound(const uint64_t* __restrict__ S, int lo, int hi, int shift, int q, int lane) {
    while (hi - lo > 64) {
        const int step = (hi - lo + 63) >> 6;
        const int idx = lo + lane * step + step - 1;        // last element of this lane's chunk
        // chunks that are empty or cut short by `hi` answer "yes": everything before them is < q
        const bool ge = idx < hi ? digit_at(S, idx, shift) >= q : true;
        const int f = __ffsll((long long)__ballot(ge)) - 1; // first chunk that may contain the answer
        const int nlo = min(lo + f * step, hi);
        hi = min(nlo + step, hi);
        lo = nlo;
    }
    const int idx = lo + lane;
    const bool ge = idx < hi ? digit_at(S, idx, shift) >= q : true;
    const int f = __ffsll((long long)__ballot(ge)) - 1;
    return min(lo + f, hi);
}

__global__ __launch_bounds__(64) void orb_octree(const PlanDev* __restrict__ plan, const uint64_t* __restrict__ sortA,
                                                 const uint64_t* __restrict__ sortB, const int* __restrict__ level_count,
                                                 uint32_t* __restrict__ kept, int* __restrict__ kept_count, int node_cap) {
    extern __shared__ __attribute__((aligned(16))) int osm[];
    const int level = blockIdx.x, frame = blockIdx.y, lane = threadIdx.x;
    const LevelDev& L = plan->lv[level];
    const int passes = L.code_bits / RADIX_BITS;
    const uint64_t* S = ((passes & 1) ? sortB : sortA) + (int64_t)frame * plan->cand_stride + L.cand_off;
    const int K = level_count[frame * MAX_LEVELS + level];
    const int N = L.nfeat;
    uint32_t* out = kept + (int64_t)frame * plan->kept_stride + L.kept_off;
    int* out_count = kept_count + frame * MAX_LEVELS + level;
    if (K == 0) { if (lane == 0) *out_count = 0; return; }

    int* n_lo = osm;               int* n_hi = n_lo + node_cap;    int* n_depth = n_hi + node_cap;
    int* n_next = n_depth + node_cap; int* n_prev = n_next + node_cap; int* n_seq = n_prev + node_cap;
    int* freel = n_seq + node_cap;
    int* e_size0 = freel + node_cap;  int* e_seq0 = e_size0 + node_cap; int* e_slot0 = e_seq0 + node_cap;
    int* e_size1 = e_slot0 + node_cap; int* e_seq1 = e_size1 + node_cap; int* e_slot1 = e_seq1 + node_cap;

    // --- list primitives (uniform; executed redundantly by every lane on LDS) ---
    int head = -1, tail = -1, count = 0, nfree = 0, seq_ctr = 0;
    for (int i = lane; i < node_cap; i += 64) freel[i] = node_cap - 1 - i;
    nfree = node_cap;
    __syncthreads();

    auto alloc_node = [&](int lo, int hi, int depth) -> int {
        const int s = freel[nfree - 1];
        nfree--;
        if (lane == 0) { n_lo[s] = lo; n_hi[s] = hi; n_depth[s] = depth; n_seq[s] = seq_ctr; }
        seq_ctr++;
        return s;
    };
    auto push_front = [&](int s) {
        if (lane == 0) { n_prev[s] = -1; n_next[s] = head; if (head >= 0) n_prev[head] = s; }
        if (head < 0) tail = s;
        head = s; count++;
    };
    auto push_back = [&](int s) {
        if (lane == 0) { n_next[s] = -1; n_prev[s] = tail; if (tail >= 0) n_next[tail] = s; }
        if (tail < 0) head = s;
        tail = s; count++;
    };
    auto erase = [&](int s) -> int {   // returns next
        __threadfence_block();
        const int p = n_prev[s], n = n_next[s];
        if (lane == 0) { if (p >= 0) n_next[p] = n; if (n >= 0) n_prev[n] = p; freel[nfree] = s; }
        if (p < 0) head = n;
        if (n < 0) tail = p;
        nfree++; count--;
        return n;
    };

    // --- initial nodes (:552-585): key ranges by initial-node index (top bits of the code) ---
    {
        const int ini_shift = 2 * L.depth;
        int lo = 0;
        for (int i = 0; i < L.nIni; i++) {
            // upper bound of index i: first element with (code >> ini_shift) > i
            int a = lo, b = K;
            while (b - a > 0) {   // plain binary search (nIni is 1 for 4:3 images; not hot)
                const int mid = (a + b) >> 1;
                if ((int)(S[mid] >> (32 + ini_shift)) > i) b = mid; else a = mid + 1;
            }
            const int hi = a;
            if (hi > lo) { const int s = alloc_node(lo, hi, 0); __threadfence_block(); push_back(s); }
            lo = hi;
        }
    }
    __threadfence_block();

    // "to expand" lists: c* = being filled by split(), p* = previous round (phase 2)
    int* c_size = e_size0; int* c_seq = e_seq0; int* c_slot = e_slot0; int c_n = 0;
    int* p_size = e_size1; int* p_seq = e_seq1; int* p_slot = e_slot1; int p_n = 0;

    // split node s: children pushed to the front in order n1..n4; children with >1 keys are
    // appended to e_*[cur]; returns number of such children (nToExpand contribution)
    auto split = [&](int s) -> int {
        __threadfence_block();
        const int lo = n_lo[s], hi = n_hi[s], depth = n_depth[s];
        const int shift = 2 * (L.depth - 1 - depth);
        int b1, b2, b3;
        const int n = hi - lo;
        if (n <= 256) {
            int c0 = 0, c1 = 0, c2 = 0;
            for (int i0 = 0; i0 < n; i0 += 64) {
                const int idx = lo + i0 + lane;
                const int dg = idx < hi ? digit_at(S, idx, shift) : 4;
                c0 += __popcll(__ballot(dg == 0)); c1 += __popcll(__ballot(dg == 1)); c2 += __popcll(__ballot(dg == 2));
            }
            b1 = lo + c0; b2 = b1 + c1; b3 = b2 + c2;
        } else {
            b1 = wave_lower_bound(S, lo, hi, shift, 1, lane);
            b2 = wave_lower_bound(S, b1, hi, shift, 2, lane);
            b3 = wave_lower_bound(S, b2, hi, shift, 3, lane);
        }
        const int bl[5] = {lo, b1, b2, b3, hi};
        int nexp = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int sz = bl[q + 1] - bl[q];
            if (sz <= 0) continue;
            const int myseq = seq_ctr;
            const int c = alloc_node(bl[q], bl[q + 1], depth + 1);
            __threadfence_block();
            push_front(c);
            if (sz > 1) {
                if (lane == 0) { c_size[c_n] = sz; c_seq[c_n] = myseq; c_slot[c_n] = c; }
                c_n++;
                nexp++;
            }
        }
        __threadfence_block();
        return nexp;
    };

    bool finish = false;
    while (!finish) {                                   // :595
        const int prevSize = count;
        int nToExpand = 0;
        c_n = 0;
        int it = head;
        while (it >= 0) {                               // :607-660
            __threadfence_block();
            const int sz = n_hi[it] - n_lo[it];
            if (sz == 1) { it = n_next[it]; continue; } // bNoMore
            nToExpand += split(it);
            it = erase(it);
        }
        if (count >= N || count == prevSize) {          // :664
            finish = true;
        } else if (count + nToExpand * 3 > N) {         // :668
            while (!finish) {
                const int prevSize2 = count;
                { int* t; t = c_size; c_size = p_size; p_size = t; t = c_seq; c_seq = p_seq; p_seq = t; t = c_slot; c_slot = p_slot; p_slot = t; }
                p_n = c_n; c_n = 0;
                __threadfence_block();
                // descending (size, seq) == back-to-front over the ascending sort (:684-685):
                // repeated wave arg-max over the previous round's list
                for (int step = 0; step < p_n; step++) {
                    unsigned long long best = 0;
                    for (int i = lane; i < p_n; i += 64) {
                        const int sz = p_size[i];
                        if (sz > 0) {
                            const unsigned long long v = ((unsigned long long)sz << 40) | ((unsigned long long)p_seq[i] << 16) | (unsigned)i;
                            best = v > best ? v : best;
                        }
                    }
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) { const unsigned long long t = __shfl_xor(best, o); best = t > best ? t : best; }
                    const int idx = (int)(best & 0xffffu);
                    const int slot = p_slot[idx];
                    if (lane == 0) p_size[idx] = 0;
                    __threadfence_block();
                    split(slot);
                    erase(slot);
                    if (count >= N) break;
                }
                if (count >= N || count == prevSize2) finish = true;
            }
        }
    }

    // --- best response per node in list order (:742-760) ---
    __threadfence_block();
    int pos = 0;
    const int cw = L.wCell, ch = L.hCell, ncols = L.nCols;
    for (int it = head; it >= 0; it = n_next[it], pos++) {
        const int lo = n_lo[it], hi = n_hi[it];
        unsigned long long best = 0;
        for (int i = lo + lane; i < hi; i += 64) {
            const uint32_t key = (uint32_t)S[i];
            const unsigned x = key & 0xfffu, y = (key >> 12) & 0xfffu, sc = key >> 24;
            // emission order of the reference: cell (row, col) row-major, then (y, x) inside the cell
            const unsigned cell = ((y - 3) / ch) * ncols + (x - 3) / cw;
            const unsigned long long ord = ((unsigned long long)cell << 24) | (y << 12) | x;
            const unsigned long long v = ((unsigned long long)sc << 48) | ((~ord) & 0xffffffffffffull);
            best = v > best ? v : best;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const unsigned long long t = __shfl_xor(best, o); best = t > best ? t : best; }
        if (lane == 0) {
            const unsigned long long ord = (~best) & 0xffffffffffffull;
            out[pos] = (uint32_t)(ord & 0xffffffu) | ((uint32_t)(best >> 48) << 24);
        }
    }
    if (lane == 0) *out_count = pos;
}

// ------------------------------------------------------------------------------------------
// K5: GaussianBlur 7x7 sigma 2 (taps 18,34,49,55,49,34,18 /256 per pass, reflect-101),
// out = min(255, (sum + 2^15) >> 16).  64x16 output tile per workgroup.
struct TileDev { short level, tx, ty, pad; };

__global__ __launch_bounds__(256) void orb_blur(const PlanDev* __restrict__ plan, const TileDev* __restrict__ tiles,
                                                const uint8_t* __restrict__ pyr, uint8_t* __restrict__ blur, int n_tiles, int B) {
    // 22 input rows x 18 words (columns x0 - 4 .. x0 + 67) as 32-bit words; horizontal sums as u16 with a row stride of 68 (8-byte aligned, banks spread)
    __shared__ uint32_t s_pw[22][19];
    __shared__ __attribute__((aligned(8))) uint16_t s_h[22][68];
    int frame, tile;
    xcd_frame_block(n_tiles, B, frame, tile);                 // neighbouring tiles share 3 rows / 4 columns of halo: a frame's tiles on one XCD (common.h)
    const TileDev T = tiles[tile];
    const LevelDev& L = plan->lv[T.level];
    const int tid = threadIdx.x;
    const int x0 = T.tx * 64, y0 = T.ty * 16;
    const uint8_t* img = pyr + (int64_t)frame * plan->pyr_stride + L.off;
    for (int i = tid; i < 22 * 18; i += 256) {
        const int r = i / 18, wq = i - r * 18;
        int y = y0 + r - 3;
        // reflect-101; tiles may overhang the right/bottom edge, where the values are unused
        y = y < 0 ? -y : y; y = y >= L.h ? 2 * L.h - 2 - y : y; y = max(0, min(y, L.h - 1));
        const uint8_t* row = img + (int64_t)y * L.pitch;
        const int xs = x0 - 4 + 4 * wq;
        uint32_t w;
        if (xs >= 0 && xs + 3 < L.w) w = *(const uint32_t*)(row + xs);          // rows and xs are 4-byte aligned
        else {
            w = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                int x = xs + k;
                x = x < 0 ? -x : x; x = x >= L.w ? 2 * L.w - 2 - x : x; x = max(0, min(x, L.w - 1));
                w |= (uint32_t)row[x] << (8 * k);
            }
        }
        s_pw[r][wq] = w;
    }
    __syncthreads();
    for (int i = tid; i < 22 * 16; i += 256) {
        const int r = i >> 4, q = i & 15;
        const uint32_t w0 = s_pw[r][q], w1 = s_pw[r][q + 1], w2 = s_pw[r][q + 2];
        // bytes 1 .. 10 of the 12 = input columns 4q - 3 .. 4q + 6 of the tile
        uint32_t b[10];
        b[0] = (w0 >> 8) & 255u; b[1] = (w0 >> 16) & 255u; b[2] = w0 >> 24;
        b[3] = w1 & 255u; b[4] = (w1 >> 8) & 255u; b[5] = (w1 >> 16) & 255u; b[6] = w1 >> 24;
        b[7] = w2 & 255u; b[8] = (w2 >> 8) & 255u; b[9] = (w2 >> 16) & 255u;
        uint32_t o[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t v = 18u * (b[j] + b[j + 6]) + 34u * (b[j + 1] + b[j + 5]) + 49u * (b[j + 2] + b[j + 4]) + 55u * b[j + 3];
            o[j] = min(v, 65535u);
        }
        *(uint2*)&s_h[r][4 * q] = make_uint2(o[0] | (o[1] << 16), o[2] | (o[3] << 16));
    }
    __syncthreads();
    {
        const int c4 = (tid & 15) * 4, r = tid >> 4;    // 16 rows x 16 quads
        uint32_t acc[4] = {0, 0, 0, 0};
        const uint32_t taps[7] = {18u, 34u, 49u, 55u, 49u, 34u, 18u};
        // (the sum below adds the seven products in the order 18 (r, r+6), 34 (r+1, r+5), 49 (r+2, r+4), 55 (r+3): integer arithmetic, any order gives the same value)
#pragma unroll
        for (int k = 0; k < 7; k++) {
            const uint2 v = *(const uint2*)&s_h[r + k][c4];
            acc[0] += taps[k] * (v.x & 0xffffu); acc[1] += taps[k] * (v.x >> 16);
            acc[2] += taps[k] * (v.y & 0xffffu); acc[3] += taps[k] * (v.y >> 16);
        }
        uint32_t packed = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) packed |= min((acc[i] + 32768u) >> 16, 255u) << (8 * i);
        const int y = y0 + r, x = x0 + c4;
        if (y < L.h && x < L.pitch)
            *(uint32_t*)(blur + (int64_t)frame * plan->pyr_stride + L.off + (int64_t)y * L.pitch + x) = packed;
    }
}

// ------------------------------------------------------------------------------------------
// K6: orientation + descriptor; 16 lanes per keypoint, 4 keypoints per wave.
__device__ __forceinline__ float fast_atan2_deg(float y, float x) {
    // cv::fastAtan2 (atan_f32): 7th-order odd polynomial, degrees; plain mul/add (no FMA)
    const float scale = (float)(180 / 3.1415926535897932384626433832795);
    const float p1 = 0.9997878412794807f * scale, p3 = -0.3258083974640975f * scale;
    const float p5 = 0.1555786518463281f * scale, p7 = -0.04432655554792128f * scale;
    const float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = __fdiv_rn(ay, __fadd_rn(ax, 2.220446049250313e-16f));
        c2 = __fmul_rn(c, c);
        a = __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c);
    } else {
        c = __fdiv_rn(ax, __fadd_rn(ay, 2.220446049250313e-16f));
        c2 = __fmul_rn(c, c);
        a = __fsub_rn(90.f, __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c));
    }
    if (x < 0) a = __fsub_rn(180.f, a);
    if (y < 0) a = __fsub_rn(360.f, a);
    return a;
}

__global__ __launch_bounds__(256) void orb_describe(const PlanDev* __restrict__ plan, const uint8_t* __restrict__ pyr,
                                                    const uint8_t* __restrict__ blur, const uint32_t* __restrict__ kept,
                                                    const int* __restrict__ kept_count, planar_keypoint* __restrict__ kps,
                                                    uint8_t* __restrict__ desc, int32_t* __restrict__ n_out, int per_frame, int B) {
    // a frame's 1000 patches (38 rows of the level image and of its blurred clone each) overlap 4-5 times over: its workgroups run on one XCD (common.h), so that
    // the frame's two pyramids go through ONE L2 once (rounds 1-4: spread over all eight, 5.0 GB of fetches per 1024 frames for 1.6 GB of pyramids)
    int frame, bx;
    xcd_frame_block(per_frame, B, frame, bx);
    const int sub = threadIdx.x & 15;
    const int kpi = bx * 16 + (threadIdx.x >> 4);               // keypoint index in the frame's output
    // locate level: prefix over per-level kept counts
    int level = -1, first = 0, total = 0;
    for (int l = 0; l < plan->nlevels; l++) {
        const int c = kept_count[frame * MAX_LEVELS + l];
        if (level < 0 && kpi < total + c) { level = l; first = total; }
        total += c;
    }
    if (bx == 0 && threadIdx.x == 0) n_out[frame] = total;
    if (level < 0) return;                                      // uniform per 16-lane group
    const LevelDev& L = plan->lv[level];
    const uint32_t key = kept[(int64_t)frame * plan->kept_stride + L.kept_off + (kpi - first)];
    const int x = (int)(key & 0xfffu) + L.minBX, y = (int)((key >> 12) & 0xfffu) + L.minBY;
    const int score = (int)(key >> 24);
    const int64_t lvl_off = (int64_t)frame * plan->pyr_stride + L.off;

    // IC_Angle (:77-104): lane `sub` handles rows +-sub (sub = 0: centre row)
    const uint8_t* c = pyr + lvl_off + (int64_t)y * L.pitch + x;
    int m10 = 0, m01 = 0;
    if (sub == 0) {
        for (int u = -HALF_PATCH; u <= HALF_PATCH; ++u) m10 += u * c[u];
    } else {
        const int d = plan->umax[sub];
        int vsum = 0;
        const uint8_t* cp = c + (int64_t)sub * L.pitch;
        const uint8_t* cm = c - (int64_t)sub * L.pitch;
        for (int u = -d; u <= d; ++u) {
            const int vp = cp[u], vm = cm[u];
            vsum += vp - vm;
            m10 += u * (vp + vm);
        }
        m01 = sub * vsum;
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) { m10 += __shfl_xor(m10, o); m01 += __shfl_xor(m01, o); }
    const float angle = fast_atan2_deg((float)m01, (float)m10);

    // steered BRIEF (:107-147): lane `sub` produces descriptor bytes 2*sub, 2*sub+1
    const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
    const float ang = __fmul_rn(angle, factorPI);
    const float a = (float)cos((double)ang), b = (float)sin((double)ang);
    const uint8_t* bc = blur + lvl_off + (int64_t)y * L.pitch + x;
    const int step = L.pitch;
    uint32_t bits = 0;
    const signed char* pat = c_pattern + sub * 64;
#pragma unroll
    for (int i = 0; i < 16; i++) {                      // (fully unrolled: the 32 taps of a lane are independent byte loads, all of them in flight)
        const float x0 = (float)pat[4 * i], y0 = (float)pat[4 * i + 1], x1 = (float)pat[4 * i + 2], y1 = (float)pat[4 * i + 3];
        const int r0 = __float2int_rn(__fadd_rn(__fmul_rn(x0, b), __fmul_rn(y0, a)));
        const int q0 = __float2int_rn(__fsub_rn(__fmul_rn(x0, a), __fmul_rn(y0, b)));
        const int r1 = __float2int_rn(__fadd_rn(__fmul_rn(x1, b), __fmul_rn(y1, a)));
        const int q1 = __float2int_rn(__fsub_rn(__fmul_rn(x1, a), __fmul_rn(y1, b)));
        const int t0 = bc[r0 * step + q0], t1 = bc[r1 * step + q1];
        bits |= (uint32_t)(t0 < t1) << i;
    }
    const int64_t o = (int64_t)frame * plan->kp_cap + kpi;
    *(uint16_t*)(desc + o * 32 + sub * 2) = (uint16_t)bits;
    if (sub == 0) {
        planar_keypoint kp;
        kp.x = __fmul_rn((float)x, L.scale);      // pt *= scale (:1094-1100); scale[0] == 1
        kp.y = __fmul_rn((float)y, L.scale);
        kp.size = L.patch;
        kp.angle = angle;
        kp.response = (float)score;
        kp.octave = level;
        kp.class_id = -1;
        kps[o] = kp;
    }
}

}  // namespace orb
}  // namespace planar

// ==========================================================================================
// host side
// ==========================================================================================
using namespace planar;
using namespace planar::orb;

struct planar_orb {
    planar_ctx* ctx = nullptr;
    planar_orb_params params{};
    int W = 0, H = 0, max_batch = 0;
    PlanDev plan{};
    std::vector<CellDev> cells;
    std::vector<TileDev> tiles;
    std::vector<float> scale, inv_scale, sigma2, inv_sigma2;
    int fast_smem = 0, node_cap = 0, oct_smem = 0;
    int last_B = 0;
    DevBuf d_plan, d_cells, d_tiles, d_tabs, d_pyr, d_blur, d_cand, d_cell_count, d_sortA, d_sortB, d_level_count,
        d_kept, d_kept_count, d_dropped;
    // staging for the host-pointer entry point
    DevBuf d_in, d_kps, d_desc, d_nout;
    // optional per-launch HIP-event timing (planar_orb_set_profiling)
    bool profiling = false;
    std::vector<std::vector<hipEvent_t>> ev_sets;   // one set of (launches+1) events per recorded call
    size_t ev_used = 0;
    std::vector<const char*> launch_names;          // kernel name per launch of one extract call
    ~planar_orb() {
        for (auto& v : ev_sets) for (hipEvent_t e : v) (void)hipEventDestroy(e);
    }
};

static inline int cv_round_f(float v) { return (int)nearbyintf(v); }
static inline int cv_round_d(double v) { return (int)nearbyint(v); }
static inline short sat_short(float v) { int i = cv_round_f(v); return (short)std::min(32767, std::max(-32768, i)); }

extern "C" {

int planar_orb_create(planar_ctx* ctx, const planar_orb_params* p, int W, int H, int max_batch, planar_orb** out) {
    PLANAR_REQUIRE(ctx && p && out, PLANAR_EINVAL, "null argument");
    *out = nullptr;
    PLANAR_REQUIRE(p->nlevels >= 1 && p->nlevels <= MAX_LEVELS, PLANAR_EINVAL, "nlevels must be in [1,16]");
    PLANAR_REQUIRE(p->nfeatures >= 1 && p->nfeatures <= 60000, PLANAR_EINVAL, "nfeatures out of range");
    PLANAR_REQUIRE(p->scale_factor > 1.0f, PLANAR_EINVAL, "scale_factor must be > 1");
    PLANAR_REQUIRE(p->min_th_fast >= 1 && p->ini_th_fast >= p->min_th_fast && p->ini_th_fast <= 254, PLANAR_EINVAL,
                   "FAST thresholds must satisfy 1 <= min <= ini <= 254");
    PLANAR_REQUIRE(W >= 64 && H >= 64 && W <= 4096 && H <= 4096, PLANAR_EINVAL, "image size must be within [64,4096]");
    PLANAR_REQUIRE(max_batch >= 1, PLANAR_EINVAL, "max_batch must be >= 1");
    PLANAR_HIP_CHECK(hipSetDevice(ctx->device));

    planar_orb* o = new (std::nothrow) planar_orb();
    PLANAR_REQUIRE(o != nullptr, PLANAR_ENOMEM, "host allocation failed");
    o->ctx = ctx; o->params = *p; o->W = W; o->H = H; o->max_batch = max_batch;
    const int nl = p->nlevels;
    const double scaleFactor = p->scale_factor;   // the reference stores the float argument in a double member
    // --- ORBextractor::ORBextractor (src/ORBextractor.cc:410-470) ---
    o->scale.resize(nl); o->sigma2.resize(nl); o->inv_scale.resize(nl); o->inv_sigma2.resize(nl);
    o->scale[0] = 1.0f; o->sigma2[0] = 1.0f;
    for (int i = 1; i < nl; i++) { o->scale[i] = (float)(o->scale[i - 1] * scaleFactor); o->sigma2[i] = o->scale[i] * o->scale[i]; }
    for (int i = 0; i < nl; i++) { o->inv_scale[i] = 1.0f / o->scale[i]; o->inv_sigma2[i] = 1.0f / o->sigma2[i]; }
    std::vector<int> nfeat(nl);
    {
        float factor = (float)(1.0f / scaleFactor);
        float nDesired = p->nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nl));
        int sum = 0;
        for (int l = 0; l < nl - 1; l++) { nfeat[l] = cv_round_f(nDesired); sum += nfeat[l]; nDesired *= factor; }
        nfeat[nl - 1] = std::max(p->nfeatures - sum, 0);
    }
    PlanDev& P = o->plan;
    memset(&P, 0, sizeof(P));
    {
        int v, v0, vmax = (int)std::floor(HALF_PATCH * std::sqrt(2.f) / 2 + 1);
        int vmin = (int)std::ceil(HALF_PATCH * std::sqrt(2.f) / 2);
        const double hp2 = HALF_PATCH * HALF_PATCH;
        for (v = 0; v <= vmax; ++v) P.umax[v] = cv_round_d(std::sqrt(hp2 - v * v));
        for (v = HALF_PATCH, v0 = 0; v >= vmin; --v) { while (P.umax[v0] == P.umax[v0 + 1]) ++v0; P.umax[v] = v0; ++v0; }
    }
    P.nlevels = nl; P.ini_th = p->ini_th_fast; P.min_th = p->min_th_fast;

    // --- per-level geometry (ComputePyramid :1107-1131, cell grid :771-786) ---
    int64_t pyr_off = 0;
    int cand_off = 0, kept_off = 0, rs_off = 0;
    int max_tile_bytes = 0, max_nfeat = 0;
    std::vector<short4> tabs;
    for (int l = 0; l < nl; l++) {
        LevelDev& L = P.lv[l];
        L.w = cv_round_f((float)W * o->inv_scale[l]);
        L.h = cv_round_f((float)H * o->inv_scale[l]);
        L.minBX = L.minBY = EDGE_THRESHOLD - 3;
        L.maxBX = L.w - EDGE_THRESHOLD + 3; L.maxBY = L.h - EDGE_THRESHOLD + 3;
        const float width = (float)(L.maxBX - L.minBX), height = (float)(L.maxBY - L.minBY);
        L.nCols = (int)(width / CELL_W); L.nRows = (int)(height / CELL_W);
        if (L.nCols < 1 || L.nRows < 1 || L.w < 2 * EDGE_THRESHOLD + 8 || L.h < 2 * EDGE_THRESHOLD + 8) {
            delete o;
            set_error("planar_orb_create: pyramid level %d (%dx%d) is too small for the 30-px FAST cell grid", l, L.w, L.h);
            return PLANAR_EINVAL;
        }
        L.wCell = (int)std::ceil(width / L.nCols); L.hCell = (int)std::ceil(height / L.nRows);
        L.pitch = align_up(L.w, 16);
        L.off = pyr_off;
        pyr_off += align_up((int64_t)L.pitch * L.h, (int64_t)256);
        L.nfeat = nfeat[l];
        max_nfeat = std::max(max_nfeat, nfeat[l]);
        L.scale = o->scale[l];
        L.patch = (float)(int)(PATCH_SIZE * o->scale[l]);
        // octree geometry (:543-545); nIni < 1 makes the reference divide by zero -> refuse
        L.nIni = (int)std::round((float)(L.maxBX - L.minBX) / (L.maxBY - L.minBY));
        if (L.nIni < 1 || L.nIni > 64) {
            delete o;
            set_error("planar_orb_create: level %d aspect ratio gives %d initial octree nodes (supported: 1..64)", l, L.nIni);
            return PLANAR_EINVAL;
        }
        L.hX = (float)(L.maxBX - L.minBX) / L.nIni;
        int ext = std::max(L.maxBX - L.minBX, L.maxBY - L.minBY), D = 1;
        while ((1 << (D - 1)) < ext) D++;          // after D halvings (ceil) every cell is <= 1 px
        L.depth = D;
        int ini_bits = 0; while ((1 << ini_bits) < L.nIni) ini_bits++;
        L.code_bits = align_up(2 * D + ini_bits, RADIX_BITS);
        if (L.code_bits > 32) { delete o; set_error("planar_orb_create: path code needs %d bits", L.code_bits); return PLANAR_EINVAL; }
        // cells (:789-826)
        L.cell_begin = (int)o->cells.size();
        L.cand_off = cand_off;
        for (int i = 0; i < L.nRows; i++) {
            const float iniY = (float)(L.minBY + i * L.hCell);
            float maxY = iniY + L.hCell + 6;
            if (iniY >= L.maxBY - 3) continue;
            if (maxY > L.maxBY) maxY = (float)L.maxBY;
            for (int j = 0; j < L.nCols; j++) {
                const float iniX = (float)(L.minBX + j * L.wCell);
                float maxX = iniX + L.wCell + 6;
                if (iniX >= L.maxBX - 6) continue;
                if (maxX > L.maxBX) maxX = (float)L.maxBX;
                CellDev C;
                C.level = (short)l;
                C.x0 = (short)((int)iniX + 3); C.y0 = (short)((int)iniY + 3);
                C.ww = (short)((int)maxX - 3 - C.x0); C.wh = (short)((int)maxY - 3 - C.y0);
                if (C.ww <= 0 || C.wh <= 0) continue;   // ROI narrower than 7: cv::FAST returns nothing
                C.slot_off = cand_off;
                C.slot_cap = ((C.ww + 1) / 2) * ((C.wh + 1) / 2);   // strict 8-neighbour maxima bound
                cand_off += C.slot_cap;
                o->cells.push_back(C);
                const int tstride = ((((int)C.x0 - 3) & 3) + C.ww + 6 + 3) & ~3;            // orb_fast_cells' tile: aligned 32-bit words per row, <= 16 of them
                if (tstride > 64 || C.ww * C.wh >= 4096) { delete o; set_error("planar_orb_create: FAST cell of %d x %d pixels at level %d is larger than the kernel's tile", (int)C.ww, (int)C.wh, l); return PLANAR_EINVAL; }
                const int bytes = ((tstride * (C.wh + 6) + 15) & ~15) + (((C.ww + 2) * (C.wh + 2) + 15) & ~15) + 2 * C.ww * C.wh;   // pixel tile, score tile, the candidate list
                max_tile_bytes = std::max(max_tile_bytes, bytes);
            }
        }
        L.ncells = (int)o->cells.size() - L.cell_begin;
        L.cand_cap = cand_off - L.cand_off;
        L.kept_off = kept_off;
        L.kept_cap = std::max(L.nfeat + 4, 4 * L.nIni + 4);
        kept_off += L.kept_cap;
        // resize tables (cv::resize INTER_LINEAR, resize.cpp) for l >= 1
        L.rs_off = rs_off;
        if (l > 0) {
            const LevelDev& S = P.lv[l - 1];
            const double sx_scale = 1. / ((double)L.w / S.w), sy_scale = 1. / ((double)L.h / S.h);
            std::vector<short4> tx(L.pitch), ty(L.h);
            for (int dx = 0; dx < L.pitch; dx++) {
                if (dx >= L.w) { tx[dx] = make_short4(0, 0, 0, 0); continue; }
                float fx = (float)((dx + 0.5) * sx_scale - 0.5);
                int sx = (int)std::floor(fx);
                fx -= sx;
                if (sx < 0) { fx = 0; sx = 0; }
                if (sx >= S.w - 1) { fx = 0; sx = S.w - 1; }
                tx[dx] = make_short4((short)sx, (short)std::min(sx + 1, S.w - 1), sat_short((1.f - fx) * 2048), sat_short(fx * 2048));
            }
            for (int dy = 0; dy < L.h; dy++) {
                float fy = (float)((dy + 0.5) * sy_scale - 0.5);
                int sy = (int)std::floor(fy);
                fy -= sy;
                ty[dy] = make_short4((short)std::min(std::max(sy, 0), S.h - 1), (short)std::min(std::max(sy + 1, 0), S.h - 1),
                                     sat_short((1.f - fy) * 2048), sat_short(fy * 2048));
            }
            tabs.insert(tabs.end(), tx.begin(), tx.end());
            tabs.insert(tabs.end(), ty.begin(), ty.end());
            rs_off += L.pitch + L.h;
        }
        // blur tiles
        for (int ty_ = 0; ty_ < (L.h + 15) / 16; ty_++)
            for (int tx_ = 0; tx_ < (L.w + 63) / 64; tx_++) o->tiles.push_back(TileDev{(short)l, (short)tx_, (short)ty_, 0});
    }
    P.ncells_total = (int)o->cells.size();
    P.pyr_stride = pyr_off;
    P.cand_stride = cand_off;
    P.kept_stride = kept_off;
    P.kp_cap = kept_off;
    o->fast_smem = max_tile_bytes;
    o->node_cap = std::max(max_nfeat + 8, 4 * 64 + 8);
    o->oct_smem = 13 * o->node_cap * (int)sizeof(int);
    if (o->oct_smem > 160 * 1024 || o->fast_smem > 64 * 1024) {
        delete o; set_error("planar_orb_create: LDS budget exceeded (octree %d B, FAST tile %d B)", o->oct_smem, o->fast_smem);
        return PLANAR_EINVAL;
    }

    int rc = PLANAR_OK;
    const size_t B = (size_t)max_batch;
    if ((rc = o->d_plan.alloc(sizeof(PlanDev))) || (rc = o->d_cells.alloc(o->cells.size() * sizeof(CellDev))) ||
        (rc = o->d_tiles.alloc(o->tiles.size() * sizeof(TileDev))) || (rc = o->d_tabs.alloc(std::max<size_t>(tabs.size(), 1) * sizeof(short4))) ||
        (rc = o->d_pyr.alloc(B * P.pyr_stride)) || (rc = o->d_blur.alloc(B * P.pyr_stride)) ||
        (rc = o->d_cand.alloc(B * P.cand_stride * sizeof(uint32_t))) || (rc = o->d_cell_count.alloc(B * P.ncells_total * sizeof(int))) ||
        (rc = o->d_sortA.alloc(B * P.cand_stride * sizeof(uint64_t))) || (rc = o->d_sortB.alloc(B * P.cand_stride * sizeof(uint64_t))) ||
        (rc = o->d_level_count.alloc(B * MAX_LEVELS * sizeof(int))) || (rc = o->d_kept.alloc(B * P.kept_stride * sizeof(uint32_t))) ||
        (rc = o->d_kept_count.alloc(B * MAX_LEVELS * sizeof(int))) || (rc = o->d_dropped.alloc(64))) {
        delete o;
        return rc;
    }
    hipError_t e = hipMemcpy(o->d_plan.p, &P, sizeof(P), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(o->d_dropped.p, 0, 64);
    if (e == hipSuccess) e = hipMemcpy(o->d_cells.p, o->cells.data(), o->cells.size() * sizeof(CellDev), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(o->d_tiles.p, o->tiles.data(), o->tiles.size() * sizeof(TileDev), hipMemcpyHostToDevice);
    if (e == hipSuccess && !tabs.empty()) e = hipMemcpy(o->d_tabs.p, tabs.data(), tabs.size() * sizeof(short4), hipMemcpyHostToDevice);
    if (e == hipSuccess && o->oct_smem > 64 * 1024)
        e = hipFuncSetAttribute((const void*)orb_octree, hipFuncAttributeMaxDynamicSharedMemorySize, o->oct_smem);
    if (e != hipSuccess) { delete o; set_error("planar_orb_create: upload failed: %s", hipGetErrorString(e)); return PLANAR_EDEVICE; }
    o->launch_names.push_back("orb_copy_level0");
    for (int l = 1; l < nl; l++) o->launch_names.push_back("orb_resize");
    for (const char* n : {"orb_fast_cells", "orb_sort", "orb_octree", "orb_blur", "orb_describe"}) o->launch_names.push_back(n);
    *out = o;
    return PLANAR_OK;
}

void planar_orb_destroy(planar_orb* o) { delete o; }

int planar_orb_set_profiling(planar_orb* o, int enable) {
    PLANAR_REQUIRE(o != nullptr, PLANAR_EINVAL, "orb is null");
    PLANAR_HIP_CHECK(hipStreamSynchronize(o->ctx->stream));
    o->profiling = enable != 0;
    o->ev_used = 0;
    return PLANAR_OK;
}

int planar_orb_profile_num_launches(const planar_orb* o) { return o ? (int)o->launch_names.size() : PLANAR_EINVAL; }

const char* planar_orb_profile_launch_name(const planar_orb* o, int i) {
    return (o && i >= 0 && i < (int)o->launch_names.size()) ? o->launch_names[i] : nullptr;
}

int planar_orb_get_profile(planar_orb* o, double* total_ms, int64_t* calls) {
    PLANAR_REQUIRE(o && total_ms && calls, PLANAR_EINVAL, "null argument");
    PLANAR_HIP_CHECK(hipStreamSynchronize(o->ctx->stream));
    const size_t nl = o->launch_names.size();
    for (size_t i = 0; i < nl; i++) total_ms[i] = 0;
    for (size_t c = 0; c < o->ev_used; c++)
        for (size_t i = 0; i < nl; i++) {
            float ms = 0;
            PLANAR_HIP_CHECK(hipEventElapsedTime(&ms, o->ev_sets[c][i], o->ev_sets[c][i + 1]));
            total_ms[i] += ms;
        }
    *calls = (int64_t)o->ev_used;
    o->ev_used = 0;
    return PLANAR_OK;
}

int planar_orb_max_keypoints(const planar_orb* o) { return o ? o->plan.kp_cap : PLANAR_EINVAL; }

int planar_orb_get_scale_factors(const planar_orb* o, float* s, float* is, float* s2, float* is2) {
    PLANAR_REQUIRE(o != nullptr, PLANAR_EINVAL, "orb is null");
    const int n = o->params.nlevels;
    if (s) memcpy(s, o->scale.data(), n * sizeof(float));
    if (is) memcpy(is, o->inv_scale.data(), n * sizeof(float));
    if (s2) memcpy(s2, o->sigma2.data(), n * sizeof(float));
    if (is2) memcpy(is2, o->inv_sigma2.data(), n * sizeof(float));
    return PLANAR_OK;
}

int planar_orb_level_size(const planar_orb* o, int level, int* w, int* h) {
    PLANAR_REQUIRE(o && w && h, PLANAR_EINVAL, "null argument");
    PLANAR_REQUIRE(level >= 0 && level < o->params.nlevels, PLANAR_EINVAL, "level out of range");
    *w = o->plan.lv[level].w; *h = o->plan.lv[level].h;
    return PLANAR_OK;
}

int planar_orb_features_per_level(const planar_orb* o, int32_t* out) {
    PLANAR_REQUIRE(o && out, PLANAR_EINVAL, "null argument");
    for (int l = 0; l < o->params.nlevels; l++) out[l] = o->plan.lv[l].nfeat;
    return PLANAR_OK;
}

int planar_orb_extract_dev(planar_orb* o, const uint8_t* d_gray, int B, int pitch, int64_t frame_stride,
                           planar_keypoint* d_kps, uint8_t* d_desc, int32_t* d_n_out) {
    PLANAR_REQUIRE(o && d_gray && d_kps && d_desc && d_n_out, PLANAR_EINVAL, "null argument");
    PLANAR_REQUIRE(B >= 1 && B <= o->max_batch, PLANAR_EINVAL, "B must be in [1, max_batch]");
    PLANAR_REQUIRE(pitch >= o->W && frame_stride >= (int64_t)pitch * o->H, PLANAR_EINVAL, "pitch/frame_stride too small");
    hipStream_t st = o->ctx->stream;
    const PlanDev& P = o->plan;
    const PlanDev* dp = o->d_plan.as<PlanDev>();
    uint8_t* pyr = o->d_pyr.as<uint8_t>();
    std::vector<hipEvent_t>* evs = nullptr;
    if (o->profiling) {
        const size_t need = (size_t)P.nlevels + 5 + 1;
        if (o->ev_used == o->ev_sets.size()) {
            std::vector<hipEvent_t> v(need);
            for (size_t i = 0; i < need; i++) PLANAR_HIP_CHECK(hipEventCreate(&v[i]));
            o->ev_sets.push_back(v);
        }
        evs = &o->ev_sets[o->ev_used++];
    }
    int li = 0;
    auto mark = [&]() { if (evs) (void)hipEventRecord((*evs)[li], st); li++; };
    mark();
    {
        const int n = (P.lv[0].pitch / 4) * P.lv[0].h;
        hipLaunchKernelGGL(orb_copy_level0, dim3((n + 255) / 256, B), dim3(256), 0, st, dp, d_gray, pitch, frame_stride, pyr);
        mark();
    }
    for (int l = 1; l < P.nlevels; l++) {
        const int n = (P.lv[l].pitch / 4) * P.lv[l].h;
        hipLaunchKernelGGL(orb_resize, dim3((n + 255) / 256, B), dim3(256), 0, st, dp, o->d_tabs.as<short4>(), pyr, l);
        mark();
    }
    hipLaunchKernelGGL(orb_fast_cells, dim3(P.ncells_total * B), dim3(256), o->fast_smem, st, dp, o->d_cells.as<CellDev>(), pyr,
                       o->d_cand.as<uint32_t>(), o->d_cell_count.as<int>(), o->d_dropped.as<int>(), B);
    mark();
    hipLaunchKernelGGL(orb_sort, dim3(P.nlevels, B), dim3(256), 0, st, dp, o->d_cells.as<CellDev>(), o->d_cand.as<uint32_t>(),
                       o->d_cell_count.as<int>(), o->d_sortA.as<uint64_t>(), o->d_sortB.as<uint64_t>(), o->d_level_count.as<int>());
    mark();
    hipLaunchKernelGGL(orb_octree, dim3(P.nlevels, B), dim3(64), o->oct_smem, st, dp, o->d_sortA.as<uint64_t>(), o->d_sortB.as<uint64_t>(),
                       o->d_level_count.as<int>(), o->d_kept.as<uint32_t>(), o->d_kept_count.as<int>(), o->node_cap);
    mark();
    hipLaunchKernelGGL(orb_blur, dim3((unsigned)o->tiles.size() * B), dim3(256), 0, st, dp, o->d_tiles.as<TileDev>(), pyr, o->d_blur.as<uint8_t>(), (int)o->tiles.size(), B);
    mark();
    hipLaunchKernelGGL(orb_describe, dim3((P.kp_cap + 15) / 16 * B), dim3(256), 0, st, dp, pyr, o->d_blur.as<uint8_t>(), o->d_kept.as<uint32_t>(),
                       o->d_kept_count.as<int>(), d_kps, d_desc, d_n_out, (P.kp_cap + 15) / 16, B);
    mark();
    PLANAR_HIP_CHECK(hipGetLastError());
    o->last_B = B;
    return PLANAR_OK;
}

int planar_orb_check(planar_orb* o) {
    PLANAR_REQUIRE(o != nullptr, PLANAR_EINVAL, "orb is null");
    int dropped = 0;
    PLANAR_HIP_CHECK(hipMemcpyAsync(&dropped, o->d_dropped.p, 4, hipMemcpyDeviceToHost, o->ctx->stream));
    PLANAR_HIP_CHECK(hipStreamSynchronize(o->ctx->stream));
    if (dropped != 0) {
        (void)hipMemsetAsync(o->d_dropped.p, 0, 4, o->ctx->stream);
        set_error("planar_orb: %d FAST candidates did not fit their cell's slot array", dropped);
        return PLANAR_ECAPACITY;
    }
    return PLANAR_OK;
}

int planar_orb_extract(planar_orb* o, const uint8_t* gray, int B, int pitch, int64_t frame_stride, planar_keypoint* kps,
                       uint8_t* desc, int32_t* n_out) {
    PLANAR_REQUIRE(o && gray && kps && desc && n_out, PLANAR_EINVAL, "null argument");
    PLANAR_REQUIRE(B >= 1 && B <= o->max_batch, PLANAR_EINVAL, "B must be in [1, max_batch]");
    PLANAR_REQUIRE(pitch >= o->W && frame_stride >= (int64_t)pitch * o->H, PLANAR_EINVAL, "pitch/frame_stride too small");
    PLANAR_HIP_CHECK(hipSetDevice(o->ctx->device));
    const size_t in_bytes = (size_t)frame_stride * (B - 1) + (size_t)pitch * o->H;
    const size_t cap = (size_t)o->plan.kp_cap;
    int rc;
    if (o->d_in.bytes < in_bytes && (rc = o->d_in.alloc(in_bytes))) return rc;
    if (o->d_kps.bytes < (size_t)o->max_batch * cap * sizeof(planar_keypoint)) {
        if ((rc = o->d_kps.alloc((size_t)o->max_batch * cap * sizeof(planar_keypoint)))) return rc;
        if ((rc = o->d_desc.alloc((size_t)o->max_batch * cap * 32))) return rc;
        if ((rc = o->d_nout.alloc((size_t)o->max_batch * sizeof(int32_t)))) return rc;
    }
    hipStream_t st = o->ctx->stream;
    PLANAR_HIP_CHECK(hipMemcpyAsync(o->d_in.p, gray, in_bytes, hipMemcpyHostToDevice, st));
    if ((rc = planar_orb_extract_dev(o, o->d_in.as<uint8_t>(), B, pitch, frame_stride, o->d_kps.as<planar_keypoint>(),
                                     o->d_desc.as<uint8_t>(), o->d_nout.as<int32_t>())))
        return rc;
    PLANAR_HIP_CHECK(hipMemcpyAsync(kps, o->d_kps.p, (size_t)B * cap * sizeof(planar_keypoint), hipMemcpyDeviceToHost, st));
    PLANAR_HIP_CHECK(hipMemcpyAsync(desc, o->d_desc.p, (size_t)B * cap * 32, hipMemcpyDeviceToHost, st));
    PLANAR_HIP_CHECK(hipMemcpyAsync(n_out, o->d_nout.p, (size_t)B * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    return planar_orb_check(o);
}

static int read_plane(planar_orb* o, const DevBuf& buf, int frame, int level, uint8_t* out) {
    PLANAR_REQUIRE(o && out, PLANAR_EINVAL, "null argument");
    PLANAR_REQUIRE(frame >= 0 && frame < o->last_B, PLANAR_ESTATE, "frame not part of the last extract call");
    PLANAR_REQUIRE(level >= 0 && level < o->params.nlevels, PLANAR_EINVAL, "level out of range");
    const LevelDev& L = o->plan.lv[level];
    PLANAR_HIP_CHECK(hipStreamSynchronize(o->ctx->stream));
    PLANAR_HIP_CHECK(hipMemcpy2D(out, L.w, buf.as<uint8_t>() + (int64_t)frame * o->plan.pyr_stride + L.off, L.pitch, L.w, L.h,
                                 hipMemcpyDeviceToHost));
    return PLANAR_OK;
}

int planar_orb_read_level(planar_orb* o, int frame, int level, uint8_t* out) { return read_plane(o, o->d_pyr, frame, level, out); }
int planar_orb_read_blurred(planar_orb* o, int frame, int level, uint8_t* out) { return read_plane(o, o->d_blur, frame, level, out); }

int planar_orb_read_candidates(planar_orb* o, int frame, int level, int32_t* out, int cap) {
    PLANAR_REQUIRE(o && out, PLANAR_EINVAL, "null argument");
    PLANAR_REQUIRE(frame >= 0 && frame < o->last_B, PLANAR_ESTATE, "frame not part of the last extract call");
    PLANAR_REQUIRE(level >= 0 && level < o->params.nlevels, PLANAR_EINVAL, "level out of range");
    const LevelDev& L = o->plan.lv[level];
    PLANAR_HIP_CHECK(hipStreamSynchronize(o->ctx->stream));
    std::vector<int> counts(L.ncells);
    std::vector<uint32_t> slots(L.cand_cap);
    PLANAR_HIP_CHECK(hipMemcpy(counts.data(), o->d_cell_count.as<int>() + (int64_t)frame * o->plan.ncells_total + L.cell_begin,
                               L.ncells * sizeof(int), hipMemcpyDeviceToHost));
    PLANAR_HIP_CHECK(hipMemcpy(slots.data(), o->d_cand.as<uint32_t>() + (int64_t)frame * o->plan.cand_stride + L.cand_off,
                               (size_t)L.cand_cap * sizeof(uint32_t), hipMemcpyDeviceToHost));
    int n = 0;
    for (int c = 0; c < L.ncells; c++) {
        const CellDev& C = o->cells[L.cell_begin + c];
        for (int i = 0; i < counts[c]; i++, n++) {
            if (n >= cap) { set_error("planar_orb_read_candidates: capacity %d too small", cap); return PLANAR_ECAPACITY; }
            const uint32_t k = slots[C.slot_off - L.cand_off + i];
            out[3 * n] = (int)(k & 0xfffu); out[3 * n + 1] = (int)((k >> 12) & 0xfffu); out[3 * n + 2] = (int)(k >> 24);
        }
    }
    return n;
}

}  // extern "C"
