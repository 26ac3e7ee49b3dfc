// planarslam_amd/csrc/isort.h — the arrangement libstdc++'s std::sort leaves, computed in parallel (gfx950, wave64).
//
// Two places of the reference path sort with std::sort and a comparator that looks at a KEY only, and then depend on where the unstable
// introsort left elements of equal key:
//   * pcl::VoxelGrid::applyFilter (src/Frame.cc:674-679 -> filters/impl/voxel_grid.hpp) sorts a plane's points by voxel index and sums each
//     voxel's points as floats IN THAT ORDER: the centroid's last bits depend on it;
//   * OpenCV's lsd.cpp (src/LSDextractor.cpp:14-16 -> LSDDetector -> ll_angle) sorts the gradient pixels by a 1024-bin norm; pixels of one bin
//     are visited in that order, which decides which seed grows first.
// std::sort(first, last, comp) of libstdc++ = __introsort_loop (median-of-three Hoare partitions down to 16 elements, depth limit 2 lg n)
// + __final_insertion_sort (a stable sort of what the partitions left).  Both are deterministic functions of the key sequence:
//   * one partition of [f, l): pivot = median of (f+1, mid, l-1) moved to f.  With L_0 < L_1 < ... the positions in (f, l) whose key is
//     >= pivot (where the left scan stops) and R_0 > R_1 > ... those whose key is <= pivot (right scan), the partition swaps L_k with R_k for
//     k < m = #{k : L_k < R_k}.  With A(x) = #L in (f, x) and B(x) = #R in [x, l), and x* the first x with A(x) >= B(x):
//     m = max(A(x*-1), B(x*)), the cut is x* - (key(x*-1) == pivot && A(x*-1) == B(x*-1) - 1), the L-stop at p is swapped iff
//     #R in (p, l) >= A(p) + 1 and the R-stop at q iff A(q) >= B(q).  All of these are prefix counts: no scan is sequential.
//   * the recursion visits disjoint ranges, so a whole level of the recursion tree is partitioned at once.
// Words are 32 bits, key = word >> SHIFT (ascending); the low bits are payload and never compared.
//
//   global_tier<SHIFT, T>   one workgroup per array set (frame): ranges longer than the LDS tier's capacity are partitioned in global memory.
//                           Stops are BITMAPS in LDS (one ballot per 64 elements), ranks are popcount prefix sums, the k-th stop is a binary
//                           search + an in-word select: the array itself is read once and only swapped elements are written.
//   lds_tier<SHIFT, T, E>   one workgroup per block of <= T * E elements (any number of ranges): staged in LDS once.  A level of the recursion is
//                           medians | flags + two segmented scans | cuts | left stops | swaps | new list, whatever the number of segments: the
//                           whole workgroup runs the levels above 2048 elements, single wavefronts everything below; the leaves of <= 16
//                           elements are ranked (stable) in the write-back, so the block goes back SORTED, ties in std::sort's order.
// A range whose depth budget runs out is heap-sorted as libstdc++ does it: the tiers record it, heap_jobs runs it afterwards (level-parallel __make_heap,
// software-pipelined __sort_heap; see "The heap-sort fallback" below).  That is not a corner case for voxel keys: the saw-tooth key sequence of a plane
// in raster order makes the median-of-three partitions degenerate for about one plane in fifty.
// tests/host_shim/isort_host.cpp compiles this file with g++ on the wave64 emulator and checks it against the real std::sort.
#pragma once
#include <stdint.h>
#ifndef PLANAR_WAVE_EMUL
#include "wave_ops.h"
#endif

namespace planar {
namespace isort {

struct Range { int f, l, d; };          // [f, l) of the array, d = depth budget left (2 * lg n at the top)
struct Block { int f, l, r0, nr; };     // LDS-tier job: the span [f, l) holds ranges r0 .. r0 + nr - 1 of the sorted range list

constexpr int ST_CAPACITY = 3;
#ifdef PLANAR_WAVE_EMUL
static long g_levels = 0, g_segs = 0;      // emulator statistics: LDS-tier levels run, segments partitioned
static long g_wlev = 0, g_wlanes = 0, g_glev = 0, g_glanes = 0, g_eqseg = 0, g_eqelem = 0;      // wavefront-scope passes and the lanes that had elements in them, workgroup-scope passes; all-equal segments finished in closed form
#endif

__device__ __forceinline__ int lg2i(int n) { return 31 - __clz(n); }          // std::__lg
__device__ __forceinline__ int depth_limit(int n) { return n > 1 ? 2 * lg2i(n) : 0; }

// exclusive prefix sum over the workgroup's T threads; s_w: [T / 64] scratch; two barriers
template <int T, typename V>
__device__ __forceinline__ V block_exscan(V v, V* s_w, V* total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    V inc = v;
    for (int o = 1; o < 64; o <<= 1) { const V t = __shfl_up(inc, o); if (lane >= o) inc += t; }
    if (lane == 63) s_w[w] = inc;
    __syncthreads();
    V base = 0, tot = 0;
    for (int i = 0; i < T / 64; i++) { const V x = s_w[i]; if (i < w) base += x; tot += x; }
    *total = tot;
    __syncthreads();
    return base + inc - v;
}

#ifndef PLANAR_WAVE_EMUL
// inclusive segmented prefix sum over the 64 lanes on DPP row operations (the Kogge-Stone ladder of wave_scan_add with the pair operator
// (v, g) <- (g ? v : v + v', g | g')): no LDS-pipe permute, no wait
__device__ __forceinline__ void dpp_seg_scan(int& v, int& g) {
#define ISORT_DPP_STEP(ctrl, rmask)                                                                                                    \
    {                                                                                                                                  \
        const int tv = __builtin_amdgcn_update_dpp(0, v, ctrl, rmask, 0xf, false), tg = __builtin_amdgcn_update_dpp(0, g, ctrl, rmask, 0xf, false); \
        v = g ? v : v + tv; g |= tg;                                                                                                   \
    }
    ISORT_DPP_STEP(0x111, 0xf) ISORT_DPP_STEP(0x112, 0xf) ISORT_DPP_STEP(0x114, 0xf) ISORT_DPP_STEP(0x118, 0xf) ISORT_DPP_STEP(0x142, 0xa) ISORT_DPP_STEP(0x143, 0xc)
#undef ISORT_DPP_STEP
}
#endif
// Two segmented scans over one wavefront at once.  Forward: (vL, resetL) -> what the lanes before this one accumulated since the last reset
// (exclusive; eL / egL: value and whether a reset lies in between).  Backward: the same from the other end.  incL / incR: the inclusive values.
__device__ __forceinline__ void wave_seg_scan2(int vL, int rL, int vR, int rR, int& eL, int& egL, int& eR, int& egR, int& incL, int& incgL, int& incR, int& incgR) {
    const int lane = threadIdx.x & 63;
#if defined(PLANAR_WAVE_EMUL) || defined(ISORT_NO_DPP)
    int aL = vL, gL = rL, aR = vR, gR = rR;
    for (int o = 1; o < 64; o <<= 1) {
        const int tv = __shfl_up(aL, o), tf = __shfl_up(gL, o), uv = __shfl_down(aR, o), uf = __shfl_down(gR, o);
        if (lane >= o) { if (!gL) aL += tv; gL |= tf; }
        if (lane + o < 64) { if (!gR) aR += uv; gR |= uf; }
    }
    eL = __shfl_up(aL, 1); egL = __shfl_up(gL, 1); eR = __shfl_down(aR, 1); egR = __shfl_down(gR, 1);
    if (lane == 0) { eL = 0; egL = 0; }
    if (lane == 63) { eR = 0; egR = 0; }
    incL = aL; incgL = gL; incR = aR; incgR = gR;
#else
    // forward on DPP; the backward scan is the forward one on the lane-reversed input (one permute each way)
    int aL = vL, gL = rL;
    dpp_seg_scan(aL, gL);
    eL = __builtin_amdgcn_update_dpp(0, aL, 0x138, 0xf, 0xf, false);                // wave_shr:1: the lane before (lane 0: nothing)
    egL = __builtin_amdgcn_update_dpp(0, gL, 0x138, 0xf, 0xf, false);
    const int packed = __shfl(vR | (rR << 24), 63 - lane);
    int aR = packed & 0xffffff, gR = packed >> 24;
    dpp_seg_scan(aR, gR);
    const int ex = __builtin_amdgcn_update_dpp(0, aR | (gR << 24), 0x138, 0xf, 0xf, false);
    const int back = __shfl(ex, 63 - lane), binc = __shfl(aR | (gR << 24), 63 - lane);
    eR = back & 0xffffff; egR = back >> 24;
    incR = binc & 0xffffff; incgR = binc >> 24;
    incL = aL; incgL = gL;
#endif
}

// Two segmented scans over the workgroup's threads at once (see wave_seg_scan2).  s_buf: [2][4 * T / 64] ints, the half alternates with `parity` so that
// one barrier is enough.
template <int T>
__device__ __forceinline__ void seg_scan2(int vL, int resetL, int vR, int resetR, int* s_buf, int parity, int& carryL, int& carryR) {
    constexpr int NW = T / 64;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int eL, egL, eR, egR, aL, gL, aR, gR;
    wave_seg_scan2(vL, resetL, vR, resetR, eL, egL, eR, egR, aL, gL, aR, gR);
    int* sb = s_buf + (parity & 1) * 4 * NW;
    if (lane == 63) { sb[w] = aL; sb[NW + w] = gL; }
    if (lane == 0) { sb[2 * NW + w] = aR; sb[3 * NW + w] = gR; }
    __syncthreads();
    int wl = 0, wr = 0;
    for (int q = 0; q < NW; q++) {
        const int v = sb[q], g = sb[NW + q];
        if (q < w) wl = g ? v : wl + v;
    }
    for (int q = NW - 1; q >= 0; q--) {
        const int v = sb[2 * NW + q], g = sb[3 * NW + q];
        if (q > w) wr = g ? v : wr + v;
    }
    carryL = lane == 0 ? wl : (egL ? eL : eL + wl);
    carryR = lane == 63 ? wr : (egR ? eR : eR + wr);
}

// __move_median_to_first(f, f + 1, mid, l - 1) with comp = key <; returns the position whose element goes to the front
template <int SHIFT>
__device__ __forceinline__ int median_pos(uint32_t xa, uint32_t xb, uint32_t xc, int A, int Bm, int Cc) {
    const uint32_t a = xa >> SHIFT, b = xb >> SHIFT, c = xc >> SHIFT;
    if (a < b) { if (b < c) return Bm; if (a < c) return Cc; return A; }
    if (a < c) return A;
    if (b < c) return Cc;
    return Bm;
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// The heap-sort fallback
// ---------------------------------------------------------------------------------------------------------------------------------------
// std::__partial_sort(first, last, last) = __make_heap + __sort_heap (bits/stl_heap.h) is what __introsort_loop does with a range of more than 16
// elements whose depth budget is used up.  That is not a corner case here: the saw-tooth voxel keys of a plane in raster order make the
// median-of-three partitions peel off a few percent per level, and about one plane in fifty ends with ranges of up to tens of thousands of elements
// in this branch.  Every pop depends on the heap the previous one left, but not on all of it:
//   * __adjust_heap(hole, value) sifts the hole down to a leaf along the larger children (the right one on ties) and __push_heap carries the value back
//     up while the parent is smaller.  Keys along that path do not increase downwards, so the value ends below the last path element that is >= it:
//     the same final state as walking DOWN and stopping at the first chosen child that is smaller than the value (nothing below is touched);
//   * __make_heap's sift-downs of the nodes of one depth touch disjoint subtrees: a whole depth at a time, one lane per node;
//   * in __sort_heap pop t + 1 reads at heap level j what pop t wrote at level j + 1: pops run two levels apart, HP of them in flight on HP lanes of
//     one wavefront.  A pop starts (it takes the last heap element as its value and puts the root there) only when no earlier pop in flight can
//     still reach that element, i.e. none of their holes is one of its ancestors.  (HP >= half the heap's depth: 8 lanes for 16 levels.)
// Jobs (ranges) are recorded by the two tiers and run by heap_jobs afterwards; a range's first H_LDS elements (the top of the heap) live in LDS.
struct HeapJob { int f, l; };
struct HeapSink { HeapJob* jobs; int* n; int cap; };      // where the tiers record the ranges that need the fallback (global memory; n: atomic counter)
__device__ __forceinline__ void push_heap_job(const HeapSink& H, int f, int l, int* status) {
    const int k = atomicAdd(H.n, 1);
    if (k < H.cap) H.jobs[k] = HeapJob{f, l}; else *status = ST_CAPACITY;
}
constexpr int HP = 8;                  // pops in flight (heap depth <= 16 levels, two levels apart)

template <bool HYBRID>
struct HeapMem {                       // element i of the range: LDS below cap, the array itself (global memory) above (HYBRID), or all of it in LDS
    uint32_t* lds; uint32_t* glb; int cap;
    __device__ __forceinline__ uint32_t ld(int i) const { if (HYBRID) return i < cap ? lds[i] : glb[i]; return lds[i]; }
    __device__ __forceinline__ void st(int i, uint32_t x) const { if (HYBRID) { if (i < cap) lds[i] = x; else glb[i] = x; } else lds[i] = x; }
    __device__ __forceinline__ void sync() const {   // one wavefront: LDS accesses execute in order; its global stores have to land before another lane's load
        if (HYBRID) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); else __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
};
__device__ __forceinline__ int heap_level(int i) { return 31 - __clz(i + 1); }

// one step of a top-down sift in a heap of `len` elements: the hole h takes its larger child (the right one on ties) if that child is not smaller than the
// value (-> true, h moves down), else the value (-> false: finished).  Branch-free: a lone wavefront pays ~8 cycles per instruction, whatever it is.
template <int SHIFT, bool HYBRID>
__device__ __forceinline__ bool sift_step(const HeapMem<HYBRID>& M, int& h, int len, uint32_t v) {
    const int c = 2 * h + 2, last = len - 1;
    const uint32_t r = M.ld(min(c, max(last, 0))), l = M.ld(min(c - 1, max(last, 0)));
    const bool has_l = c - 1 < len, take_r = c < len && !((r >> SHIFT) < (l >> SHIFT));
    const uint32_t pick = take_r ? r : l;
    const bool cont = has_l && !((pick >> SHIFT) < (v >> SHIFT));
    M.st(h, cont ? pick : v);
    h = cont ? (take_r ? c : c - 1) : h;
    return cont;
}

// heap-sorts the k elements behind M (one wavefront calls it)
template <int SHIFT, bool HYBRID>
__device__ __forceinline__ void heap_sort_wave(const HeapMem<HYBRID>& M, int k) {
    const int lane = threadIdx.x & 63;
    if (k < 2) return;
    // __make_heap: depth by depth from the last parent's, one lane per node
    for (int dd = heap_level((k - 2) / 2); dd >= 0; dd--) {
        const int first = (1 << dd) - 1, last = min((1 << (dd + 1)) - 2, (k - 2) / 2);
        for (int i0 = first; i0 <= last; i0 += 64) {
            int h = min(i0 + lane, last);
            const uint32_t v = M.ld(h);
            bool go = i0 + lane <= last;
            while (__ballot(go) != 0ull) { if (go) go = sift_step<SHIFT, HYBRID>(M, h, k, v); }
        }
        M.sync();
    }
    // __sort_heap: pop t (t = 0 .. k - 2) takes the last element z = k - 1 - t of the heap as its value, puts the root there and sifts in the heap of z
    // elements; lane t % HP runs it, TWO levels per step (two half-steps with a fence between them), and a new pop starts every step: pop t + 1 reads level
    // j + 1 in the half-step after pop t wrote it, so by construction nothing else has to be checked from step to step.
    bool on = false;                       // this lane has a pop in flight
    int h = 0, lv = 0, len = 0;            // its hole, the hole's level, its heap size
    uint32_t v = 0;
    int next = 0;                          // the next pop to start (uniform)
    while (true) {
        if (next <= k - 2) {
            // ... unless a pop in flight can still reach z (its hole is z or one of z's ancestors: it may yet write there), or the lane is still busy
            const int z = k - 1 - next, zl = heap_level(z), owner = next % HP;
            const bool blocks = on && (lane == owner || (lv <= zl && ((z + 1) >> (zl - min(lv, zl))) == h + 1));
            if (__ballot(blocks) == 0ull) {
                const uint32_t vz = M.ld(z), root = M.ld(0);
                if (lane == owner) { on = true; h = 0; lv = 0; len = z; v = vz; M.st(z, root); }
                M.sync();
                next++;
            }
        }
        if (on) { on = sift_step<SHIFT, HYBRID>(M, h, len, v); lv++; }
        M.sync();
        if (on) { on = sift_step<SHIFT, HYBRID>(M, h, len, v); lv++; }
        M.sync();
        if (next > k - 2 && __ballot(on) == 0ull) break;
    }
}

// The jobs first, first + stride, ... of a job list whose ranges hold min_len .. max_len elements, by one wavefront; lds: cap 32-bit words of this
// wavefront.  A range longer than cap keeps its first cap elements (the top of the heap) in LDS and the rest in place.
template <int SHIFT>
__device__ void heap_jobs(uint32_t* __restrict__ arr, const HeapJob* __restrict__ jobs, int njobs, int first, int stride, uint32_t* lds, int cap, int min_len, int max_len) {
    const int lane = threadIdx.x & 63;
    for (int j = first; j < njobs; j += stride) {
        const HeapJob J = jobs[j];
        const int k = J.l - J.f, in_lds = min(k, cap);
        if (k < min_len || k > max_len) continue;
        for (int i = lane; i < in_lds; i += 64) lds[i] = arr[J.f + i];
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); __builtin_amdgcn_wave_barrier();
        if (k <= cap) { const HeapMem<false> M{lds, arr + J.f, cap}; heap_sort_wave<SHIFT, false>(M, k); }
        else { const HeapMem<true> M{lds, arr + J.f, cap}; heap_sort_wave<SHIFT, true>(M, k); }
        for (int i = lane; i < in_lds; i += 64) arr[J.f + i] = lds[i];
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); __builtin_amdgcn_wave_barrier();
    }
}

// position of the k-th (0-based) set bit of w (which has more than k set bits)
__device__ __forceinline__ int select64(unsigned long long w, int k) {
    int pos = 0;
    uint32_t lo = (uint32_t)w;
    const int c = __popc(lo);
    if (k >= c) { k -= c; pos = 32; lo = (uint32_t)(w >> 32); }
#pragma unroll
    for (int sh = 16; sh > 0; sh >>= 1) {
        const int cc = __popc(lo & ((1u << sh) - 1u));
        if (k >= cc) { k -= cc; lo >>= sh; pos += sh; }
    }
    return pos;
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// LDS tier
// ---------------------------------------------------------------------------------------------------------------------------------------
// The level loop below (sort_levels) is written once and run at two scopes: by the whole workgroup on the segments of more than W_CAP elements
// (a level costs a handful of workgroup barriers whatever it holds, so only the few top levels of a block run there), and by single wavefronts on
// segments of <= W_CAP = 1984 elements, which they take from a task list and finish on their own - no workgroup barrier, sixteen of them side by side.
// Nothing is insertion-sorted on the way: every cut sets a bit, and the block's last pass ranks every element inside its <= 16-element leaf
// (stable: what __final_insertion_sort would do) while it writes the block back.
//
// Segments whose keys are ALL EQUAL (a voxel's points once the partitions have isolated it: where every chain of the recursion ends) are not partitioned at all.
// With equal keys every comparison of __move_median_to_first is false (it swaps first with mid), both scans of __unguarded_partition stop at once (it swaps k with
// n - k for k = 1 .. while k < n - k and returns ceil(n / 2)), and both children are all-equal again: where an element ends is a function of its position and n
// alone (equal_dest below, <= 11 steps of integer arithmetic).  A wavefront-scope pass recognises such a segment (every element stops both scans: the two segmented
// scans carry both stop counts), puts its pivot back, flags its start (eb) and drops it from the list; the write-back sends its elements straight to their places.
// On a plane's voxel keys that takes the last three to six levels off every chain.
// levels of __introsort_loop an all-equal range of n elements goes through (each costs one unit of the depth budget; it halves)
__device__ __forceinline__ int equal_levels(int n) { int k = 0; while (n > 16) { n = (n + 1) >> 1; k++; } return k; }
// where the element at position p of an all-equal range of n elements is when std::sort returns (the depth budget must cover equal_levels(n): no heap sort on the way;
// __final_insertion_sort moves nothing: it is stable).  One level: swap(0, n / 2); positions 1 .. n - 1 reversed (k <-> n - k); cut at ceil(n / 2).
__device__ __forceinline__ int equal_dest(int p, int n) {
    int base = 0;
    while (n > 16) {
        const int mid = n >> 1, c = (n + 1) >> 1;
        p = p == mid ? 0 : (p == 0 ? n - mid : n - p);
        const bool right = p >= c;
        base += right ? c : 0; p -= right ? c : 0; n = right ? n - c : c;
    }
    return base + p;
}
constexpr int EQ_CUT = 0xffff;            // scut[] value of a segment recognised as all-equal (a real cut is < 2^16 - 1: blocks hold fewer elements)

constexpr int W_E = 31, W_CAP = 64 * W_E, W_LIST = 128, G_LIST = 32, TASKS = 256;      // (an odd stride: lane l's chunk starts at bank 31 l mod 32)

template <int T>
struct WgScope {
    static constexpr int NT = T;
    int* s_buf;                                               // [2 * T / 64] (exscan, as 64-bit slots elsewhere) + [2][4 * T / 64] (seg_scan)
    __device__ __forceinline__ int tid() const { return threadIdx.x; }
    __device__ __forceinline__ void sync() const { __syncthreads(); }
    __device__ __forceinline__ int exscan(int v, int* total) const { return block_exscan<T, int>(v, s_buf, total); }
    __device__ __forceinline__ void seg_scan(int vL, int rL, int vR, int rR, int parity, int& cL, int& cR) const { seg_scan2<T>(vL, rL, vR, rR, s_buf + 2 * T / 64, parity, cL, cR); }
};
struct WaveScope {                                            // one wavefront: its LDS accesses execute in order, a "barrier" only pins the compiler
    static constexpr int NT = 64;
    __device__ __forceinline__ int tid() const { return threadIdx.x & 63; }
    __device__ __forceinline__ void sync() const { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); }
    __device__ __forceinline__ int exscan(int v, int* total) const {
        const int inc = ::planar::wave_scan_add(v);
        *total = ::planar::wave_lane(inc, 63);
        return inc - v;
    }
    __device__ __forceinline__ void seg_scan(int vL, int rL, int vR, int rR, int, int& cL, int& cR) const {
        int egL, egR, a0, a1, a2, a3;
        wave_seg_scan2(vL, rL, vR, rR, cL, egL, cR, egR, a0, a1, a2, a3);
    }
    __device__ __forceinline__ void sync_lists() const { sync(); }
};

struct Lists {                                                // a scope's segment list and per-segment results, all in LDS ([cap] each).  One buffer: a pass reads its
    uint16_t *f, *l, *pre, *cut, *mm;                         // segments into registers before the scan of phase F and writes the children behind it.
    uint8_t* d;                                               // pre[s]: the elements behind their pivots of the segments listed before s
    int cap;
};
constexpr int LIST_BYTES = 2 + 2 + 2 + 2 + 2 + 1;             // per list entry
struct Tasks { uint16_t *f, *l; uint8_t* d; int* n; };        // segments of 17 .. W_CAP elements waiting for a wavefront; n[0] count, n[1] next

template <int T, int E>
struct LdsLayout {
    static constexpr int N = T * E, NW = T / 64;
    static_assert(E >= 1 && E <= 32, "a thread's chunk is a 32-bit mask");
    static constexpr int off_a = 0;
    static constexpr int off_posh = off_a + N * 4;                                  // u16 [N / 2 + 2]
    static constexpr int off_mb = off_posh + ((N / 2 + 2) * 2 + 3) / 4 * 4;         // u32 [N / 32 + 2]: a cut / range boundary at this position
    static constexpr int off_kb = off_mb + (N / 32 + 2) * 4;                        // u32 [N / 32 + 2]: the leaf that starts here lies inside a range
    static constexpr int off_eb = off_kb + (N / 32 + 2) * 4;                        // u32 [N / 32 + 2]: an all-equal segment finished in closed form starts here (no cut inside it)
    static constexpr int off_fwd = off_eb + (N / 32 + 2) * 4;                       // u16 [N / 32 + 2]: the closed-form segment that covers the first element of this 32-element word starts at .. (0xffff: none)
    static constexpr int off_wl = off_fwd + ((N / 32 + 2) * 2 + 3) / 4 * 4;         // per wavefront: lists of W_LIST entries: f, l [2][W_LIST] u16; cut, mm u16; d [2][W_LIST] u8
    static constexpr int wl_bytes = (W_LIST * LIST_BYTES + 3) / 4 * 4;
    static constexpr int off_gl = off_wl + NW * wl_bytes;                           // the workgroup's lists, G_LIST entries, same layout
    static constexpr int gl_bytes = (G_LIST * LIST_BYTES + 3) / 4 * 4;
    static constexpr int off_tk = off_gl + gl_bytes;                                // tasks: f, l u16 [TASKS]; d u8 [TASKS]; group starts u16 [TASKS + 2]
    static constexpr int off_buf = (off_tk + TASKS * 8 + 3) / 4 * 4;                // int [10 * NW + 4]
    static constexpr int bytes = off_buf + (10 * NW + 4) * 4;
    static_assert(bytes <= 160 * 1024, "one workgroup per CU: 160 KB of LDS");
    static_assert(bytes >= (N + W_CAP + 32) * 4, "a chunk's key loads may run W_CAP words past the block (sort_levels, B)");
};
__device__ __forceinline__ Lists carve_lists(uint8_t* p, int cap) {
    Lists L;
    L.f = (uint16_t*)p; L.l = L.f + cap; L.pre = L.l + cap; L.cut = L.pre + cap; L.mm = L.cut + cap; L.d = (uint8_t*)(L.mm + cap); L.cap = cap;
    return L;
}

#ifdef ISORT_TIMING
#define ISORT_MARK(k) do { const long long _t = __builtin_readcyclecounter(); if (threadIdx.x == 0) g_isort_t[(blockIdx.x % ISORT_TBLK) * 16 + (k)] += _t - _tm; _tm = _t; } while (0)
constexpr int ISORT_TBLK = 8192;                    // a row of 16 buckets per block (thread 0 adds, nobody else touches the row): the host sums the rows
__device__ long long g_isort_t[ISORT_TBLK * 16];
#else
#define ISORT_MARK(k) do { } while (0)
#endif

// __move_median_to_first for the segment [f, l) in LDS
template <int SHIFT>
__device__ __forceinline__ void move_median(uint32_t* a, int f, int l) {
    const int A = f + 1, Bm = f + (l - f) / 2, Cc = l - 1;
    const uint32_t xa = a[A], xb = a[Bm], xc = a[Cc], xf = a[f];
    const int t = median_pos<SHIFT>(xa, xb, xc, A, Bm, Cc);
    a[f] = t == A ? xa : (t == Bm ? xb : xc); a[t] = xf;
}

// All levels of the recursion below the nseg segments listed in L (buffer 0; any order, disjoint, each of more than 16 elements, together at most NT * E elements),
// by the threads of scope S.  Children of more than keep_above elements stay in the scope's list, smaller ones of more than 16 go to the task list TK (none when
// keep_above == 16); a child whose depth budget is used up is recorded for the heap-sort fallback.
//
// The threads share the ACTIVE elements - the elements behind the pivots of the segments still listed, concatenated in list order - evenly: with A of them a thread owns
// Q = ceil(A / NT) consecutive ones (bit j of its masks = its j-th), which lie in at most three segments (a segment has more than 16, Q <= E <= 32) as contiguous pieces;
// pre[] (the running count of active elements per listed segment) maps a thread's range to its pieces.  A pass therefore costs what is still being partitioned, not the
// span the job started with: the passes at the end of a chain, when most segments are leaves or closed-form segments, touch a few elements per lane.
// A level is latency, not arithmetic (a wavefront's ~40 dependent LDS round trips, four wavefronts per SIMD to hide them behind), so the code keeps the
// number of dependent LDS accesses down: the three segments a thread can touch are read together, swaps go four at a time, the pivots of the next level are placed
// by the thread that lists the segment.
template <int SHIFT, int E, class S>
__device__ __forceinline__ void sort_levels(const S& sc, uint32_t* a, uint16_t* posh, uint32_t* mb, uint32_t* kb, uint32_t* eb, uint16_t* fwd, const Lists& L, int nseg,
                                            int keep_above, const Tasks& TK, const HeapSink& HS, int span_f, int* status, uint32_t skip_key) {
#ifdef ISORT_TIMING
    long long _tm = __builtin_readcyclecounter();
#endif
    constexpr int NT = S::NT;
    constexpr bool PACK = NT == 64;      // (wavefront scope: the scans carry both stop counts, see B)
    const int tid = sc.tid(), cap = L.cap;
    int A;                                                       // active elements of the current list
    {   // the initial list: pivots to the front, pre[]
        int m[2] = {0, 0};
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int s_ = 2 * tid + u;
            if (s_ < nseg) { const int f = L.f[s_], l = L.l[s_]; move_median<SHIFT>(a, f, l); m[u] = l - f - 1; }
        }
        const int off = sc.exscan(m[0] + m[1], &A);
        if (2 * tid < nseg) L.pre[2 * tid] = (uint16_t)off;
        if (2 * tid + 1 < nseg) L.pre[2 * tid + 1] = (uint16_t)(off + m[0]);
        if (A > NT * E) { *status = ST_CAPACITY; nseg = 0; }
        sc.sync();
    }
    int cur = 0;
    while (nseg > 0) {
        const int Q = (A + NT - 1) / NT;                         // (uniform; <= E)
#ifdef PLANAR_WAVE_EMUL
        if (tid == 0) { g_levels++; g_segs += nseg; if (NT == 64) { g_wlev++; g_wlanes += Q; } else { g_glev++; g_glanes += Q; } }
#endif
        uint16_t* sf = L.f; uint16_t* sl = L.l; uint16_t* sp = L.pre; uint8_t* sd = L.d;
        uint16_t* scut = L.cut; uint16_t* smm = L.mm;
        ISORT_MARK(1);
        // ---- B: this thread's active elements [a0, a1): the (at most three) segments they belong to, where the two scans stop ----
        const int a0 = min(tid * Q, A), a1 = min(a0 + Q, A);
        int s0 = 0;                                              // the listed segment that holds active element a0: the last one with pre <= a0
        { int hi_ = nseg; while (hi_ - s0 > 1) { const int mid = (s0 + hi_) >> 1; if ((int)sp[mid] <= a0) s0 = mid; else hi_ = mid; } }
        uint32_t mL = 0, mR = 0;
        int npc = 0;
        int pf[3], pl_[3], plo[3], phi[3], pb[3];                // piece q: bits [plo, phi) of the masks; bit j is the element at position pb[q] + j
        uint32_t ppv[3];
        bool contL = false, contR = false;
        {
#pragma unroll
            for (int q = 0; q < 3; q++) { const int s_ = min(s0 + q, nseg - 1); pf[q] = sf[s_]; pl_[q] = sl[s_]; pb[q] = sp[s_]; }
#pragma unroll
            for (int q = 0; q < 3; q++) ppv[q] = a[pf[q]] >> SHIFT;
#pragma unroll
            for (int q = 0; q < 3; q++) {
                const int g0 = pb[q], g1 = g0 + (pl_[q] - pf[q] - 1);              // the segment's active range
                const int lo = max(g0, a0) - a0, hi = min(g1, a1) - a0;
                const bool on = s0 + q < nseg && lo < hi;
                plo[q] = on ? lo : 99; phi[q] = on ? hi : 99;
                pb[q] = on ? pf[q] + 1 + a0 - g0 : 0;
                if (on) { npc = q + 1; contR = g1 > a1; if (q == 0) contL = g0 < a0; }
            }
            const uint32_t rm_all = a1 - a0 >= 32 ? 0xffffffffu : ((1u << (a1 - a0)) - 1u);
            uint32_t ge = 0, le = 0;
            for (int j0 = 0; j0 < Q; j0 += 4) {                  // (elements behind the thread's last one: masked; their reads stay inside the block's LDS, see LdsLayout)
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int j = j0 + u;
                    const bool q2 = j >= plo[2], q1 = j >= plo[1];
                    const uint32_t pv = q2 ? ppv[2] : (q1 ? ppv[1] : ppv[0]);
                    const uint32_t k = a[(q2 ? pb[2] : (q1 ? pb[1] : pb[0])) + j] >> SHIFT;
                    ge |= (uint32_t)(k >= pv) << (j & 31); le |= (uint32_t)(k <= pv) << (j & 31);
                }
            }
            mL = ge & rm_all; mR = le & rm_all;
        }
        auto rmask = [&](int q) { return (phi[q] >= 32 ? 0xffffffffu : ((1u << phi[q]) - 1u)) & ~((1u << plo[q]) - 1u); };
        // contL / contR: the first piece's segment has active elements before this thread's / the last piece's segment behind them
        const int lastq = npc > 0 ? npc - 1 : 0;
        uint32_t rm_last = 0;
#pragma unroll
        for (int q = 0; q < 3; q++) if (q == lastq && npc > 0) rm_last = rmask(q);
        const uint32_t rm_first = npc > 0 ? rmask(0) : 0u;
        int carryL, carryR, carryL2 = 0, carryR2 = 0;
        ISORT_MARK(2);
        // (wavefront scope: both scans carry BOTH stop counts, 12 bits each - a wavefront's job has at most W_CAP < 4096 elements -, so that the piece that holds x* knows
        // the segment's totals: every element stops both scans = all keys equal the pivot's)
        sc.seg_scan(__popc(mL & rm_last) | (PACK ? __popc(mR & rm_last) << 12 : 0), !(npc == 1 && contL),
                    __popc(mR & rm_first) | (PACK ? __popc(mL & rm_first) << 12 : 0), !(npc == 1 && contR), cur, carryL, carryR);
        if (PACK) { carryL2 = carryL >> 12; carryL &= 0xfff; carryR2 = carryR >> 12; carryR &= 0xfff; }
        ISORT_MARK(3);
        // ---- D1: the piece with g false at its start and true behind its end holds x*: it writes the segment's cut and its number of swaps m ----
        int pA0[3] = {0, 0, 0}, pBe[3] = {0, 0, 0};
#pragma unroll
        for (int q = 0; q < 3; q++) {
            if (q >= npc) continue;
            const uint32_t rm = rmask(q), Lp = mL & rm, Rp = mR & rm;
            const int A0 = (q == 0 && contL) ? carryL : 0, Be = (q == lastq && contR) ? carryR : 0;
            pA0[q] = A0; pBe[q] = Be;
            if (!(A0 >= Be + __popc(Rp)) && A0 + __popc(Lp) >= Be) {
                int jl = plo[q], jh = phi[q];                    // g(jl) false, g(jh) true: x* in (jl, jh]
                while (jh - jl > 1) {
                    const int mid = (jl + jh) >> 1;
                    if (A0 + __popc(Lp & ((1u << mid) - 1u)) >= Be + __popc(Rp >> mid)) jh = mid; else jl = mid;
                }
                const int e = jh - 1;                            // the element before x*: in this piece
                const bool eL = (Lp >> e) & 1u, eR = (Rp >> e) & 1u;
                const int Ap = A0 + __popc(Lp & ((1u << e) - 1u)), Bp = Be + __popc(Rp >> e);
                scut[s0 + q] = (uint16_t)(pb[q] + jh - ((eL && eR && Ap == Bp - 1) ? 1 : 0));
                smm[s0 + q] = (uint16_t)max(Ap, Bp - (int)eR);
                if (PACK) {                                      // all n - 1 elements behind the pivot stop both scans, and the budget covers the halvings: closed form (equal_dest)
                    const int n1 = pl_[q] - pf[q] - 1, totL = A0 + __popc(Lp) + ((q == lastq && contR) ? carryR2 : 0), totR = Be + __popc(Rp) + ((q == 0 && contL) ? carryL2 : 0);
                    if (totL == n1 && totR == n1 && (int)sd[s0 + q] >= equal_levels(n1 + 1)) { scut[s0 + q] = (uint16_t)EQ_CUT; smm[s0 + q] = 0; }
                }
            }
        }
        sc.sync();
        ISORT_MARK(4);
        // ---- D2: the first m stops of the left scan publish their positions (rank order) ----
        int pm[3];
#pragma unroll
        for (int q = 0; q < 3; q++) pm[q] = smm[min(s0 + q, nseg - 1)];
#pragma unroll
        for (int q = 0; q < 3; q++) {
            if (q >= npc) continue;
            uint32_t bits = mL & rmask(q);
            const int base = ((pf[q] + 1) >> 1) + pA0[q];
            const int gl = min(max(pm[q] - pA0[q], 0), __popc(bits));
            for (int k = 0; k < gl; k++) { const int j = __ffs((int)bits) - 1; bits &= bits - 1u; posh[base + k] = (uint16_t)(pb[q] + j); }
        }
        sc.sync();
        ISORT_MARK(5);
        // ---- E: the first m stops of the right scan (it walks downwards) fetch their partners and swap, four at a time ----
#pragma unroll
        for (int q = 0; q < 3; q++) {
            if (q >= npc) continue;
            uint32_t bits = mR & rmask(q);
            const int base = ((pf[q] + 1) >> 1) + pBe[q];
            const int gr = min(max(pm[q] - pBe[q], 0), __popc(bits));
            constexpr int SU = 4;
            for (int t0 = 0; t0 < gr; t0 += SU) {
                int qq[SU], pL[SU];
                uint32_t x[SU], y[SU];
#pragma unroll
                for (int u = 0; u < SU; u++) {
                    const int j = bits ? 31 - __clz((int)bits) : plo[q];
                    if (t0 + u < gr) bits &= ~(1u << j);
                    qq[u] = pb[q] + j;
                    pL[u] = posh[base + min(t0 + u, gr - 1)];
                }
#pragma unroll
                for (int u = 0; u < SU; u++) { x[u] = a[pL[u]]; y[u] = a[qq[u]]; }
#pragma unroll
                for (int u = 0; u < SU; u++) if (t0 + u < gr) { a[pL[u]] = y[u]; a[qq[u]] = x[u]; }
            }
        }
        sc.sync();
        ISORT_MARK(6);
        // ---- F: the cuts become leaf boundaries; children: stay listed (their pivot is placed now) / become a wavefront's task / are finished (leaf, closed-form
        //      segment, or heap sort at depth 0) ----
        {
            uint16_t* nf = sf; uint16_t* nl = sl; uint16_t* np = sp; uint8_t* nd = sd;      // (in place: every thread has read its segments before the scan below lets any write)
            int cf[4], cl[4], cd[4], nk = 0, nact = 0;
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int s = 2 * tid + u;
                if (s < nseg) {
                    const int f = sf[s], l = sl[s], cut = scut[s], d = sd[s] - 1;      // (listed segments have a budget of at least one)
                    const bool equal = PACK && cut == EQ_CUT;
                    if (equal) {                                                   // all keys equal: the pivot goes back where __move_median_to_first took it from (mid), the
                        const int mid = f + (l - f) / 2;                           // segment is flagged and leaves the list; the write-back places its elements (equal_dest)
                        const uint32_t x = a[f]; a[f] = a[mid]; a[mid] = x;
                        atomicOr(&eb[f >> 5], 1u << (f & 31));
                        posh[(f + 1) >> 1] = (uint16_t)l;                          // (its own, now idle, part of the rendezvous table) the write-back finds the segment's end here,
                        for (int w = (f >> 5) + 1; w <= ((l - 1) >> 5); w++) fwd[w] = (uint16_t)f;   // and its start from any word whose first element it covers
#ifdef PLANAR_WAVE_EMUL
                        g_eqseg++; g_eqelem += l - f;
#endif
                    } else { atomicOr(&mb[cut >> 5], 1u << (cut & 31)); atomicOr(&kb[cut >> 5], 1u << (cut & 31)); }
                    const bool drop_right = (a[f] >> SHIFT) > skip_key;            // everything from the cut on is >= the pivot (it sits at f): nothing the caller wants is in there
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        const int xf = h ? cut : f, xl = h ? l : cut;
                        if (equal || (h == 1 && drop_right)) continue;
                        if (xl - xf > 16 && d == 0) push_heap_job(HS, span_f + xf, span_f + xl, status);   // budget used up: the heap-sort fallback, later (a leaf of more than 16: the write-back leaves it alone)
                        else if (xl - xf > keep_above) {
#pragma unroll
                            for (int z = 0; z < 4; z++) if (nk == z) { cf[z] = xf; cl[z] = xl; cd[z] = d; }
                            nk++; nact += xl - xf - 1;
                            move_median<SHIFT>(a, xf, xl);
                        } else if (xl - xf > 16) {
                            const int k = atomicAdd(&TK.n[0], 1);
                            if (k < TASKS) { TK.f[k] = (uint16_t)xf; TK.l[k] = (uint16_t)xl; TK.d[k] = (uint8_t)d; } else *status = ST_CAPACITY;
                        }
                    }
                }
            }
            int tot;                                             // one scan for both: listed children (low 9 bits: at most 4 per thread, `cap` in all) and their active elements
            const int offp = sc.exscan(nk | (nact << 9), &tot);
            int off = offp & 0x1ff, run = offp >> 9;
#pragma unroll
            for (int z = 0; z < 4; z++)
                if (z < nk && off + z < cap) { nf[off + z] = (uint16_t)cf[z]; nl[off + z] = (uint16_t)cl[z]; nd[off + z] = (uint8_t)cd[z]; np[off + z] = (uint16_t)run; run += cl[z] - cf[z] - 1; }
            A = tot >> 9; tot &= 0x1ff;
            if (tot > cap) { *status = ST_CAPACITY; tot = 0; }
            sc.sync();
            nseg = tot;
            cur ^= 1;                                            // (the segmented scan's buffer parity; no barrier here: the next writes to scut / smm are behind the scan's)
        }
        ISORT_MARK(7);
    }
}

// Sorts arr[span_f, span_l) (<= T * E elements), which consists of the nr <= T ranges `ranges` (sorted by f, disjoint; anything between them is left
// alone), as std::sort would have finished each of them.  All T threads of the workgroup call it.
template <int SHIFT, int T, int E>
__device__ void lds_tier(uint32_t* __restrict__ arr, const Range* __restrict__ ranges, int nr, int span_f, int span_l, uint8_t* lds, const HeapSink& HS, int* status,
                         uint32_t skip_key = 0xffffffffu) {
    using LL = LdsLayout<T, E>;
    uint32_t* a = (uint32_t*)(lds + LL::off_a);
    uint16_t* posh = (uint16_t*)(lds + LL::off_posh);
    uint32_t* mb = (uint32_t*)(lds + LL::off_mb);
    uint32_t* kb = (uint32_t*)(lds + LL::off_kb);
    uint32_t* eb = (uint32_t*)(lds + LL::off_eb);
    uint16_t* fwd = (uint16_t*)(lds + LL::off_fwd);
    int* s_buf = (int*)(lds + LL::off_buf);
    int* s_tn = s_buf + 10 * LL::NW;                             // [0] tasks, [1] next group, [2] groups
    Tasks TK;
    TK.f = (uint16_t*)(lds + LL::off_tk); TK.l = TK.f + TASKS; TK.d = (uint8_t*)(TK.l + TASKS); TK.n = s_tn;
    const Lists GL = carve_lists(lds + LL::off_gl, G_LIST);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = span_l - span_f;
#ifdef ISORT_TIMING
    long long _tm = __builtin_readcyclecounter();
#endif
    for (int i = tid; i < n; i += T) a[i] = arr[span_f + i];
    for (int i = tid; i < LL::N / 32 + 2; i += T) { mb[i] = 0u; kb[i] = 0u; eb[i] = 0u; fwd[i] = (uint16_t)0xffff; }
    if (tid == 0) { s_tn[0] = 0; s_tn[1] = 0; s_tn[2] = 0; }
    __syncthreads();
    const WgScope<T> wg{s_buf};
    int nseg;
    {   // the ranges: boundaries; more than W_CAP elements: the workgroup's list; more than 16: a task; depth 0 (cannot be, but harmless): heap sort
        int keep = 0;
        Range R{0, 0, 0};
        if (tid < nr) {
            R = ranges[tid];
            R.f -= span_f; R.l -= span_f;
            atomicOr(&mb[R.f >> 5], 1u << (R.f & 31)); atomicOr(&kb[R.f >> 5], 1u << (R.f & 31));
            atomicOr(&mb[R.l >> 5], 1u << (R.l & 31));
            if (R.l - R.f > 16 && R.d == 0) push_heap_job(HS, span_f + R.f, span_f + R.l, status);
            else if (R.l - R.f > W_CAP) keep = 1;
            else if (R.l - R.f > 16) {
                const int k = atomicAdd(&s_tn[0], 1);
                if (k < TASKS) { TK.f[k] = (uint16_t)R.f; TK.l[k] = (uint16_t)R.l; TK.d[k] = (uint8_t)R.d; } else *status = ST_CAPACITY;
            }
        }
        if (tid == 0) { atomicOr(&mb[0], 1u); atomicOr(&mb[n >> 5], 1u << (n & 31)); }
        int tot;
        const int off = wg.exscan(keep, &tot);
        if (keep && off < G_LIST) { GL.f[off] = (uint16_t)R.f; GL.l[off] = (uint16_t)R.l; GL.d[off] = (uint8_t)R.d; }
        if (tot > G_LIST) { *status = ST_CAPACITY; tot = G_LIST; }
        nseg = tot;
        __syncthreads();
    }
    ISORT_MARK(0);
    sort_levels<SHIFT, E, WgScope<T>>(wg, a, posh, mb, kb, eb, fwd, GL, nseg, W_CAP, TK, HS, span_f, status, skip_key);
    __syncthreads();
#ifdef ISORT_TIMING
    _tm = __builtin_readcyclecounter();
#endif
    {   // the tasks, in groups of at most W_CAP elements (consecutive entries of the task list: sort_levels takes segments in any order); a wavefront per group, first
        // come first served.  A block's tasks add up to at most its span, so there are about as many groups as wavefronts and every lane of them has elements
        const WaveScope wv;
        const Lists WL = carve_lists(lds + LL::off_wl + wave * LL::wl_bytes, W_LIST);
        const int ntasks = min(s_tn[0], TASKS);
        const Tasks none{nullptr, nullptr, nullptr, s_tn};
        uint16_t* gstart = (uint16_t*)(lds + LL::off_tk + 5 * TASKS);       // [groups + 1]
        if (tid == 0) {
            int total = 0;
            for (int t = 0; t < ntasks; t++) total += TK.l[t] - TK.f[t];
            // as many groups as wavefronts where that fits, of about equal size: the block ends with its longest group, and a pass costs what the group holds
            const int target = min(W_CAP, max((total + LL::NW - 1) / LL::NW, 64));
            int g = 0, sum = 0;
            gstart[0] = 0;
            for (int t = 0; t < ntasks; t++) {
                const int len = TK.l[t] - TK.f[t];
                if (sum > 0 && (sum + len > W_CAP || sum + len / 2 > target)) { gstart[++g] = (uint16_t)t; sum = 0; }
                sum += len;
            }
            if (ntasks > 0) gstart[++g] = (uint16_t)ntasks;
            s_tn[2] = g;
        }
        __syncthreads();
        const int ngroups = s_tn[2];
        while (true) {
            int g = 0;
            if (lane == 0) g = atomicAdd(&s_tn[1], 1);
            g = __shfl(g, 0);
            if (g >= ngroups) break;
            const int t0 = gstart[g], cnt = gstart[g + 1] - t0;              // (<= W_CAP / 17 < W_LIST tasks)
            for (int i = lane; i < cnt; i += 64) { WL.f[i] = TK.f[t0 + i]; WL.l[i] = TK.l[t0 + i]; WL.d[i] = TK.d[t0 + i]; }
            wv.sync();
            sort_levels<SHIFT, W_E, WaveScope>(wv, a, posh, mb, kb, eb, fwd, WL, cnt, 16, none, HS, span_f, status, skip_key);
        }
    }
    __syncthreads();
    ISORT_MARK(8);
    // the write-back; an element of a leaf of <= 16 elements inside a range goes to its stable rank in the leaf (__final_insertion_sort), an element of a closed-form
    // segment to equal_dest of its position, anything else stays.  A leaf's boundaries are within a word of the element; a closed-form segment's start is the nearest
    // boundary below where that is within a word, else fwd[] of the element's word, and its end is in the segment's slot of the rendezvous table: no searches.
    // Four elements per thread at a time (each a row of 64 consecutive elements per wavefront); the leaf walk (16 predicated reads) is skipped by rows without a leaf
    constexpr int WU = 4;
    for (int ib = 0; ib < n; ib += T * WU) {                 // (the same trip count for every lane: the ballots below)
        const int i0 = ib + tid;
        uint32_t v[WU];
        int s[WU], e[WU], dest[WU];
        bool leaf[WU];
#pragma unroll
        for (int u = 0; u < WU; u++) {
            const int i = min(i0 + u * T, n - 1), w = i >> 5, bi = i & 31;
            v[u] = a[i];
            const uint32_t mw = mb[w], mp = mb[max(w - 1, 0)], mn = mb[w + 1];
            const int fw = fwd[w];
            const uint32_t below = mw & (0xffffffffu >> (31 - bi)), above = bi == 31 ? 0u : (mw & (0xffffffffu << (bi + 1)));
            s[u] = below ? (w << 5) + 31 - __clz((int)below) : ((w > 0 && mp) ? ((w - 1) << 5) + 31 - __clz((int)mp) : (fw != 0xffff ? fw : -1));
            e[u] = above ? (w << 5) + __ffs((int)above) - 1 : (mn ? ((w + 1) << 5) + __ffs((int)mn) - 1 : -1);
        }
#pragma unroll
        for (int u = 0; u < WU; u++) {
            const int i = min(i0 + u * T, n - 1), sb = max(s[u], 0);
            const bool flagged = s[u] >= 0 && ((eb[sb >> 5] >> (sb & 31)) & 1u);
            leaf[u] = s[u] >= 0 && e[u] >= 0 && e[u] - s[u] <= 16 && ((kb[sb >> 5] >> (sb & 31)) & 1u);
            dest[u] = i;
            if (flagged) dest[u] = sb + equal_dest(i - sb, (int)posh[(sb + 1) >> 1] - sb);
        }
#pragma unroll
        for (int u = 0; u < WU; u++) {
            if (__ballot(leaf[u]) == 0ull) continue;
            const int i = min(i0 + u * T, n - 1);
            const uint32_t kv = v[u] >> SHIFT;
            int rank = 0;
#pragma unroll
            for (int t = 0; t < 16; t++) {
                const int jj = max(s[u], 0) + t;
                const uint32_t kj = a[jj] >> SHIFT;                  // (jj < e[u] <= n where it counts; behind the block: readable, see LdsLayout)
                rank += (leaf[u] && jj < e[u] && (kj < kv || (kj == kv && jj < i))) ? 1 : 0;
            }
            if (leaf[u]) dest[u] = s[u] + rank;
        }
#pragma unroll
        for (int u = 0; u < WU; u++) if (i0 + u * T < n) arr[span_f + dest[u]] = v[u];
    }
    __syncthreads();
    ISORT_MARK(9);
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// Global tier
// ---------------------------------------------------------------------------------------------------------------------------------------
constexpr int G_QMAX = 64;       // ranges longer than the LDS tier's capacity alive at one level
constexpr int G_FMAX = 512;      // ranges handed to the LDS tier, per array set
constexpr int G_POSH = 6144;     // swaps of a partition go through a table of this many positions at a time (24 KB: beside the bitmaps of a 640 x 480 range in 160 KB)

template <int T>
struct GlobalLayout {            // LDS of global_tier: stop bitmaps + rank prefixes of a range of up to `rows` 64-element rows, the rendezvous table, then the tier's lists
    static constexpr int NW = T / 64;
    static constexpr int LDS_LIMIT = 160 * 1024 - 2560;                                                                // (a kernel's few static LDS words beside the block)
    __host__ __device__ static constexpr int off_Lw(int) { return 0; }
    __host__ __device__ static constexpr int off_Rw(int rows) { return rows * 8; }
    __host__ __device__ static constexpr int off_pL(int rows) { return rows * 16; }
    __host__ __device__ static constexpr int off_pR(int rows) { return rows * 16 + (rows + 1) * 4; }
    __host__ __device__ static constexpr int off_posh(int rows) { return (rows * 16 + (rows + 1) * 8 + 7) / 8 * 8; }  // u32 [G_POSH]: the swaps' rendezvous table
    __host__ __device__ static constexpr int off_q(int rows) { return off_posh(rows) + G_POSH * 4; }                  // Range [2][G_QMAX]
    __host__ __device__ static constexpr int off_fin(int rows) { return off_q(rows) + 2 * G_QMAX * 12; }              // Range [G_FMAX]
    __host__ __device__ static constexpr int off_rank(int rows) { return off_fin(rows) + G_FMAX * 12; }               // Range [G_FMAX] (sorted copy)
    __host__ __device__ static constexpr int off_buf(int rows) { return off_rank(rows) + G_FMAX * 12; }               // u64 [NW] + ints
    __host__ __device__ static constexpr int bytes(int rows) { return off_buf(rows) + NW * 8 + 64; }
    __host__ __device__ static constexpr int rows_for(int n) { return (n + 63) / 64 + 1; }
    // A range LONGER than the bitmaps hold (round 6: frames beyond ~390 000 sort words - 1280x720) is partitioned by wg_partition_long, which keeps only the rank prefixes
    // of its rows in LDS - two u32 per row, OVER the bitmaps, prefixes and rendezvous table of the ordinary path, none of which is live then - and recomputes a row's
    // stops where it needs them.  plan(n): rows_cap for the ordinary path and rows_long (0: not needed) for an array of n words; false: n does not fit at all.
    __host__ __device__ static constexpr int long_bytes(int rows_long) { return (rows_long + 2) * 8; }
    static bool plan(int n, int& rows_cap, int& rows_long) {
        rows_cap = rows_for(n); rows_long = 0;
        if (bytes(rows_cap) <= LDS_LIMIT) return true;
        rows_long = rows_for(n);
        rows_cap = 1024;
        while (off_q(rows_cap) < long_bytes(rows_long)) rows_cap += 64;
        return bytes(rows_cap) <= LDS_LIMIT;
    }
};

// One Hoare partition of arr[f, l) (l - f > 16) by the whole workgroup; returns the cut in every thread.
// The same partition for a range whose rows do not fit the bitmaps (nrow <= rows_long).  LDS: pL / pR [nrow + 1] only.  Pass 1 counts the stops of every row; the row that
// holds x* and the rows whose stops take part in the swaps have their ballots recomputed from the array (every row is read at most three times instead of once); the positions
// of the m left and m right stops go to a scratch array in global memory (gpos [2][n / 2 + 1]) BEFORE any swap, so every ballot sees the unpartitioned range; then thread k
// swaps pair k.  Same cut, same swaps as wg_partition.
template <int SHIFT, int T>
__device__ int wg_partition_long(uint32_t* __restrict__ arr, int f, int l, uint8_t* lds, int rows_cap, uint32_t* __restrict__ gpos, int gpos_half) {
    using GL = GlobalLayout<T>;
    constexpr int NW = T / 64;
    uint32_t* pL = (uint32_t*)lds;
    const int nrow = (l - f - 1 + 63) / 64;
    uint32_t* pR = pL + nrow + 1;
    unsigned long long* s_w = (unsigned long long*)(lds + GL::off_buf(rows_cap));
    int* s_i = (int*)(s_w + NW);          // [0] pivot key, [1] tpos, [2] old front, [3] cut, [4] m, [5] the row that holds x*
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) {
        const int A = f + 1, Bm = f + (l - f) / 2, Cc = l - 1;
        const uint32_t xa = arr[A], xb = arr[Bm], xc = arr[Cc], xf = arr[f];
        const int t = median_pos<SHIFT>(xa, xb, xc, A, Bm, Cc);
        const uint32_t xt = t == A ? xa : (t == Bm ? xb : xc);
        arr[f] = xt; arr[t] = xf;
        s_i[0] = (int)(xt >> SHIFT); s_i[1] = t; s_i[2] = (int)xf;
    }
    __syncthreads();
    const uint32_t pv = (uint32_t)s_i[0];
    const int tpos = s_i[1];
    const uint32_t xfront = (uint32_t)s_i[2];
    auto row_ballots = [&](int r, unsigned long long& bl, unsigned long long& br) {          // the whole wavefront calls it
        const int i = f + 1 + 64 * r + lane;
        const uint32_t x = arr[min(i, l - 1)];
        const uint32_t k = (i == tpos ? xfront : x) >> SHIFT;
        bl = __ballot(i < l && k >= pv); br = __ballot(i < l && k <= pv);
    };
    // ---- pass 1: the rows' stop counts ----
    constexpr int U = 8;
    for (int rb = wave; rb < nrow; rb += NW * U) {
        uint32_t x[U];
#pragma unroll
        for (int u = 0; u < U; u++) x[u] = arr[min(f + 1 + 64 * (rb + u * NW) + lane, l - 1)];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int r = rb + u * NW, i = f + 1 + 64 * r + lane;
            if (r < nrow) {
                const uint32_t k = (i == tpos ? xfront : x[u]) >> SHIFT;
                const unsigned long long bl = __ballot(i < l && k >= pv), br = __ballot(i < l && k <= pv);
                if (lane == 0) { pL[r] = (uint32_t)__popcll(bl); pR[r] = (uint32_t)__popcll(br); }
            }
        }
    }
    __syncthreads();
    // ---- ranks: exclusive prefixes in place ----
    const int RP = (nrow + T - 1) / T, r_begin = min(nrow, tid * RP), r_end = min(nrow, r_begin + RP);
    unsigned long long mine = 0;
    for (int r = r_begin; r < r_end; r++) mine += ((unsigned long long)pL[r] << 32) | (unsigned long long)pR[r];
    unsigned long long tot;
    unsigned long long run = block_exscan<T, unsigned long long>(mine, s_w, &tot);
    for (int r = r_begin; r < r_end; r++) {
        const unsigned long long c = ((unsigned long long)pL[r] << 32) | (unsigned long long)pR[r];
        pL[r] = (uint32_t)(run >> 32); pR[r] = (uint32_t)run;
        run += c;
    }
    const int totL = (int)(tot >> 32), totR = (int)(uint32_t)tot;
    if (tid == 0) { pL[nrow] = (uint32_t)totL; pR[nrow] = (uint32_t)totR; }
    __syncthreads();
    // ---- x*: the row where A >= B turns true; its ballots again, then the bit inside it (first wavefront) ----
    for (int r = r_begin; r < r_end; r++) {
        const int A0 = (int)pL[r], B0 = totR - (int)pR[r], A1 = (int)pL[r + 1], B1 = totR - (int)pR[r + 1];
        if (!(A0 >= B0) && A1 >= B1) s_i[5] = r;
    }
    __syncthreads();
    if (wave == 0) {
        const int r = s_i[5];
        unsigned long long Lb, Rb;
        row_ballots(r, Lb, Rb);
        if (lane == 0) {
            const int A0 = (int)pL[r], B0 = totR - (int)pR[r];
            auto low = [](int j_) { return j_ >= 64 ? ~0ull : ((1ull << j_) - 1ull); };
            int jl = 0, j = 64;
            while (j - jl > 1) {
                const int mid = (jl + j) >> 1;
                if (A0 + __popcll(Lb & low(mid)) >= B0 - __popcll(Rb & low(mid))) j = mid; else jl = mid;
            }
            const bool eL = (Lb >> (j - 1)) & 1ull, eR = (Rb >> (j - 1)) & 1ull;
            const int A = A0 + __popcll(Lb & low(j)), Bf = B0 - __popcll(Rb & low(j));
            const int xs = f + 1 + 64 * r + j, Ap = A - (int)eL, Bp = Bf + (int)eR;
            s_i[3] = xs - ((eL && eR && Ap == Bp - 1) ? 1 : 0);
            s_i[4] = max(Ap, Bf);
        }
    }
    __syncthreads();
    const int cut = s_i[3], m = s_i[4];
    // ---- the positions of the first m stops of the left scan and of the last m of the right scan, by rank ----
    if (m > 0) {
        auto row_of = [&](const uint32_t* pref, int rank) {       // the last row whose prefix is <= rank
            int lo = 0, hi = nrow;
            while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if ((int)pref[mid] <= rank) lo = mid; else hi = mid; }
            return lo;
        };
        const unsigned long long below = (1ull << lane) - 1ull;
        uint32_t* gL = gpos; uint32_t* gR = gpos + gpos_half;
        const int lb = row_of(pL, m - 1), ra = row_of(pR, totR - m);
        for (int r = wave; r <= lb; r += NW) {
            unsigned long long Lb, Rb;
            row_ballots(r, Lb, Rb);
            const int k = (int)pL[r] + __popcll(Lb & below);
            if (((Lb >> lane) & 1ull) && k < m) gL[k] = (uint32_t)(f + 1 + 64 * r + lane);
        }
        for (int r = ra + wave; r < nrow; r += NW) {
            unsigned long long Lb, Rb;
            row_ballots(r, Lb, Rb);
            const int k = totR - 1 - ((int)pR[r] + __popcll(Rb & below));
            if (((Rb >> lane) & 1ull) && k >= 0 && k < m) gR[k] = (uint32_t)(f + 1 + 64 * r + lane);
        }
        __threadfence_block();
        __syncthreads();
        for (int k = tid; k < m; k += T) {
            const int pp = (int)gL[k], qq = (int)gR[k];
            const uint32_t x = arr[pp], y = arr[qq];
            arr[pp] = y; arr[qq] = x;
        }
    }
    __threadfence_block();
    __syncthreads();
    return cut;
}

template <int SHIFT, int T>
__device__ int wg_partition(uint32_t* __restrict__ arr, int f, int l, uint8_t* lds, int rows_cap, int* status, int rows_long = 0, uint32_t* gpos = nullptr, int gpos_half = 0) {
    using GL = GlobalLayout<T>;
    constexpr int NW = T / 64;
    unsigned long long* Lw = (unsigned long long*)(lds + GL::off_Lw(rows_cap));
    unsigned long long* Rw = (unsigned long long*)(lds + GL::off_Rw(rows_cap));
    uint32_t* pL = (uint32_t*)(lds + GL::off_pL(rows_cap));
    uint32_t* pR = (uint32_t*)(lds + GL::off_pR(rows_cap));
    unsigned long long* s_w = (unsigned long long*)(lds + GL::off_buf(rows_cap));
    int* s_i = (int*)(s_w + NW);          // [0] pivot key, [1] tpos, [2] old front, [3] cut, [4] m
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nrow = (l - f - 1 + 63) / 64;
    if (nrow > rows_cap) {
        if (nrow <= rows_long && gpos && (l - f) / 2 + 1 <= gpos_half) return wg_partition_long<SHIFT, T>(arr, f, l, lds, rows_cap, gpos, gpos_half);
        if (tid == 0) *status = ST_CAPACITY;
        return f + (l - f) / 2;
    }
#ifdef ISORT_TIMING
    long long _tm = __builtin_readcyclecounter();
#endif
    if (tid == 0) {
        const int A = f + 1, Bm = f + (l - f) / 2, Cc = l - 1;
        const uint32_t xa = arr[A], xb = arr[Bm], xc = arr[Cc], xf = arr[f];
        const int t = median_pos<SHIFT>(xa, xb, xc, A, Bm, Cc);
        const uint32_t xt = t == A ? xa : (t == Bm ? xb : xc);
        arr[f] = xt; arr[t] = xf;
        s_i[0] = (int)(xt >> SHIFT); s_i[1] = t; s_i[2] = (int)xf;
    }
    __syncthreads();
    const uint32_t pv = (uint32_t)s_i[0];
    const int tpos = s_i[1];
    const uint32_t xfront = (uint32_t)s_i[2];
    ISORT_MARK(10);
    // ---- pass 1: one read of the range; a row of 64 elements = one ballot per scan ----
    constexpr int U = 8;
    for (int rb = wave; rb < nrow; rb += NW * U) {
        uint32_t x[U];
#pragma unroll
        for (int u = 0; u < U; u++) x[u] = arr[min(f + 1 + 64 * (rb + u * NW) + lane, l - 1)];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int r = rb + u * NW, i = f + 1 + 64 * r + lane;
            if (r < nrow) {
                const uint32_t k = (i == tpos ? xfront : x[u]) >> SHIFT;
                const unsigned long long bl = __ballot(i < l && k >= pv), br = __ballot(i < l && k <= pv);
                if (lane == 0) { Lw[r] = bl; Rw[r] = br; }
            }
        }
    }
    __syncthreads();
    ISORT_MARK(11);
    // ---- ranks: exclusive popcount prefix per row ----
    const int RP = (nrow + T - 1) / T, r_begin = min(nrow, tid * RP), r_end = min(nrow, r_begin + RP);
    unsigned long long mine = 0;
    for (int r = r_begin; r < r_end; r++) mine += ((unsigned long long)__popcll(Lw[r]) << 32) | (unsigned long long)__popcll(Rw[r]);
    unsigned long long tot;
    unsigned long long run = block_exscan<T, unsigned long long>(mine, s_w, &tot);
    for (int r = r_begin; r < r_end; r++) {
        pL[r] = (uint32_t)(run >> 32); pR[r] = (uint32_t)run;
        run += ((unsigned long long)__popcll(Lw[r]) << 32) | (unsigned long long)__popcll(Rw[r]);
    }
    const int totL = (int)(tot >> 32), totR = (int)(uint32_t)tot;
    if (tid == 0) { pL[nrow] = (uint32_t)totL; pR[nrow] = (uint32_t)totR; }
    __syncthreads();
    ISORT_MARK(12);
    // ---- x*: the row where A >= B turns true, then the bit inside it ----
    for (int r = r_begin; r < r_end; r++) {
        const int A0 = (int)pL[r], B0 = totR - (int)pR[r], A1 = (int)pL[r + 1], B1 = totR - (int)pR[r + 1];
        if (!(A0 >= B0) && A1 >= B1) {
            // the first j in 1 .. 64 with A0 + #L among the row's first j elements >= B0 - #R among them (the difference only grows with j: six halvings
            // instead of a walk of up to 64 steps by one thread while the other 1023 wait at the barrier)
            const unsigned long long Lb = Lw[r], Rb = Rw[r];
            auto low = [](int j_) { return j_ >= 64 ? ~0ull : ((1ull << j_) - 1ull); };
            int jl = 0, j = 64;
            while (j - jl > 1) {
                const int mid = (jl + j) >> 1;
                if (A0 + __popcll(Lb & low(mid)) >= B0 - __popcll(Rb & low(mid))) j = mid; else jl = mid;
            }
            const bool eL = (Lb >> (j - 1)) & 1ull, eR = (Rb >> (j - 1)) & 1ull;
            const int A = A0 + __popcll(Lb & low(j)), Bf = B0 - __popcll(Rb & low(j));
            const int xs = f + 1 + 64 * r + j, Ap = A - (int)eL, Bp = Bf + (int)eR;
            s_i[3] = xs - ((eL && eR && Ap == Bp - 1) ? 1 : 0);
            s_i[4] = max(Ap, Bf);
        }
    }
    __syncthreads();
    const int cut = s_i[3], m = s_i[4];
    ISORT_MARK(13);
    // ---- swaps: the k-th stop of the left scan (k < m) with the k-th of the right scan (counted from the right), G_POSH ranks at a time: the right stops write
    //      their positions into a table indexed by rank, the left stops read their partner there (a binary search over the row prefixes + an in-word select per
    //      element before: two thirds of the tier's time).  lane = element of a row; a chunk's rows are found once per wavefront ----
    {
        uint32_t* posh = (uint32_t*)(lds + GL::off_posh(rows_cap));
        auto row_of = [&](const uint32_t* pref, int rank) {       // the last row whose prefix is <= rank (pref[0] = 0 <= rank)
            int lo = 0, hi = nrow;
            while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if ((int)pref[mid] <= rank) lo = mid; else hi = mid; }
            return lo;
        };
        const unsigned long long below = (1ull << lane) - 1ull;
        for (int base = 0; base < m; base += G_POSH) {
            const int kend = min(m, base + G_POSH);
            const int rr_lo = totR - kend, rr_hi = totR - 1 - base;          // the right stops of these ranks, counted from the left
            const int ra = row_of(pR, rr_lo), rb = row_of(pR, rr_hi);
            for (int r = ra + wave; r <= rb; r += NW) {
                const unsigned long long Rb = Rw[r];
                const int rr = (int)pR[r] + __popcll(Rb & below);
                if (((Rb >> lane) & 1ull) && rr >= rr_lo && rr <= rr_hi) posh[(totR - 1 - rr) - base] = (uint32_t)(f + 1 + 64 * r + lane);
            }
            __syncthreads();
            const int la = row_of(pL, base), lb = row_of(pL, kend - 1);
            for (int r = la + wave; r <= lb; r += NW) {
                const unsigned long long Lb = Lw[r];
                const int k = (int)pL[r] + __popcll(Lb & below);
                if (((Lb >> lane) & 1ull) && k >= base && k < kend) {
                    const int q = (int)posh[k - base], p = f + 1 + 64 * r + lane;
                    const uint32_t x = arr[p], y = arr[q];
                    arr[p] = y; arr[q] = x;
                }
            }
            __syncthreads();
        }
    }
    __threadfence_block();
    __syncthreads();
    ISORT_MARK(14);
    return cut;
}

// The workgroup partitions every range of `init` (n_init <= G_FMAX, disjoint) that is longer than n_stage until none is left, then writes the
// resulting ranges sorted by position to out_ranges and packs consecutive ones into LDS-tier blocks (span <= n_stage, <= nr_cap ranges).
// out_counts: [0] ranges, [1] blocks.
// skip_key: the caller only needs the elements whose key is <= skip_key where std::sort would put them (LSD: the pixels with a defined angle).  A partition's right part
// holds nothing below its pivot, so when the pivot's key is above skip_key that part is left as it is - never queued, never handed to the LDS tier, no fallback job.
// What is left unsorted are whole index ranges of unwanted elements: every wanted element still ends exactly where std::sort puts it.
template <int SHIFT, int T>
__device__ void global_tier(uint32_t* __restrict__ arr, const Range* __restrict__ init, int n_init, int n_stage, int nr_cap, Range* __restrict__ out_ranges,
                            Block* __restrict__ out_blocks, int max_blocks, int* __restrict__ out_counts, uint8_t* lds, int rows_cap, const HeapSink& HS, int* status,
                            uint32_t skip_key = 0xffffffffu, int rows_long = 0, uint32_t* gpos = nullptr, int gpos_half = 0) {
    using GL = GlobalLayout<T>;
    Range* qb = (Range*)(lds + GL::off_q(rows_cap));
    Range* fin = (Range*)(lds + GL::off_fin(rows_cap));
    Range* srt = (Range*)(lds + GL::off_rank(rows_cap));
    int* s_c = (int*)(lds + GL::off_buf(rows_cap)) + 2 * GL::NW + 8;      // [0] ncur, [1] nnext, [2] nfin
    const int tid = threadIdx.x;
    if (tid == 0) {
        int nq = 0, nf = 0;
        for (int i = 0; i < n_init; i++) {
            const Range R = init[i];
            if (R.l - R.f > n_stage && R.d > 0) { if (nq < G_QMAX) qb[nq++] = R; else *status = ST_CAPACITY; }
            else if (R.l - R.f > n_stage) push_heap_job(HS, R.f, R.l, status);
            else if (R.l - R.f > 1) { if (nf < G_FMAX) fin[nf++] = R; else *status = ST_CAPACITY; }
        }
        s_c[0] = nq; s_c[1] = 0; s_c[2] = nf;
    }
    __syncthreads();
    int cur = 0;
    while (true) {
        const int ncur = s_c[0];
        if (ncur == 0) break;
        for (int r = 0; r < ncur; r++) {
            const Range R = qb[cur * G_QMAX + r];
            const int cut = wg_partition<SHIFT, T>(arr, R.f, R.l, lds, rows_cap, status, rows_long, gpos, gpos_half);
            if (tid == 0) {
                const bool drop_right = (uint32_t)((const int*)(lds + GL::off_buf(rows_cap)))[2 * GL::NW] > skip_key;   // the pivot's key (wg_partition's s_i[0])
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    if (h == 1 && drop_right) continue;
                    const Range C{h ? cut : R.f, h ? R.l : cut, R.d - 1};       // (queued ranges have a budget of at least one)
                    if (C.l - C.f > n_stage && C.d > 0) { if (s_c[1] < G_QMAX) qb[(cur ^ 1) * G_QMAX + s_c[1]++] = C; else *status = ST_CAPACITY; }
                    else if (C.l - C.f > n_stage) push_heap_job(HS, C.f, C.l, status);   // budget used up on a range too long for an LDS block: the fallback, later
                    else if (C.l - C.f > 1) { if (s_c[2] < G_FMAX) fin[s_c[2]++] = C; else *status = ST_CAPACITY; }
                }
            }
        }
        __syncthreads();
        if (tid == 0) { s_c[0] = s_c[1]; s_c[1] = 0; }
        cur ^= 1;
        __syncthreads();
    }
    const int nf = s_c[2];
    for (int i = tid; i < nf; i += T) {         // ranges are disjoint: the rank of a range is the number of ranges that start before it
        const Range R = fin[i];
        int rank = 0;
        for (int j = 0; j < nf; j++) rank += fin[j].f < R.f;
        srt[rank] = R;
    }
    __syncthreads();
    for (int i = tid; i < nf; i += T) out_ranges[i] = srt[i];
    if (tid == 0) {
        int nb = 0, i = 0;
        while (i < nf) {
            const int f0 = srt[i].f;
            int j = i + 1;
            while (j < nf && srt[j].l - f0 <= n_stage && j - i < nr_cap) j++;
            if (nb < max_blocks) out_blocks[nb] = Block{f0, srt[j - 1].l, i, j - i}; else *status = ST_CAPACITY;
            nb++;
            i = j;
        }
        out_counts[0] = nf; out_counts[1] = min(nb, max_blocks);
    }
    __syncthreads();
}

}  // namespace isort
}  // namespace planar
