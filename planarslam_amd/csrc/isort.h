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
//   lds_tier<SHIFT, T, E>   one workgroup per block of <= T * E elements (any number of ranges): staged in LDS once, every level of the
//                           recursion is six barriers (medians | flags + two segmented scans | ranks, cuts, left stops | swaps | new list),
//                           whatever the number of segments; ranges of <= 16 elements are insertion-sorted (stable) on the spot, so the block
//                           goes back to global memory SORTED, ties in std::sort's order.
// A depth-limit overflow (heap-sort fallback) is not reproduced: status 2.  It needs ~2 lg n maximally unbalanced partitions in a row.
// tests/host_shim/isort_host.cpp compiles this file with g++ on the wave64 emulator and checks it against the real std::sort.
#pragma once
#include <stdint.h>

namespace planar {
namespace isort {

struct Range { int f, l, d; };          // [f, l) of the array, d = depth budget left (2 * lg n at the top)
struct Block { int f, l, r0, nr; };     // LDS-tier job: the span [f, l) holds ranges r0 .. r0 + nr - 1 of the sorted range list

constexpr int ST_DEPTH = 2, ST_CAPACITY = 3;

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

// Two segmented scans over the workgroup's threads at once.  Forward: (vL, resetL) -> what the threads before this one accumulated since the
// last reset (exclusive).  Backward: the same from the other end.  s_buf: [4 * T / 64] ints; two barriers.
template <int T>
__device__ __forceinline__ void seg_scan2(int vL, int resetL, int vR, int resetR, int* s_buf, int& carryL, int& carryR) {
    constexpr int NW = T / 64;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int aL = vL, gL = resetL, aR = vR, gR = resetR;
    for (int o = 1; o < 64; o <<= 1) {
        const int tv = __shfl_up(aL, o), tf = __shfl_up(gL, o), uv = __shfl_down(aR, o), uf = __shfl_down(gR, o);
        if (lane >= o) { if (!gL) aL += tv; gL |= tf; }
        if (lane + o < 64) { if (!gR) aR += uv; gR |= uf; }
    }
    const int eL = __shfl_up(aL, 1), egL = __shfl_up(gL, 1), eR = __shfl_down(aR, 1), egR = __shfl_down(gR, 1);
    if (lane == 63) { s_buf[w] = aL; s_buf[NW + w] = gL; }
    if (lane == 0) { s_buf[2 * NW + w] = aR; s_buf[3 * NW + w] = gR; }
    __syncthreads();
    int wl = 0, wr = 0;
    for (int q = 0; q < NW; q++) {
        const int v = s_buf[q], g = s_buf[NW + q];
        if (q < w) wl = g ? v : wl + v;
    }
    for (int q = NW - 1; q >= 0; q--) {
        const int v = s_buf[2 * NW + q], g = s_buf[3 * NW + q];
        if (q > w) wr = g ? v : wr + v;
    }
    carryL = lane == 0 ? wl : (egL ? eL : eL + wl);
    carryR = lane == 63 ? wr : (egR ? eR : eR + wr);
    __syncthreads();
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

// stable insertion sort of a[f, l) by key: what __final_insertion_sort makes of a range the partitions are done with
template <int SHIFT>
__device__ __forceinline__ void insertion(uint32_t* a, int f, int l) {
    for (int i = f + 1; i < l; i++) {
        const uint32_t v = a[i], kv = v >> SHIFT;
        int j = i;
        while (j > f) { const uint32_t u = a[j - 1]; if (!((u >> SHIFT) > kv)) break; a[j] = u; j--; }
        a[j] = v;
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
template <int T, int E>
struct LdsLayout {
    static constexpr int N = T * E, NW = T / 64, SMAX = N / 17 + 2;
    static_assert(E >= 1 && E <= 32, "a thread's chunk is a 32-bit mask");
    static_assert(SMAX <= 2 * T, "two segments per thread in the list steps");
    static constexpr int off_a = 0;
    static constexpr int off_posh = off_a + N * 4;                                  // u16 [N / 2 + 2]
    static constexpr int off_sf = off_posh + ((N / 2 + 2) * 2 + 3) / 4 * 4;         // u16 [2][SMAX]
    static constexpr int off_sl = off_sf + 2 * SMAX * 2;
    static constexpr int off_scut = off_sl + 2 * SMAX * 2;                          // u16 [SMAX]
    static constexpr int off_sd = off_scut + SMAX * 2;                              // u8 [2][SMAX]
    static constexpr int off_buf = (off_sd + 2 * SMAX + 3) / 4 * 4;                 // int [4 * NW + 4]
    static constexpr int bytes = off_buf + (4 * NW + 4) * 4;
};

// Sorts arr[span_f, span_l) (<= T * E elements), which consists of the nr ranges `ranges` (sorted by f, disjoint; gaps are left alone), as
// std::sort would have finished each of them.  All T threads of the workgroup call it.
template <int SHIFT, int T, int E>
__device__ void lds_tier(uint32_t* __restrict__ arr, const Range* __restrict__ ranges, int nr, int span_f, int span_l, uint8_t* lds, int* status) {
    using LL = LdsLayout<T, E>;
    constexpr int SMAX = LL::SMAX;
    uint32_t* a = (uint32_t*)(lds + LL::off_a);
    uint16_t* posh = (uint16_t*)(lds + LL::off_posh);
    uint16_t* sfb = (uint16_t*)(lds + LL::off_sf);
    uint16_t* slb = (uint16_t*)(lds + LL::off_sl);
    uint16_t* scut = (uint16_t*)(lds + LL::off_scut);
    uint8_t* sdb = lds + LL::off_sd;
    int* s_buf = (int*)(lds + LL::off_buf);
    const int tid = threadIdx.x;
    const int n = span_l - span_f;
    for (int i = tid; i < n; i += T) a[i] = arr[span_f + i];
    __syncthreads();
    int nseg;
    {   // the initial list: ranges of more than 16 elements; shorter ones are finished here
        int keep = 0;
        Range R{0, 0, 0};
        if (tid < nr) {
            R = ranges[tid];
            R.f -= span_f; R.l -= span_f;
            if (R.l - R.f > 16) keep = 1;
            else if (R.l - R.f > 1) insertion<SHIFT>(a, R.f, R.l);
        }
        int tot;
        const int off = block_exscan<T, int>(keep, s_buf, &tot);
        if (keep) { sfb[off] = (uint16_t)R.f; slb[off] = (uint16_t)R.l; sdb[off] = (uint8_t)R.d; }
        nseg = tot;
        __syncthreads();
    }
    const int c0 = tid * E, c1 = min(c0 + E, n);
    int cur = 0;
    while (nseg > 0) {
        uint16_t* sf = sfb + cur * SMAX; uint16_t* sl = slb + cur * SMAX; uint8_t* sd = sdb + cur * SMAX;
        // ---- A: pivots (median of three to the front) ----
        for (int s = tid; s < nseg; s += T) {
            const int f = sf[s], l = sl[s];
            if (sd[s] == 0) *status = ST_DEPTH;
            const int A = f + 1, Bm = f + (l - f) / 2, Cc = l - 1;
            const int t = median_pos<SHIFT>(a[A], a[Bm], a[Cc], A, Bm, Cc);
            const uint32_t x = a[f]; a[f] = a[t]; a[t] = x;
        }
        __syncthreads();
        // ---- B: this thread's E elements: which segments they belong to (at most three pieces), where the scans stop ----
        uint32_t mL = 0, mR = 0;
        int npc = 0;
        int ps[3] = {0, 0, 0}, plo[3] = {0, 0, 0}, phi[3] = {0, 0, 0};
        if (c0 < n) {
            uint32_t kk[E];
#pragma unroll
            for (int j = 0; j < E; j++) kk[j] = a[min(c0 + j, n - 1)] >> SHIFT;
            int lo_ = 0, hi_ = nseg;
            while (lo_ < hi_) { const int mid = (lo_ + hi_) >> 1; if ((int)sl[mid] > c0) hi_ = mid; else lo_ = mid + 1; }
            for (int s = lo_; s < nseg && npc < 3; s++) {
                const int f = sf[s], l = sl[s];
                if (f >= c1) break;
                const int lo = max(f + 1, c0) - c0, hi = min(l, c1) - c0;
                if (lo < hi) {
                    const uint32_t pv = a[f] >> SHIFT;
                    uint32_t ge = 0, le = 0;
#pragma unroll
                    for (int j = 0; j < E; j++) { ge |= (uint32_t)(kk[j] >= pv) << j; le |= (uint32_t)(kk[j] <= pv) << j; }
                    const uint32_t rm = (hi >= 32 ? 0xffffffffu : ((1u << hi) - 1u)) & ~((1u << lo) - 1u);
                    mL |= ge & rm; mR |= le & rm;
#pragma unroll
                    for (int q = 0; q < 3; q++) if (npc == q) { ps[q] = s; plo[q] = lo; phi[q] = hi; }
                    npc++;
                }
                if (l >= c1) break;
            }
        }
        auto rmask = [&](int q) { return (phi[q] >= 32 ? 0xffffffffu : ((1u << phi[q]) - 1u)) & ~((1u << plo[q]) - 1u); };
        // does the first piece's segment have elements before this chunk / the last piece's segment elements behind it
        const bool contL = npc > 0 && (int)sf[ps[0]] + 1 < c0;
        int lastq = 0;
#pragma unroll
        for (int q = 1; q < 3; q++) if (q < npc) lastq = q;
        int last_s = ps[0];
#pragma unroll
        for (int q = 1; q < 3; q++) if (q < npc) last_s = ps[q];
        const bool contR = npc > 0 && (int)sl[last_s] > c1;
        uint32_t rm_last = 0;
#pragma unroll
        for (int q = 0; q < 3; q++) if (q == lastq && npc > 0) rm_last = rmask(q);
        const uint32_t rm_first = npc > 0 ? rmask(0) : 0u;
        int carryL, carryR;
        seg_scan2<T>(__popc(mL & rm_last), !(npc == 1 && contL), __popc(mR & rm_first), !(npc == 1 && contR), s_buf, carryL, carryR);
        // ---- D: ranks; the left-scan stops that will be swapped publish their positions; the thread that sees x* writes the cut ----
#pragma unroll
        for (int q = 0; q < 3; q++) {
            if (q >= npc) continue;
            const int s = ps[q], f = sf[s], l = sl[s];
            const uint32_t rm = rmask(q);
            int A = (q == 0 && contL) ? carryL : 0;
            int Bf = ((q == lastq && contR) ? carryR : 0) + __popc(mR & rm);
            const uint32_t pv = a[f] >> SHIFT;
            const int base = (f + 1) >> 1;
            bool gprev = false, eLp = false, eRp = false;        // g and the flags of the element before the piece
            if (c0 + plo[q] > f + 1) {                           // (then plo == 0: the element is the previous thread's last one)
                const uint32_t kp = a[c0 - 1] >> SHIFT;
                eLp = kp >= pv; eRp = kp <= pv;
                gprev = (A - (int)eLp) >= (Bf + (int)eRp);
            }
            for (int j = plo[q]; j <= phi[q]; j++) {
                const bool end = j == phi[q];
                if (end && c0 + j != l) break;                   // the virtual position l belongs to the thread that holds l - 1
                const bool isL = !end && ((mL >> j) & 1u), isR = !end && ((mR >> j) & 1u);
                const bool g = A >= Bf;
                if (g && !gprev) {
                    const int Ap = A - (int)eLp, Bp = Bf + (int)eRp;
                    scut[s] = (uint16_t)(c0 + j - ((eLp && eRp && Ap == Bp - 1) ? 1 : 0));
                }
                if (isL && Bf - (int)isR >= A + 1) posh[base + A] = (uint16_t)(c0 + j);
                gprev = g; eLp = isL; eRp = isR;
                A += (int)isL; Bf -= (int)isR;
            }
        }
        __syncthreads();
        // ---- E: the right-scan stops that are swapped fetch their partners ----
#pragma unroll
        for (int q = 0; q < 3; q++) {
            if (q >= npc) continue;
            const int s = ps[q], f = sf[s];
            const uint32_t rm = rmask(q);
            int A = (q == 0 && contL) ? carryL : 0;
            int Bf = ((q == lastq && contR) ? carryR : 0) + __popc(mR & rm);
            const int base = (f + 1) >> 1;
            for (int j = plo[q]; j < phi[q]; j++) {
                const bool isL = (mL >> j) & 1u, isR = (mR >> j) & 1u;
                if (isR && A >= Bf) {
                    const int pL = posh[base + Bf - 1], qq = c0 + j;
                    const uint32_t x = a[pL], y = a[qq];
                    a[pL] = y; a[qq] = x;
                }
                A += (int)isL; Bf -= (int)isR;
            }
        }
        __syncthreads();
        // ---- F: the next level's list (children of more than 16 elements, in order); shorter children are finished now ----
        {
            uint16_t* nf = sfb + (cur ^ 1) * SMAX; uint16_t* nl = slb + (cur ^ 1) * SMAX; uint8_t* nd = sdb + (cur ^ 1) * SMAX;
            int cf[4], cl[4], cd[4], nk = 0;
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int s = 2 * tid + u;
                if (s < nseg) {
                    const int f = sf[s], l = sl[s], cut = scut[s], d = sd[s] > 0 ? sd[s] - 1 : 0;
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        const int xf = h ? cut : f, xl = h ? l : cut;
                        if (xl - xf > 16) {
#pragma unroll
                            for (int z = 0; z < 4; z++) if (nk == z) { cf[z] = xf; cl[z] = xl; cd[z] = d; }
                            nk++;
                        } else if (xl - xf > 1) insertion<SHIFT>(a, xf, xl);
                    }
                }
            }
            int tot;
            const int off = block_exscan<T, int>(nk, s_buf, &tot);
#pragma unroll
            for (int z = 0; z < 4; z++) if (z < nk) { nf[off + z] = (uint16_t)cf[z]; nl[off + z] = (uint16_t)cl[z]; nd[off + z] = (uint8_t)cd[z]; }
            nseg = tot;
            cur ^= 1;
            __syncthreads();
        }
    }
    for (int i = tid; i < n; i += T) arr[span_f + i] = a[i];
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// Global tier
// ---------------------------------------------------------------------------------------------------------------------------------------
constexpr int G_QMAX = 64;       // ranges longer than the LDS tier's capacity alive at one level
constexpr int G_FMAX = 512;      // ranges handed to the LDS tier, per array set

template <int T>
struct GlobalLayout {            // LDS of global_tier for arrays whose longest range has `rows` 64-element rows
    static constexpr int NW = T / 64;
    __host__ __device__ static constexpr int off_Lw(int) { return 0; }
    __host__ __device__ static constexpr int off_Rw(int rows) { return rows * 8; }
    __host__ __device__ static constexpr int off_pL(int rows) { return rows * 16; }
    __host__ __device__ static constexpr int off_pR(int rows) { return rows * 16 + (rows + 1) * 4; }
    __host__ __device__ static constexpr int off_q(int rows) { return (rows * 16 + (rows + 1) * 8 + 7) / 8 * 8; }     // Range [2][G_QMAX]
    __host__ __device__ static constexpr int off_fin(int rows) { return off_q(rows) + 2 * G_QMAX * 12; }              // Range [G_FMAX]
    __host__ __device__ static constexpr int off_rank(int rows) { return off_fin(rows) + G_FMAX * 12; }               // Range [G_FMAX] (sorted copy)
    __host__ __device__ static constexpr int off_buf(int rows) { return off_rank(rows) + G_FMAX * 12; }               // u64 [NW] + ints
    __host__ __device__ static constexpr int bytes(int rows) { return off_buf(rows) + NW * 8 + 64; }
    __host__ __device__ static constexpr int rows_for(int n) { return (n + 63) / 64 + 1; }
};

// One Hoare partition of arr[f, l) (l - f > 16) by the whole workgroup; returns the cut in every thread.
template <int SHIFT, int T>
__device__ int wg_partition(uint32_t* __restrict__ arr, int f, int l, uint8_t* lds, int rows_cap, int* status) {
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
    if (nrow > rows_cap) { if (tid == 0) *status = ST_CAPACITY; return f + (l - f) / 2; }
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
    // ---- pass 1: one read of the range; a row of 64 elements = one ballot per scan ----
    constexpr int U = 4;
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
    // ---- x*: the row where A >= B turns true, then the bit inside it ----
    for (int r = r_begin; r < r_end; r++) {
        const int A0 = (int)pL[r], B0 = totR - (int)pR[r], A1 = (int)pL[r + 1], B1 = totR - (int)pR[r + 1];
        if (!(A0 >= B0) && A1 >= B1) {
            const unsigned long long Lb = Lw[r], Rb = Rw[r];
            int j = 1, A = A0, Bf = B0;
            bool eL = false, eR = false;
            for (; j <= 64; j++) {
                eL = (Lb >> (j - 1)) & 1ull; eR = (Rb >> (j - 1)) & 1ull;
                A += (int)eL; Bf -= (int)eR;
                if (A >= Bf) break;
            }
            const int xs = f + 1 + 64 * r + j, Ap = A - (int)eL, Bp = Bf + (int)eR;
            s_i[3] = xs - ((eL && eR && Ap == Bp - 1) ? 1 : 0);
            s_i[4] = max(Ap, Bf);
        }
    }
    __syncthreads();
    const int cut = s_i[3], m = s_i[4];
    // ---- swaps: the k-th stop of the left scan (k < m) with the k-th of the right scan, lane = element of a row ----
    for (int r = wave; r < nrow; r += NW) {
        const int pl = (int)pL[r];
        if (pl >= m) break;
        const unsigned long long Lb = Lw[r];
        const int k = pl + __popcll(Lb & ((1ull << lane) - 1ull));
        if (((Lb >> lane) & 1ull) && k < m) {
            const int rk = totR - 1 - k;
            int lo = 0, hi = nrow;
            while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if ((int)pR[mid] <= rk) lo = mid; else hi = mid; }
            const int q = f + 1 + 64 * lo + select64(Rw[lo], rk - (int)pR[lo]), p = f + 1 + 64 * r + lane;
            const uint32_t x = arr[p], y = arr[q];
            arr[p] = y; arr[q] = x;
        }
    }
    __threadfence_block();
    __syncthreads();
    return cut;
}

// The workgroup partitions every range of `init` (n_init <= G_FMAX, disjoint) that is longer than n_stage until none is left, then writes the
// resulting ranges sorted by position to out_ranges and packs consecutive ones into LDS-tier blocks (span <= n_stage, <= nr_cap ranges).
// out_counts: [0] ranges, [1] blocks.
template <int SHIFT, int T>
__device__ void global_tier(uint32_t* __restrict__ arr, const Range* __restrict__ init, int n_init, int n_stage, int nr_cap, Range* __restrict__ out_ranges,
                            Block* __restrict__ out_blocks, int max_blocks, int* __restrict__ out_counts, uint8_t* lds, int rows_cap, int* status) {
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
            if (R.l - R.f > n_stage) { if (nq < G_QMAX) qb[nq++] = R; else *status = ST_CAPACITY; }
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
            if (R.d == 0) { if (tid == 0) *status = ST_DEPTH; }
            const int cut = wg_partition<SHIFT, T>(arr, R.f, R.l, lds, rows_cap, status);
            if (tid == 0) {
                const int d = R.d > 0 ? R.d - 1 : 0;
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const Range C{h ? cut : R.f, h ? R.l : cut, d};
                    if (C.l - C.f > n_stage) { if (s_c[1] < G_QMAX) qb[(cur ^ 1) * G_QMAX + s_c[1]++] = C; else *status = ST_CAPACITY; }
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
