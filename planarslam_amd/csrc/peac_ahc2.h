// planarslam_amd/csrc/peac_ahc2.h — the order-dependent clustering of PEAC (ahCluster, reference include/peac/AHCPlaneFitter.hpp:983-1189,
// PlaneSeg::mergeNbsFrom / disconnectAllNbs include/peac/AHCPlaneSeg.hpp:379-404) for ONE wavefront per frame, with LAZY adjacency.
// Included by peac.hip; also compiled for the host by tests/host_shim (wave64 emulator), so it only depends on peac_common.h.
//
// The reference keeps std::set<PlaneSeg*> neighbour sets and, on every merge m = p + nb, erases p and nb from all their neighbours' sets and
// inserts m: for a wavefront that is a chain of dependent global-memory round trips per merge (the previous kernel spent 13 % of a frame in
// those appends and 18 % in list unions).  Here a node's neighbour list is written ONCE, when the node is created ("bag"), and is never
// touched by later merges.  What a bag entry x stands for NOW is found through merge-parent pointers in LDS:
//     mp[x] == x      x is alive              mp[x] == m   x was merged into m (follow the chain)
//     mp[x] == TOMB   x left the graph without a merge (disconnectAllNbs): the adjacency is gone
// By induction over the merge sequence, { live end of the chain of x : x in bag(a) } is exactly the reference's nbs(a) for every live a:
// a merge replaces its two nodes by m in every neighbour's set, which is what following mp does; a disconnect erases the node from every
// set, which is what TOMB does.  Chains are shortened by path halving (only ever re-pointing a node to one of its ancestors: results do not
// depend on it).  The union of a merge is a 6144-bit LDS bitmap: both bags' resolved entries set their bit, rank = prefix popcount, so the
// new bag comes out deduplicated and in ascending node id (= the reference's std::set order) with no searches.
//
// Candidate merges of a node are a pure function of its live-neighbour set, so they are evaluated ahead of time (the popped node together
// with the not-yet-evaluated nodes at the top of the heap, one lane per (node, bag entry): one 3x3 eigen-solve of latency for all of them)
// and kept in the node's candidate record; a `valid` bit per node (LDS) is cleared for every neighbour of a merge / disconnect, exactly when
// the reference's candidate loop would see a different set.  A pop of an evaluated node costs: heap pop (LDS; the node's 272-byte record is
// requested before the pop and arrives behind it), the partner's record load, bitmap union (LDS), stores.
//
// What bounds the kernel: a lone wavefront issues about one instruction every five cycles (profiles/README.md, -DPLANAR_PEAC_TIMING buckets), so
// the code is written for instruction count: heap entries are one 64-bit LDS word (float key | node id | bag size), wave-uniform values are kept in
// SGPRs (wave_ops.h), prefix sums run on DPP, and the common cases (bag entries still alive, both bags in registers) have straight-line paths.
//
// Candidate record, 68 dwords per node (frame workspace, L2-resident):
//   d0      n_roots (low 16: entries of the bag) | flags << 16   (1 have: some neighbour passed the normal test; 2 merge_ok: best mse below
//           T_mse(merge); 4 big: more than 64 entries, the bag lives in the pool at d3)
//   d1      best neighbour (low 16) | own N / 100 << 16          d2  own rid (DisjointSet root id)     d3  pool offset of a big bag
//   d4,d5   mse of the best merge (double)
//   d6..37  the bag: 64 x u16, resolved in place whenever the node is evaluated (TOMB = dropped)
//   d38..67 moments[9], centre[3], normal[3] of the best merge: they become node m's record when the merge is taken
#pragma once
#include "peac_common.h"
#include "wave_ops.h"

namespace planar {
namespace peac {

constexpr unsigned TOMB = 0xFFFFu;
// Single-wavefront code: the LDS traffic of one wavefront is processed in program order, so lanes only need the compiler to keep that order
// (wavefront-scope fence, no s_waitcnt).  GFENCE additionally waits for the wavefront's global stores (workgroup scope): used where lanes read
// global memory other lanes have just written.  Macros, so that the host emulator's divergence check sees the call site.
#define WFENCE() do { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
#define GFENCE() __threadfence_block()

__device__ __forceinline__ void stats_compute_u(const double s[9], int N, Geo& g) {   // PlaneSeg::Stats::compute (AHCPlaneSeg.hpp:125-156), wavefront eigen-solver
    const double sc = 1.0 / N;
    g.center[0] = s[0] * sc; g.center[1] = s[1] * sc; g.center[2] = s[2] * sc;
    const double k00 = s[3] - s[0] * s[0] * sc, k01 = s[6] - s[0] * s[1] * sc, k02 = s[8] - s[0] * s[2] * sc;
    const double k11 = s[4] - s[1] * s[1] * sc, k12 = s[7] - s[1] * s[2] * sc, k22 = s[5] - s[2] * s[2] * sc;
    double ev[3], v[3];
    eig33u(k00, k01, k11, k02, k12, k22, ev, v);
    const bool keep = v[0] * g.center[0] + v[1] * g.center[1] + v[2] * g.center[2] <= 0;
    g.normal[0] = keep ? v[0] : -v[0]; g.normal[1] = keep ? v[1] : -v[1]; g.normal[2] = keep ? v[2] : -v[2];
    g.mse = ev[0] * sc;
}

__host__ __device__ static inline size_t al8(size_t v) { return (v + 7) & ~(size_t)7; }
// LDS bytes of peac_ahc2 for a layout
static inline int ahc2_smem_bytes(const Layout& L) {
    const int W32 = (L.NB2 + 31) / 32;
    return (int)((size_t)L.NB * 8 + al8((size_t)L.NB2 * 2) + (size_t)W32 * 4 * 2 + al8((size_t)W32 * 2));
}

typedef unsigned long long u64;

// host emulator only: after a pruned evaluation of a pooled bag (eval_big) also run the in-order evaluation of ALL its candidates and compare (err 9)
#ifdef PLANAR_WAVE_EMUL
static int g_peac_check_prune = 0;
#define PEAC_CHECK_PRUNE (g_peac_check_prune != 0)
#else
#define PEAC_CHECK_PRUNE false
#endif

constexpr int ST_RETRY = 100;              // status of a frame the fast kernel gave up on (exact FP64 tie between live nodes): peac_ahc2 redoes it
constexpr unsigned K_EMPTY = 0xffffff80u;  // tournament queue: no node (above every key)

// FAST = false: the queue is libstdc++'s binary heap, restated exactly (layout, hence the pop order among EQUAL keys, is the reference's).
// FAST = true:  as long as no two live nodes have bit-equal mse the pop order does not depend on the heap's layout at all - it is "smallest key
//               first" - so the queue may be anything.  Here: K[id] (LDS) = key rounded to float, mapped to an unsigned integer of the same
//               order, low 7 bits replaced by the node's bag size (a monotonic proxy of the FP64 key: proxy(a) < proxy(b) implies a < b; equal
//               proxies are decided on the FP64 keys), K_EMPTY = absent.  Lane L keeps the minimum of column L = ids congruent L mod 64 in registers; top = DPP minimum over the 64 lanes; removing
//               a node clears its slot and recomputes one column (one LDS round trip).  Nodes that die are removed at once, so there are no pops
//               of dead nodes (the reference pops and skips ~2600 of them per 640x480 frame).  Wherever equal proxies meet (inside a column when
//               it is recomputed, across columns at the top, at a push) the FP64 keys are compared, and bit-equal FP64 keys of two live nodes
//               end the attempt with ST_RETRY: the launch of peac_ahc2 that follows redoes exactly those frames with the exact heap.
template <bool FAST>
__device__ __forceinline__ int ahc_frame(const Layout& L, const Consts& C, uint8_t* __restrict__ ws, int32_t* __restrict__ status,
                                          long long* __restrict__ timing, const int frame) {
    PLANAR_DYN_SMEM(smem);
    __shared__ int s_ext[MAX_PLANES];
    __shared__ unsigned s_mark[64];
    const int lane = threadIdx.x;
    uint8_t* F = ws + (size_t)frame * L.frame_bytes;
    double* g_stats = (double*)(F + L.off_stats);
    double* g_geo = (double*)(F + L.off_geo);
    int* g_N = (int*)(F + L.off_N);
    const uint8_t* g_flags = F + L.off_flags;
    uint32_t* crec = (uint32_t*)(F + L.off_crec);
    u16* bpool = (u16*)(F + L.off_bpool);
    u16* h_dsp = (u16*)(F + L.off_h_dsp); u16* h_dss = (u16*)(F + L.off_h_dss); u16* h_rid = (u16*)(F + L.off_h_rid);
    int* g_hand = (int*)(F + L.off_h_hand);
    const int NB = L.NB, NB2 = L.NB2, Nw = L.Nw, Nh = L.Nh, W32 = (NB2 + 31) / 32;

    // merge heap: one 64-bit word per entry = key rounded to float (bits 0..31) | node id (32..47) | bag size at creation, 255 = in the pool (48..55)
    u64* hp = (u64*)smem;
    u16* mp = (u16*)(smem + (size_t)NB * 8);                                            // merge-parent pointers
    unsigned* cval = (unsigned*)((uint8_t*)mp + al8((size_t)NB2 * 2));                  // bit per node: its candidate record is valid
    unsigned* bmp = cval + W32;                                                         // union bitmap (all zero between merges)
    u16* pre = (u16*)(bmp + W32);                                                       // exclusive popcount prefix of the bitmap words

    long long tphase[4];
    int nph = 0;
    auto mark = [&]() { if (nph < 4) tphase[nph++] = (long long)wall_clock64(); };
    mark();
    auto rec = [&](int id) -> uint32_t* { return crec + (size_t)id * CREC_DW; };
    auto roots_of = [&](int id) -> u16* { return (u16*)(rec(id) + 6); };
    auto geo_of = [&](int id) -> double* { return g_geo + (size_t)id * 7; };
    auto nsim = [&](int a, int b) {
        const double* ga = geo_of(a) + 3; const double* gb = geo_of(b) + 3;
        return fabs(ga[0] * gb[0] + ga[1] * gb[1] + ga[2] * gb[2]);
    };
    auto is_valid = [&](int id) { return (cval[id >> 5] >> (id & 31)) & 1u; };
    auto inval = [&](unsigned id) { atomicAnd(&cval[id >> 5], ~(1u << (id & 31))); };
    auto lane_below = [&](int k) -> u64 { return k >= 64 ? ~0ull : (1ull << k) - 1ull; };
    auto e_key = [](u64 e) -> float { return __uint_as_float((unsigned)e); };
    auto e_id = [](u64 e) -> int { return (int)((e >> 32) & 0xffffu); };
    auto e_cnt = [](u64 e) -> int { return (int)((e >> 48) & 0xffu); };
    auto e_make = [](float key, int id, int cnt) -> u64 { return (u64)__float_as_uint(key) | (u64)(unsigned)id << 32 | (u64)(unsigned)(cnt > 64 ? 255 : cnt) << 48; };

    // ---- init ----
    unsigned char* cnt8 = (unsigned char*)hp;                // bag sizes while the graph is built (the heap is built afterwards)
    for (int b = lane; b < NB; b += 64) { cnt8[b] = 0; h_dsp[b] = (u16)b; h_dss[b] = 1; h_rid[b] = (u16)b; }
    for (int t = lane; t < NB2; t += 64) mp[t] = (u16)t;
    for (int t = lane; t < W32; t += 64) { cval[t] = 0; bmp[t] = 0; }
    __syncthreads();

    // ---- initGraph edges (AHCPlaneFitter.hpp:896-954): a lane owns a whole row, then a whole column (the reference's --j / ++j skip logic is a
    //      sequential scan of one row / column).  A block's bag (<= 4 entries, ascending) goes straight into its candidate record.
    auto connect = [&](int a, int b) {
        int ca = cnt8[a]; lst_insert(roots_of(a), ca, b); cnt8[a] = (unsigned char)ca;
        int cb = cnt8[b]; lst_insert(roots_of(b), cb, a); cnt8[b] = (unsigned char)cb;
    };
    auto inG = [&](int idx) { return (g_flags[idx] & 1) != 0; };
    for (int i = lane; i < Nh; i += 64) {
        for (int j = 1; j < Nw; j += 2) {
            const int c = i * Nw + j;
            if (!inG(c - 1)) { --j; continue; }
            if (!inG(c)) continue;
            if (j < Nw - 1 && !inG(c + 1)) { ++j; continue; }
            const double th = T_ang_init(C, geo_of(c)[2]);
            if ((j < Nw - 1 && nsim(c - 1, c + 1) >= th) || (j == Nw - 1 && nsim(c, c - 1) >= th)) {
                connect(c, c - 1);
                if (j < Nw - 1) connect(c, c + 1);
            } else --j;
        }
    }
    __syncthreads();
    for (int j = lane; j < Nw; j += 64) {
        for (int i = 1; i < Nh; i += 2) {
            const int c = i * Nw + j;
            if (!inG(c - Nw)) { --i; continue; }
            if (!inG(c)) continue;
            if (i < Nh - 1 && !inG(c + Nw)) { ++i; continue; }
            const double th = T_ang_init(C, geo_of(c)[2]);
            if ((i < Nh - 1 && nsim(c - Nw, c + Nw) >= th) || (i == Nh - 1 && nsim(c, c - Nw) >= th)) {
                connect(c, c - Nw);
                if (i < Nh - 1) connect(c, c + Nw);
            } else --i;
        }
    }
    __syncthreads();
    for (int b = lane; b < NB; b += 64) {
        uint32_t* r = rec(b);
        r[0] = (uint32_t)cnt8[b];
        r[1] = (uint32_t)(g_N[b] / (WIN * WIN)) << 16;
        r[2] = (uint32_t)b;
        r[3] = 0;
    }
    __syncthreads();
    mark();

    int heap_n = 0, n_nodes = NB, n_ext = 0, err = 0;
    unsigned pool_top = 0;
    long long cyc[24] = {0}, c0 = 0;   // cycle buckets (PEAC_TICK)
    (void)c0;
    // ---- libstdc++ binary heap (std::priority_queue with PlaneSegMinMSECmp), keys rounded to float; equal floats fall back to the FP64 keys in the
    //      node records, so the comparisons - hence layout and pop order - are those of the FP64 heap.  __push_heap: the <= 12 ancestors of the hole
    //      are read by one lane each; the leading run of larger ancestors moves down one level in parallel.
    auto heap_sift_up = [&](int hole, u64 ent) {
        const float mf = e_key(ent);
        const int anc = lane < 16 ? ((hole + 1) >> lane) - 1 : -1;
        const bool isanc = lane >= 1 && anc >= 0;
        u64 E = 0;
        if (isanc) E = hp[anc];
        const float K = e_key(E);
        bool less = isanc && mf < K;
        if (__ballot(isanc && mf == K)) {
            GFENCE();
            const double dv = geo_of(e_id(ent))[6];
            if (isanc && mf == K) less = dv < geo_of(e_id(E))[6];
        }
        const u64 up = __ballot(less) >> 1;
        const int n = __builtin_ctzll(~up);
        if (lane >= 1 && lane <= n) hp[((hole + 1) >> (lane - 1)) - 1] = E;
        if (lane == 0) hp[((hole + 1) >> n) - 1] = ent;
        WFENCE();
    };
    auto heap_push = [&](int id, double mse, int cnt) { heap_n++; heap_sift_up(heap_n - 1, e_make((float)mse, id, cnt)); };
    // pop_heap = __adjust_heap(first, 0, len, last value): the hole sinks to the bottom along the smaller child (no early exit), then the value
    // climbs back.  Lane t = 1..63 stands for node t of the subtree under the hole: six levels per LDS round trip.
    const int dl = 31 - __clz(max(lane, 1));                 // level of local node `lane` in the 63-node subtree
    auto heap_pop = [&]() {
        const u64 vlast = hp[heap_n - 1];
        heap_n--;
        const int len = heap_n;
        if (len == 0) return;
        const int half = (len - 1) / 2;
        int hole = 0;
        while (hole < half) {
            const int g = (hole << dl) + lane - 1;
            const bool inner = lane >= 1 && g < half;
            u64 el = 0, er = 0;
            if (inner) { el = hp[2 * g + 1]; er = hp[2 * g + 2]; }
            const float kl = e_key(el), kr = e_key(er);
            bool lt = inner && kl < kr;
            if (__ballot(inner && kl == kr)) { GFENCE(); if (inner && kl == kr) lt = geo_of(e_id(el))[6] < geo_of(e_id(er))[6]; }
            const u64 two = __ballot(inner);
            const u64 takel = __ballot(lt);
            int cur = 1;
            u64 path = 0;
#pragma unroll
            for (int d = 0; d < 6; d++) {
                if (!((two >> cur) & 1ull)) break;
                path |= 1ull << cur;
                cur = 2 * cur + 1 - (int)((takel >> cur) & 1ull);
            }
            if ((path >> lane) & 1ull) hp[g] = ((takel >> lane) & 1ull) ? el : er;
            const int dc = 31 - __clz(cur);
            hole = (hole << dc) + cur - 1;
        }
        WFENCE();
        PEAC_TICK(0);
        if ((len & 1) == 0 && hole == (len - 2) / 2) {
            const int c = 2 * hole + 1;
            const u64 ce = hp[c];
            if (lane == 0) hp[hole] = ce;
            WFENCE(); hole = c;
        }
        heap_sift_up(hole, vlast);
        PEAC_TICK(1);
    };


    // ---- FAST: the tournament queue ----
    unsigned* K = (unsigned*)smem;                            // [NB2] (the heap's LDS)
    float top_key = 0.f, la_margin = 0.5f;                    // proxy key of the node being popped; lookahead margin (eval_phase)
    unsigned cm_v = K_EMPTY;                                  // this lane's column minimum: raw K value ...
    int cm_id = -1;                                           // ... and node id
    auto kproxy = [](unsigned v) -> unsigned { return v & ~0x7fu; };
    auto k_of = [](double mse, int cnt) -> unsigned {         // float bits -> unsigned of the same order (sign flip / complement), bag size in the low bits
        const unsigned b = __float_as_uint((float)mse);
        return ((b ^ ((b >> 31) ? 0xffffffffu : 0x80000000u)) & ~0x7fu) | (unsigned)(cnt > 64 ? 127 : cnt);
    };
    auto k_float = [](unsigned v) -> float { const unsigned o = v & ~0x7fu; return __uint_as_float((o >> 31) ? (o ^ 0x80000000u) : ~o); };
    struct Pick { int id; double d; bool any, tie; };
    auto pick_add = [&](Pick& P, int id) {                    // smallest FP64 key among nodes whose proxies are equal; bit-equal keys are a tie
        const double d = geo_of(id)[6];
        if (!P.any || d < P.d) { P.d = d; P.id = id; P.any = true; P.tie = false; }
        else if (d == P.d) P.tie = true;
    };
    // (frames of more than 4096 blocks - 8192 node ids, two per lane and column - take the general form: any number of ids per lane)
    auto col_recompute_large = [&](int Lc) {
        unsigned mn = K_EMPTY;
        for (int i = Lc + 64 * lane; i < NB2; i += 4096) mn = min(mn, kproxy(K[i]));
        mn = wave_min_u32(mn);
        int id = -1; unsigned v = K_EMPTY;
        if (mn != K_EMPTY) {
            int nmatch = 0, first = -1;
            for (int base = 0; base < NB2; base += 4096) {
                const int i = base + Lc + 64 * lane;
                const u64 m = __ballot(i < NB2 && kproxy(K[min(i, NB2 - 1)]) == mn);
                if (m) { if (first < 0) first = base + Lc + 64 * (__ffsll((long long)m) - 1); nmatch += __popcll(m); }
            }
            if (nmatch == 1) id = first;
            else {
                GFENCE();
                Pick P{-1, 0.0, false, false};
                for (int base = 0; base < NB2; base += 4096) {
                    const int i = base + Lc + 64 * lane;
                    for (u64 m = __ballot(i < NB2 && kproxy(K[min(i, NB2 - 1)]) == mn); m; m &= m - 1) pick_add(P, base + Lc + 64 * (__ffsll((long long)m) - 1));
                }
                if (P.tie) err = ST_RETRY;
                id = P.id;
            }
            v = K[id];
        }
        if (lane == Lc) { cm_v = v; cm_id = id; }
    };
    auto col_recompute = [&](int Lc) {                        // Lc wave-uniform
        if (NB2 > 8192) { col_recompute_large(Lc); return; }
        const int i1 = Lc + 64 * lane, i2 = i1 + 4096;
        const unsigned v1 = i1 < NB2 ? K[i1] : K_EMPTY, v2 = i2 < NB2 ? K[i2] : K_EMPTY;
        const unsigned p1 = kproxy(v1), p2 = kproxy(v2);
        const unsigned mn = wave_min_u32(min(p1, p2));
        int id = -1; unsigned v = K_EMPTY;
        if (mn != K_EMPTY) {
            const u64 e1 = __ballot(p1 == mn), e2 = __ballot(p2 == mn);
            if (__popcll(e1) + __popcll(e2) == 1) {
                const int j = e1 ? __ffsll((long long)e1) - 1 : __ffsll((long long)e2) - 1;
                id = Lc + 64 * j + (e1 ? 0 : 4096);
                v = e1 ? wave_lane(v1, j) : wave_lane(v2, j);
            } else {
                GFENCE();
                Pick P{-1, 0.0, false, false};
                for (u64 m = e1; m; m &= m - 1) pick_add(P, Lc + 64 * (__ffsll((long long)m) - 1));
                for (u64 m = e2; m; m &= m - 1) pick_add(P, Lc + 64 * (__ffsll((long long)m) - 1) + 4096);
                if (P.tie) err = ST_RETRY;
                id = P.id; v = K[id];
            }
        }
        if (lane == Lc) { cm_v = v; cm_id = id; }
    };
    auto pq_top = [&]() -> int {                              // the live node with the smallest key, -1: the queue is empty
        const unsigned pm = kproxy(cm_v);
        const unsigned mn = wave_min_u32(pm);
        if (mn == K_EMPTY) return -1;
        top_key = k_float(mn);
        const u64 eq = __ballot(pm == mn);
        if (__popcll(eq) == 1) return wave_lane(cm_id, __ffsll((long long)eq) - 1);
        GFENCE();
        Pick P{-1, 0.0, false, false};
        for (u64 m = eq; m; m &= m - 1) pick_add(P, wave_lane(cm_id, __ffsll((long long)m) - 1));
        if (P.tie) err = ST_RETRY;
        return P.id;
    };
    auto pq_remove = [&](int id) {                            // id wave-uniform; its column is recomputed when it was the column's minimum
        if (lane == 0) K[id] = K_EMPTY;
        WFENCE();
        if (wave_lane(cm_id, id & 63) == id) col_recompute(id & 63);
    };
    auto pq_push = [&](int id, double mse, int cnt) {
        if (!(mse < 3.0e38) || !(mse > -3.0e38)) err = ST_RETRY;   // NaN / infinite key: leave it to the exact kernel
        const unsigned v = k_of(mse, cnt);
        if (lane == 0) K[id] = v;
        WFENCE();
        const int Lc = id & 63;
        const unsigned pn = kproxy(v), pc = kproxy(wave_lane(cm_v, Lc));
        if (pn < pc) { if (lane == Lc) { cm_v = v; cm_id = id; } }
        else if (pn == pc) col_recompute(Lc);
    };

    // ---- what a bag entry stands for now: follow mp to the live node (or TOMB), halving the path on the way.  All 64 lanes call it together.
    auto chase = [&](unsigned x) -> unsigned {
        bool done = x == TOMB;
        int guard = 0;
        while (true) {                                         // two levels per round: an entry that is alive, or whose owner is, is settled in the first
            const unsigned par = done ? x : (unsigned)mp[x];
            const unsigned g = mp[(done || par == TOMB) ? 0u : par];
            if (!done) {
                if (par == TOMB) { x = TOMB; done = true; }
                else if (g == par) { x = par; done = true; }   // par is alive (par == x: the entry itself)
                else { mp[x] = (u16)g; x = g; if (g == TOMB) done = true; }
            }
            if (!__ballot(!done)) break;
            if (++guard > 8192) { err = 7; break; }            // mp only ever points to newer nodes: cannot happen; keeps a corrupted workspace from hanging the GPU
        }
        return x;
    };
    // exclusive prefix of the bitmap words' popcounts -> pre[]; returns the number of set bits.  Lane j owns words 3j .. 3j+2.
    unsigned bw[3];
    // (frames of more than 3072 blocks: more than 192 bitmap words = three per lane; the general forms walk the bitmap 192 words at a time and re-read the words)
    auto bitmap_prefix_large = [&]() -> int {
        int carry = 0;
        for (int w0 = 0; w0 < W32; w0 += 192) {
            int c[3];
#pragma unroll
            for (int k = 0; k < 3; k++) { const int w = w0 + 3 * lane + k; c[k] = w < W32 ? __popc(bmp[w]) : 0; }
            const int tot = c[0] + c[1] + c[2];
            const int incl = wave_scan_add(tot);
            const int ex = carry + incl - tot;
            if (w0 + 3 * lane < W32) pre[w0 + 3 * lane] = (u16)ex;
            if (w0 + 3 * lane + 1 < W32) pre[w0 + 3 * lane + 1] = (u16)(ex + c[0]);
            if (w0 + 3 * lane + 2 < W32) pre[w0 + 3 * lane + 2] = (u16)(ex + c[0] + c[1]);
            carry += wave_lane(incl, 63);
        }
        WFENCE();
        return carry;
    };
    auto bitmap_emit_large = [&](u16* dst, bool with_inval) {
        for (int w0 = 0; w0 < W32; w0 += 192) {
            if (w0 + 3 * lane >= W32) continue;
            int o = (int)pre[w0 + 3 * lane];
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const int wi = w0 + 3 * lane + k;
                if (wi >= W32) break;
                unsigned w = bmp[wi];
                const int base = wi * 32;
                if (w) bmp[wi] = 0;
                while (w) {
                    const int b = __ffs((int)w) - 1;
                    w &= w - 1;
                    dst[o++] = (u16)(base + b);
                    if (with_inval) inval((unsigned)(base + b));
                }
            }
        }
        WFENCE();
    };
    auto bitmap_prefix = [&]() -> int {
        if (W32 > 192) return bitmap_prefix_large();
        int c[3];
#pragma unroll
        for (int k = 0; k < 3; k++) { const int w = 3 * lane + k; bw[k] = w < W32 ? bmp[w] : 0u; c[k] = __popc(bw[k]); }
        const int tot = c[0] + c[1] + c[2];
        const int incl = wave_scan_add(tot);
        const int ex = incl - tot;
        if (3 * lane < W32) pre[3 * lane] = (u16)ex;
        if (3 * lane + 1 < W32) pre[3 * lane + 1] = (u16)(ex + c[0]);
        if (3 * lane + 2 < W32) pre[3 * lane + 2] = (u16)(ex + c[0] + c[1]);
        WFENCE();
        return wave_lane(incl, 63);
    };
    // the lanes that own bitmap words write their set bits, ascending, to dst[ex ...] and clear the words; with_inval: the ids' valid bits are cleared
    auto bitmap_emit = [&](u16* dst, bool with_inval) {
        if (W32 > 192) { bitmap_emit_large(dst, with_inval); return; }
        int o = (int)pre[min(3 * lane, W32 - 1)];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            unsigned w = bw[k];
            const int base = (3 * lane + k) * 32;
            while (w) {
                const int b = __ffs((int)w) - 1;
                w &= w - 1;
                dst[o++] = (u16)(base + b);
                if (with_inval) inval((unsigned)(base + b));
            }
            if (bw[k]) bmp[3 * lane + k] = 0;
        }
        WFENCE();
    };
    auto set_bit = [&](unsigned id) { atomicOr(&bmp[id >> 5], 1u << (id & 31)); };

    int dbg_hits = 0, dbg_phases = 0, dbg_nodes = 0, dbg_big = 0, dbg_prune_skip = 0, dbg_bigsolves = 0;
    (void)dbg_prune_skip;

    // One candidate merge per lane: node `nd` with its live neighbour `r` (or r == TOMB: none).
    // Returns ok (the neighbour passed the normal-similarity test) and the merged moments / plane / N.
    auto eval_pair = [&](int nd, unsigned r, double ms[9], Geo& mg, int& mN) -> bool {
        bool ok = false;
        mN = 0; mg.mse = 0;
#pragma unroll
        for (int t = 0; t < 9; t++) ms[t] = 0;
#pragma unroll
        for (int t = 0; t < 3; t++) { mg.center[t] = 0; mg.normal[t] = 0; }
        if (r != TOMB) {
            const double* sp = g_stats + (size_t)nd * 9;
            const double* gp = geo_of(nd) + 3;
            const double* gn = geo_of((int)r) + 3;
            const double* sb = g_stats + (size_t)r * 9;
            const double pn0 = gp[0], pn1 = gp[1], pn2 = gp[2], n0 = gn[0], n1 = gn[1], n2 = gn[2];
            double ps[9];
#pragma unroll
            for (int t = 0; t < 9; t++) { ps[t] = sp[t]; ms[t] = sb[t]; }
            const int Na = g_N[nd], Nb = g_N[r];
            if (!(fabs(pn0 * n0 + pn1 * n1 + pn2 * n2) < C.cos_merge)) {        // AHCPlaneFitter.hpp:1035
#pragma unroll
                for (int t = 0; t < 9; t++) ms[t] = ps[t] + ms[t];
                mN = Na + Nb;
                ok = true;
            }
        }
        const u64 any_ok = __ballot(ok);
        PEAC_TICK(14);
        if (any_ok) {
            Geo g2;
            double s2[9];
#pragma unroll
            for (int t = 0; t < 9; t++) s2[t] = ok ? ms[t] : (t >= 3 && t < 6 ? 1.0 : 0.0);   // idle lanes solve a harmless diagonal matrix
            stats_compute_u(s2, ok ? mN : 1, g2);
            if (ok) mg = g2;
        }
        return ok;
    };
    // the winner's record: merged moments / centre / normal into dwords 38..67 of node nd's candidate record
    auto write_merged = [&](int nd, const double ms[9], const Geo& mg) {
        double* o = (double*)(rec(nd) + 38);
#pragma unroll
        for (int t = 0; t < 9; t++) o[t] = ms[t];
#pragma unroll
        for (int t = 0; t < 3; t++) { o[9 + t] = mg.center[t]; o[12 + t] = mg.normal[t]; }
    };

    // ---- evaluation phase for a popped node p with a small bag (cp entries) whose record is not valid: p and the live, not-yet-valid small
    //      nodes among the first 64 heap slots are packed into the 64 lanes (one lane per bag entry), resolved, evaluated and folded.
    auto eval_phase = [&](int p, int cp) {
        // lookahead candidates: the first 64 heap slots / the 64 column minima
        int hq, hc;
        bool cand;
        if constexpr (FAST) { hq = cm_id; hc = (int)(cm_v & 0x7fu); cand = hq >= 0 && hc >= 1 && hc <= 64 && !is_valid(hq); }
        else {
            const u64 he = lane < heap_n ? hp[lane] : 0ull;
            hq = e_id(he); hc = e_cnt(he);
            cand = lane < heap_n && hc >= 1 && hc <= 64 && mp[hq] == hq && !is_valid(hq);
        }
        if constexpr (FAST) {
            // The column minima are not the 64 smallest keys; evaluating a node long before it pops is wasted when a neighbour dies in between.
            // Only minima within a margin above the popped node's key are taken: the margin tunes itself towards ~16 candidates per phase
            // (the choice affects nothing but the number of phases).
            const float kp = k_float(cm_v), span = fabsf(top_key) + 1e-20f;
            bool c2 = cand;
            for (int it = 0; it < 4; it++) {
                c2 = cand && kp <= top_key + la_margin * span;
                const int nc = __popcll(__ballot(c2));
                if (nc < 5) la_margin *= 1.6f; else if (nc > 10) la_margin *= 0.7f; else break;
            }
            cand = c2;
        }
        const int v = cand ? hc : 0;
        const int incl = wave_scan_add(v);
        const bool sel = cand && cp + incl <= 64;             // the prefix is monotonic: the selected nodes are a prefix of the candidates
        const u64 selm = __ballot(sel);
        const int total = cp + (selm ? wave_lane(incl, 63 - __builtin_clzll(selm | 1ull)) : 0);
        const int mypos = cp + incl - v;
        // segment heads: node | bag size << 16 | 1 << 31 at the segment's first lane; the popped node's segment starts at lane 0
        s_mark[lane] = lane == 0 ? ((unsigned)p | (unsigned)cp << 16 | 0x80000000u) : 0u;
        WFENCE();
        if (sel) s_mark[mypos] = (unsigned)hq | (unsigned)hc << 16 | 0x80000000u;
        WFENCE();
        const unsigned mk = s_mark[lane];
        const u64 heads = __ballot(mk != 0);
        const int hpos = 63 - __builtin_clzll(heads & (lane_below(lane) | (1ull << lane)));   // my segment's first lane
        const unsigned mkh = __shfl(mk, hpos);
        const int nd = (int)(mkh & 0xffffu), cnt = (int)((mkh >> 16) & 0xffu);
        const int k = lane - hpos;
        const bool active = lane < total;
        PEAC_TICK(12);
        unsigned e = TOMB;
        if (active) e = roots_of(nd)[k];
        const unsigned r = chase(e);
        if (active && r != e) roots_of(nd)[k] = (u16)r;       // the bag is resolved in place
        PEAC_TICK(13);
        double ms[9]; Geo mg; int mN;
        const bool ok = eval_pair(active ? nd : 0, active ? r : TOMB, ms, mg, mN);
        PEAC_TICK(15);
        // fold per segment.  The reference scans the neighbours in ascending id (:1043-1049): take a candidate if none yet, or its mse is smaller, or
        // (equal mse and best.N < mse - quirk).  Without exact ties / NaNs that is the minimum mse: a segmented prefix-min of the mse alone; the
        // winner is the first lane of the segment that holds the minimum (entries that resolved to the same node hold equal values).
        double rm = ok ? mg.mse : 1.7976931348623157e308;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const double om = __shfl_up(rm, o);
            if (lane - o >= hpos && om < rm) rm = om;
        }
        const int last = min(hpos + max(cnt, 1) - 1, 63);
        double min_m = __shfl(rm, last);
        const u64 segm = lane_below(hpos + max(cnt, 1)) & ~lane_below(hpos);
        const u64 eqm = __ballot(ok && mg.mse == min_m) & segm;
        bool have = eqm != 0;
        int win = have ? __ffsll((long long)eqm) - 1 : hpos;
        int w_nb = __shfl((int)r, win);
        const bool odd = active && ok && (mg.mse != mg.mse || (mg.mse == min_m && (int)r != w_nb));
        u64 oddm = __ballot(odd);
        while (oddm) {   // exact ties or NaNs inside a segment: the reference's in-order rule over the DISTINCT neighbours in ascending id (rare)
            const int ol = __ffsll((long long)oddm) - 1;
            const int shp = wave_lane(hpos, ol), scnt = wave_lane(cnt, ol);
            const bool mine = active && lane >= shp && lane < shp + scnt;
            bool f_have = false; double f_mse = 0; int f_N = 0, f_lane = 0, last_id = -1;
            while (true) {
                int cid = (mine && ok && (int)r > last_id) ? (int)r : 0x7fffffff;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) cid = min(cid, __shfl_xor(cid, o));
                cid = wave_uni(cid);
                if (cid == 0x7fffffff) break;
                const u64 who = __ballot(mine && ok && (int)r == cid);
                const int sl = __ffsll((long long)who) - 1;
                const double c_mse = wave_lane(mg.mse, sl); const int c_N = wave_lane(mN, sl);
                if (!f_have || f_mse > c_mse || (f_mse == c_mse && (double)f_N < c_mse)) { f_have = true; f_mse = c_mse; f_N = c_N; f_lane = sl; }   // quirk :1045
                last_id = cid;
            }
            if (mine) { have = f_have; win = f_lane; min_m = f_mse; }
            oddm &= ~(lane_below(shp + scnt) & ~lane_below(shp));
            w_nb = __shfl((int)r, win);
        }
        PEAC_TICK(16);
        // the winner lane publishes the merged record; the segment's first lane the header
        const double w_z = __shfl(mg.center[2], win);
        if (active && have && lane == win) write_merged(nd, ms, mg);
        if (active && k == 0) {
            uint32_t* rr = rec(nd);
            const bool mok = have && min_m < T_mse_merge(w_z);                     // AHCPlaneFitter.hpp:1057
            const int ownN = g_N[nd] / (WIN * WIN);
            rr[0] = (uint32_t)cnt | ((have ? 1u : 0u) | (mok ? 2u : 0u)) << 16;
            rr[1] = (uint32_t)(have ? w_nb : 0) | (uint32_t)ownN << 16;
            *(double*)(rr + 4) = have ? min_m : 0.0;
            atomicOr(&cval[nd >> 5], 1u << (nd & 31));
        }
        dbg_phases++; dbg_nodes += __popcll(heads & lane_below(total));
    };

    // ---- a popped node whose bag lives in the pool (more than 64 entries when it was created).
    // In-order evaluation of a resolved, deduplicated bag of n2 entries SORTED by id, 64 at a time: the reference's scan (:1043-1049), quirk included.
    auto eval_big_inorder = [&](int p, int n2, unsigned off) {
        bool have = false; double best_mse = 0; int best_nb = 0, best_N = 0;
        double best_stats[9]; Geo best_geo;
#pragma unroll
        for (int t = 0; t < 9; t++) best_stats[t] = 0;
#pragma unroll
        for (int t = 0; t < 3; t++) { best_geo.center[t] = 0; best_geo.normal[t] = 0; }
        best_geo.mse = 0;
        for (int k0 = 0; k0 < n2; k0 += 64) {
            const unsigned r = k0 + lane < n2 ? (unsigned)bpool[off + k0 + lane] : TOMB;
            double ms[9]; Geo mg; int mN;
            const bool ok = eval_pair(p, r, ms, mg, mN);
            PEAC_TICK(15);
            const u64 okm = __ballot(ok);
            if (okm) {
                double rm = ok ? mg.mse : 1.7976931348623157e308;
                int rl = ok ? lane : 64;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    const double om = __shfl_xor(rm, o); const int ol = __shfl_xor(rl, o);
                    if (om < rm || (om == rm && ol < rl)) { rm = om; rl = ol; }
                }
                rl = wave_uni(rl);
                const bool tie = __popcll(__ballot(ok && mg.mse == rm)) > 1 || __ballot(ok && mg.mse != mg.mse);
                u64 scan = tie ? okm : (1ull << rl);
                while (scan) {   // lanes are in ascending id: this is the reference's scan
                    const int src = __ffsll((long long)scan) - 1;
                    scan &= scan - 1;
                    const double c_mse = wave_lane(mg.mse, src);
                    if (!have || best_mse > c_mse || (best_mse == c_mse && (double)best_N < c_mse)) {   // quirk :1045
                        have = true; best_mse = c_mse; best_nb = wave_lane((int)r, src); best_N = wave_lane(mN, src);
#pragma unroll
                        for (int t = 0; t < 9; t++) best_stats[t] = wave_lane(ms[t], src);
#pragma unroll
                        for (int t = 0; t < 3; t++) { best_geo.center[t] = wave_lane(mg.center[t], src); best_geo.normal[t] = wave_lane(mg.normal[t], src); }
                        best_geo.mse = c_mse;
                    }
                }
            }
        }
        if (lane == 0) {
            uint32_t* rr = rec(p);
            const bool mok = have && best_mse < T_mse_merge(best_geo.center[2]);
            if (have) write_merged(p, best_stats, best_geo);
            rr[0] = (uint32_t)n2 | ((have ? 1u : 0u) | (mok ? 2u : 0u) | 4u) << 16;
            rr[1] = (uint32_t)(have ? best_nb : 0) | (rr[1] & 0xffff0000u);
            *(double*)(rr + 4) = have ? best_mse : 0.0;
        }
    };
    // The bag is resolved, deduplicated (first occurrence through the bitmap) and packed in place, unsorted.  Such a node has 100-250 live neighbours
    // - three or four eigen-solves of latency if all are evaluated - but only the SMALLEST merged mse matters.  So: the merged moments of every
    // candidate are formed in registers (<= 4 per lane) together with a rigorous lower bound of the mse the solver would return
    // (merged_mse_lower_bound, peac_eig.h); every lane solves its most promising candidate; candidates whose bound lies above the best solved mse
    // cannot win or tie and are dropped; the few that remain go through another round.  Typically one solve per node.  Exact ties / NaNs (the
    // reference's in-order rule matters) and bags above 256 entries sort the bag and take the in-order path.
    auto eval_big = [&](int p, int cp, unsigned off) -> int {
        int n2 = 0;
        const double INF = __builtin_inf(), BIG = 1.7976931348623157e308;
        bool fallback = false;
        {
            const double* sp = g_stats + (size_t)p * 9;
            const double* gp = geo_of(p) + 3;
            double ps[9];
#pragma unroll
            for (int t = 0; t < 9; t++) ps[t] = sp[t];
            const double pn0 = gp[0], pn1 = gp[1], pn2 = gp[2];
            const int Na = g_N[p];
            double b_ms[9], b_mse = INF; Geo b_g; unsigned b_r = TOMB; bool b_have = false, odd = false;
#pragma unroll
            for (int t = 0; t < 9; t++) b_ms[t] = 0;
#pragma unroll
            for (int t = 0; t < 3; t++) { b_g.center[t] = 0; b_g.normal[t] = 0; }
            b_g.mse = 0;
            // the best solved mse, wave-uniform.  "None yet" is DBL_MAX, not infinity: hipcc (ROCm 7.2) materialises a uniform +inf with
            // s_mov_b64 and a 64-bit literal, which gfx950 does not have - the register ends up 0 (tests/test_build_sanity.py scans for it).
            double bestm = BIG;
            // 256 bag entries at a time, four per lane.  In a batch the frame workspaces do not fit the caches: every dependent global access is an HBM
            // round trip (2-4 k cycles), so each kind of load is issued for all four chunks before anything waits for it: entries, then the
            // candidates' normals and N, then their moments.
            for (int g0 = 0; g0 < cp; g0 += 256) {
                unsigned cr[4];
#pragma unroll
                for (int c = 0; c < 4; c++) cr[c] = g0 + c * 64 + lane < cp ? (unsigned)bpool[off + g0 + c * 64 + lane] : TOMB;
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    if (g0 + c * 64 >= cp) break;
                    const unsigned r = chase(cr[c]);
                    bool keep = false;
                    if (r != TOMB) { const unsigned bit = 1u << (r & 31); keep = !(atomicOr(&bmp[r >> 5], bit) & bit); }
                    const u64 km = __ballot(keep);
                    if (keep) bpool[off + n2 + __popcll(km & lane_below(lane))] = (u16)r;    // the resolved bag, packed in place: at or below the entries read so far
                    n2 += __popcll(km);
                    cr[c] = keep ? r : TOMB;                                                  // this lane's candidate of chunk c
                }
                PEAC_TICK(19);
                double cms[4][9], clb[4], cn[4][3]; int cN[4];
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const int idx = cr[c] == TOMB ? p : (int)cr[c];
                    const double* gn = geo_of(idx) + 3;
                    cn[c][0] = gn[0]; cn[c][1] = gn[1]; cn[c][2] = gn[2];
                    cN[c] = g_N[idx];
                }
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const double* sb = g_stats + (size_t)(cr[c] == TOMB ? p : (int)cr[c]) * 9;
#pragma unroll
                    for (int t = 0; t < 9; t++) cms[c][t] = sb[t];
                }
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const bool ok = cr[c] != TOMB && !(fabs(pn0 * cn[c][0] + pn1 * cn[c][1] + pn2 * cn[c][2]) < C.cos_merge);   // AHCPlaneFitter.hpp:1035
#pragma unroll
                    for (int t = 0; t < 9; t++) cms[c][t] = ps[t] + cms[c][t];
                    cN[c] = Na + cN[c];
                    if (!ok) cr[c] = TOMB;
                    const double lb = merged_mse_lower_bound(cms[c], cN[c]);
                    clb[c] = ok ? lb : INF;
                }
                PEAC_TICK(20);
                while (true) {
                    int sel = 0; double lbs = clb[0];
#pragma unroll
                    for (int c = 1; c < 4; c++) if (clb[c] < lbs) { lbs = clb[c]; sel = c; }
                    const bool go = lbs < INF && lbs <= bestm;
                    if (!__ballot(go)) break;
                    double s2[9]; int N2 = 1; unsigned r2 = TOMB;
#pragma unroll
                    for (int t = 0; t < 9; t++) s2[t] = t >= 3 && t < 6 ? 1.0 : 0.0;          // idle lanes solve a harmless diagonal matrix
#pragma unroll
                    for (int c = 0; c < 4; c++) if (go && sel == c) {
#pragma unroll
                        for (int t = 0; t < 9; t++) s2[t] = cms[c][t];
                        N2 = cN[c]; r2 = cr[c]; clb[c] = INF;
                    }
                    Geo g2;
                    stats_compute_u(s2, N2, g2);
                    if (go) {
                        if (!(g2.mse < BIG)) odd = true;                            // NaN / infinite: the in-order path decides
                        else if (!b_have || g2.mse < b_mse) {
                            b_have = true; b_mse = g2.mse; b_g = g2; b_r = r2;
#pragma unroll
                            for (int t = 0; t < 9; t++) b_ms[t] = s2[t];
                        } else if (g2.mse == b_mse) odd = true;
                    }
                    bestm = wave_min_f64(b_have ? b_mse : BIG);
                    dbg_bigsolves++;
                    PEAC_TICK(15);
                }
            }
            for (int t = lane; t < W32; t += 64) bmp[t] = 0;
            GFENCE();
            const u64 eqm = __ballot(b_have && b_mse == bestm);
            if (__ballot(odd) || __popcll(eqm) > 1) { fallback = true; dbg_prune_skip = 1; }
            else {
                const bool have = eqm != 0;
                const int wl = have ? __ffsll((long long)eqm) - 1 : 0;
                if (have && lane == wl) write_merged(p, b_ms, b_g);
                const double w_z = wave_lane(b_g.center[2], wl);
                const unsigned w_nb = wave_lane(b_r, wl);
                if (lane == 0) {
                    uint32_t* rr = rec(p);
                    const bool mok = have && bestm < T_mse_merge(w_z);
                    rr[0] = (uint32_t)n2 | ((have ? 1u : 0u) | (mok ? 2u : 0u) | 4u) << 16;
                    rr[1] = (uint32_t)(have ? w_nb : 0u) | (rr[1] & 0xffff0000u);
                    *(double*)(rr + 4) = have ? bestm : 0.0;
                }
                if (PEAC_CHECK_PRUNE) fallback = true;
            }
        }
        if (fallback) {
            uint32_t chk0 = 0, chk1 = 0; double chkm = 0;
            if (PEAC_CHECK_PRUNE) { GFENCE(); chk0 = rec(p)[0]; chk1 = rec(p)[1]; chkm = *(const double*)(rec(p) + 4); WFENCE(); }
            for (int k0 = 0; k0 < n2; k0 += 64) if (k0 + lane < n2) set_bit((unsigned)bpool[off + k0 + lane]);
            WFENCE();
            bitmap_prefix();
            bitmap_emit(bpool + off, false);
            GFENCE();
            eval_big_inorder(p, n2, off);
            dbg_bigsolves += (n2 + 63) / 64;
            GFENCE();
            if (PEAC_CHECK_PRUNE && dbg_prune_skip == 0 && (rec(p)[0] != chk0 || ((rec(p)[0] >> 16) & 1u && (rec(p)[1] != chk1 || *(const double*)(rec(p) + 4) != chkm)))) err = 9;
            dbg_prune_skip = 0;
        }
        dbg_big++;
        return n2;
    };

    // ---- the queue of the initial blocks ----
    if constexpr (FAST) {
        for (int b = lane; b < NB2; b += 64) {
            unsigned v = K_EMPTY;
            if (b < NB && (g_flags[b] & 1)) v = k_of(geo_of(b)[6], (int)(rec(b)[0] & 0x7fu));
            K[b] = v;
        }
        WFENCE();
        bool tie = false;
        for (int i = lane; i < NB2; i += 64) {                 // every lane scans its own column
            const unsigned v = K[i];
            const unsigned pv = kproxy(v), pc = kproxy(cm_v);
            if (pv < pc) { cm_v = v; cm_id = i; }
            else if (pv == pc && pv != K_EMPTY) {
                const double a = geo_of(i)[6], b = geo_of(cm_id)[6];
                if (a < b) { cm_v = v; cm_id = i; } else if (a == b) tie = true;
            }
        }
        if (__ballot(tie)) err = ST_RETRY;
    } else
    // heap of the initial blocks, in block order (:809)
    for (int b0 = 0; b0 < NB; b0 += 64) {
        const int b = b0 + lane;
        const bool in = b < NB && (g_flags[b] & 1);
        const double m = in ? geo_of(b)[6] : 0.0;
        const int c = in ? (int)(rec(b)[0] & 0xffffu) : 0;      // bag size (cnt8 shares its LDS with the heap)
        u64 mask = __ballot(in);
        while (mask) {
            const int src = __ffsll((long long)mask) - 1;
            mask &= mask - 1;
            heap_push(b0 + src, wave_lane(m, src), wave_lane(c, src));
        }
    }
    mark();

    // ---- ahCluster (:983-1189) ----
    // A record is read the way it is used: the header as wave-uniform loads (every lane the same address: the values land in SGPRs without
    // shuffles), bag entry `lane` and dword `lane` of the best merge per lane.
    struct RecView { uint4 h; double mse; unsigned root; uint32_t mw; };
    auto load_rec = [&](const uint32_t* rp) -> RecView {
        RecView r;
        r.h = *(const uint4*)rp;                               // d0..d3
        r.mse = *(const double*)(rp + 4);                      // d4, d5
        r.root = ((const u16*)(rp + 6))[lane];                 // bag entry `lane` (the bag area holds 64 entries)
        r.mw = lane < 30 ? rp[38 + lane] : 0u;                 // moments / centre / normal of the best merge, one dword per lane
        return r;
    };
    int step = 0;
    while (step <= MAX_STEP && !err) {
        c0 = PEAC_CYCLES();
        // the node this iteration pops: its record is requested now and arrives while the queue is being repaired
        int p;
        if constexpr (FAST) { p = pq_top(); if (p < 0 || err) break; }
        else { if (heap_n <= 0) break; p = e_id(hp[0]); }
        const uint32_t* rp = rec(p);
        RecView A = load_rec(rp);
        bool dead_p = false;
        if constexpr (!FAST) dead_p = mp[p] != p;
        WFENCE();                                                 // every lane has looked at mp[p] before lane 0 may change it below
        PEAC_TICK(2);
        if constexpr (FAST) { pq_remove(p); PEAC_TICK(0); } else heap_pop();
        if (dead_p) continue;                                     // nouse (merged away earlier)
        int n = (int)(A.h.x & 0xffffu);
        PEAC_TICK(3);
        if (n > 0 && !((A.h.x >> 16) & 4u) && !is_valid(p)) {
            eval_phase(p, n);
            GFENCE();
            A = load_rec(rp);
            PEAC_TICK(17);
        } else if (n > 0 && ((A.h.x >> 16) & 4u)) {
            eval_big(p, n, A.h.w);
            GFENCE();
            A = load_rec(rp);
            n = (int)(A.h.x & 0xffffu);
            PEAC_TICK(18);
        } else dbg_hits++;
        const unsigned fl = n > 0 ? A.h.x >> 16 : 0u;
        const bool bigA = (fl & 4u) != 0;
        const int N100p = (int)(A.h.y >> 16);
        const unsigned offA = A.h.w;
        const unsigned rootA = A.root;                              // p's entry: alive or TOMB (the record is valid)
        if (fl & 2u) {
            // ---------------- merge p with its best neighbour ----------------
            const int nb = (int)(A.h.y & 0xffffu);
            const int ridp = (int)(A.h.z & 0xffffu);
            const uint32_t* rq = rec(nb);
            const uint4 hB = *(const uint4*)rq;                     // the partner's header ...
            const unsigned rootB = ((const u16*)(rq + 6))[lane];    // ... and bag
            if constexpr (FAST) { pq_remove(nb); PEAC_TICK(1); }    // nb dies with this merge (behind the load's latency)
            const int nbn = (int)(hB.x & 0xffffu);
            const bool bigB = ((hB.x >> 16) & 4u) != 0;
            const int N100n = (int)(hB.y >> 16);
            const int ridn = (int)(hB.z & 0xffffu);
            const unsigned offB = hB.w;
            PEAC_TICK(4);
            const int m = n_nodes++;
            if (m >= NB2) { err = 1; break; }
            int nm;
            unsigned offM = 0;
            bool bigM = false;
            if (!bigA && !bigB) {
                // both bags in registers.  Deduplication through the bitmap: an entry whose bit was clear before its own atomic OR is the
                // first of its node and stays; the survivors are packed in lane order (ballot + popcount: no prefix arrays).  The new bag is
                // NOT sorted by id: nothing reads a bag in order (the fold takes the minimum mse, exact ties sort by id themselves).
                unsigned ra = lane < n ? rootA : TOMB;
                unsigned rb = chase(lane < nbn ? rootB : TOMB);
                PEAC_TICK(5);
                if (ra == (unsigned)nb || ra == (unsigned)p) ra = TOMB;
                if (rb == (unsigned)p || rb == (unsigned)nb) rb = TOMB;
                bool keepA = false, keepB = false;
                if (ra != TOMB) { const unsigned bit = 1u << (ra & 31); keepA = !(atomicOr(&bmp[ra >> 5], bit) & bit); }
                if (rb != TOMB) { const unsigned bit = 1u << (rb & 31); keepB = !(atomicOr(&bmp[rb >> 5], bit) & bit); }
                const u64 mA = __ballot(keepA), mB = __ballot(keepB);
                const int nA = __popcll(mA);
                nm = nA + __popcll(mB);
                PEAC_TICK(6);
                u16* dst;
                if (nm <= 64) dst = roots_of(m);
                else {
                    const int cap = nm + nm / 4 + 16;
                    if (pool_top + 1 + cap > (unsigned)L.bpool_cap) { err = 2; break; }
                    if (lane == 0) bpool[pool_top] = (u16)cap;
                    offM = pool_top + 1; pool_top += 1 + cap; bigM = true;
                    dst = bpool + offM;
                }
                if (keepA) { dst[__popcll(mA & lane_below(lane))] = (u16)ra; inval(ra); bmp[ra >> 5] = 0; }
                if (keepB) { dst[nA + __popcll(mB & lane_below(lane))] = (u16)rb; inval(rb); bmp[rb >> 5] = 0; }
                WFENCE();
                PEAC_TICK(7);
            } else {
                // a bag in the pool on either side.  The new bag is written while the old ones are read, 64 entries at a time: first occurrence through
                // the bitmap (atomic OR), survivors packed by ballot / popcount.  Destination: the pool slot of a dying bag that can hold n + nbn
                // entries - that bag is read first, the write position never passes the read position - else a new slot (a region that absorbs
                // its neighbours one by one keeps its slot).  p's entries are live or TOMB (its record was evaluated in this iteration).
                const int ub = n + nbn;
                const unsigned saved_top = pool_top;
                bool fresh = false, firstB = false;
                u16* dst;
                if (ub <= 64) dst = roots_of(m);
                else {
                    const int capA = bigA ? (int)bpool[offA - 1] : 0, capB = bigB ? (int)bpool[offB - 1] : 0;
                    if (capA >= ub && (capA <= capB || capB < ub)) offM = offA;
                    else if (capB >= ub) { offM = offB; firstB = true; }
                    else {
                        const int cap = ub + ub / 4 + 16;
                        if (pool_top + 1 + cap > (unsigned)L.bpool_cap) { err = 2; break; }
                        if (lane == 0) bpool[pool_top] = (u16)cap;
                        offM = pool_top + 1; pool_top += 1 + cap; fresh = true;
                    }
                    dst = bpool + offM;
                }
                nm = 0;
                for (int side = 0; side < 2; side++) {
                    const bool isA = (side == 0) != firstB;
                    const int cnt = isA ? n : nbn;
                    const bool bigS = isA ? bigA : bigB;
                    const unsigned offS = isA ? offA : offB, rootS = isA ? rootA : rootB;
                    for (int g0 = 0; g0 < cnt; g0 += 256) {           // four chunks are requested together: one memory latency per 256 entries
                        unsigned e4[4];
#pragma unroll
                        for (int c = 0; c < 4; c++) {
                            e4[c] = TOMB;
                            if (g0 + c * 64 + lane < cnt) e4[c] = bigS ? (unsigned)bpool[offS + g0 + c * 64 + lane] : rootS;
                        }
#pragma unroll
                        for (int c = 0; c < 4; c++) {
                            if (g0 + c * 64 >= cnt) break;
                            unsigned r = e4[c];
                            if (!isA) r = chase(r);
                            bool keep = false;
                            if (r != TOMB && r != (unsigned)p && r != (unsigned)nb) { const unsigned bit = 1u << (r & 31); keep = !(atomicOr(&bmp[r >> 5], bit) & bit); }
                            const u64 km = __ballot(keep);
                            if (keep) { dst[nm + __popcll(km & lane_below(lane))] = (u16)r; inval(r); }
                            nm += __popcll(km);
                        }
                    }
                }
                for (int t = lane; t < W32; t += 64) bmp[t] = 0;
                if (ub > 64) {
                    if (nm <= 64) {                                   // the union fits the record again: the bag moves there, a new slot is given back
                        GFENCE();
                        if (lane < nm) roots_of(m)[lane] = dst[lane];
                        if (fresh) pool_top = saved_top;
                        offM = 0;
                    } else bigM = true;
                }
                WFENCE();
                PEAC_TICK(8);
            }
            // node m: its moments / plane are the best merge of p's record (AHCPlaneSeg.hpp:301-315)
            {
                uint32_t* gs = (uint32_t*)(g_stats + (size_t)m * 9);
                uint32_t* gg = (uint32_t*)(g_geo + (size_t)m * 7);
                uint32_t* dstw = lane < 18 ? gs + lane : gg + (lane - 18);     // 18 dwords of moments, then centre and normal
                if (lane < 30) *dstw = A.mw;
            }
            const int N100m = N100p + N100n;
            const int ridm = N100p >= N100n ? ridp : ridn;
            if (lane == 0) {
                geo_of(m)[6] = A.mse;
                g_N[m] = N100m * (WIN * WIN);
                *(uint4*)rec(m) = make_uint4((uint32_t)nm | (bigM ? 4u << 16 : 0u), (uint32_t)N100m << 16, (uint32_t)ridm, offM);
                h_rid[m] = (u16)ridm;
                // ds.Union(pa.rid, pb.rid) (DisjointSet.hpp:64-84): the rid of a live node is its set's root, so Find() returns its argument;
                // union by size, size(root) * 100 == N of the live node whose rid it is
                if (ridp != ridn) {
                    if (N100p < N100n) { h_dsp[ridp] = (u16)ridn; h_dss[ridn] = (u16)N100m; }
                    else { h_dsp[ridn] = (u16)ridp; h_dss[ridp] = (u16)N100m; }
                }
                mp[p] = (u16)m; mp[nb] = (u16)m;
            }
            WFENCE();
            PEAC_TICK(9);
            if constexpr (FAST) pq_push(m, A.mse, nm);
            else heap_push(m, A.mse, nm);
            PEAC_TICK(10);
        } else {
            // ---------------- no merge: extract p if it is large enough, disconnect it (:1160-1170) ----------------
            if (N100p * (WIN * WIN) >= MIN_SUPPORT) { if (n_ext < MAX_PLANES) { if (lane == 0) s_ext[n_ext] = p; n_ext++; } else err = 4; }
            for (int k0 = 0; k0 < n; k0 += 64) {
                unsigned e = TOMB;
                if (k0 + lane < n) e = bigA ? (unsigned)bpool[offA + k0 + lane] : rootA;
                if (e != TOMB) inval(e);                                // p leaves their live-neighbour sets
            }
            if (lane == 0) mp[p] = (u16)TOMB;
            WFENCE();
            PEAC_TICK(11);
        }
        ++step;
    }
    if constexpr (FAST) { if (!err && step > MAX_STEP) err = ST_RETRY; }   // maxStep reached (never for NB <= 3072): the exact kernel knows what to do
    while (!FAST && heap_n > 0 && !err) {                          // only after MAX_STEP: the reference extracts what is left without looking at nouse
        const int p = e_id(hp[0]);
        heap_pop();
        if (g_N[p] >= MIN_SUPPORT) { if (n_ext < MAX_PLANES) { if (lane == 0) s_ext[n_ext] = p; n_ext++; } else err = 4; }
        if (lane == 0 && mp[p] == p) mp[p] = (u16)TOMB;
        WFENCE();
    }
    GFENCE();
    if (lane == 0) {   // std::sort(extractedPlanes, b->N < a->N): insertion sort (stable)
        for (int i = 1; i < n_ext; i++) {
            const int v = s_ext[i];
            int j = i;
            while (j > 0 && g_N[s_ext[j - 1]] < g_N[v]) { s_ext[j] = s_ext[j - 1]; j--; }
            s_ext[j] = v;
        }
    }
    WFENCE();
    mark();
    // ---- hand the clustering state over to peac_refine: set sizes / root ids / parents are in the workspace already; dead bits, extracted planes
    {
        unsigned* o_nouse = (unsigned*)(F + L.off_h_nouse); unsigned* o_cval = (unsigned*)(F + L.off_h_cval);
        for (int b0 = 0; b0 < NB2; b0 += 64) {
            const int id = b0 + lane;
            const u64 dead = __ballot(id < NB2 && mp[id] != id);
            if (lane == 0) { o_nouse[b0 >> 5] = (unsigned)dead; if ((b0 >> 5) + 1 < W32) o_nouse[(b0 >> 5) + 1] = (unsigned)(dead >> 32); }
        }
        for (int t = lane; t < W32; t += 64) o_cval[t] = 0;
        for (int t = lane; t < MAX_PLANES; t += 64) g_hand[4 + t] = t < n_ext ? s_ext[t] : 0;
        if (lane == 0) {
            g_hand[0] = n_ext; g_hand[1] = err; g_hand[2] = n_nodes;
            status[frame] = err;
            if (timing) {
                for (int t = 0; t < 4; t++) timing[(size_t)frame * TSLOTS + t] = t < nph ? tphase[t] - tphase[0] : 0;
                timing[(size_t)frame * TSLOTS + 9] = n_nodes;
                timing[(size_t)frame * TSLOTS + 7] = ((long long)dbg_phases << 40) | ((long long)dbg_nodes << 20) | dbg_hits;
                timing[(size_t)frame * TSLOTS + 10] = dbg_big;
                timing[(size_t)frame * TSLOTS + 11] = dbg_bigsolves;
                for (int t = 0; t < 24; t++) timing[(size_t)frame * TSLOTS + 16 + t] = cyc[t];
            }
        }
    }
    return wave_uni(err);
}

// The fast attempt: every frame of the batch, taken from a start-order counter (longest first, see peac_order), not from the block index.
// retry_inline: a frame the fast attempt gives up on (ST_RETRY) is redone with the exact heap by the same workgroup, which holds the LDS already - a separate
// launch of the exact kernel over the batch would have to be granted 38 KB of LDS per workgroup just to find, almost always, nothing to do.
__global__ __launch_bounds__(64) void peac_ahc3(Layout L, Consts C, uint8_t* __restrict__ ws, int32_t* __restrict__ status,
                                                long long* __restrict__ timing, int* __restrict__ next_frame, const int* __restrict__ order, int retry_inline) {
    __shared__ int s_frame;
#if defined(PLANAR_AHC_PRIO) && !defined(PLANAR_WAVE_EMUL)
    __builtin_amdgcn_s_setprio(PLANAR_AHC_PRIO);             // (experiment: issue priority of the clustering wavefront over the wide kernels' wavefronts on its SIMD)
#endif
    if (threadIdx.x == 0) { const int k = atomicAdd(next_frame, 1); s_frame = order ? order[k] : k; }
    __syncthreads();
    const int st = ahc_frame<true>(L, C, ws, status, timing, s_frame);
    if (st == ST_RETRY && retry_inline) {
        __syncthreads();
        ahc_frame<false>(L, C, ws, status, timing, s_frame);
        if (timing && threadIdx.x == 0) timing[(size_t)s_frame * TSLOTS + 12] = 1;      // the frame went through both kernels
    }
}
// The exact kernel.  only_retry != 0: workgroup b redoes frame b if the fast attempt left ST_RETRY there and exits at once otherwise.
__global__ __launch_bounds__(64) void peac_ahc2(Layout L, Consts C, uint8_t* __restrict__ ws, int32_t* __restrict__ status,
                                                long long* __restrict__ timing, int* __restrict__ next_frame, const int* __restrict__ order, int only_retry) {
    __shared__ int s_frame;
    if (threadIdx.x == 0) {
        if (only_retry) s_frame = status[blockIdx.x] == ST_RETRY ? (int)blockIdx.x : -1;
        else { const int k = atomicAdd(next_frame, 1); s_frame = order ? order[k] : k; }
    }
    __syncthreads();
    if (s_frame < 0) return;
    ahc_frame<false>(L, C, ws, status, timing, s_frame);
}

}  // namespace peac
}  // namespace planar
