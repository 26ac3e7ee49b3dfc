// planarslam_amd/csrc/peac.hip — batched PEAC / AHC plane segmentation on depth for MI355X (gfx950).
//
// Replaces PlaneDetection::readDepthImage + runPlaneDetection (reference src/PlaneExtractor.cpp:26-65), i.e.
// ahc::PlaneFitter::run (reference include/peac/AHCPlaneFitter.hpp:211) with PlanarSLAM's defaults, for a batch of
// B independent depth frames.  Plane labels, plane normals/centres/MSE are bit-exact against the CPU oracle, which
// is pinned label-for-label against the real reference sources.
//
//   peac_blocks   one THREAD per 10x10 block: depth -> XYZ in FP64, validity / depth-discontinuity tests, the nine moment sums
//                 accumulated in the reference's raster order (bit-exact FP64 sums), PCA by the iterative 3x3 symmetric
//                 eigen-solver (Eigen's algorithm, restated)                                                            (a13-a15)
//   peac_ahc      one WAVEFRONT (64 lanes) per frame, the order-dependent clustering: initGraph edges -> ahCluster with a
//                 libstdc++-exact binary min-MSE heap.  24.5 KB LDS per frame (heap keys as floats 12 K, heap ids 6 K, DisjointSet parents / sizes,
//                 flag bits); the neighbour-list POOL, list offsets / counts and the candidate cache live in the frame's global
//                 workspace, lists are staged through LDS when two are merged.  234 VGPRs: two such wavefronts per SIMD, up to six frames per CU.
//   peac_order    ranks the frames by the clustering time of the previous call (longest first) for the next launch
//   peac_refine   256 threads per frame (24.5 KB LDS): block erosion + seed queue (prefix sums) -> floodFill (512 queue entries x
//                 4 neighbours per step, same-pixel conflicts replayed in order; membership image in bytes, 4-byte queue entries) -> final
//                 ahCluster over the surviving planes -> relabel; the node arrays stay in the global workspace            (a16-a17)
// Frame-level batch parallelism supplies the occupancy: the clustering is a chain of dependent FP64 operations, so a
// frame is latency-bound and the launch time is (frames / resident frames) x the slowest frame.  DESIGN.md §PEAC has the numbers.
#include "common.h"

#include "peac_common.h"

namespace planar {
namespace peac {

// ------------------------------------------------------------------------------------------------------------
// K2: one 256-thread workgroup per frame.  Everything the sequential part chases pointers through lives in LDS
// (merge heap with its MSE keys, the neighbour lists as u16, the disjoint set, the block map); the per-node
// moments / plane parameters stay in the frame's global workspace and are read once per merge step.
constexpr int NT_AHC = 64;       // clustering: one wavefront per frame (a second one, 128 threads, leaves the chain as long and costs 12 ms per step: measured)
constexpr int NT_REFINE = 256;   // refinement: four wavefronts per frame (throughput: several frames per CU)
constexpr int NT_REFINE_WIDE = 1024;   // few frames in the batch (the reference's one-camera operating point): sixteen wavefronts per frame, the flood fill takes a quarter of the steps

struct Lds {
    float* h_key; u16* h_id; u16* pool; u16* nb_off; u16* nb_cnt; unsigned char* nb_cntb; u16* dsp; u16* dss; u16* rid; unsigned* nouse; unsigned* cval; signed char* blk;
};

__device__ __forceinline__ int lds_find(u16* parent, int x) {   // DisjointSet::Find with path compression
    int r = x, guard = 0;
    while (parent[r] != r && ++guard < 8192) r = parent[r];           // bounded: a corrupted workspace must not hang the GPU
    guard = 0;
    while (parent[x] != r && ++guard < 8192) { const int nx = parent[x]; parent[x] = (u16)r; x = nx; }
    return r;
}
// PHASE 0 = peac_ahc: initGraph + the first ahCluster (heap, neighbour lists, set sizes, root ids in LDS).  PHASE 1 = peac_refine:
// refineDetails (block membership, erosion, seeds, flood fill), the final ahCluster over the <= 128 extracted planes and the relabelling;
// its node-indexed arrays live in the frame's global workspace (a handful of nodes are touched), so it needs ~17 KB of LDS and runs
// beside the clustering workgroups of other frames.  The code of ahCluster is shared: the pointers of `Lds` point into LDS or global memory.
template <int PHASE, int NT>
__device__ __forceinline__ void segment_frame(const Layout& L, const Intr& K, const Consts& C, const uint16_t* __restrict__ depth, int pitch_px,
                                              int64_t frame_stride_px, uint8_t* __restrict__ ws, int32_t* __restrict__ labels,
                                              int64_t label_stride, double* __restrict__ planes, int32_t* __restrict__ n_planes,
                                              int32_t* __restrict__ status, long long* __restrict__ timing, const int frame) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint8_t* F = ws + (size_t)frame * L.frame_bytes;
    double* g_stats = (double*)(F + L.off_stats);
    double* g_geo = (double*)(F + L.off_geo);
    int* g_N = (int*)(F + L.off_N);
    uint8_t* g_flags = F + L.off_flags;
    // membershipImg as signed bytes (plane ids 0..127, -1 = none, -2..-6 = the flood fill's rejection trail), the distance map WITHOUT its FLT_MAX fill (a
    // pixel carries a distance exactly when the flood fill made it a member: read as FLT_MAX otherwise), queue entries packed pixel | plane << 24:
    // the reference's int32 image, float image and (index, plane) pairs cost 2.3x the bytes per frame
    signed char* member = (signed char*)(F + L.off_member);
    float* distMap = (float*)(F + L.off_dist);
    unsigned* queue = (unsigned*)(F + L.off_queue);
    auto qent = [](int pixel, int plane) { return (unsigned)pixel | ((unsigned)plane << 24); };
    int* seedcnt = (int*)(F + L.off_seedcnt);
    int* g_cint = (int*)(F + L.off_cint);
    double* g_cdbl = (double*)(F + L.off_cdbl);
    const uint16_t* D = depth + (size_t)frame * frame_stride_px;
    int32_t* lab = labels + (size_t)frame * label_stride;
    const int NB = L.NB, Nw = L.Nw, Nh = L.Nh, W = L.W, H = L.H;

    __shared__ float s_hm[PHASE == 1 ? MAX_PLANES : 1];      // refinement: the heap of the final clustering holds <= MAX_PLANES nodes
    __shared__ u16 s_hi[PHASE == 1 ? MAX_PLANES : 1];
    __shared__ signed char s_blk[PHASE == 1 ? 3072 : 1];      // block -> plane id (NB <= 3072 is checked at create time for this path)
    Lds S;
    if (PHASE == 0) {
        S.h_key = (float*)smem;
        S.h_id = (u16*)(S.h_key + NB);
        S.pool = (u16*)(F + L.off_h_pool);        // neighbour lists: global memory (L2 resident); LDS holds what every pop / merge chases
        // LDS holds what every pop / merge chases: the heap, the list offsets / counts of the MERGED nodes (a block's list sits at
        // 4 * id, its count fits a byte) and the dead / cache-valid bits.  Set sizes, root ids and DisjointSet parents are only written
        // here (the node's own N travels with its moments), so they live in the frame workspace where peac_refine reads them.
        // list offsets / counts of the merged nodes: frame workspace too (12 KB of LDS less per frame: with 25 KB the four clustering wavefronts of a CU leave
        // 60 KB to the kernels of the other streams); a merge reads them in the same round trip as the list heads
        S.nb_off = (u16*)(F + L.off_h_nboff);     // [NB] merged node NB + i
        S.nb_cnt = (u16*)(F + L.off_h_nbcnt);     // [NB] merged node NB + i
        S.nouse = (unsigned*)(S.h_id + NB);       // bit per node: out of the graph (merged away = PlaneSeg::nouse, or disconnected)
        S.cval = S.nouse + (L.NB2 + 31) / 32;     // bit per node: its cached candidate record (g_cint / g_cdbl) is valid for its current live-neighbour set
        S.nb_cntb = (unsigned char*)(S.cval + (L.NB2 + 31) / 32);   // [NB] blocks
        S.dss = (u16*)(F + L.off_h_dss); S.rid = (u16*)(F + L.off_h_rid); S.dsp = (u16*)(F + L.off_h_dsp);
        S.blk = nullptr;
    } else {
        S.h_key = s_hm; S.h_id = s_hi;
        S.pool = (u16*)(F + L.off_h_pool); S.nb_off = (u16*)(F + L.off_h_nboff); S.nb_cnt = (u16*)(F + L.off_h_nbcnt); S.nb_cntb = nullptr;
        S.dsp = (u16*)(F + L.off_h_dsp); S.dss = (u16*)(F + L.off_h_dss); S.rid = (u16*)(F + L.off_h_rid);
        S.nouse = (unsigned*)(F + L.off_h_nouse); S.cval = (unsigned*)(F + L.off_h_cval);
        S.blk = s_blk;
    }
    int* g_hand = (int*)(F + L.off_h_hand);       // [0] n_ext, [1] err, [2] n_nodes, [4 + q] extracted node ids
    // list capacities are only consulted when a node is appended to a full list: the 4-slot lists of the initial blocks have it implicit,
    // every other list keeps it in the pool slot in front of its first entry (no LDS array: the line detector's wavefront, lsd_detect,
    // needs 8 KB on the same CU; no global array: the append would wait a memory round trip per merge)
    bool hdr_all = false;                                     // second ahCluster: every list has the header slot
    auto noff = [&](int q) -> int { return PHASE == 0 ? (q < NB ? 4 * q : (int)S.nb_off[q - NB]) : (int)S.nb_off[q]; };
    auto set_noff = [&](int q, int v) { if (PHASE == 0) S.nb_off[q - NB] = (u16)v; else S.nb_off[q] = (u16)v; };
    auto ncnt = [&](int q) -> int { return PHASE == 0 ? (q < NB ? (int)S.nb_cntb[q] : (int)S.nb_cnt[q - NB]) : (int)S.nb_cnt[q]; };
    auto set_ncnt = [&](int q, int v) { if (PHASE == 0) { if (q < NB) S.nb_cntb[q] = (unsigned char)v; else S.nb_cnt[q - NB] = (u16)v; } else S.nb_cnt[q] = (u16)v; };
    auto list_cap = [&](int q) -> int { return (q < NB && !hdr_all) ? 4 : (int)S.pool[noff(q) - 1]; };

    constexpr int RP = PHASE == 1 ? MAX_PLANES : 1;           // refinement-only arrays take no LDS in the clustering kernel
    __shared__ int s_ext[MAX_PLANES], s_old[RP], s_plidmap[RP];
    __shared__ uint8_t s_valid[RP];
    __shared__ unsigned s_adj[RP][MAX_PLANES / 32];
    constexpr int NW = NT / 64;                               // wavefronts of the workgroup
    constexpr int NSLOT = 16 * NT;                            // pixel -> slot hash of the flood fill (pairs of one step: 8 * NT)
    __shared__ int s_slot[PHASE == 1 ? NSLOT : 1];
    __shared__ int s_pcnt[8 * NW];
    __shared__ int s_scalar[4];   // [0] n_ext, [1] err, [2] q_tail, [3] scratch
    constexpr int EVAL_MAX = 64;   // nodes evaluated per cooperative phase
    __shared__ int s_cmd, s_nlist;
    __shared__ u16 s_stageA[64], s_stageB[64];                // the two neighbour lists of a merge (read once from global memory)
    __shared__ unsigned short s_lnode[EVAL_MAX];
    __shared__ unsigned char s_lwave[EVAL_MAX], s_loff[EVAL_MAX];
    long long tphase[8];
    int nph = 0;
    auto mark = [&]() { if (nph < 8) tphase[nph++] = (long long)wall_clock64(); };
    mark();
    // Single-wave sections: the LDS traffic of one wavefront is processed in program order, so lanes only need the compiler to keep
    // that order (wavefront-scope fence, no s_waitcnt).  gfence additionally waits for the wave's global stores (workgroup scope):
    // used where lane 0 publishes a new node's moments / plane in global memory.
    auto wfence = [&]() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); };
    auto gfence = [&]() { __threadfence_block(); };

    auto geo_of = [&](int id) { return g_geo + (size_t)id * 7; };
    auto cvalid = [&](int id) { return (S.cval[id >> 5] >> (id & 31)) & 1u; };
    auto cinval = [&](int id) { atomicAnd(&S.cval[id >> 5], ~(1u << (id & 31))); };
    auto nsim = [&](int a, int b) {
        const double* ga = geo_of(a) + 3; const double* gb = geo_of(b) + 3;
        return fabs(ga[0] * gb[0] + ga[1] * gb[1] + ga[2] * gb[2]);
    };

    // ---- init (all threads) ----
    if (PHASE == 0) {
        for (int b = tid; b < NB; b += NT) { S.dsp[b] = (u16)b; S.dss[b] = 1; S.nb_cntb[b] = 0; S.rid[b] = (u16)b; }
        for (int t = tid; t < (L.NB2 + 31) / 32; t += NT) { S.nouse[t] = 0; S.cval[t] = 0; }
        if (tid < 4) s_scalar[tid] = 0;
    } else {
        for (int t = tid; t < MAX_PLANES; t += NT) { s_valid[t] = 0; s_plidmap[t] = -1; s_ext[t] = g_hand[4 + t]; }
        for (int t = tid; t < MAX_PLANES * (MAX_PLANES / 32); t += NT) (&s_adj[0][0])[t] = 0;
        if (tid < 4) s_scalar[tid] = tid == 1 ? g_hand[1] : 0;
    }
    __syncthreads();

    // ---- initGraph edges (AHCPlaneFitter.hpp:896-954).  The horizontal pass only links nodes of one row and the
    //      vertical pass nodes of one column, each a sequential scan with the reference's --j/++j skip logic, so a
    //      thread owns a whole row / column.  Every block has its own 4-slot list.
    auto connect = [&](int a, int b) {
        int ca = ncnt(a); lst_insert(S.pool + noff(a), ca, b); set_ncnt(a, ca);
        int cb = ncnt(b); lst_insert(S.pool + noff(b), cb, a); set_ncnt(b, cb);
    };
    auto inG = [&](int idx) { return (g_flags[idx] & 1) != 0; };
    if (PHASE == 0) {
    for (int i = tid; i < Nh; i += NT) {
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
    for (int j = tid; j < Nw; j += NT) {
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
    }
    mark();

    // =========================== sequential section: wave 0 only ===========================
    int heap_n = 0, n_nodes = PHASE == 0 ? NB : g_hand[2], n_ext = PHASE == 0 ? 0 : g_hand[0], pool_top = 4 * NB, err = PHASE == 0 ? 0 : g_hand[1];
    // libstdc++ binary heap (std::priority_queue with PlaneSegMinMSECmp: comp(a,b) = b.mse < a.mse).  Entries carry their key ROUNDED TO FLOAT
    // (rounding is monotonic: two different floats order like the doubles they came from) so a comparison is one LDS read and the heap is
    // 6 bytes per node; only when two floats are EQUAL are the FP64 keys fetched from the nodes' records in the frame workspace.  The
    // comparisons, hence the heap layout and the pop order, are those of the FP64 heap.
    // __push_heap: the value climbs from `hole` while it is smaller than the parent.  The <= 12 ancestors are read by one lane each in a
    // single LDS round trip; the leading run of larger ancestors moves down one level in parallel.
    auto heap_sift_up = [&](int hole, int id, float mf) {
        const int anc = lane < 16 ? ((hole + 1) >> lane) - 1 : -1;                          // lane j: the j-th ancestor of the hole (lane 0: the hole)
        const bool isanc = lane >= 1 && anc >= 0;
        float K = 0; int I = 0;
        if (isanc) { K = S.h_key[anc]; I = S.h_id[anc]; }
        bool less = isanc && mf < K;
        if (__ballot(isanc && mf == K)) {                                    // float tie somewhere on the path: decide those on the doubles
            const double dv = geo_of(id)[6];
            if (isanc && mf == K) less = dv < geo_of(I)[6];
        }
        const unsigned long long up = __ballot(less) >> 1;                   // bit j-1: ancestor j is larger than the value
        const int n = __builtin_ctzll(~up);                                  // the loop stops at the first ancestor that is not
        if (lane >= 1 && lane <= n) { const int dst = ((hole + 1) >> (lane - 1)) - 1; S.h_key[dst] = K; S.h_id[dst] = (u16)I; }
        if (lane == 0) { const int dst = ((hole + 1) >> n) - 1; S.h_key[dst] = mf; S.h_id[dst] = (u16)id; }
        wfence();
    };
    auto heap_push = [&](int id, double mse) {      // the node's record (geo_of(id)[6] == mse) has been written before
        heap_n++;
        heap_sift_up(heap_n - 1, id, (float)mse);
    };
    // pop_heap = __adjust_heap(first, 0, len, last value): the hole sinks to the bottom along the smaller child (no early exit), then
    // the value climbs back.  Lane t = 1..63 stands for node t of the subtree under the hole and compares its two children (one LDS round
    // trip for everything); the six decisions are then bit operations on the ballot of those comparisons (scalar unit) and every node on
    // the path pulls its chosen child up in one parallel store.
    auto heap_pop = [&]() -> int {
        const int top = S.h_id[0];
        const float vm = S.h_key[heap_n - 1]; const int vi = S.h_id[heap_n - 1];
        heap_n--;
        const int len = heap_n;
        if (len == 0) return top;
        const int half = (len - 1) / 2;                                      // nodes below `half` have two children
        const int dl = 31 - __clz(max(lane, 1));                             // level of local node `lane`; its heap index is (hole << dl) + lane - 1
        int hole = 0;
        while (hole < half) {
            // lane t = 1..63 is node t of the subtree under the hole: it reads its two children, decides which one it would pull up, and the
            // nodes that turn out to lie on the path do so - six levels per LDS round trip
            const int g = (hole << dl) + lane - 1;
            const bool inner = lane >= 1 && g < half;
            float kl = 0, kr = 0; int il = 0, ir = 0;
            if (inner) { kl = S.h_key[2 * g + 1]; kr = S.h_key[2 * g + 2]; il = S.h_id[2 * g + 1]; ir = S.h_id[2 * g + 2]; }
            bool lt = inner && kl < kr;
            if (inner && kl == kr) lt = geo_of(il)[6] < geo_of(ir)[6];       // float tie: the FP64 keys decide
            const unsigned long long two = __ballot(inner);
            const unsigned long long takel = __ballot(lt);                   // comp(first[second], first[second - 1]): take second - 1
            int cur = 1;
            unsigned long long path = 0;
#pragma unroll
            for (int d = 0; d < 6; d++) {
                if (!((two >> cur) & 1ull)) break;
                path |= 1ull << cur;
                cur = 2 * cur + 1 - (int)((takel >> cur) & 1ull);
            }
            if ((path >> lane) & 1ull) {
                const bool left = (takel >> lane) & 1ull;
                S.h_key[g] = left ? kl : kr; S.h_id[g] = (u16)(left ? il : ir);
            }
            const int dc = 31 - __clz(cur);
            hole = (hole << dc) + cur - 1;
        }
        wfence();
        if ((len & 1) == 0 && hole == (len - 2) / 2) {                       // a last node with a single (left) child
            const int c = 2 * hole + 1;
            const float cm = S.h_key[c]; const int ci = S.h_id[c];
            if (lane == 0) { S.h_key[hole] = cm; S.h_id[hole] = (u16)ci; }
            wfence(); hole = c;
        }
        heap_sift_up(hole, vi, vm);
        return top;
    };
    // Neighbour lists use LAZY deletion: a node that leaves the graph (merged away or disconnected) only gets its
    // `dead` bit set; readers skip dead entries and a list is compacted when it is full.  Lists stay sorted by node
    // id because new ids are always the largest.
    auto is_dead = [&](int id) { return (S.nouse[id >> 5] >> (id & 31)) & 1u; };
    auto mark_dead = [&](int id) { if (lane == 0) S.nouse[id >> 5] |= 1u << (id & 31); wfence(); };   // PlaneSeg::disconnectAllNbs
    auto node_N = [&](int id) { return g_N[id]; };   // == dss[rid[id]] * 100 for a live node (its rid is its set's root); one load, fetched with the moments
    auto extract = [&](int p) {
        if (node_N(p) >= MIN_SUPPORT) { if (n_ext < MAX_PLANES) { if (lane == 0) s_ext[n_ext] = p; n_ext++; } else err = 4; }
    };
    // ahCluster (:983-1189)
    long long cyc[6] = {0, 0, 0, 0, 0, 0};
    int dbg_hits = 0, dbg_phases = 0, dbg_nodes = 0;
    // ahCluster (:983-1189).  coop = true: called by ALL four wavefronts.  The candidate evaluation of a popped node (one 3x3
    // eigen-solve per neighbour, ~27k cycles of latency however few lanes it uses) dominates this loop, but its result is a pure
    // function of the node's live-neighbour set.  So every time wavefront 0 pops a node without a valid cached result, all four
    // wavefronts evaluate it TOGETHER WITH the next nodes of the heap (up to 64 nodes, one lane per (node, neighbour) pair) and
    // store each result and set the node's `valid` bit (LDS); the bit is cleared whenever the node's live-neighbour set changes.
    // Later pops find their result in the cache unless a merge touched their neighbourhood.  Phases are separated by workgroup
    // barriers, so the schedule - and with it every bit of the output - is deterministic.
    auto eval_phase = [&]() {
        const int nl = s_nlist;
        int my_node = -1, my_k = 0, seg_first = 0, seg_cnt = 0;
        for (int i = 0; i < nl; i++)
            if (s_lwave[i] == wave) {
                const int off = s_loff[i], nd = s_lnode[i], c = ncnt(nd);
                if (lane >= off && lane < off + c) { my_node = nd; my_k = lane - off; seg_first = off; seg_cnt = c; }
            }
        bool ok = false;
        double ms[9]; Geo mg; int mN = 0, nb = -1;
        for (int t = 0; t < 9; t++) ms[t] = 0;
        for (int t = 0; t < 3; t++) { mg.center[t] = 0; mg.normal[t] = 0; }
        mg.mse = 0;
        if (my_node >= 0) {
            const int x = (S.pool + noff(my_node))[my_k];
            if (!is_dead(x)) {
                nb = x;
                const double* sp = g_stats + (size_t)my_node * 9;
                const double* gp = geo_of(my_node) + 3;
                const double* gn = geo_of(nb) + 3;
                const double* sb = g_stats + (size_t)nb * 9;
                const double pn0 = gp[0], pn1 = gp[1], pn2 = gp[2], n0 = gn[0], n1 = gn[1], n2 = gn[2];
                double ps[9];
                for (int t = 0; t < 9; t++) { ps[t] = sp[t]; ms[t] = sb[t]; }
                if (!(fabs(pn0 * n0 + pn1 * n1 + pn2 * n2) < C.cos_merge)) {
                    for (int t = 0; t < 9; t++) ms[t] = ps[t] + ms[t];
                    mN = node_N(my_node) + node_N(nb);
                    stats_compute(ms, mN, mg);
                    ok = true;
                }
            }
        }
        // the reference's in-order rule (:1043-1049) inside every node's lane segment.  Without exact ties (and NaNs) the rule keeps the
        // minimum mse: a segmented prefix-min over (mse, k) in 6 shuffle steps; a tie anywhere in the wavefront falls back to the scan.
        bool have = false; double best_mse = 0; int best_k = 0, best_N = 0;
        {
            double rm = ok ? mg.mse : 1.7976931348623157e308;
            int rk = ok ? my_k : 64;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const double om = __shfl_up(rm, o); const int ok2 = __shfl_up(rk, o);
                if (my_node >= 0 && lane - o >= seg_first && (om < rm || (om == rm && ok2 < rk))) { rm = om; rk = ok2; }
            }
            const int last = min(seg_first + max(seg_cnt, 1) - 1, 63);
            const double min_m = __shfl(rm, last); const int min_k = __shfl(rk, last);
            const bool odd = ok && (mg.mse != mg.mse || (mg.mse == min_m && my_k != min_k));
            if (!__ballot(odd)) {
                have = min_k < 64; best_k = have ? min_k : 0; best_mse = have ? min_m : 0;
                best_N = __shfl(mN, min(seg_first + best_k, 63));
                if (!have) best_N = 0;
            } else {
                int maxc = seg_cnt;
                for (int o = 32; o > 0; o >>= 1) maxc = max(maxc, __shfl_xor(maxc, o));
                for (int k = 0; k < maxc; k++) {
                    const int src = min(seg_first + k, 63);
                    const int c_ok = __shfl((int)ok, src); const double c_mse = __shfl(mg.mse, src); const int c_N = __shfl(mN, src);
                    if (k < seg_cnt && c_ok && (!have || best_mse > c_mse || (best_mse == c_mse && (double)best_N < c_mse))) {   // quirk :1045
                        have = true; best_mse = c_mse; best_k = k; best_N = c_N;
                    }
                }
            }
        }
        const int bl = min(seg_first + best_k, 63);
        const int w_nb = __shfl(nb, bl);
        double w[15];
        for (int t = 0; t < 9; t++) w[t] = __shfl(ms[t], bl);
        for (int t = 0; t < 3; t++) { w[9 + t] = __shfl(mg.center[t], bl); w[12 + t] = __shfl(mg.normal[t], bl); }
        if (my_node >= 0 && my_k == 0) {
            int* ci = g_cint + (size_t)my_node * 4;
            double* cd = g_cdbl + (size_t)my_node * 16;
            ci[0] = have ? 1 : 0; ci[1] = w_nb; ci[2] = best_N;
            cd[0] = best_mse;
            for (int t = 0; t < 15; t++) cd[1 + t] = w[t];
            atomicOr(&S.cval[my_node >> 5], 1u << (my_node & 31));
        }
    };
    auto ah_cluster = [&](const bool coop) {
        int step = 0, pending = -1;
        while (true) {
            bool need_eval = false;
            long long e0 = 0;
            if (wave == 0) {
                while ((pending >= 0 || heap_n > 0) && step <= MAX_STEP && !err) {
                    long long c0 = PEAC_CYCLES();
                    int p;
                    if (pending >= 0) { p = pending; pending = -1; }
                    else {
                        p = heap_pop();
                        cyc[0] += PEAC_CYCLES() - c0; c0 = PEAC_CYCLES();
                        if (is_dead(p)) continue;                           // nouse
                    }
                    const int cnt = ncnt(p);
                    const u16* lst = S.pool + noff(p);
                    const double* sp = g_stats + (size_t)p * 9;
                    const int Np = node_N(p);
                    const double* gp = geo_of(p) + 3;
                    const double pn0 = gp[0], pn1 = gp[1], pn2 = gp[2];
                    double ps[9];
                    for (int t = 0; t < 9; t++) ps[t] = sp[t];
                    // candidate merges, one per lane; the in-order fold reproduces "first minimum wins (+ the N<mse quirk)"
                    double best_mse = 0; int best_nb = -1, best_N = 0; bool have = false;
                    double best_stats[9]; Geo best_geo;
                    bool from_cache = false;
                    if (coop && cnt > 0 && cnt <= 64) {
                        const bool hit = cvalid(p) != 0;
                        const int* ci = g_cint + (size_t)p * 4;
                        const double* cd = g_cdbl + (size_t)p * 16;
                        const int c_have = ci[0], c_nb = ci[1], c_N = ci[2];
                        double cv[16];
                        for (int t = 0; t < 16; t++) cv[t] = cd[t];
                        if (hit) {
                            have = c_have != 0; best_nb = c_nb; best_N = c_N; best_mse = cv[0];
                            for (int t = 0; t < 9; t++) best_stats[t] = cv[1 + t];
                            for (int t = 0; t < 3; t++) { best_geo.center[t] = cv[10 + t]; best_geo.normal[t] = cv[13 + t]; }
                            best_geo.mse = best_mse;
                            from_cache = true; dbg_hits++;
                        } else {
                            // miss: evaluate p and the nodes waiting at the top of the heap (live, uncached, <= 64 neighbours), packed
                            // into the 4 x 64 lanes; p goes first
                            const int hp = lane < heap_n ? (int)S.h_id[lane] : -1;
                            int hc = 0;
                            bool cand = false;
                            if (hp >= 0 && !is_dead(hp)) { hc = ncnt(hp); cand = hc > 0 && hc <= 64 && !cvalid(hp); }
                            unsigned long long cm = __ballot(cand);
                            int nl = 0, cw = 0, co = cnt;
                            if (lane == 0) { s_lnode[0] = (unsigned short)p; s_lwave[0] = 0; s_loff[0] = 0; }
                            nl = 1;
                            while (cm && nl < EVAL_MAX) {
                                const int src = __ffsll((long long)cm) - 1;
                                cm &= cm - 1;
                                const int nd = __shfl(hp, src), c = __shfl(hc, src);
                                if (co + c > 64) { cw++; co = 0; if (cw >= NT / 64) break; }
                                if (lane == 0) { s_lnode[nl] = (unsigned short)nd; s_lwave[nl] = (unsigned char)cw; s_loff[nl] = (unsigned char)co; }
                                nl++; co += c;
                            }
                            if (lane == 0) s_nlist = nl;
                            pending = p;
                            need_eval = true;
                            break;
                        }
                    }
                    if (!from_cache) {
                        for (int k0 = 0; k0 < cnt; k0 += 64) {
                            const int k = k0 + lane;
                            bool ok = false;
                            double ms[9]; Geo mg; int mN = 0, nb = -1;
                            mg.mse = 0;
                            if (k < cnt && !is_dead(lst[k])) {
                                nb = lst[k];
                                // one memory round trip: the neighbour's normal and moments are fetched together
                                const double* gn = geo_of(nb) + 3;
                                const double* sb = g_stats + (size_t)nb * 9;
                                const double n0 = gn[0], n1 = gn[1], n2 = gn[2];
                                for (int t = 0; t < 9; t++) ms[t] = sb[t];
                                if (!(fabs(pn0 * n0 + pn1 * n1 + pn2 * n2) < C.cos_merge)) {
                                    for (int t = 0; t < 9; t++) ms[t] = ps[t] + ms[t];
                                    mN = Np + node_N(nb);
                                    stats_compute(ms, mN, mg);
                                    ok = true;
                                }
                            }
                            // Reference rule (:1043-1049), candidates in ascending node id: take a candidate if none yet, or its mse is
                            // smaller, or (equal mse and best.N < mse — quirk).  Without exact ties that is "first minimum": a butterfly
                            // arg-min on (mse, lane); exact ties among this round's candidates fall back to the in-order scan.
                            unsigned long long okm = __ballot(ok);
                            if (okm) {
                                double rm = ok ? mg.mse : 1.7976931348623157e308;
                                int rl = ok ? lane : 64;
            #pragma unroll
                                for (int o = 32; o > 0; o >>= 1) {
                                    const double om = __shfl_xor(rm, o); const int ol = __shfl_xor(rl, o);
                                    if (om < rm || (om == rm && ol < rl)) { rm = om; rl = ol; }
                                }
                                const bool tie = __popcll(__ballot(ok && mg.mse == rm)) > 1;
                                unsigned long long scan = tie ? okm : (1ull << rl);
                                while (scan) {
                                    const int src = __builtin_amdgcn_readfirstlane(__ffsll((long long)scan) - 1);
                                    scan &= scan - 1;
                                    const double c_mse = __shfl(mg.mse, src);
                                    if (!have || best_mse > c_mse || (best_mse == c_mse && (double)best_N < c_mse)) {   // quirk :1045
                                        have = true; best_mse = c_mse; best_nb = __shfl(nb, src); best_N = __shfl(mN, src);
                                        for (int t = 0; t < 9; t++) best_stats[t] = __shfl(ms[t], src);
                                        for (int t = 0; t < 3; t++) { best_geo.center[t] = __shfl(mg.center[t], src); best_geo.normal[t] = __shfl(mg.normal[t], src); }
                                        best_geo.mse = c_mse;
                                    }
                                }
                            }
                        }
                    }
                    cyc[1] += PEAC_CYCLES() - c0; c0 = PEAC_CYCLES();
                    if (have && best_mse < T_mse_merge(best_geo.center[2])) {
                        const int m = n_nodes++;
                        const int nb = best_nb;
                        if (m >= L.NB2) { err = 1; break; }
                        const int rp = S.rid[p], rn = S.rid[nb];
                        const int Nn = node_N(nb);
                        if (lane == 0) {
                            for (int t = 0; t < 9; t++) g_stats[(size_t)m * 9 + t] = best_stats[t];
                            for (int t = 0; t < 3; t++) { g_geo[(size_t)m * 7 + t] = best_geo.center[t]; g_geo[(size_t)m * 7 + 3 + t] = best_geo.normal[t]; }
                            g_geo[(size_t)m * 7 + 6] = best_geo.mse;
                            g_N[m] = best_N;
                            S.rid[m] = (u16)(Np >= Nn ? rp : rn);
                            S.nouse[p >> 5] |= 1u << (p & 31);
                            S.nouse[nb >> 5] |= 1u << (nb & 31);
                            // ds.Union(pa.rid, pb.rid) (DisjointSet.hpp:64-84).  The rid of a live node is its set's root (node_N relies on the
                            // same invariant), so the two Find() calls return their arguments and compress nothing.
                            // Union by size: size(root) * 100 is the N of the live node whose rid the root is, i.e. Np and Nn
                            const int xr = rp, yr = rn;
                            if (xr != yr) {
                                const u16 sz = (u16)((Np + Nn) / (WIN * WIN));
                                if (Np < Nn) { S.dsp[xr] = (u16)yr; S.dss[yr] = sz; }
                                else { S.dsp[yr] = (u16)xr; S.dss[xr] = sz; }
                            }
                        }
                        // no wait for these stores: later loads of this wavefront are performed behind them in order, and the other
                        // wavefronts only read them behind the fence + barrier that starts an evaluation phase
                        heap_push(m, best_geo.mse);
                        cyc[2] += PEAC_CYCLES() - c0; c0 = PEAC_CYCLES();
                        // mergeNbsFrom (AHCPlaneSeg.hpp:379-404): union of the two sorted lists minus {p, nb} (both dead by now),
                        // built cooperatively: prefix counts of the surviving entries, then every entry writes itself to its rank.
                        const int ca = ncnt(p), cb = ncnt(nb);
                        const int need = 2 * (ca + cb) + 3 + (ca + cb) / 4 + 8;
                        if (pool_top + need > L.pool_cap) {             // compact the pool: live merged nodes only (rare)
                            if (lane == 0) {
                                int top = 4 * NB + 1;                    // first list entry (its capacity header sits at top - 1)
                                for (int id = NB; id < m; id++) {
                                    if (is_dead(id) && id != p && id != nb) { set_ncnt(id, 0); continue; }
                                    const int c = ncnt(id), o = noff(id);
                                    if (top > o) { top = 0x7fff0000; break; }   // a list would grow over unread ones: report a capacity error
                                    int n2 = 0;
                                    for (int t = 0; t < c; t++) { const int v = S.pool[o + t]; if (!is_dead(v) || id == p || id == nb) S.pool[top + n2++] = (u16)v; }   // top <= o: in place
                                    const int cap = (id != p && id != nb) ? n2 + max(8, n2 / 4) : n2;
                                    S.pool[top - 1] = (u16)cap; set_noff(id, top); set_ncnt(id, n2);
                                    top += cap + 1;
                                }
                                s_scalar[3] = top - 1;
                            }
                            wfence();
                            pool_top = s_scalar[3];
                            if (pool_top > L.pool_cap) { err = 2; break; }
                        }
                        const int ca2 = ncnt(p), cb2 = ncnt(nb);
                        if (pool_top + 2 * (ca2 + cb2) + 3 + (ca2 + cb2) / 4 + 8 > L.pool_cap) { err = 2; break; }
                        const u16* A = S.pool + noff(p);
                        const u16* Bl = S.pool + noff(nb);
                        const int off = pool_top + 1;                   // pool_top itself becomes the capacity header
                        int n;
                        if (ca2 <= 64 && cb2 <= 64) {
                            // both lists fit the wavefront: lane i owns A[i] and B[i]; the two lower-bound searches (A[i] in B, B[i] in A) run
                            // interleaved, the ranks of the survivors are popcounts of two ballots and every survivor is written straight to
                            // its final slot (no prefix arrays, no second pass, no copy).  The lists live in global memory: one round trip
                            // brings both into an LDS stage, the searches then run on LDS.
                            const bool inA = lane < ca2, inB = lane < cb2;
                            const int xa = inA ? (int)A[lane] : 0, xb = inB ? (int)Bl[lane] : 0;
                            s_stageA[lane] = (u16)xa; s_stageB[lane] = (u16)xb;
                            wfence();
                            const u16* A = s_stageA; const u16* Bl = s_stageB;
                            const bool liveA = inA && !is_dead(xa), liveB = inB && !is_dead(xb);
                            int loA = 0, hiA = liveA ? cb2 : 0;         // lower bound of xa in B
                            int loB = 0, hiB = liveB ? ca2 : 0;         // lower bound of xb in A
                            while (__ballot(loA < hiA || loB < hiB)) {
                                const int mA = (loA + hiA) >> 1, mB = (loB + hiB) >> 1;
                                const int vB = Bl[min(mA, max(cb2 - 1, 0))], vA = A[min(mB, max(ca2 - 1, 0))];
                                if (loA < hiA) { if (vB < xa) loA = mA + 1; else hiA = mA; }
                                if (loB < hiB) { if (vA < xb) loB = mB + 1; else hiB = mB; }
                            }
                            const bool dup = liveB && loB < ca2 && (int)A[min(loB, max(ca2 - 1, 0))] == xb;   // then it is alive in A too
                            const bool keepB = liveB && !dup;
                            const unsigned long long mkA = __ballot(liveA), mkB = __ballot(keepB);
                            auto below = [](int k) -> unsigned long long { return k >= 64 ? ~0ull : (1ull << k) - 1ull; };
                            if (liveA) S.pool[off + __popcll(mkA & below(lane)) + __popcll(mkB & below(loA))] = (u16)xa;
                            if (keepB) S.pool[off + __popcll(mkB & below(lane)) + __popcll(mkA & below(loB))] = (u16)xb;
                            n = __popcll(mkA) + __popcll(mkB);
                            wfence();
                        } else {
                            u16* PA = S.pool + pool_top;                    // [ca2+1] exclusive counts of surviving A entries
                            u16* PB = PA + ca2 + 1;                         // [cb2+1] ... of surviving, non-duplicate B entries
                            u16* out = PB + cb2 + 1;
                            int na = 0, nbk = 0;
                            for (int i0 = 0; i0 < ca2; i0 += 64) {
                                const int i = i0 + lane;
                                const bool keep = i < ca2 && !is_dead(A[i]);
                                const unsigned long long mk = __ballot(keep);
                                if (i < ca2) PA[i] = (u16)(na + __popcll(mk & ((1ull << lane) - 1ull)));
                                na += __popcll(mk);
                            }
                            for (int j0 = 0; j0 < cb2; j0 += 64) {
                                const int j = j0 + lane;
                                bool keep = false;
                                if (j < cb2) {
                                    const int x = Bl[j];
                                    if (!is_dead(x)) {
                                        int lo = 0, hi = ca2;               // duplicate test: x in A (then it is alive there too)
                                        while (lo < hi) { const int mid = (lo + hi) >> 1; if (A[mid] < x) lo = mid + 1; else hi = mid; }
                                        keep = !(lo < ca2 && A[lo] == x);
                                    }
                                }
                                const unsigned long long mk = __ballot(keep);
                                if (j < cb2) PB[j] = (u16)(nbk + __popcll(mk & ((1ull << lane) - 1ull)));
                                nbk += __popcll(mk);
                            }
                            if (lane == 0) { PA[ca2] = (u16)na; PB[cb2] = (u16)nbk; }
                            wfence();
                            for (int i0 = 0; i0 < ca2; i0 += 64) {
                                const int i = i0 + lane;
                                if (i < ca2 && PA[i + 1] != PA[i]) {
                                    const int x = A[i];
                                    int lo = 0, hi = cb2;
                                    while (lo < hi) { const int mid = (lo + hi) >> 1; if (Bl[mid] < x) lo = mid + 1; else hi = mid; }
                                    out[PA[i] + PB[lo]] = (u16)x;
                                }
                            }
                            for (int j0 = 0; j0 < cb2; j0 += 64) {
                                const int j = j0 + lane;
                                if (j < cb2 && PB[j + 1] != PB[j]) {
                                    const int x = Bl[j];
                                    int lo = 0, hi = ca2;
                                    while (lo < hi) { const int mid = (lo + hi) >> 1; if (A[mid] < x) lo = mid + 1; else hi = mid; }
                                    out[PB[j] + PA[lo]] = (u16)x;
                                }
                            }
                            wfence();
                            n = na + nbk;
                            // move the list down over the scratch counters and reserve some slack for later appends
                            for (int k0 = 0; k0 < n; k0 += 64) {            // forward copy, destination below source: chunk-safe
                                const int k = k0 + lane;
                                const int v = k < n ? out[k] : 0;
                                wfence();
                                if (k < n) S.pool[off + k] = (u16)v;
                                wfence();
                            }
                        }
                        const int cap = n + max(8, n / 4);
                        pool_top = off + cap;
                        if (lane == 0) { S.pool[off - 1] = (u16)cap; set_noff(m, off); set_ncnt(m, n); set_ncnt(p, 0); set_ncnt(nb, 0); }
                        wfence();
                        cyc[3] += PEAC_CYCLES() - c0; c0 = PEAC_CYCLES();
                        {   // nb->nbs.insert(this): m is the largest id so far -> append; a full list is compacted first
                            const u16* lstm = S.pool + off;
                            for (int k = lane; k < n; k += 64) {
                                const int q = lstm[k];
                                u16* ql = S.pool + noff(q);
                                int c = ncnt(q);
                                if (q < NB && !hdr_all) {
                                    // a block's list: four u16 in one 8-byte word (offset 4 * q).  One load, compaction in registers, one store.
                                    if (c >= 4) {
                                        const uint2 w = *(const uint2*)ql;
                                        const int e0 = w.x & 0xffff, e1 = w.x >> 16, e2 = w.y & 0xffff, e3 = w.y >> 16;
                                        int n2 = 0;
                                        if (!is_dead(e0)) { ql[n2] = (u16)e0; n2++; }
                                        if (!is_dead(e1)) { ql[n2] = (u16)e1; n2++; }
                                        if (!is_dead(e2)) { ql[n2] = (u16)e2; n2++; }
                                        if (!is_dead(e3)) { ql[n2] = (u16)e3; n2++; }
                                        c = n2;
                                    }
                                } else if (c >= list_cap(q)) { int n2 = 0; for (int t = 0; t < c; t++) { const int v = ql[t]; if (!is_dead(v)) ql[n2++] = (u16)v; } c = n2; }
                                ql[c] = (u16)m; set_ncnt(q, c + 1);
                                cinval(q);                               // q's live-neighbour set changed: its cached candidates are stale
                            }
                        }
                        wfence();
                        cyc[5] += PEAC_CYCLES() - c0;
                    } else {
                        extract(p);
                        for (int k = lane; k < cnt; k += 64) { const int q = lst[k]; if (!is_dead(q)) cinval(q); }   // p leaves their live sets
                        mark_dead(p);
                    }
                    ++step;
                }
                if (lane == 0) s_cmd = need_eval ? 1 : 0;
                gfence();
            }
            if (!coop) break;
            __syncthreads();
            if (s_cmd == 0) break;
            e0 = PEAC_CYCLES();
            eval_phase();
            __threadfence_block();
            __syncthreads();
            cyc[4] += PEAC_CYCLES() - e0; dbg_phases++; dbg_nodes += s_nlist;
        }
        if (wave != 0) return;
        while (heap_n > 0 && !err) { const int p = heap_pop(); extract(p); mark_dead(p); }
        wfence();
        if (lane == 0) {   // std::sort(extractedPlanes, b->N < a->N): insertion sort (stable)
            for (int i = 1; i < n_ext; i++) {
                const int v = s_ext[i];
                int j = i;
                while (j > 0 && node_N(s_ext[j - 1]) < node_N(v)) { s_ext[j] = s_ext[j - 1]; j--; }
                s_ext[j] = v;
            }
        }
        wfence();
    };
    if (PHASE == 0) {
    if (wave == 0) {
        for (int b0 = 0; b0 < NB; b0 += 64) {   // minQ.push in block order (:809); 64 blocks fetched per round
            const int b = b0 + lane;
            const bool in = b < NB && (g_flags[b] & 1);
            const double m = in ? geo_of(b)[6] : 0.0;
            unsigned long long mask = __ballot(in);
            while (mask) {
                const int src = __ffsll((long long)mask) - 1;
                mask &= mask - 1;
                heap_push(b0 + src, __shfl(m, src));
            }
        }
        mark();
    } else mark();
    ah_cluster(true);            // all four wavefronts (cooperative candidate evaluation)
    if (wave == 0 && lane == 0) { s_scalar[0] = n_ext; s_scalar[1] = err; g_hand[0] = n_ext; g_hand[1] = err; g_hand[2] = n_nodes; }
    __syncthreads();
    n_ext = s_scalar[0]; err = s_scalar[1];
    mark();
    // hand the clustering state over to peac_refine: set sizes, root ids, dead bits, the extracted planes
    {
        unsigned* o_nouse = (unsigned*)(F + L.off_h_nouse); unsigned* o_cval = (unsigned*)(F + L.off_h_cval);
        for (int t = tid; t < (L.NB2 + 31) / 32; t += NT) { o_nouse[t] = S.nouse[t]; o_cval[t] = 0; }
        for (int t = tid; t < MAX_PLANES; t += NT) g_hand[4 + t] = t < n_ext ? s_ext[t] : 0;
    }
    if (tid == 0) {
        status[frame] = err;
        if (timing) {
            for (int t = 0; t < 4; t++) timing[(size_t)frame * TSLOTS + t] = t < nph ? tphase[t] - tphase[0] : 0;
            timing[(size_t)frame * TSLOTS + 9] = n_nodes;
            timing[(size_t)frame * TSLOTS + 7] = ((long long)dbg_phases << 40) | ((long long)dbg_nodes << 20) | dbg_hits;   // cooperative ahCluster: phases, nodes evaluated, cache hits
            for (int t = 0; t < 6; t++) timing[(size_t)frame * TSLOTS + 10 + t] = cyc[t];
        }
    }
    return;
    }

    // ---- refineDetails (:299-379): findBlockMembership (:485-587), all threads ----
    {
        unsigned* m4 = (unsigned*)member;
        for (int i = tid; i < (W * H + 3) / 4; i += NT) m4[i] = 0xffffffffu;
    }
    for (int b = tid; b < NB; b += NT) {
        const int i = b / Nw, j = b - i * Nw;
        const int setid = lds_find(S.dsp, b);   // concurrent path compression only ever stores true roots: benign
        int plid = -1;
        if (S.dss[setid] * (WIN * WIN) >= MIN_SUPPORT) {
            bool same = true;
            if (j > 0 && lds_find(S.dsp, b - 1) != setid) same = false;
            if (j < Nw - 1 && lds_find(S.dsp, b + 1) != setid) same = false;
            if (i > 0 && lds_find(S.dsp, b - Nw) != setid) same = false;
            if (i < Nh - 1 && lds_find(S.dsp, b + Nw) != setid) same = false;
            if (same) {
                plid = 0;                                    // std::map::operator[] default when absent
                for (int q = 0; q < n_ext; q++) if (S.rid[s_ext[q]] == setid) { plid = q; break; }
                s_valid[plid] = 1;
            }
        }
        S.blk[b] = (signed char)plid;
    }
    __syncthreads();
    for (int b = tid; b < NB; b += NT) {       // seeds per block (depend on blkMap of self / up / left only)
        const int i = b / Nw, j = b - i * Nw, me = S.blk[b];
        int c = 0;
        if (me < 0) { if (i > 0 && S.blk[b - Nw] >= 0) c += WIN - 1; if (j > 0 && S.blk[b - 1] >= 0) c += WIN - 1; }
        else { if (i > 0 && S.blk[b - Nw] != me) c += WIN - 1; if (j > 0 && S.blk[b - 1] != me) c += WIN - 1; }
        seedcnt[b] = c;
    }
    __syncthreads();
    if (wave == 0) {                           // exclusive scan in block order
        int run = 0;
        for (int b0 = 0; b0 < NB; b0 += 64) {
            const int b = b0 + lane;
            const int c = b < NB ? seedcnt[b] : 0;
            int incl = c;
            for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o); if (lane >= o) incl += t; }
            if (b < NB) seedcnt[b] = run + incl - c;
            run += __shfl(incl, 63);
        }
        if (lane == 0) { s_scalar[2] = run; if (run > L.q_cap) s_scalar[1] = 5; }
    }
    __syncthreads();
    err = s_scalar[1];
    for (int b = tid; b < NB && !err; b += NT) {
        const int i = b / Nw, j = b - i * Nw, me = S.blk[b];
        int o = seedcnt[b];
        if (me >= 0)   // membershipImg(block) = plid
            for (int y = i * WIN; y < (i + 1) * WIN; y++) for (int x = j * WIN; x < (j + 1) * WIN; x++) member[y * W + x] = (signed char)me;
        if (me < 0) {
            if (i > 0 && S.blk[b - Nw] >= 0) { const int up = S.blk[b - Nw]; const int sp = (i * WIN - 1) * W + j * WIN; for (int k = 1; k < WIN; ++k) queue[o++] = qent(sp + k, up); }
            if (j > 0 && S.blk[b - 1] >= 0) { const int lp = S.blk[b - 1]; const int sp = (i * WIN) * W + j * WIN - 1; for (int k = 0; k < WIN - 1; ++k) queue[o++] = qent(sp + k * W, lp); }
        } else {
            if (i > 0 && S.blk[b - Nw] != me) { const int sp = (i * WIN) * W + j * WIN; for (int k = 0; k < WIN - 1; ++k) queue[o++] = qent(sp + k, me); }
            if (j > 0 && S.blk[b - 1] != me) { const int sp = (i * WIN) * W + j * WIN; for (int k = 1; k < WIN; ++k) queue[o++] = qent(sp + k * W, me); }
        }
    }
    __threadfence_block();
    __syncthreads();
    mark();

    // ---- floodFill (:428-476), all threads.  Per step: 512 queue entries x 4 neighbours = 2048 (entry, neighbour) pairs, eight per thread
    //      (pair p = entry * 4 + direction is the reference's processing order).  Pairs that hit the same pixel are replayed in pair order:
    //      per round the smallest pair index wins the pixel's slot (atomicMin on a key whose high bits count the rounds DOWN, so stale
    //      entries of earlier rounds lose by themselves and the slots are never reset); plane-plane connect() is a commutative set
    //      insertion and goes to an LDS bit matrix; queue pushes are appended in pair order.
    {
        constexpr int FJ = 8;                                      // pairs per thread and step (512 queue entries per step: the step count, not the work, sets the time)
        unsigned* slot = (unsigned*)s_slot;
        for (int t = tid; t < NSLOT; t += NT) slot[t] = 0xffffffffu;
        __syncthreads();
        unsigned epoch = 0;
        const double factor = (double)K.factor;
        int q_head = 0, q_tail = s_scalar[2];
        while (q_head < q_tail && !err) {
            const int nent = min(FJ * NT / 4, q_tail - q_head);
            bool act[FJ], geo_ok[FJ], done[FJ], push[FJ];
            int cIdx[FJ], plid[FJ], pidx[FJ], hsl[FJ];
            float cdist[FJ];
#pragma unroll
            for (int j = 0; j < FJ; j++) {
                pidx[j] = tid + NT * j;
                const int e = pidx[j] >> 2, dir = pidx[j] & 3;
                act[j] = e < nent; cIdx[j] = -1; plid[j] = -1; geo_ok[j] = false; cdist[j] = -1.f; push[j] = false;
                int cx = 0, cy = 0;
                if (act[j]) {
                    const unsigned ent_p = queue[q_head + e];
                    const int2 ent = make_int2((int)(ent_p & 0xffffffu), (int)(ent_p >> 24));
                    plid[j] = ent.y;
                    const int sy = ent.x / W, sx = ent.x - sy * W;
                    // getValid4Neighbor order (:398-410): left, right, up, down, invalid ones skipped
                    if (dir == 0) { act[j] = sx > 0; cIdx[j] = ent.x - 1; }
                    else if (dir == 1) { act[j] = sx < W - 1; cIdx[j] = ent.x + 1; }
                    else if (dir == 2) { act[j] = sy > 0; cIdx[j] = ent.x - W; }
                    else { act[j] = sy < H - 1; cIdx[j] = ent.x + W; }
                    if (act[j]) { cy = cIdx[j] / W; cx = cIdx[j] - cy * W; }
                }
                // slot of the target pixel: x + (W | 1) * y: an odd row stride, so neither a horizontal nor a vertical run of the frontier
                // folds onto a few slots (with stride W = 640 a vertical run of 256 pixels has 16 distinct slots)
                hsl[j] = (cIdx[j] + ((W & 1) ? 0 : cy)) & (NSLOT - 1);
                if (act[j]) {
                    const int by = cy / WIN, bx = cx / WIN;
                    const int blkid = (by < Nh && bx < Nw) ? by * Nw + bx : -1;
                    if (blkid >= 0 && S.blk[blkid] >= 0) act[j] = false;          // only "black" blocks are refined
                }
                if (act[j]) {
                    const double z = (double)D[(size_t)cy * pitch_px + cx] * factor;
                    if (z != 0) {
                        const double x = ((double)cx - (double)K.cx) * z / (double)K.fx;
                        const double y = ((double)cy - (double)K.cy) * z / (double)K.fy;
                        const double* g = geo_of(s_ext[plid[j]]);
                        const double sd = g[3] * (x - g[0]) + g[4] * (y - g[1]) + g[5] * (z - g[2]);
                        cdist[j] = (float)fabs(sd);
                        geo_ok[j] = (double)cdist[j] * (double)cdist[j] < 9 * g[6] + 1e-5;
                    }
                }
                done[j] = !act[j];
            }
            while (true) {
                bool pend = false;
#pragma unroll
                for (int j = 0; j < FJ; j++) pend = pend || !done[j];
                if (!__syncthreads_or(pend)) break;
                const unsigned ek = (0x7ffffu - epoch) << 13;           // pair indices are < 8192 (NT <= 1024)
                epoch++;
#pragma unroll
                for (int j = 0; j < FJ; j++) if (!done[j]) atomicMin(&slot[hsl[j]], ek | (unsigned)pidx[j]);
                __syncthreads();
#pragma unroll
                for (int j = 0; j < FJ; j++) {
                    if (!done[j] && slot[hsl[j]] == (ek | (unsigned)pidx[j])) {
                        const int trail = (int)member[cIdx[j]];
                        if (!(trail <= -6) && !(trail >= 0 && trail == plid[j])) {
                            if (geo_ok[j]) {
                                if (trail >= 0 && nsim(s_ext[plid[j]], s_ext[trail]) >= C.cos_refine) {   // n_pl.connect(pl)
                                    atomicOr(&s_adj[trail][plid[j] >> 5], 1u << (plid[j] & 31));
                                    atomicOr(&s_adj[plid[j]][trail >> 5], 1u << (trail & 31));
                                }
                                if (cdist[j] < (trail >= 0 ? distMap[cIdx[j]] : 3.4028234663852886e38f)) { member[cIdx[j]] = (signed char)plid[j]; distMap[cIdx[j]] = cdist[j]; push[j] = true; }
                                else if (trail < 0) member[cIdx[j]] = (signed char)(trail - 1);
                            } else if (trail < 0) member[cIdx[j]] = (signed char)(trail - 1);
                        }
                        done[j] = true;
                    }
                    // a later pair of this thread may target the pixel an earlier one just wrote: it lost the slot (smaller pair index wins)
                    // and is replayed in the next round, after the fence below
                }
                __threadfence_block();
            }
            // pushes in pair order (pair p = tid + NT * j): j-major, then wavefront, then lane
            unsigned long long pm[FJ];
#pragma unroll
            for (int j = 0; j < FJ; j++) { pm[j] = __ballot(push[j]); if (lane == 0) s_pcnt[j * NW + wave] = __popcll(pm[j]); }
            __syncthreads();
            int base = 0, mybase[FJ];
#pragma unroll
            for (int j = 0; j < FJ; j++)
                for (int w = 0; w < NW; w++) { if (w == wave) mybase[j] = base; base += s_pcnt[j * NW + w]; }
            if (q_tail + base > L.q_cap) err = 5;
            else {
#pragma unroll
                for (int j = 0; j < FJ; j++)
                    if (push[j]) queue[q_tail + mybase[j] + __popcll(pm[j] & ((1ull << lane) - 1ull))] = qent(cIdx[j], plid[j]);
            }
            q_tail += base;
            q_head += nent;
            __threadfence_block();
            __syncthreads();
        }
        if (tid == 0) { s_scalar[2] = q_tail; if (err) s_scalar[1] = err; }
    }
    __syncthreads();
    err = s_scalar[1];
    mark();

    // ---- final ahCluster over the surviving planes (:319-326): wave 0 ----
    const int n_old = n_ext;
    if (wave == 0) {
        for (int q = lane; q < n_old; q += 64) s_old[q] = s_ext[q];
        wfence();
        // neighbour lists in ascending NODE id (== std::set<PlaneSeg*> order) from the bit matrix.  Every list is empty
        // after the first ahCluster (all nodes were disconnected), so the pool is reused from the start.
        for (int q = lane; q < n_old; q += 64) {
            const int id = s_old[q];
            const int off = q * (n_old + 1) + 1;
            int c = 0;
            for (int r = 0; r < n_old; r++)
                if (s_adj[q][r >> 5] & (1u << (r & 31))) lst_insert(S.pool + off, c, s_old[r]);
            S.pool[off - 1] = (u16)n_old; set_noff(id, off); set_ncnt(id, c);
            atomicAnd(&S.nouse[id >> 5], ~(1u << (id & 31)));   // back in the graph
        }
        pool_top = n_old * (n_old + 1);
        hdr_all = true;
        wfence();
        n_ext = 0;
        heap_n = 0;
        for (int q = 0; q < n_old; q++) if (s_valid[q]) heap_push(s_old[q], geo_of(s_old[q])[6]);
        if (!err) ah_cluster(false);
        for (int q = lane; q < n_old; q += 64) {
            int m = -1;
            if (s_valid[q]) {
                const int np_rid = lds_find(S.dsp, S.rid[s_old[q]]);
                for (int j = 0; j < n_ext; j++) if (S.rid[s_ext[j]] == np_rid) { m = j; break; }
            }
            s_plidmap[q] = m;
        }
        if (lane == 0) { s_scalar[0] = n_ext; if (err) s_scalar[1] = err; }
    }
    __threadfence_block();
    __syncthreads();
    n_ext = s_scalar[0]; err = s_scalar[1];

    // ---- relabel (:327-372) and plane parameters, all threads ----
    for (int i = tid; i < W * H; i += NT) {
        const int plid = (int)member[i];
        lab[i] = (plid >= 0 && s_plidmap[plid] >= 0) ? s_plidmap[plid] : -1;
    }
    for (int j = tid; j < n_ext; j += NT) {
        const int id = s_ext[j];
        double* o = planes + ((size_t)frame * MAX_PLANES + j) * 8;
        o[0] = (double)g_N[id];
        for (int t = 0; t < 3; t++) { o[1 + t] = geo_of(id)[3 + t]; o[4 + t] = geo_of(id)[t]; }
        o[7] = geo_of(id)[6];
    }
    mark();
    if (tid == 0) {
        n_planes[frame] = n_ext; status[frame] = err;
        if (timing) {   // marks of this kernel: [1] seeds done, [2] floodFill done, [3] end; stored behind the clustering kernel's [0..3]
            const long long base = timing[(size_t)frame * TSLOTS + 3];
            for (int t = 2; t < 5; t++) timing[(size_t)frame * TSLOTS + 2 + t] = base + (t < nph ? tphase[t] - tphase[0] : 0);
            timing[(size_t)frame * TSLOTS + 8] = s_scalar[2];
        }
    }
}

// peac_ahc: one workgroup (one wavefront) per frame, up to six per CU.  A workgroup takes
// the next frame from a counter when it STARTS instead of using its block index: frames differ by up to 1.7x in merge steps, the dispatcher
// deals block indices round-robin over the 8 XCDs, and a periodic mix of frames would otherwise send every slow frame to the same XCD.
// (A persistent one-workgroup-per-CU loop costs ~50 more VGPRs and with them the co-residency of lsd_detect's wavefront on the same SIMDs.)
__global__ __launch_bounds__(NT_AHC) void peac_ahc(Layout L, Intr K, Consts C, const uint16_t* __restrict__ depth, int pitch_px,
                                               int64_t frame_stride_px, uint8_t* __restrict__ ws, int32_t* __restrict__ status,
                                               long long* __restrict__ timing, int* __restrict__ next_frame, const int* __restrict__ order) {
    __shared__ int s_frame;
    if (threadIdx.x == 0) { const int k = atomicAdd(next_frame, 1); s_frame = order ? order[k] : k; }
    __syncthreads();
    segment_frame<0, NT_AHC>(L, K, C, depth, pitch_px, frame_stride_px, ws, nullptr, 0, nullptr, nullptr, status, timing, s_frame);
}

// peac_refine: one workgroup per frame, ~17 KB of LDS: several per CU, and beside the clustering workgroups of the next launch.
// (four wavefronts per SIMD: all 1024 frames of a batch resident at once instead of two rounds of 512; 128 VGPRs with 77 spilled: 16.4 -> 14.4 ms alone)
__global__ __launch_bounds__(NT_REFINE, 4) void peac_refine(Layout L, Intr K, Consts C, const uint16_t* __restrict__ depth, int pitch_px,
                                                  int64_t frame_stride_px, uint8_t* __restrict__ ws, int32_t* __restrict__ labels,
                                                  int64_t label_stride, double* __restrict__ planes, int32_t* __restrict__ n_planes,
                                                  int32_t* __restrict__ status, long long* __restrict__ timing) {
    segment_frame<1, NT_REFINE>(L, K, C, depth, pitch_px, frame_stride_px, ws, labels, label_stride, planes, n_planes, status, timing, (int)blockIdx.x);
}

// The same refinement with 1024 threads per frame, for small batches: the flood fill processes 2048 queue entries per step instead of 512 (its result does not
// depend on the step size: pairs that meet at a pixel are replayed in the reference's order), one frame per CU.
__global__ __launch_bounds__(NT_REFINE_WIDE) void peac_refine_wide(Layout L, Intr K, Consts C, const uint16_t* __restrict__ depth, int pitch_px,
                                                  int64_t frame_stride_px, uint8_t* __restrict__ ws, int32_t* __restrict__ labels,
                                                  int64_t label_stride, double* __restrict__ planes, int32_t* __restrict__ n_planes,
                                                  int32_t* __restrict__ status, long long* __restrict__ timing) {
    segment_frame<1, NT_REFINE_WIDE>(L, K, C, depth, pitch_px, frame_stride_px, ws, labels, label_stride, planes, n_planes, status, timing, (int)blockIdx.x);
}

// Longest-first order for the NEXT call with the same batch size: slot b of a batch is one camera stream, consecutive frames of a stream cost
// about the same, and a launch ends with its slowest frames - so they should start first.  order[rank] = frame, rank by the duration the
// last call measured (ties by frame index).  Only the schedule depends on it, never a result.
__global__ void peac_order(const long long* __restrict__ timing, int B, int* __restrict__ order) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B) return;
    const long long ci = timing[(size_t)i * TSLOTS + 3];         // clock ticks of the clustering kernel (entry to the end of ahCluster)
    int rank = 0;
    for (int j = 0; j < B; j++) {
        const long long cj = timing[(size_t)j * TSLOTS + 3];
        rank += (cj > ci || (cj == ci && j < i)) ? 1 : 0;
    }
    order[rank] = i;
}

}  // namespace peac
}  // namespace planar

#include "peac_ahc2.h"

// ==========================================================================================================
// host side
// ==========================================================================================================
using namespace planar;

struct planar_peac {
    planar_ctx* ctx = nullptr;
    int W = 0, H = 0, max_batch = 0;
    peac::Layout L{};
    peac::Consts C{};
    int smem = 0, smem2 = 0;
    // kernel variants, set only through planar_peac_set_variant (tools' A/B runs and tests; no environment switch reaches the product path)
    int wide_below = 64;                                      // batches up to this size refine with 1024 threads per frame (0 = never)
    bool exact_only = false;                                  // skip the fast clustering attempt: every frame through the exact heap
    bool legacy_ahc = false;                                  // the round-2 clustering kernel (eager neighbour lists)
    DevBuf d_ws, d_status, d_timing, d_next, d_order;
    int order_B = 0;                                          // batch size d_order was computed for (0: none yet)
    DevBuf d_depth, d_labels, d_planes, d_nplanes;   // staging for the host-pointer entry point
    bool profiling = false;                                   // planar_peac_set_profiling: five events per recorded call
    std::vector<std::vector<hipEvent_t>> ev_sets;
    size_t ev_used = 0;
    ~planar_peac() { for (auto& v : ev_sets) for (hipEvent_t e : v) (void)hipEventDestroy(e); }
};

extern "C" {

int planar_peac_create(planar_ctx* ctx, int width, int height, int max_batch, planar_peac** out) {
    PLANAR_REQUIRE(ctx && out, PLANAR_EINVAL, "null argument");
    *out = nullptr;
    PLANAR_REQUIRE(width >= 2 * peac::WIN && height >= 2 * peac::WIN && width <= 4096 && height <= 4096, PLANAR_EINVAL, "image size out of range");
    PLANAR_REQUIRE(max_batch >= 1, PLANAR_EINVAL, "max_batch must be >= 1");
    PLANAR_HIP_CHECK(hipSetDevice(ctx->device));
    planar_peac* o = new (std::nothrow) planar_peac();
    PLANAR_REQUIRE(o != nullptr, PLANAR_ENOMEM, "host allocation failed");
    o->ctx = ctx; o->W = width; o->H = height; o->max_batch = max_batch;
    o->L = peac::make_layout(width, height);
    o->C = peac::make_consts();
    const peac::Layout& L = o->L;
    o->smem2 = peac::ahc2_smem_bytes(L);
    // peac_ahc (legacy): heap keys + ids, neighbour-list pool, list offsets / counts, set sizes, root ids, dead / cache-valid bits
    o->smem = L.NB * 4 + L.NB * 2 + 2 * ((L.NB2 + 31) / 32) * 4 + L.NB + 16;
    if (L.pool_cap > 65535 || L.NB2 > 65535 || o->smem > 150 * 1024 || L.NB > 3072) { delete o; set_error("planar_peac_create: %dx%d needs %d B of LDS for the merge heap", width, height, o->smem); return PLANAR_EINVAL; }
    int rc;
    if ((rc = o->d_ws.alloc((size_t)max_batch * L.frame_bytes)) || (rc = o->d_status.alloc((size_t)max_batch * 4)) ||
        (rc = o->d_timing.alloc((size_t)max_batch * peac::TSLOTS * 8)) || (rc = o->d_next.alloc(256)) || (rc = o->d_order.alloc((size_t)max_batch * 4))) { delete o; return rc; }
    if (o->smem > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)peac::peac_ahc, hipFuncAttributeMaxDynamicSharedMemorySize, o->smem);
        if (e != hipSuccess) { delete o; set_error("hipFuncSetAttribute failed: %s", hipGetErrorString(e)); return PLANAR_EDEVICE; }
    }
    *out = o;
    return PLANAR_OK;
}

void planar_peac_destroy(planar_peac* p) { delete p; }
// A/B aid (tools/peac_ab.py, tests): clustering variant 0 = product (fast attempt + exact redo), 1 = exact only, 2 = the round-2 kernel; wide_below < 0 keeps the default.
// All variants produce the same labels and planes; nothing in the product calls this.
int planar_peac_set_variant(planar_peac* p, int clustering, int wide_below) {
    PLANAR_REQUIRE(p && clustering >= 0 && clustering <= 2, PLANAR_EINVAL, "bad argument");
    p->exact_only = clustering == 1; p->legacy_ahc = clustering == 2;
    if (wide_below >= 0) p->wide_below = wide_below;
    return PLANAR_OK;
}
int planar_peac_max_planes(void) { return peac::MAX_PLANES; }

int planar_peac_segment_dev(planar_peac* p, const uint16_t* d_depth, int B, int pitch_px, int64_t frame_stride_px, float fx, float fy,
                            float cx, float cy, float depth_factor, int32_t* d_labels, double* d_planes, int32_t* d_n_planes) {
    PLANAR_REQUIRE(p && d_depth && d_labels && d_planes && d_n_planes, PLANAR_EINVAL, "null argument");
    PLANAR_REQUIRE(B >= 1 && B <= p->max_batch, PLANAR_EINVAL, "B must be in [1, max_batch]");
    PLANAR_REQUIRE(pitch_px >= p->W && frame_stride_px >= (int64_t)pitch_px * p->H, PLANAR_EINVAL, "pitch/frame_stride too small");
    hipStream_t st = p->ctx->stream;
    const peac::Intr K{fx, fy, cx, cy, depth_factor};
    // optional HIP-event timing of each launch (planar_peac_set_profiling): events on the stream the kernels run on, right before / after each launch
    std::vector<hipEvent_t>* evs = nullptr;
    if (p->profiling) {
        if (p->ev_used == p->ev_sets.size()) {
            std::vector<hipEvent_t> v(5);
            for (hipEvent_t& e : v) PLANAR_HIP_CHECK(hipEventCreate(&e));
            p->ev_sets.push_back(v);
        }
        evs = &p->ev_sets[p->ev_used++];
    }
    int li = 0;
    auto mark = [&]() { if (evs) (void)hipEventRecord((*evs)[li], st); li++; };
    PLANAR_HIP_CHECK(hipMemsetAsync(p->d_next.p, 0, 4, st));
    mark();
    hipLaunchKernelGGL(peac::peac_blocks, dim3((p->L.NB + 63) / 64, B), dim3(64), 0, st, p->L, K, d_depth, pitch_px, frame_stride_px, p->d_ws.as<uint8_t>());
    mark();
    if (p->legacy_ahc)
        hipLaunchKernelGGL(peac::peac_ahc, dim3(B), dim3(peac::NT_AHC), p->smem, st, p->L, K, p->C, d_depth, pitch_px, frame_stride_px,
                           p->d_ws.as<uint8_t>(), p->d_status.as<int32_t>(), p->d_timing.as<long long>(), p->d_next.as<int>(),
                           p->order_B == B ? p->d_order.as<int>() : nullptr);
    else {
        // the fast attempt (tournament queue), then the exact kernel for the frames it gave up on (bit-equal keys of two live nodes: degenerate input)
        // (the workgroup that gave up redoes its frame itself: no second launch)
        if (!p->exact_only)
            hipLaunchKernelGGL(peac::peac_ahc3, dim3(B), dim3(64), p->smem2, st, p->L, p->C, p->d_ws.as<uint8_t>(), p->d_status.as<int32_t>(),
                               p->d_timing.as<long long>(), p->d_next.as<int>(), p->order_B == B ? p->d_order.as<int>() : nullptr, 1);
        else
            hipLaunchKernelGGL(peac::peac_ahc2, dim3(B), dim3(64), p->smem2, st, p->L, p->C, p->d_ws.as<uint8_t>(), p->d_status.as<int32_t>(),
                               p->d_timing.as<long long>(), p->d_next.as<int>(), p->order_B == B ? p->d_order.as<int>() : nullptr, 0);
    }
    mark();
    hipLaunchKernelGGL(peac::peac_order, dim3((B + 255) / 256), dim3(256), 0, st, p->d_timing.as<long long>(), B, p->d_order.as<int>());
    mark();
    if (B <= p->wide_below)
        hipLaunchKernelGGL(peac::peac_refine_wide, dim3(B), dim3(peac::NT_REFINE_WIDE), 0, st, p->L, K, p->C, d_depth, pitch_px, frame_stride_px,
                           p->d_ws.as<uint8_t>(), d_labels, (int64_t)p->W * p->H, d_planes, d_n_planes, p->d_status.as<int32_t>(), p->d_timing.as<long long>());
    else
        hipLaunchKernelGGL(peac::peac_refine, dim3(B), dim3(peac::NT_REFINE), 0, st, p->L, K, p->C, d_depth, pitch_px, frame_stride_px,
                           p->d_ws.as<uint8_t>(), d_labels, (int64_t)p->W * p->H, d_planes, d_n_planes, p->d_status.as<int32_t>(), p->d_timing.as<long long>());
    mark();
    p->order_B = B;
    PLANAR_HIP_CHECK(hipGetLastError());
    return PLANAR_OK;
}

// Per-launch HIP-event timing (bench.py's roofline leg): slots peac_blocks, peac_ahc, peac_order, peac_refine
int planar_peac_set_profiling(planar_peac* p, int enable) {
    PLANAR_REQUIRE(p != nullptr, PLANAR_EINVAL, "peac is null");
    PLANAR_HIP_CHECK(hipStreamSynchronize(p->ctx->stream));
    p->profiling = enable != 0;
    p->ev_used = 0;
    return PLANAR_OK;
}
int planar_peac_get_profile(planar_peac* p, double* total_ms /* [4] */, int64_t* calls) {
    PLANAR_REQUIRE(p && total_ms && calls, PLANAR_EINVAL, "null argument");
    PLANAR_HIP_CHECK(hipStreamSynchronize(p->ctx->stream));
    for (int i = 0; i < 4; i++) total_ms[i] = 0;
    for (size_t c = 0; c < p->ev_used; c++)
        for (int i = 0; i < 4; i++) {
            float ms = 0;
            PLANAR_HIP_CHECK(hipEventElapsedTime(&ms, p->ev_sets[c][i], p->ev_sets[c][i + 1]));
            total_ms[i] += ms;
        }
    *calls = (int64_t)p->ev_used;
    p->ev_used = 0;
    return PLANAR_OK;
}

// Returns PLANAR_ECAPACITY if a frame overflowed an internal capacity (node pool, flood-fill queue, > MAX_PLANES planes).
// Debug/profiling: per-frame phase timestamps of the last call (100 MHz wall clock ticks since kernel entry):
// [1] graph edges, [2] heap built, [3]/[4] ahCluster done, [5] seeds done, [6] floodFill done, [7] end, [8] queue entries, [9] nodes
int planar_peac_read_timing(planar_peac* p, int B, int64_t* out) {
    PLANAR_REQUIRE(p && out && B >= 1 && B <= p->max_batch, PLANAR_EINVAL, "bad argument");
    PLANAR_HIP_CHECK(hipStreamSynchronize(p->ctx->stream));
    PLANAR_HIP_CHECK(hipMemcpy(out, p->d_timing.p, (size_t)B * peac::TSLOTS * 8, hipMemcpyDeviceToHost));
    return PLANAR_OK;
}

// Debug aid (tools/peac_ab.py): the per-frame workspace layout {frame_bytes, NB, NB2, off_stats, off_geo, off_N, off_h_dsp, off_h_dss, off_h_rid,
// off_h_nouse, off_h_hand, off_crec} and a raw read of one frame's workspace after the last call.
int planar_peac_debug_layout(planar_peac* p, int64_t* out /* [12] */) {
    PLANAR_REQUIRE(p && out, PLANAR_EINVAL, "null argument");
    const peac::Layout& L = p->L;
    const int64_t v[12] = {(int64_t)L.frame_bytes, L.NB, L.NB2, (int64_t)L.off_stats, (int64_t)L.off_geo, (int64_t)L.off_N, (int64_t)L.off_h_dsp,
                           (int64_t)L.off_h_dss, (int64_t)L.off_h_rid, (int64_t)L.off_h_nouse, (int64_t)L.off_h_hand, (int64_t)L.off_crec};
    for (int i = 0; i < 12; i++) out[i] = v[i];
    return PLANAR_OK;
}
int planar_peac_debug_read(planar_peac* p, int frame, int64_t offset, int64_t bytes, void* out) {
    PLANAR_REQUIRE(p && out && frame >= 0 && frame < p->max_batch && offset >= 0 && bytes >= 0 && (size_t)(offset + bytes) <= p->L.frame_bytes, PLANAR_EINVAL, "bad argument");
    PLANAR_HIP_CHECK(hipStreamSynchronize(p->ctx->stream));
    PLANAR_HIP_CHECK(hipMemcpy(out, p->d_ws.as<uint8_t>() + (size_t)frame * p->L.frame_bytes + offset, (size_t)bytes, hipMemcpyDeviceToHost));
    return PLANAR_OK;
}

int planar_peac_check(planar_peac* p, int B) {
    PLANAR_REQUIRE(p && B >= 1 && B <= p->max_batch, PLANAR_EINVAL, "bad argument");
    std::vector<int32_t> st(B);
    PLANAR_HIP_CHECK(hipStreamSynchronize(p->ctx->stream));
    PLANAR_HIP_CHECK(hipMemcpy(st.data(), p->d_status.p, (size_t)B * 4, hipMemcpyDeviceToHost));
    for (int b = 0; b < B; b++)
        if (st[b] != 0) { set_error("planar_peac: frame %d overflowed an internal capacity (code %d)", b, st[b]); return PLANAR_ECAPACITY; }
    return PLANAR_OK;
}

int planar_peac_segment(planar_peac* p, const uint16_t* depth, int B, int pitch_px, int64_t frame_stride_px, float fx, float fy, float cx,
                        float cy, float depth_factor, int32_t* labels, double* planes, int32_t* n_planes) {
    PLANAR_REQUIRE(p && depth && labels && planes && n_planes, PLANAR_EINVAL, "null argument");
    PLANAR_REQUIRE(B >= 1 && B <= p->max_batch, PLANAR_EINVAL, "B must be in [1, max_batch]");
    PLANAR_REQUIRE(pitch_px >= p->W && frame_stride_px >= (int64_t)pitch_px * p->H, PLANAR_EINVAL, "pitch/frame_stride too small");
    PLANAR_HIP_CHECK(hipSetDevice(p->ctx->device));
    const size_t in_bytes = ((size_t)frame_stride_px * (B - 1) + (size_t)pitch_px * p->H) * 2;
    const size_t npx = (size_t)p->W * p->H;
    int rc;
    if (p->d_depth.bytes < in_bytes && (rc = p->d_depth.alloc(in_bytes))) return rc;
    if (!p->d_labels.p) {
        if ((rc = p->d_labels.alloc((size_t)p->max_batch * npx * 4)) || (rc = p->d_planes.alloc((size_t)p->max_batch * peac::MAX_PLANES * 64)) ||
            (rc = p->d_nplanes.alloc((size_t)p->max_batch * 4)))
            return rc;
    }
    hipStream_t st = p->ctx->stream;
    PLANAR_HIP_CHECK(hipMemcpyAsync(p->d_depth.p, depth, in_bytes, hipMemcpyHostToDevice, st));
    if ((rc = planar_peac_segment_dev(p, p->d_depth.as<uint16_t>(), B, pitch_px, frame_stride_px, fx, fy, cx, cy, depth_factor,
                                      p->d_labels.as<int32_t>(), p->d_planes.as<double>(), p->d_nplanes.as<int32_t>())))
        return rc;
    PLANAR_HIP_CHECK(hipMemcpyAsync(labels, p->d_labels.p, (size_t)B * npx * 4, hipMemcpyDeviceToHost, st));
    PLANAR_HIP_CHECK(hipMemcpyAsync(planes, p->d_planes.p, (size_t)B * peac::MAX_PLANES * 64, hipMemcpyDeviceToHost, st));
    PLANAR_HIP_CHECK(hipMemcpyAsync(n_planes, p->d_nplanes.p, (size_t)B * 4, hipMemcpyDeviceToHost, st));
    return planar_peac_check(p, B);
}

}  // extern "C"
