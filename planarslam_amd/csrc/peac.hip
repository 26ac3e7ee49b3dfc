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
//   peac_ahc3     one WAVEFRONT (64 lanes) per frame, the order-dependent clustering (peac_ahc2.h: lazy adjacency, tournament queue; frames with
//                 bit-equal keys are redone with libstdc++'s exact heap by the same workgroup)
//   peac_order    ranks the frames by the clustering time of the previous call (longest first) for the next launch
//   peac_refine   256 threads per frame: block erosion + seed queue (prefix sums) -> floodFill (512 queue entries x 4 neighbours per step;
//                 all pairs of a step that meet at one pixel are folded IN THE REFERENCE'S ORDER by one thread: round 5) -> final
//                 ahCluster over the surviving planes -> relabel; the node arrays stay in the global workspace            (a16-a17)
// Frame-level batch parallelism supplies the occupancy: the clustering is a chain of dependent FP64 operations, so a
// frame is latency-bound and the launch time is (frames / resident frames) x the slowest frame.  DESIGN.md §PEAC has the numbers.
#include "common.h"

#include "peac_common.h"

namespace planar {
namespace peac {

constexpr int NT_REFINE = 256;   // refinement: four wavefronts per frame (throughput: several frames per CU)
constexpr int NT_REFINE_WIDE = 1024;   // few frames in the batch (the reference's one-camera operating point): sixteen wavefronts per frame, a quarter of the flood-fill steps

struct Lds {
    float* h_key; u16* h_id; u16* pool; u16* nb_off; u16* nb_cnt; u16* dsp; u16* dss; u16* rid; unsigned* nouse; signed char* blk;
};

__device__ __forceinline__ int lds_find(u16* parent, int x) {   // DisjointSet::Find with path compression
    int r = x, guard = 0;
    while (parent[r] != r && ++guard < 8192) r = parent[r];           // bounded: a corrupted workspace must not hang the GPU
    guard = 0;
    while (parent[x] != r && ++guard < 8192) { const int nx = parent[x]; parent[x] = (u16)r; x = nx; }
    return r;
}
// peac_refine: refineDetails (block membership, erosion, seeds, flood fill), the final ahCluster over the <= 128 extracted planes and the relabelling.
// The clustering kernel (peac_ahc2.h) left the DisjointSet, the root ids, the dead bits and the extracted planes in the frame's workspace; the node-indexed
// arrays stay there (a handful of nodes are touched).  FJ = (entry, neighbour) pairs per thread and flood-fill step.
template <int NT, int FJ>
__device__ __forceinline__ void refine_frame(const Layout& L, const Intr& K, const Consts& C, const uint16_t* __restrict__ depth, int pitch_px,
                                             int64_t frame_stride_px, uint8_t* __restrict__ ws, int32_t* __restrict__ labels,
                                             int64_t label_stride, double* __restrict__ planes, int32_t* __restrict__ n_planes,
                                             int32_t* __restrict__ status, long long* __restrict__ timing, const int frame) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint8_t* F = ws + (size_t)frame * L.frame_bytes;
    double* g_stats = (double*)(F + L.off_stats);
    double* g_geo = (double*)(F + L.off_geo);
    int* g_N = (int*)(F + L.off_N);
    // membershipImg as signed bytes (plane ids 0..127, -1 = none, -2..-6 = the flood fill's rejection trail), MEANINGFUL ONLY INSIDE "BLACK" BLOCKS (blocks without
    // a plane after the erosion: the only pixels the flood fill reads or writes; a pixel of any other block carries its block's plane, s_blk); the distance map
    // WITHOUT its FLT_MAX fill (a pixel carries a distance exactly when the flood fill made it a member: read as FLT_MAX otherwise); queue entries packed
    // x | y << 12 | plane << 24 (no divisions in the flood fill): the reference's int32 image, float image and (index, plane) pairs cost 2.3x the bytes per frame
    signed char* member = (signed char*)(F + L.off_member);
    float* distMap = (float*)(F + L.off_dist);
    unsigned* queue = (unsigned*)(F + L.off_queue);
    auto qent = [&](int pixel, int plane) { const int y = pixel / L.W; return (unsigned)(pixel - y * L.W) | ((unsigned)y << 12) | ((unsigned)plane << 24); };
    int* seedcnt = (int*)(F + L.off_seedcnt);
    const uint16_t* D = depth + (size_t)frame * frame_stride_px;
    int32_t* lab = labels + (size_t)frame * label_stride;
    const int NB = L.NB, Nw = L.Nw, Nh = L.Nh, W = L.W, H = L.H;

    __shared__ float s_hm[MAX_PLANES];      // the heap of the final clustering holds <= MAX_PLANES nodes
    __shared__ u16 s_hi[MAX_PLANES];
    PLANAR_DYN_SMEM(s_dyn);
    signed char* s_blk = (signed char*)s_dyn;      // [NB] block -> plane id (dynamic LDS: the launch passes refine_smem_bytes(L))
    Lds S;
    S.h_key = s_hm; S.h_id = s_hi;
    S.pool = (u16*)(F + L.off_h_pool); S.nb_off = (u16*)(F + L.off_h_nboff); S.nb_cnt = (u16*)(F + L.off_h_nbcnt);
    S.dsp = (u16*)(F + L.off_h_dsp); S.dss = (u16*)(F + L.off_h_dss); S.rid = (u16*)(F + L.off_h_rid);
    S.nouse = (unsigned*)(F + L.off_h_nouse);
    S.blk = s_blk;
    int* g_hand = (int*)(F + L.off_h_hand);       // [0] n_ext, [1] err, [2] n_nodes, [4 + q] extracted node ids
    // every list of the final clustering keeps its capacity in the pool slot in front of its first entry
    auto noff = [&](int q) -> int { return (int)S.nb_off[q]; };
    auto set_noff = [&](int q, int v) { S.nb_off[q] = (u16)v; };
    auto ncnt = [&](int q) -> int { return (int)S.nb_cnt[q]; };
    auto set_ncnt = [&](int q, int v) { S.nb_cnt[q] = (u16)v; };
    auto list_cap = [&](int q) -> int { return (int)S.pool[noff(q) - 1]; };

    __shared__ int s_ext[MAX_PLANES], s_old[MAX_PLANES], s_plidmap[MAX_PLANES];
    __shared__ uint8_t s_valid[MAX_PLANES];
    __shared__ unsigned s_adj[MAX_PLANES][MAX_PLANES / 32];
    constexpr int NPAIR = FJ * NT;                            // (queue entry, neighbour) pairs of one flood-fill step
    constexpr int NSLOT = NPAIR / 4;                          // pixel -> chain of the step's work items that target it
    // the work items of a step (pairs whose target may change): target x | y << 12 | plane << 24 | geometric test << 31; point-plane distance; pair index; the
    // target's membership byte at the start of the step; next item of the slot's chain; then a bit per PAIR: pushed, with the words' exclusive prefix counts
    __shared__ unsigned s_word[NPAIR];
    __shared__ float s_cd[NPAIR];
    __shared__ unsigned short s_next[NPAIR], s_pair[NPAIR];
    __shared__ signed char s_m0[NPAIR];
    __shared__ unsigned s_head[NSLOT];
    __shared__ unsigned s_pushbits[NPAIR / 32];
    __shared__ unsigned short s_pushpre[NPAIR / 32];
    __shared__ unsigned s_ring[NPAIR / 4];                    // the head of the queue while the frontier is shorter than a step (no global round trip between steps)
    constexpr int NGEO = 16;
    __shared__ double s_geo[NGEO][7];                         // centre, normal, mse of the first extracted planes (the others are read from the workspace)
    __shared__ int s_scalar[4];   // [0] n_ext, [1] err, [2] q_tail, [3] scratch
    __shared__ u16 s_stageA[64], s_stageB[64];                // the two neighbour lists of a merge (read once from global memory)
    long long tphase[8];
    int nph = 0;
    auto mark = [&]() { if (nph < 8) tphase[nph++] = (long long)wall_clock64(); };
    mark();
    // Single-wave sections: the LDS traffic of one wavefront is processed in program order, so lanes only need the compiler to keep
    // that order (wavefront-scope fence, no s_waitcnt).
    auto wfence = [&]() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); };

    auto geo_of = [&](int id) { return g_geo + (size_t)id * 7; };
    auto nsim = [&](int a, int b) {
        const double* ga = geo_of(a) + 3; const double* gb = geo_of(b) + 3;
        return fabs(ga[0] * gb[0] + ga[1] * gb[1] + ga[2] * gb[2]);
    };

    // ---- init (all threads) ----
    for (int t = tid; t < MAX_PLANES; t += NT) { s_valid[t] = 0; s_plidmap[t] = -1; s_ext[t] = g_hand[4 + t]; }
    for (int t = tid; t < MAX_PLANES * (MAX_PLANES / 32); t += NT) (&s_adj[0][0])[t] = 0;
    if (tid < 4) s_scalar[tid] = tid == 1 ? g_hand[1] : 0;
    __syncthreads();
    mark();

    // =========================== the final clustering's queue and graph: wave 0 only ===========================
    int heap_n = 0, n_nodes = g_hand[2], n_ext = g_hand[0], pool_top = 4 * NB, err = g_hand[1];
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
    // ahCluster (:983-1189) over the surviving planes: one wavefront, one candidate neighbour per lane; lists with lazy deletion in the frame's pool
    auto ah_cluster = [&]() {
        int step = 0;
        while (heap_n > 0 && step <= MAX_STEP && !err) {
            const int p = heap_pop();
            if (is_dead(p)) continue;                           // nouse
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
            for (int k0 = 0; k0 < cnt; k0 += 64) {
                const int k = k0 + lane;
                bool ok = false;
                double ms[9]; Geo mg; int mN = 0, nb = -1;
                mg.mse = 0;
                if (k < cnt && !is_dead(lst[k])) {
                    nb = lst[k];
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
                    // ds.Union(pa.rid, pb.rid) (DisjointSet.hpp:64-84).  The rid of a live node is its set's root, so the two Find() calls return
                    // their arguments and compress nothing.  Union by size: size(root) * 100 is the N of the live node whose rid the root is, i.e. Np and Nn
                    const int xr = rp, yr = rn;
                    if (xr != yr) {
                        const u16 sz = (u16)((Np + Nn) / (WIN * WIN));
                        if (Np < Nn) { S.dsp[xr] = (u16)yr; S.dss[yr] = sz; }
                        else { S.dsp[yr] = (u16)xr; S.dss[xr] = sz; }
                    }
                }
                __threadfence_block();                          // the new node's record is read back (other lanes) when it is pushed / popped
                heap_push(m, best_geo.mse);
                // mergeNbsFrom (AHCPlaneSeg.hpp:379-404): union of the two sorted lists minus {p, nb} (both dead by now); <= 128 planes, so both lists fit the
                // wavefront: lane i owns A[i] and B[i], two interleaved lower-bound searches, ranks by popcount, every survivor written to its final slot
                // (the pool holds 16 lists' worth per block of the image: the <= 127 merges of <= 128 planes cannot fill it unless nearly all planes touch each other;
                //  then the frame reports a capacity error, as the round-2 kernel did)
                const int ca2 = ncnt(p), cb2 = ncnt(nb);
                if (pool_top + 2 * (ca2 + cb2) + 3 + (ca2 + cb2) / 4 + 8 > L.pool_cap) { err = 2; break; }
                const u16* A = S.pool + noff(p);
                const u16* Bl = S.pool + noff(nb);
                const int off = pool_top + 1;                   // pool_top itself becomes the capacity header
                int n;
                if (ca2 <= 64 && cb2 <= 64) {
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
                } else {                                        // a list of more than 64 planes: prefix counts in the pool, binary searches in global memory
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
                    __threadfence_block();
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
                    __threadfence_block();
                    n = na + nbk;
                    for (int k0 = 0; k0 < n; k0 += 64) {            // forward copy, destination below source: chunk-safe
                        const int k = k0 + lane;
                        const int v = k < n ? out[k] : 0;
                        __threadfence_block();
                        if (k < n) S.pool[off + k] = (u16)v;
                        __threadfence_block();
                    }
                }
                const int cap = n + max(8, n / 4);
                pool_top = off + cap;
                if (lane == 0) { S.pool[off - 1] = (u16)cap; set_noff(m, off); set_ncnt(m, n); set_ncnt(p, 0); set_ncnt(nb, 0); }
                __threadfence_block();
                {   // nb->nbs.insert(this): m is the largest id so far -> append; a full list is compacted first
                    const u16* lstm = S.pool + off;
                    for (int k = lane; k < n; k += 64) {
                        const int q = lstm[k];
                        u16* ql = S.pool + noff(q);
                        int c = ncnt(q);
                        if (c >= list_cap(q)) { int n2 = 0; for (int t = 0; t < c; t++) { const int v = ql[t]; if (!is_dead(v)) ql[n2++] = (u16)v; } c = n2; }
                        ql[c] = (u16)m; set_ncnt(q, c + 1);
                    }
                }
                __threadfence_block();
            } else {
                extract(p);
                mark_dead(p);
            }
            ++step;
        }
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

    // ---- refineDetails (:299-379): findBlockMembership (:485-587), all threads ----
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
    // membershipImg = -1 where the flood fill may look: the pixels of the black blocks (and what the block grid leaves over at the right / bottom edge), a row of
    // WIN bytes per (block, row); every other pixel's membership is its block's (s_blk) and is never stored
    for (int r = tid; r < NB * WIN; r += NT) {
        const int b = r / WIN, y = r - b * WIN, i = b / Nw, j = b - i * Nw;
        if (S.blk[b] < 0) { signed char* m = member + (size_t)(i * WIN + y) * W + j * WIN; for (int x = 0; x < WIN; x++) m[x] = -1; }
    }
    for (int y = tid; y < H; y += NT) for (int x = Nw * WIN; x < W; x++) member[(size_t)y * W + x] = -1;
    for (int y = Nh * WIN + tid / 64; y < H; y += NT / 64) for (int x = lane; x < Nw * WIN; x += 64) member[(size_t)y * W + x] = -1;
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

    // ---- floodFill (:428-476), all threads.  A step takes up to NENT = NPAIR / 4 queue entries x 4 neighbours = NPAIR (entry, neighbour) pairs; pair p = entry * 4
    //      + direction is the reference's processing order, and the result does not depend on the step size: pairs interact only through the pixel they target (its
    //      membership, distance and rejection trail), and the pushes are appended in pair order.  What bounds the kernel is the number of instructions a step
    //      issues (measured, round 5: ~220 steps per frame, each ~2 200 instructions per wavefront whatever it did), and three out of four pairs find their target
    //      already a member of their own plane (the reference's `continue` before any geometry).  So a step only carries on with the pairs that can change something:
    //      A1  a thread per queue ENTRY: its four neighbours, the block test, the target's membership byte m0; a pair whose target is neither a member of its own
    //          plane nor a dead trail becomes a WORK ITEM (dense list, ballot ranks)
    //      A2  per work item: depth -> point -> distance to the plane, in FP64 as the reference; the item is hung into the LDS chain of its pixel's slot
    //      B   the item with the SMALLEST pair index among those that target one pixel folds all of them, in ascending pair index = the order the reference would
    //          process them in, over the pixel's state in registers: ONE read and ONE write of the membership / distance per pixel and step, no replay rounds;
    //          plane-plane connect() is a commutative set insertion and goes to an LDS bit matrix.  The pairs left out in A1 (target already in the pair's own
    //          plane A) cannot change the pixel: if another plane took it earlier in the step, A's distance (the one it owned the pixel with) is not below the new
    //          one.  What they CAN still do is connect(A, current owner) - new information only when the pixel changed hands TWICE before them in one step
    //          (the first taker connected itself to A already): that case re-reads the step's entries and replays them for this pixel (extra_connects)
    //      C   the queue pushes in pair order: a bit per pair, prefix popcounts, one write per pushed item. ----
    {
        constexpr int NENT = NPAIR / 4, FE = NENT / NT;
        static_assert((NENT & (NENT - 1)) == 0 && NPAIR <= 0x7fff && FE >= 1 && FE * NT * 4 == NPAIR && NPAIR / 32 <= 64 * 4, "chain links are 15 bits; whole entries per thread; push bitmap scanned by one wavefront");
        constexpr unsigned END = 0x7fffu, NONE = 0xffffffffu;
        constexpr int PBW = NPAIR / 32;                               // words of the push bitmap
        for (int t = tid; t < NSLOT; t += NT) s_head[t] = END;
        const double factor = (double)K.factor;
        auto slot_of = [&](int x, int y) { return (x + y * 83) & (NSLOT - 1); };   // an odd row stride: neither a horizontal nor a vertical run of the frontier folds onto a few slots
        // point-plane distance of pixel (cx, cy) to extracted plane plid (float, as the reference narrows it) and the refinement test; -1 / false without depth
        for (int t = tid; t < NGEO * 7; t += NT) { const int q = t / 7; if (q < n_ext) s_geo[q][t - q * 7] = geo_of(s_ext[q])[t - q * 7]; }
        __syncthreads();
        auto eval_geo = [&](int cx, int cy, unsigned short dv, int plid, bool& geo_ok) -> float {
            float cdist = -1.f;
            geo_ok = false;
            const double z = (double)dv * factor;
            if (z != 0) {
                const double x = ((double)cx - (double)K.cx) * z / (double)K.fx;
                const double y = ((double)cy - (double)K.cy) * z / (double)K.fy;
                double g[7];
                if (plid < NGEO) { for (int c = 0; c < 7; c++) g[c] = s_geo[plid][c]; }
                else { const double* gg = geo_of(s_ext[plid]); for (int c = 0; c < 7; c++) g[c] = gg[c]; }
                const double sd = g[3] * (x - g[0]) + g[4] * (y - g[1]) + g[5] * (z - g[2]);
                cdist = (float)fabs(sd);
                geo_ok = (double)cdist * (double)cdist < 9 * g[6] + 1e-5;
            }
            return cdist;
        };
        // |n_a . n_b| of two extracted planes (PlaneSeg::normalSimilarity), normals from LDS for the first NGEO planes
        auto nsim_planes = [&](int a, int b2) {
            if (a < NGEO && b2 < NGEO) return fabs(s_geo[a][3] * s_geo[b2][3] + s_geo[a][4] * s_geo[b2][4] + s_geo[a][5] * s_geo[b2][5]);
            return nsim(s_ext[a], s_ext[b2]);
        };
        // queue[a], through the LDS ring where it holds it: entries [q_head, cached_hi) of the queue live in s_ring[a % NENT] and nowhere else
        int cached_hi = 0;
        auto entry_at = [&](int a) -> unsigned { return a < cached_hi ? s_ring[a & (NENT - 1)] : queue[a]; };
        int q_head = 0, q_tail = s_scalar[2], n_steps = 0;
#ifdef PLANAR_REFINE_PHASES
        long long ph[6] = {0, 0, 0, 0, 0, 0}, pt0 = 0;
#define PH_T0() pt0 = (long long)wall_clock64()
#define PH(i) do { const long long t_ = (long long)wall_clock64(); ph[i] += t_ - pt0; pt0 = t_; } while (0)
#else
#define PH_T0() do { } while (0)
#define PH(i) do { } while (0)
#endif
        while (q_head < q_tail && !err) {
            const int nent = min(NENT, q_tail - q_head);
            n_steps++;
            PH_T0();
            // ---- A1 ----
            if (tid == 0) s_scalar[3] = 0;
            if (tid < PBW) s_pushbits[tid] = 0u;
            __syncthreads();
#pragma unroll
            for (int jj = 0; jj < FE; jj++) {
                if (NT * jj >= nent) break;                            // (uniform: a short step costs what it holds)
                if (NT * jj + (tid & ~63) >= nent) continue;           // (this wavefront's 64 entries lie beyond the step's: no ballot below is shared across wavefronts)
                const int e = tid + NT * jj;
                const unsigned ent = e < nent ? entry_at(q_head + e) : NONE;
                const bool have = ent != NONE;
                const int sx = (int)(ent & 0xfffu), sy = (int)((ent >> 12) & 0xfffu), plid = (int)((ent >> 24) & 0x7fu);
                bool act[4];
                int m0[4];
                const int ebx = sx / WIN, eby = sy / WIN, erx = sx - ebx * WIN, ery = sy - eby * WIN;   // the entry's block: a neighbour is in it or in the next one over
#pragma unroll
                for (int dir = 0; dir < 4; dir++) {                    // getValid4Neighbor order (:398-410): left, right, up, down, invalid ones skipped
                    const int cx = sx + (dir == 0 ? -1 : (dir == 1 ? 1 : 0)), cy = sy + (dir == 2 ? -1 : (dir == 3 ? 1 : 0));
                    act[dir] = have && cx >= 0 && cx < W && cy >= 0 && cy < H;
                    if (act[dir]) {
                        const int bx = ebx + (dir == 0 ? -(int)(erx == 0) : (dir == 1 ? (int)(erx == WIN - 1) : 0));
                        const int by = eby + (dir == 2 ? -(int)(ery == 0) : (dir == 3 ? (int)(ery == WIN - 1) : 0));
                        if (by < Nh && bx < Nw && S.blk[by * Nw + bx] >= 0) act[dir] = false;      // only "black" blocks are refined
                    }
                    m0[dir] = act[dir] ? (int)member[(size_t)cy * W + cx] : -128;
                }
                unsigned long long nm[4];
                int total = 0;
#pragma unroll
                for (int dir = 0; dir < 4; dir++) {
                    // INVARIANT behind `m0 != plid` (the reference's `continue` before any geometry, AHCPlaneFitter.hpp:441-447): inside a black block (or outside the block
                    // grid) a membership >= 0 is only ever written by a fold of this flood fill, i.e. the pixel passed plane m0's geometric test when it became a member,
                    // and it was pushed then.  A later pair of the SAME plane can neither change the byte nor the distance (same plane, same point: same distance) nor push
                    // again; what it can still do is connect() its plane with a new owner when the pixel changes hands within the step - the case phase B replays
                    // (extra_connects, when a pixel takes two claims in one step).  The -DPLANAR_REFINE_PARANOID build (tests/test_peac_gpu.py) drops this filter's
                    // consequences: every pixel through the generic fold, every changed pixel replayed; its labels must equal the product's.
                    act[dir] = act[dir] && m0[dir] > -6 && m0[dir] != plid;       // a dead trail stays dead; the pair's own plane: nothing to do (see B)
                    nm[dir] = __ballot(act[dir]);
                    total += __popcll(nm[dir]);
                }
                if (total) {
                    int base = 0;
                    if (lane == 0) base = atomicAdd(&s_scalar[3], total);
                    base = __builtin_amdgcn_readfirstlane(base);
                    const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
                    for (int dir = 0; dir < 4; dir++) {
                        if (act[dir]) {
                            const int cx = sx + (dir == 0 ? -1 : (dir == 1 ? 1 : 0)), cy = sy + (dir == 2 ? -1 : (dir == 3 ? 1 : 0));
                            const int i = base + __popcll(nm[dir] & below);
                            s_word[i] = (unsigned)cx | ((unsigned)cy << 12) | ((unsigned)plid << 24);
                            s_pair[i] = (unsigned short)(4 * e + dir);
                            s_m0[i] = (signed char)m0[dir];
                        }
                        base += __popcll(nm[dir]);
                    }
                }
            }
            __syncthreads();
            PH(0);
            // ---- A2 ----
            const int nwork = s_scalar[3];
            for (int t = tid; t < nwork; t += NT) {
                const unsigned w = s_word[t];
                const int cx = (int)(w & 0xfffu), cy = (int)((w >> 12) & 0xfffu);
                bool gok;
                s_cd[t] = eval_geo(cx, cy, D[(size_t)cy * pitch_px + cx], (int)((w >> 24) & 0x7fu), gok);
                if (gok) s_word[t] = w | 0x80000000u;
                s_next[t] = (unsigned short)atomicExch(&s_head[slot_of(cx, cy)], (unsigned)t);
            }
            __syncthreads();
            PH(1);
            // ---- B ----
            for (int t = tid; t < nwork; t += NT) {
                const unsigned wt = s_word[t];
                const unsigned pixw = wt & 0xffffffu;
                const int px = (int)(wt & 0xfffu), py = (int)((wt >> 12) & 0xfffu);
                const int myp = (int)s_pair[t];
                const size_t pix = (size_t)py * W + px;
                const int trail0 = (int)s_m0[t];
                float dcur = 3.4028234663852886e38f;
                if (trail0 >= 0) dcur = distMap[pix];                      // (asked for before the chain is walked: the round trip overlaps the walk)
                const unsigned h0 = s_head[slot_of(px, py)];
                // ONE walk of the slot's chain: am I the pixel's first pair, and which are its items - up to four are kept in registers, sorted by pair index
                unsigned qi[4] = {END, END, END, END};
                int qp[4] = {0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff};
                int m = 0;
                bool leader = true;
                for (unsigned q = h0; q != END; q = s_next[q]) {
                    if ((s_word[q] & 0xffffffu) != pixw) continue;
                    int p_ = (int)s_pair[q];
                    unsigned i_ = q;
                    if (p_ < myp) { leader = false; break; }
#pragma unroll
                    for (int k = 0; k < 4; k++)
                        if (p_ < qp[k]) { const int tp = qp[k]; const unsigned ti = qi[k]; qp[k] = p_; qi[k] = i_; p_ = tp; i_ = ti; }
                    m++;
                }
                if (!leader) continue;
                // one pair of this pixel, in the reference's order: `trail` is the membership byte as it stands, dcur the distance that goes with a membership >= 0
                auto step_item = [&](unsigned it, int itp, bool commit, int& trail, float& dc, int& nclaims) {
                    const unsigned w = s_word[it];
                    const int plid = (int)((w >> 24) & 0x7fu);
                    if (!(trail <= -6) && !(trail >= 0 && trail == plid)) {
                        if (w >> 31) {
                            if (commit && trail >= 0 && nsim_planes(plid, trail) >= C.cos_refine) {   // n_pl.connect(pl)
                                atomicOr(&s_adj[trail][plid >> 5], 1u << (plid & 31));
                                atomicOr(&s_adj[plid][trail >> 5], 1u << (trail & 31));
                            }
                            const float cd = s_cd[it];
                            if (cd < (trail >= 0 ? dc : 3.4028234663852886e38f)) {
                                trail = plid; dc = cd; nclaims++;
                                if (commit) atomicOr(&s_pushbits[itp >> 5], 1u << (itp & 31));
                            } else if (trail < 0) trail -= 1;
                        } else if (trail < 0) trail -= 1;
                    }
                };
                // the fold over this pixel's items with pair index < limit, in ascending pair index; commit: with its side effects (connects, push bits)
                auto fold = [&](int limit, bool commit, int& trail, float& dc, int& nclaims) {
#ifdef PLANAR_REFINE_PARANOID               // test build (tests/test_peac_gpu.py::test_rare_paths...): the generic path for every pixel
                    if (false) {
#else
                    if (m <= 4) {
#endif
#pragma unroll
                        for (int k = 0; k < 4; k++) if (k < m && qp[k] < limit) step_item(qi[k], qp[k], commit, trail, dc, nclaims);
                        return;
                    }
                    int lastp = -1;                                        // more than four pairs at one pixel in one step: pick them off the chain one by one
                    while (true) {
                        unsigned best = END; int bestp = 0x7fffffff;
                        for (unsigned q = h0; q != END; q = s_next[q]) {
                            const int qp_ = (int)s_pair[q];
                            if ((s_word[q] & 0xffffffu) == pixw && qp_ > lastp && qp_ < bestp) { best = q; bestp = qp_; }
                        }
                        if (best == END || bestp >= limit) break;
                        lastp = bestp;
                        step_item(best, bestp, commit, trail, dc, nclaims);
                    }
                };
                int trail = trail0, nclaims = 0;
                const float dcur0 = dcur;
                fold(0x7fffffff, true, trail, dcur, nclaims);
                if (trail != trail0) member[pix] = (signed char)trail;
                if (nclaims) distMap[pix] = dcur;
#ifdef PLANAR_REFINE_PARANOID               // ... and the replay whenever the pixel changed hands at all (then it only repeats connects the claims made themselves)
                if (trail0 >= 0 && nclaims >= 1) {
#else
                if (trail0 >= 0 && nclaims >= 2) {
#endif
                    // extra_connects: the pixel was plane A's at the start of the step and changed hands at least twice within it.  A pair (A -> this pixel) that
                    // the reference processes after the second change connects A with the owner of that moment; such pairs were left out in A1: find them among
                    // the step's entries (an entry of plane A on a 4-neighbour of the pixel), replay the fold up to each (no side effects) and connect
                    for (int e = 0; e < nent; e++) {
                        const unsigned en = entry_at(q_head + e);
                        if ((int)((en >> 24) & 0x7fu) != trail0) continue;
                        const int ex = (int)(en & 0xfffu), ey = (int)((en >> 12) & 0xfffu);
                        int dir = -1;
                        if (ey == py && ex - 1 == px) dir = 0; else if (ey == py && ex + 1 == px) dir = 1; else if (ex == px && ey - 1 == py) dir = 2; else if (ex == px && ey + 1 == py) dir = 3;
                        if (dir < 0) continue;
                        int tr = trail0, nc = 0; float dc = dcur0;
                        fold(4 * e + dir, false, tr, dc, nc);
                        if (tr >= 0 && tr != trail0 && nsim_planes(trail0, tr) >= C.cos_refine) {
                            atomicOr(&s_adj[tr][trail0 >> 5], 1u << (trail0 & 31));
                            atomicOr(&s_adj[trail0][tr >> 5], 1u << (tr & 31));
                        }
                    }
                }
            }
            PH(2);
            __threadfence_block();
            __syncthreads();
            PH(3);
            // ---- C: pushes in pair order ----
            if (wave == 0) {                                           // exclusive prefix popcounts of the push bitmap's words (lane l: words l * PBW / 64 ..)
                constexpr int WPL = (PBW + 63) / 64;
                int mine = 0;
#pragma unroll
                for (int k = 0; k < WPL; k++) { const int wi = lane * WPL + k; if (wi < PBW) mine += __popc(s_pushbits[wi]); }
                int incl = mine;
                for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o); if (lane >= o) incl += v; }
                int run = incl - mine;
#pragma unroll
                for (int k = 0; k < WPL; k++) { const int wi = lane * WPL + k; if (wi < PBW) { s_pushpre[wi] = (unsigned short)run; run += __popc(s_pushbits[wi]); } }
                if (lane == 63) s_scalar[3] = incl;
            }
            __syncthreads();
            const int npush = s_scalar[3];
            const int q_next = q_head + nent;                            // the queue's head after this step
            if (cached_hi < q_next) cached_hi = q_next;                  // (the ring is empty)
            const bool ring_ok = cached_hi == q_tail;                    // the ring holds everything up to the tail: pushes may continue it
            if (q_tail + npush > L.q_cap) err = 5;
            else
                for (int t = tid; t < nwork; t += NT) {
                    const unsigned w = s_word[t];
                    const int p = (int)s_pair[t];
                    const unsigned bits = s_pushbits[p >> 5];
                    if ((bits >> (p & 31)) & 1u) {
                        const int a = q_tail + (int)s_pushpre[p >> 5] + __popc(bits & ((1u << (p & 31)) - 1u));
                        if (ring_ok && a - q_next < NENT) s_ring[a & (NENT - 1)] = w & 0x7fffffffu;   // = qent(x, y, plane)
                        else queue[a] = w & 0x7fffffffu;
                    }
                    s_head[slot_of((int)(w & 0xfffu), (int)((w >> 12) & 0xfffu))] = END;          // the chains are done with
                }
            if (ring_ok) cached_hi = min(q_tail + npush, q_next + NENT);
            q_tail += npush;
            q_head = q_next;
            __syncthreads();                                             // (global pushes are read at the earliest one step later: B's fence of that step is behind them)
            PH(4);
        }
#ifdef PLANAR_REFINE_PHASES
        if (tid == 0 && timing) for (int i = 0; i < 5; i++) timing[(size_t)frame * TSLOTS + 20 + i] = ph[i];
#endif
        if (tid == 0) { s_scalar[2] = q_tail; s_scalar[3] = n_steps; if (err) s_scalar[1] = err; }
    }
    __syncthreads();
    err = s_scalar[1];
    const int flood_steps = s_scalar[3];
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
        wfence();
        n_ext = 0;
        heap_n = 0;
        for (int q = 0; q < n_old; q++) if (s_valid[q]) heap_push(s_old[q], geo_of(s_old[q])[6]);
        if (!err) ah_cluster();
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
    // a pixel's membership: its block's plane, or - in a black block / outside the block grid - what the flood fill left in the membership image
    auto label_of = [&](int y, int x, int m) {
        const int by = y / WIN, bx = x / WIN;
        int plid = m;
        if (by < Nh && bx < Nw) { const int bp = S.blk[by * Nw + bx]; if (bp >= 0) plid = bp; }
        return (plid >= 0 && s_plidmap[plid] >= 0) ? s_plidmap[plid] : -1;
    };
    if ((W & 3) == 0 && (((size_t)lab | (size_t)member) & 15) == 0) {        // four pixels per thread: one 4-byte read, one 16-byte write
        const int W4 = W >> 2;
        for (int t = tid; t < W4 * H; t += NT) {
            const int y = t / W4, x0 = (t - y * W4) * 4;
            const unsigned m4 = *(const unsigned*)(member + (size_t)y * W + x0);
            int4 o;
            o.x = label_of(y, x0, (int)(signed char)(m4 & 0xffu)); o.y = label_of(y, x0 + 1, (int)(signed char)((m4 >> 8) & 0xffu));
            o.z = label_of(y, x0 + 2, (int)(signed char)((m4 >> 16) & 0xffu)); o.w = label_of(y, x0 + 3, (int)(signed char)(m4 >> 24));
            *(int4*)(lab + (size_t)y * W + x0) = o;
        }
    } else {
        for (int i = tid; i < W * H; i += NT) { const int y = i / W; lab[i] = label_of(y, i - y * W, (int)member[i]); }
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
            timing[(size_t)frame * TSLOTS + 7] = flood_steps;
        }
    }
}

// peac_refine: one workgroup per frame, ~37 KB of LDS: four per CU.
__global__ __launch_bounds__(NT_REFINE, 4) void peac_refine(Layout L, Intr K, Consts C, const uint16_t* __restrict__ depth, int pitch_px,
                                                  int64_t frame_stride_px, uint8_t* __restrict__ ws, int32_t* __restrict__ labels,
                                                  int64_t label_stride, double* __restrict__ planes, int32_t* __restrict__ n_planes,
                                                  int32_t* __restrict__ status, long long* __restrict__ timing) {
    refine_frame<NT_REFINE, 8>(L, K, C, depth, pitch_px, frame_stride_px, ws, labels, label_stride, planes, n_planes, status, timing, (int)blockIdx.x);
}

// The same refinement with 1024 threads per frame, for small batches: the flood fill takes 1024 queue entries per step instead of 512 (its result does not
// depend on the step size), one frame per CU.
__global__ __launch_bounds__(NT_REFINE_WIDE) void peac_refine_wide(Layout L, Intr K, Consts C, const uint16_t* __restrict__ depth, int pitch_px,
                                                  int64_t frame_stride_px, uint8_t* __restrict__ ws, int32_t* __restrict__ labels,
                                                  int64_t label_stride, double* __restrict__ planes, int32_t* __restrict__ n_planes,
                                                  int32_t* __restrict__ status, long long* __restrict__ timing) {
    refine_frame<NT_REFINE_WIDE, 4>(L, K, C, depth, pitch_px, frame_stride_px, ws, labels, label_stride, planes, n_planes, status, timing, (int)blockIdx.x);
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
    int smem2 = 0, smem_refine = 0;
    // kernel variants, set only through planar_peac_set_variant (tools' A/B runs and tests; no environment switch reaches the product path)
    int wide_below = 64;                                      // batches up to this size refine with 1024 threads per frame (0 = never)
    bool exact_only = false;                                  // skip the fast clustering attempt: every frame through the exact heap
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
    // Frame sizes: node ids are 16 bits, and the clustering wavefront keeps the queue keys (8 B per block), the merge-parent table (4 B per block) and three bitmaps in LDS:
    // up to ~10 900 blocks = 160 KB (1280x720 = 9 216 blocks: 114 KB, one frame per CU at a time; 640x480 = 3 072: 38 KB, four per CU)
    o->smem_refine = (int)((L.NB + 15) / 16 * 16);
    if (L.pool_cap > 65535 || L.NB2 > 65535 || o->smem2 > 160 * 1024 - 2048) {
        delete o;
        set_error("planar_peac_create: %dx%d = %d blocks of 10x10 pixels: the clustering kernel's LDS-resident queue holds about 10 900 (1280x720 fits, 1920x1080 does not)", width, height, L.NB);
        return PLANAR_EINVAL;
    }
    if (o->smem2 > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)peac::peac_ahc3, hipFuncAttributeMaxDynamicSharedMemorySize, o->smem2);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)peac::peac_ahc2, hipFuncAttributeMaxDynamicSharedMemorySize, o->smem2);
        if (e != hipSuccess) { (void)hipGetLastError(); delete o; set_error("planar_peac_create: %d bytes of LDS per workgroup are not available: %s", o->smem2, hipGetErrorString(e)); return PLANAR_EINVAL; }
    }
    int rc;
    if ((rc = o->d_ws.alloc((size_t)max_batch * L.frame_bytes)) || (rc = o->d_status.alloc((size_t)max_batch * 4)) ||
        (rc = o->d_timing.alloc((size_t)max_batch * peac::TSLOTS * 8)) || (rc = o->d_next.alloc(256)) || (rc = o->d_order.alloc((size_t)max_batch * 4))) { delete o; return rc; }
    *out = o;
    return PLANAR_OK;
}

void planar_peac_destroy(planar_peac* p) { delete p; }
// A/B aid (tools, tests): clustering variant 0 = product (fast attempt + exact redo), 1 = exact heap only; wide_below < 0 keeps the default.
// All variants produce the same labels and planes; nothing in the product calls this.
int planar_peac_set_variant(planar_peac* p, int clustering, int wide_below) {
    PLANAR_REQUIRE(p && clustering >= 0 && clustering <= 1, PLANAR_EINVAL, "bad argument");
    p->exact_only = clustering == 1;
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
    {
        // the fast attempt (tournament queue), then the exact kernel for the frames it gave up on (bit-equal keys of two live nodes: degenerate input)
        // (the workgroup that gave up redoes its frame itself: no second launch)
        // (on the context's side stream when it has one: planar_ctx_set_seq_stream)
        hipStream_t sq = p->ctx->seq_begin();
        if (!p->exact_only)
            hipLaunchKernelGGL(peac::peac_ahc3, dim3(B), dim3(64), p->smem2, sq, p->L, p->C, p->d_ws.as<uint8_t>(), p->d_status.as<int32_t>(),
                               p->d_timing.as<long long>(), p->d_next.as<int>(), p->order_B == B ? p->d_order.as<int>() : nullptr, 1);
        else
            hipLaunchKernelGGL(peac::peac_ahc2, dim3(B), dim3(64), p->smem2, sq, p->L, p->C, p->d_ws.as<uint8_t>(), p->d_status.as<int32_t>(),
                               p->d_timing.as<long long>(), p->d_next.as<int>(), p->order_B == B ? p->d_order.as<int>() : nullptr, 0);
        p->ctx->seq_end();
    }
    mark();
    hipLaunchKernelGGL(peac::peac_order, dim3((B + 255) / 256), dim3(256), 0, st, p->d_timing.as<long long>(), B, p->d_order.as<int>());
    mark();
    if (B <= p->wide_below)
        hipLaunchKernelGGL(peac::peac_refine_wide, dim3(B), dim3(peac::NT_REFINE_WIDE), p->smem_refine, st, p->L, K, p->C, d_depth, pitch_px, frame_stride_px,
                           p->d_ws.as<uint8_t>(), d_labels, (int64_t)p->W * p->H, d_planes, d_n_planes, p->d_status.as<int32_t>(), p->d_timing.as<long long>());
    else
        hipLaunchKernelGGL(peac::peac_refine, dim3(B), dim3(peac::NT_REFINE), p->smem_refine, st, p->L, K, p->C, d_depth, pitch_px, frame_stride_px,
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
