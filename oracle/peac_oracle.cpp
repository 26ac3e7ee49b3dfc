// oracle/peac_oracle.cpp — TEST INFRASTRUCTURE (CPU oracle), not product code.
//
// CPU restatement of the reference's plane extractor (PEAC / AHC, Feng et al.) as PlanarSLAM configures it:
//   PlaneDetection::readDepthImage / runPlaneDetection     reference src/PlaneExtractor.cpp:26-65
//   ImagePointCloud::get                                   reference include/PlaneExtractor.h:25-33
//   PlaneSeg (block stats, PCA, merge)                     reference include/peac/AHCPlaneSeg.hpp
//   PlaneFitter::run/initGraph/ahCluster/refineDetails/findBlockMembership/floodFill
//                                                          reference include/peac/AHCPlaneFitter.hpp
//   DisjointSet                                            reference include/peac/DisjointSet.hpp
//   ParamSet thresholds                                    reference include/peac/AHCParamSet.hpp
//
// Pinning: checked label-for-label against the REAL reference sources compiled against oracle/shim
// (oracle/_ref/ref_peac, tests/test_oracle_peac.py); only Eigen's 3x3 eigen-solver is a restatement there too.
//
// Chosen refinements of behaviour the reference leaves to the platform:
//  (1) PlaneSeg::NbSet is std::set<PlaneSeg*>: neighbours are visited in heap-address order.  We use creation
//      order (== address order under the monotonic allocator oracle/_ref is built with).
//  (2) std::priority_queue / std::sort tie behaviour: the libstdc++ binary-heap sift rules are restated
//      (push_heap/__adjust_heap); the final size sort is an insertion sort (what libstdc++'s std::sort does
//      for <= 16 elements; more than 16 extracted planes with equal sizes may order differently).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <map>
#include <memory>
#include <vector>

#include "eigprim.h"

namespace orc {
namespace {

struct Params {   // AHCParamSet.hpp:55-66 + PlaneFitter defaults AHCPlaneFitter.hpp:154-158
    double depthSigma = 1.6e-6, stdTol_init = 5, stdTol_merge = 8, z_near = 500, z_far = 4000;
    double angle_near = 15.0 * M_PI / 180.0, angle_far = 90.0 * M_PI / 180.0;
    double similarityTh_merge = std::cos(60.0 * M_PI / 180.0), similarityTh_refine = std::cos(30.0 * M_PI / 180.0);
    double depthAlpha = 0.04, depthChangeTol = 0.02;
    int maxStep = 100000, minSupport = 3000, windowWidth = 10, windowHeight = 10;
    enum Phase { P_INIT, P_MERGING, P_REFINE };
    double T_mse(Phase ph, double z) const {   // :87-99
        return ph == P_INIT ? std::pow(depthSigma * z * z + stdTol_init, 2) : std::pow(depthSigma * z * z + stdTol_merge, 2);
    }
    double T_ang(Phase ph, double z) const {   // :112-134
        if (ph == P_INIT) {
            double cz = std::max(z, z_near);
            cz = std::min(cz, z_far);
            const double factor = (angle_far - angle_near) / (z_far - z_near);
            return std::cos(factor * cz + angle_near - factor * z_near);
        }
        return ph == P_MERGING ? similarityTh_merge : similarityTh_refine;
    }
    double T_dz(double z) const { return depthAlpha * std::fabs(z) + depthChangeTol; }   // :144-146
};

struct Cloud {   // ImagePointCloud
    int w, h;
    std::vector<double> xyz;
    bool get(int row, int col, double& x, double& y, double& z) const {
        const size_t i = ((size_t)row * w + col) * 3;
        z = xyz[i + 2];
        if (z == 0 || std::isnan(z)) return false;
        x = xyz[i]; y = xyz[i + 1];
        return true;
    }
};

struct Stats {
    double sx = 0, sy = 0, sz = 0, sxx = 0, syy = 0, szz = 0, sxy = 0, syz = 0, sxz = 0;
    int N = 0;
    void push(double x, double y, double z) {
        sx += x; sy += y; sz += z; sxx += x * x; syy += y * y; szz += z * z; sxy += x * y; syz += y * z; sxz += x * z; ++N;
    }
    static Stats sum(const Stats& a, const Stats& b) {
        Stats s;
        s.sx = a.sx + b.sx; s.sy = a.sy + b.sy; s.sz = a.sz + b.sz; s.sxx = a.sxx + b.sxx; s.syy = a.syy + b.syy; s.szz = a.szz + b.szz;
        s.sxy = a.sxy + b.sxy; s.syz = a.syz + b.syz; s.sxz = a.sxz + b.sxz; s.N = a.N + b.N;
        return s;
    }
    void compute(double center[3], double normal[3], double& mse, double& curvature) const {   // AHCPlaneSeg.hpp:125-156
        const double sc = 1.0 / N;
        center[0] = sx * sc; center[1] = sy * sc; center[2] = sz * sc;
        double K[3][3] = {{sxx - sx * sx * sc, sxy - sx * sy * sc, sxz - sx * sz * sc}, {0, syy - sy * sy * sc, syz - sy * sz * sc}, {0, 0, szz - sz * sz * sc}};
        K[1][0] = K[0][1]; K[2][0] = K[0][2]; K[2][1] = K[1][2];
        double sv[3], V[3][3];
        eig33_selfadjoint(K, sv, V);
        if (V[0][0] * center[0] + V[1][0] * center[1] + V[2][0] * center[2] <= 0) { normal[0] = V[0][0]; normal[1] = V[1][0]; normal[2] = V[2][0]; }
        else { normal[0] = -V[0][0]; normal[1] = -V[1][0]; normal[2] = -V[2][0]; }
        mse = sv[0] * sc;
        curvature = sv[0] / (sv[0] + sv[1] + sv[2]);
    }
};

struct Seg {
    Stats stats;
    int rid = 0, N = 0;
    double mse = 0, center[3] = {0, 0, 0}, normal[3] = {0, 0, 0}, curvature = 0;
    bool nouse = false;
    std::vector<int> nbs;   // sorted ids (creation order)
    double normalSimilarity(const Seg& p) const { return std::abs(normal[0] * p.normal[0] + normal[1] * p.normal[1] + normal[2] * p.normal[2]); }
    double signedDist(const double pt[3]) const {
        return normal[0] * (pt[0] - center[0]) + normal[1] * (pt[1] - center[1]) + normal[2] * (pt[2] - center[2]);
    }
};

struct DisjointSet {   // DisjointSet.hpp
    std::vector<int> parent, size;
    explicit DisjointSet(int n) : parent(n), size(n, 1) { for (int i = 0; i < n; i++) parent[i] = i; }
    int Find(int x) { if (parent[x] != x) parent[x] = Find(parent[x]); return parent[x]; }
    int getSetSize(int x) { return size[Find(x)]; }
    int Union(int x, int y) {
        const int xr = Find(x), yr = Find(y);
        if (xr == yr) return xr;
        if (size[xr] < size[yr]) { parent[xr] = yr; size[yr] += size[xr]; return yr; }
        parent[yr] = xr; size[xr] += size[yr]; return xr;
    }
};

struct Fitter {
    Params prm;
    const Cloud* pts = nullptr;
    int width = 0, height = 0;
    std::vector<Seg> segs;          // all PlaneSeg objects that ever enter the graph, id = creation order
    std::vector<int> extracted;     // extractedPlanes (seg ids)
    std::unique_ptr<DisjointSet> ds;
    std::vector<int> membership;    // membershipImg
    std::vector<int> blkMap;
    std::vector<std::pair<int, int>> rfQueue;

    // --- libstdc++ heap with PlaneSegMinMSECmp: comp(a,b) = b.mse < a.mse ---
    std::vector<int> heap;
    bool cmp(int a, int b) const { return segs[b].mse < segs[a].mse; }
    void heap_push(int v) {
        heap.push_back(v);
        int hole = (int)heap.size() - 1, parent = (hole - 1) / 2;
        while (hole > 0 && cmp(heap[parent], v)) { heap[hole] = heap[parent]; hole = parent; parent = (hole - 1) / 2; }
        heap[hole] = v;
    }
    int heap_pop() {
        const int top = heap[0];
        const int value = heap.back();
        heap.pop_back();
        const int len = (int)heap.size();
        if (len == 0) return top;
        int hole = 0, second = 0;
        while (second < (len - 1) / 2) {
            second = 2 * (second + 1);
            if (cmp(heap[second], heap[second - 1])) second--;
            heap[hole] = heap[second]; hole = second;
        }
        if ((len & 1) == 0 && second == (len - 2) / 2) { second = 2 * (second + 1); heap[hole] = heap[second - 1]; hole = second - 1; }
        int parent = (hole - 1) / 2;
        while (hole > 0 && cmp(heap[parent], value)) { heap[hole] = heap[parent]; hole = parent; parent = (hole - 1) / 2; }
        heap[hole] = value;
        return top;
    }

    static void set_insert(std::vector<int>& v, int x) { auto it = std::lower_bound(v.begin(), v.end(), x); if (it == v.end() || *it != x) v.insert(it, x); }
    static void set_erase(std::vector<int>& v, int x) { auto it = std::lower_bound(v.begin(), v.end(), x); if (it != v.end() && *it == x) v.erase(it); }
    void connect(int a, int b) { set_insert(segs[a].nbs, b); set_insert(segs[b].nbs, a); }
    void disconnectAllNbs(int a) { for (int nb : segs[a].nbs) set_erase(segs[nb].nbs, a); segs[a].nbs.clear(); }

    // PlaneSeg window constructor, AHCPlaneSeg.hpp:211-285
    Seg make_block(int rid, int seed_row, int seed_col) const {
        Seg s;
        s.rid = rid;
        bool valid = true;
        for (int i = seed_row, ic = 0; ic < prm.windowHeight && i < height; ++i, ++ic) {
            for (int j = seed_col, jc = 0; jc < prm.windowWidth && j < width; ++j, ++jc) {
                double x = 0, y = 0, z = 10000;
                if (!pts->get(i, j, x, y, z)) { valid = false; break; }   // INIT_STRICT
                double xn = 0, yn = 0, zn = 10000;
                if (j + 1 < width && (pts->get(i, j + 1, xn, yn, zn) && std::fabs(z - zn) > prm.T_dz(z))) { valid = false; break; }
                if (i + 1 < height && (pts->get(i + 1, j, xn, yn, zn) && std::fabs(z - zn) > prm.T_dz(z))) { valid = false; break; }
                s.stats.push(x, y, z);
            }
            if (!valid) break;
        }
        if (valid) { s.nouse = false; s.N = s.stats.N; }
        else { s.N = 0; s.stats = Stats(); s.nouse = true; }
        if (s.N < 4) s.mse = s.curvature = std::numeric_limits<double>::quiet_NaN();
        else s.stats.compute(s.center, s.normal, s.mse, s.curvature);
        return s;
    }
    Seg make_merge(const Seg& pa, const Seg& pb) const {   // :301-315
        Seg s;
        s.stats = Stats::sum(pa.stats, pb.stats);
        s.nouse = false;
        s.rid = pa.N >= pb.N ? pa.rid : pb.rid;
        s.N = s.stats.N;
        s.stats.compute(s.center, s.normal, s.mse, s.curvature);
        return s;
    }

    void initGraph() {   // AHCPlaneFitter.hpp:786-972
        const int Nh = height / prm.windowHeight, Nw = width / prm.windowWidth;
        std::vector<int> G(Nh * Nw, -1);
        segs.reserve(2 * Nh * Nw);
        for (int i = 0; i < Nh; i++)
            for (int j = 0; j < Nw; j++) {
                segs.push_back(make_block(i * Nw + j, i * prm.windowHeight, j * prm.windowWidth));
                const Seg& p = segs.back();
                if (p.mse < prm.T_mse(Params::P_INIT, p.center[2]) && !p.nouse) { G[i * Nw + j] = (int)segs.size() - 1; heap_push((int)segs.size() - 1); }
            }
        for (int i = 0; i < Nh; ++i) {   // :896-925 horizontal triples
            for (int j = 1; j < Nw; j += 2) {
                const int c = i * Nw + j;
                if (G[c - 1] < 0) { --j; continue; }
                if (G[c] < 0) continue;
                if (j < Nw - 1 && G[c + 1] < 0) { ++j; continue; }
                const double th = prm.T_ang(Params::P_INIT, segs[G[c]].center[2]);
                if ((j < Nw - 1 && segs[G[c - 1]].normalSimilarity(segs[G[c + 1]]) >= th) ||
                    (j == Nw - 1 && segs[G[c]].normalSimilarity(segs[G[c - 1]]) >= th)) {
                    connect(G[c], G[c - 1]);
                    if (j < Nw - 1) connect(G[c], G[c + 1]);
                } else --j;
            }
        }
        for (int j = 0; j < Nw; ++j) {   // :927-954 vertical triples
            for (int i = 1; i < Nh; i += 2) {
                const int c = i * Nw + j;
                if (G[c - Nw] < 0) { --i; continue; }
                if (G[c] < 0) continue;
                if (i < Nh - 1 && G[c + Nw] < 0) { ++i; continue; }
                const double th = prm.T_ang(Params::P_INIT, segs[G[c]].center[2]);
                if ((i < Nh - 1 && segs[G[c - Nw]].normalSimilarity(segs[G[c + Nw]]) >= th) ||
                    (i == Nh - 1 && segs[G[c]].normalSimilarity(segs[G[c - Nw]]) >= th)) {
                    connect(G[c], G[c - Nw]);
                    if (i < Nh - 1) connect(G[c], G[c + Nw]);
                } else --i;
            }
        }
    }

    int ahCluster() {   // :983-1189
        int step = 0;
        while (!heap.empty() && step <= prm.maxStep) {
            const int p = heap_pop();
            if (segs[p].nouse) continue;
            bool have = false;
            Seg cand;
            int cand_nb = -1;
            for (int nb : segs[p].nbs) {
                if (segs[p].normalSimilarity(segs[nb]) < prm.T_ang(Params::P_MERGING, segs[p].center[2])) continue;
                Seg merge = make_merge(segs[p], segs[nb]);
                if (!have || cand.mse > merge.mse || (cand.mse == merge.mse && cand.N < merge.mse)) { cand = std::move(merge); cand_nb = nb; have = true; }   // quirk :1045
            }
            if (have && cand.mse < prm.T_mse(Params::P_MERGING, cand.center[2])) {
                segs.push_back(std::move(cand));
                const int m = (int)segs.size() - 1;
                heap_push(m);
                // mergeNbsFrom (AHCPlaneSeg.hpp:379-404)
                ds->Union(segs[p].rid, segs[cand_nb].rid);
                std::vector<int> u;
                std::set_union(segs[p].nbs.begin(), segs[p].nbs.end(), segs[cand_nb].nbs.begin(), segs[cand_nb].nbs.end(), std::back_inserter(u));
                set_erase(u, p); set_erase(u, cand_nb);
                disconnectAllNbs(p); disconnectAllNbs(cand_nb);
                segs[m].nbs = u;
                for (int nb : u) set_insert(segs[nb].nbs, m);
                segs[p].nouse = segs[cand_nb].nouse = true;
            } else {
                if (segs[p].N >= prm.minSupport) extracted.push_back(p);
                disconnectAllNbs(p);
            }
            ++step;
        }
        while (!heap.empty()) {
            const int p = heap_pop();
            if (segs[p].N >= prm.minSupport) extracted.push_back(p);
            disconnectAllNbs(p);
        }
        // std::sort(extractedPlanes, sizecmp): insertion sort (refinement 2)
        for (size_t i = 1; i < extracted.size(); i++) {
            const int v = extracted[i];
            size_t j = i;
            while (j > 0 && segs[extracted[j - 1]].N < segs[v].N) { extracted[j] = extracted[j - 1]; j--; }
            extracted[j] = v;
        }
        return step;
    }

    static int valid4(int i, int j, int H, int W, int nbs[4]) {   // :398-410
        const int id = i * W + j;
        int cnt = 0;
        if (j > 0) nbs[cnt++] = id - 1;
        if (j < W - 1) nbs[cnt++] = id + 1;
        if (i > 0) nbs[cnt++] = id - W;
        if (i < H - 1) nbs[cnt++] = id + W;
        return cnt;
    }
    int blockIdx(int px, int py) const {   // :418-426
        const int Nw = width / prm.windowWidth, Nh = height / prm.windowHeight;
        const int by = py / prm.windowHeight, bx = px / prm.windowWidth;
        return (by < Nh && bx < Nw) ? (by * Nw + bx) : -1;
    }

    void findBlockMembership(std::vector<char>& isValid) {   // :485-587 (ERODE_ALL_BORDER)
        std::map<int, int> rid2plid;
        for (int pl = 0; pl < (int)extracted.size(); ++pl) rid2plid.insert({segs[extracted[pl]].rid, pl});
        const int Nh = height / prm.windowHeight, Nw = width / prm.windowWidth, NptsPerBlk = prm.windowHeight * prm.windowWidth;
        const int wh = prm.windowHeight, ww = prm.windowWidth;
        membership.assign((size_t)width * height, -1);
        blkMap.assign(Nh * Nw, 0);
        isValid.assign(extracted.size(), 0);
        for (int i = 0, blkid = 0; i < Nh; ++i) {
            for (int j = 0; j < Nw; ++j, ++blkid) {
                const int setid = ds->Find(blkid);
                const int setSize = ds->getSetSize(setid) * NptsPerBlk;
                if (setSize >= prm.minSupport) {
                    int nbs[4] = {-1};
                    const int nN = valid4(i, j, Nh, Nw, nbs);
                    bool same = true;
                    for (int k = 0; k < nN; ++k) if (ds->Find(nbs[k]) != setid) { same = false; break; }
                    const int plid = rid2plid[setid];   // operator[]: inserts 0 when absent, as the reference does
                    if (same) {
                        blkMap[blkid] = plid;
                        for (int y = i * wh; y < (i + 1) * wh; y++) for (int x = j * ww; x < (j + 1) * ww; x++) membership[(size_t)y * width + x] = plid;
                        isValid[plid] = 1;
                    } else blkMap[blkid] = -1;
                } else blkMap[blkid] = -1;
                if (blkMap[blkid] < 0) {
                    if (i > 0) { const int u = blkid - Nw; if (blkMap[u] >= 0) { const int up = blkMap[u]; const int sp = (i * wh - 1) * width + j * ww; for (int k = 1; k < ww; ++k) rfQueue.push_back({sp + k, up}); } }
                    if (j > 0) { const int l = blkid - 1; if (blkMap[l] >= 0) { const int lp = blkMap[l]; const int sp = (i * wh) * width + j * ww - 1; for (int k = 0; k < wh - 1; ++k) rfQueue.push_back({sp + k * width, lp}); } }
                } else {
                    const int plid = blkMap[blkid];
                    if (i > 0) { const int u = blkid - Nw; if (blkMap[u] != plid) { const int sp = (i * wh) * width + j * ww; for (int k = 0; k < ww - 1; ++k) rfQueue.push_back({sp + k, plid}); } }
                    if (j > 0) { const int l = blkid - 1; if (blkMap[l] != plid) { const int sp = (i * wh) * width + j * ww; for (int k = 1; k < wh; ++k) rfQueue.push_back({sp + k * width, plid}); } }
                }
            }
        }
    }

    void floodFill() {   // :428-476
        std::vector<float> distMap((size_t)height * width, std::numeric_limits<float>::max());
        for (int k = 0; k < (int)rfQueue.size(); ++k) {
            const int sIdx = rfQueue[k].first, seedy = sIdx / width, seedx = sIdx - seedy * width, plid = rfQueue[k].second;
            const Seg& pl = segs[extracted[plid]];
            int nbs[4] = {-1};
            const int Nn = valid4(seedy, seedx, height, width, nbs);
            for (int it = 0; it < Nn; ++it) {
                const int cIdx = nbs[it];
                int& trail = membership[cIdx];
                if (trail <= -6) continue;
                if (trail >= 0 && trail == plid) continue;
                const int cy = cIdx / width, cx = cIdx - cy * width;
                const int blkid = blockIdx(cx, cy);
                if (blkid >= 0 && blkMap[blkid] >= 0) continue;
                double pt[3] = {0};
                float cdist = -1;
                if (pts->get(cy, cx, pt[0], pt[1], pt[2]) && std::pow(cdist = (float)std::abs(pl.signedDist(pt)), 2) < 9 * pl.mse + 1e-5) {
                    if (trail >= 0) {
                        Seg& n_pl = segs[extracted[trail]];
                        if (pl.normalSimilarity(n_pl) >= prm.T_ang(Params::P_REFINE, pl.center[2])) connect(extracted[trail], extracted[plid]);
                    }
                    float& old = distMap[cIdx];
                    if (cdist < old) { trail = plid; old = cdist; rfQueue.push_back({cIdx, plid}); }
                    else if (trail < 0) trail -= 1;
                } else if (trail < 0) trail -= 1;
            }
        }
    }

    // returns final labels (-1 = none) ; refineDetails :299-379
    void refineDetails(std::vector<int>& labels) {
        std::vector<char> isValid;
        findBlockMembership(isValid);
        floodFill();
        std::vector<int> old;
        old.swap(extracted);
        heap.clear();
        for (int i = 0; i < (int)old.size(); ++i) if (isValid[i]) heap_push(old[i]);
        ahCluster();
        std::vector<int> plidmap(old.size(), -1);
        for (int i = 0; i < (int)old.size(); ++i) {
            if (!isValid[i]) continue;
            const int np_rid = ds->Find(segs[old[i]].rid);
            for (size_t j = 0; j < extracted.size(); ++j) if (np_rid == segs[extracted[j]].rid) { plidmap[i] = (int)j; break; }
        }
        labels.assign((size_t)width * height, -1);
        for (size_t i = 0; i < labels.size(); i++) {
            const int plid = membership[i];
            if (plid >= 0 && plidmap[plid] >= 0) labels[i] = plidmap[plid];
        }
    }
};

}  // namespace
}  // namespace orc

extern "C" {
// planes: [max_planes][8] = N, normal[3], center[3], mse.  blocks (optional, for stage tests): per 10x10 block
// [Nh*Nw][6] = in_graph flag, N, mse, normal[3].  Returns the number of planes (may exceed max_planes: truncated).
int orc_peac_run(const uint16_t* depth, int W, int H, float fx, float fy, float cx, float cy, float factor, int32_t* labels,
                 double* planes, int max_planes, double* blocks) {
    using namespace orc;
    Cloud cloud;
    cloud.w = W; cloud.h = H; cloud.xyz.resize((size_t)W * H * 3);
    for (int i = 0; i < H; i++)        // PlaneDetection::readDepthImage (src/PlaneExtractor.cpp:40-55)
        for (int j = 0; j < W; j++) {
            const double z = (double)depth[(size_t)i * W + j] * factor;
            const double x = ((double)j - cx) * z / fx;
            const double y = ((double)i - cy) * z / fy;
            double* p = &cloud.xyz[((size_t)i * W + j) * 3];
            p[0] = x; p[1] = y; p[2] = z;
        }
    Fitter f;
    f.pts = &cloud; f.width = W; f.height = H;
    const int Nh = H / f.prm.windowHeight, Nw = W / f.prm.windowWidth;
    f.ds.reset(new DisjointSet(Nh * Nw));
    f.initGraph();
    if (blocks) {
        for (int b = 0; b < Nh * Nw; b++) {
            const Seg& s = f.segs[b];
            const bool in = s.mse < f.prm.T_mse(Params::P_INIT, s.center[2]) && !s.nouse;
            double* o = blocks + (size_t)b * 6;
            o[0] = in; o[1] = s.N; o[2] = in ? s.mse : 0; o[3] = in ? s.normal[0] : 0; o[4] = in ? s.normal[1] : 0; o[5] = in ? s.normal[2] : 0;
        }
    }
    f.ahCluster();
    std::vector<int> lab;
    f.refineDetails(lab);
    std::memcpy(labels, lab.data(), lab.size() * sizeof(int32_t));
    const int n = (int)f.extracted.size();
    for (int i = 0; i < n && i < max_planes; i++) {
        const Seg& s = f.segs[f.extracted[i]];
        double* o = planes + (size_t)i * 8;
        o[0] = s.N; o[1] = s.normal[0]; o[2] = s.normal[1]; o[3] = s.normal[2]; o[4] = s.center[0]; o[5] = s.center[1]; o[6] = s.center[2]; o[7] = s.mse;
    }
    return n;
}

// State of the first ahCluster (AHCPlaneFitter.hpp:983-1189) for kernel-level checks (tests/test_peac_emul.py, tools/peac_ab.py): every PlaneSeg ever
// created, in creation order.  nodes: [cap][18] = N, rid, mse, center[3], normal[3], stats[9]; extracted: node ids in extractedPlanes order (after the
// size sort); set_root / set_size: DisjointSet::Find / getSetSize of every block.  Returns the number of nodes; *n_extracted the number of planes.
int orc_peac_cluster_state(const uint16_t* depth, int W, int H, float fx, float fy, float cx, float cy, float factor, double* nodes, int cap,
                           int32_t* extracted, int32_t* n_extracted, int32_t* set_root, int32_t* set_size) {
    using namespace orc;
    Cloud cloud;
    cloud.w = W; cloud.h = H; cloud.xyz.resize((size_t)W * H * 3);
    for (int i = 0; i < H; i++)
        for (int j = 0; j < W; j++) {
            const double z = (double)depth[(size_t)i * W + j] * factor;
            double* p = &cloud.xyz[((size_t)i * W + j) * 3];
            p[0] = ((double)j - cx) * z / fx; p[1] = ((double)i - cy) * z / fy; p[2] = z;
        }
    Fitter f;
    f.pts = &cloud; f.width = W; f.height = H;
    const int Nh = H / f.prm.windowHeight, Nw = W / f.prm.windowWidth;
    f.ds.reset(new DisjointSet(Nh * Nw));
    f.initGraph();
    f.ahCluster();
    const int n = (int)f.segs.size();
    for (int i = 0; i < n && i < cap; i++) {
        const Seg& g = f.segs[i];
        double* o = nodes + (size_t)i * 18;
        o[0] = g.N; o[1] = g.rid; o[2] = g.mse;
        for (int k = 0; k < 3; k++) { o[3 + k] = g.center[k]; o[6 + k] = g.normal[k]; }
        const Stats& t = g.stats;
        const double st[9] = {t.sx, t.sy, t.sz, t.sxx, t.syy, t.szz, t.sxy, t.syz, t.sxz};
        for (int k = 0; k < 9; k++) o[9 + k] = st[k];
    }
    *n_extracted = (int)f.extracted.size();
    for (size_t i = 0; i < f.extracted.size(); i++) extracted[i] = f.extracted[i];
    for (int b = 0; b < Nh * Nw; b++) { set_root[b] = f.ds->Find(b); set_size[b] = f.ds->getSetSize(b); }
    return n;
}
}
