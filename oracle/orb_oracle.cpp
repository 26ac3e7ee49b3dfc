// oracle/orb_oracle.cpp — TEST INFRASTRUCTURE (CPU oracle), not product code.
//
// CPU restatement of PlanarSLAM's ORB extractor.  Every function cites the
// reference lines it follows (paths relative to the PlanarSLAM tree).  The
// OpenCV calls are replaced by the restatements in cvprim.cpp.
//
// Pinning status: the in-tree arithmetic (cell grid, FAST threshold fallback,
// octree distribution, IC_Angle, steered BRIEF, level concatenation) is checked
// against the *real* reference translation unit compiled against the shim
// headers (oracle/_ref, see oracle/Makefile and tests/test_oracle_vs_ref.py).
// The OpenCV primitives themselves are unpinned (no OpenCV in the container).
//
// Chosen refinements of under-specified reference behaviour:
//  (1) octree expansion order among equal-sized nodes: the reference sorts
//      pair<int,ExtractorNode*> (src/ORBextractor.cc:684), i.e. ties are broken
//      by heap address.  We break ties by node creation order, which equals the
//      address order under a monotonic (bump) allocator — that is how oracle/_ref
//      is built, so the two agree exactly.
//  (2) cos/sin of the keypoint angle (src/ORBextractor.cc:113): the reference
//      calls libm cosf/sinf whose last bit is libm-version dependent; we use the
//      correctly rounded value (float)cos((double)angle).
//  (3) no FMA contraction in `x*b + y*a` (src/ORBextractor.cc:119-120).
#include "orb_oracle.h"

#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstring>

#include "cvprim.h"

namespace orc {

const int8_t kBriefPattern[1024] = {
#include "../planarslam_amd/csrc/brief_pattern.inc"
};

static const int PATCH_SIZE = 31, HALF_PATCH_SIZE = 15, EDGE_THRESHOLD = 19;

// reference src/ORBextractor.cc:410-470
OrbOracle::OrbOracle(int nf, float sf, int nl, int ini, int mn)
    : nfeatures(nf), nlevels(nl), ini_th(ini), min_th(mn), scale_factor(sf) {
    scale.resize(nl); sigma2.resize(nl); inv_scale.resize(nl); inv_sigma2.resize(nl);
    scale[0] = 1.0f; sigma2[0] = 1.0f;
    for (int i = 1; i < nl; i++) {
        scale[i] = (float)(scale[i - 1] * scale_factor);      // float*double -> float (:418)
        sigma2[i] = scale[i] * scale[i];
    }
    for (int i = 0; i < nl; i++) { inv_scale[i] = 1.0f / scale[i]; inv_sigma2[i] = 1.0f / sigma2[i]; }

    features_per_level.resize(nl);
    float factor = (float)(1.0f / scale_factor);
    float nDesired = nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nl));
    int sum = 0;
    for (int l = 0; l < nl - 1; l++) {
        features_per_level[l] = cv_round(nDesired);
        sum += features_per_level[l];
        nDesired *= factor;
    }
    features_per_level[nl - 1] = std::max(nfeatures - sum, 0);

    umax.assign(HALF_PATCH_SIZE + 1, 0);
    int v, v0, vmax = cv_floor(HALF_PATCH_SIZE * std::sqrt(2.f) / 2 + 1);
    int vmin = cv_ceil(HALF_PATCH_SIZE * std::sqrt(2.f) / 2);
    const double hp2 = HALF_PATCH_SIZE * HALF_PATCH_SIZE;
    for (v = 0; v <= vmax; ++v) umax[v] = cv_round(std::sqrt(hp2 - v * v));
    for (v = HALF_PATCH_SIZE, v0 = 0; v >= vmin; --v) {
        while (umax[v0] == umax[v0 + 1]) ++v0;
        umax[v] = v0;
        ++v0;
    }
}

// reference src/ORBextractor.cc:1107-1131.  The 19-px reflect border the reference adds
// is never read by any consumer (resize/FAST/IC_Angle stay inside the ROI, the blur works
// on a borderless clone), so levels are stored borderless.
void OrbOracle::compute_pyramid(const uint8_t* gray, int W, int H, int pitch) {
    pyramid.assign(nlevels, Image());
    for (int l = 0; l < nlevels; l++) {
        float s = inv_scale[l];
        Image& im = pyramid[l];
        im.w = cv_round((float)W * s);
        im.h = cv_round((float)H * s);
        im.px.resize((size_t)im.w * im.h);
        if (l == 0) {
            for (int y = 0; y < H; y++) std::memcpy(&im.px[(size_t)y * W], gray + (size_t)y * pitch, W);
        } else {
            const Image& p = pyramid[l - 1];
            resize_linear_u8(p.px.data(), p.w, p.h, p.w, im.px.data(), im.w, im.h, im.w);
        }
    }
}

// reference src/ORBextractor.cc:771-827 (cell loop of ComputeKeyPointsOctTree)
void OrbOracle::detect_level(int level, std::vector<Candidate>& cand) const {
    cand.clear();
    const Image& im = pyramid[level];
    const int minBorderX = EDGE_THRESHOLD - 3, minBorderY = minBorderX;
    const int maxBorderX = im.w - EDGE_THRESHOLD + 3, maxBorderY = im.h - EDGE_THRESHOLD + 3;
    const float W = 30;
    const float width = (float)(maxBorderX - minBorderX), height = (float)(maxBorderY - minBorderY);
    const int nCols = (int)(width / W), nRows = (int)(height / W);
    if (nCols < 1 || nRows < 1) return;
    const int wCell = (int)std::ceil(width / nCols), hCell = (int)std::ceil(height / nRows);
    std::vector<FastKp> cell;
    for (int i = 0; i < nRows; i++) {
        const float iniY = (float)(minBorderY + i * hCell);
        float maxY = iniY + hCell + 6;
        if (iniY >= maxBorderY - 3) continue;
        if (maxY > maxBorderY) maxY = (float)maxBorderY;
        for (int j = 0; j < nCols; j++) {
            const float iniX = (float)(minBorderX + j * wCell);
            float maxX = iniX + wCell + 6;
            if (iniX >= maxBorderX - 6) continue;
            if (maxX > maxBorderX) maxX = (float)maxBorderX;
            const int x0 = (int)iniX, x1 = (int)maxX, y0 = (int)iniY, y1 = (int)maxY;
            const uint8_t* roi = im.row(y0) + x0;
            fast9_16(roi, x1 - x0, y1 - y0, im.w, ini_th, true, cell);
            if (cell.empty()) fast9_16(roi, x1 - x0, y1 - y0, im.w, min_th, true, cell);
            for (const FastKp& k : cell) cand.push_back({k.x + j * wCell, k.y + i * hCell, k.score});
        }
    }
}

// ---- octree (reference src/ORBextractor.cc:481-537 DivideNode, :539-763 DistributeOctTree) ----
namespace {
struct Node {
    int x0, x1, y0, y1;            // UL.x, UR.x, UL.y, BL.y
    std::vector<Candidate> keys;   // order preserved (stable split)
    bool no_more = false;
    int prev = -1, next = -1;      // list links (index into pool); pool index == creation order
};
struct NodeList {
    std::vector<Node> pool;
    int head = -1, tail = -1, count = 0;
    int push_back(Node&& n) {
        pool.push_back(std::move(n));
        int id = (int)pool.size() - 1;
        pool[id].prev = tail; pool[id].next = -1;
        if (tail >= 0) pool[tail].next = id; else head = id;
        tail = id; count++;
        return id;
    }
    int push_front(Node&& n) {
        pool.push_back(std::move(n));
        int id = (int)pool.size() - 1;
        pool[id].prev = -1; pool[id].next = head;
        if (head >= 0) pool[head].prev = id; else tail = id;
        head = id; count++;
        return id;
    }
    int erase(int id) {   // returns the following element
        int p = pool[id].prev, n = pool[id].next;
        if (p >= 0) pool[p].next = n; else head = n;
        if (n >= 0) pool[n].prev = p; else tail = p;
        count--;
        std::vector<Candidate>().swap(pool[id].keys);
        return n;
    }
};
// :481-537
void divide(const Node& p, Node c[4]) {
    const int halfX = (int)std::ceil((float)(p.x1 - p.x0) / 2);
    const int halfY = (int)std::ceil((float)(p.y1 - p.y0) / 2);
    const int xm = p.x0 + halfX, ym = p.y0 + halfY;
    c[0].x0 = p.x0; c[0].x1 = xm;   c[0].y0 = p.y0; c[0].y1 = ym;
    c[1].x0 = xm;   c[1].x1 = p.x1; c[1].y0 = p.y0; c[1].y1 = ym;
    c[2].x0 = p.x0; c[2].x1 = xm;   c[2].y0 = ym;   c[2].y1 = p.y1;
    c[3].x0 = xm;   c[3].x1 = p.x1; c[3].y0 = ym;   c[3].y1 = p.y1;
    for (const Candidate& k : p.keys) {
        int q = ((float)k.x < xm ? 0 : 1) + ((float)k.y < ym ? 0 : 2);
        c[q].keys.push_back(k);
    }
    for (int q = 0; q < 4; q++) c[q].no_more = c[q].keys.size() == 1;
}
}  // namespace

std::vector<Candidate> OrbOracle::distribute_octree(const std::vector<Candidate>& keys, int minX, int maxX,
                                                    int minY, int maxY, int N) const {
    std::vector<Candidate> result;
    const int nIni = (int)std::round((float)(maxX - minX) / (maxY - minY));   // :543
    if (nIni < 1) return result;   // the reference divides by zero here; we refuse
    const float hX = (float)(maxX - minX) / nIni;
    NodeList L;
    std::vector<int> ini(nIni);
    for (int i = 0; i < nIni; i++) {   // :552-563
        Node n;
        n.x0 = (int)(hX * (float)i); n.x1 = (int)(hX * (float)(i + 1));
        n.y0 = 0; n.y1 = maxY - minY;
        ini[i] = L.push_back(std::move(n));
    }
    for (const Candidate& k : keys) L.pool[ini[(int)((float)k.x / hX)]].keys.push_back(k);   // :566-570
    for (int it = L.head; it >= 0;) {   // :572-585
        Node& n = L.pool[it];
        if (n.keys.size() == 1) { n.no_more = true; it = n.next; }
        else if (n.keys.empty()) it = L.erase(it);
        else it = n.next;
    }

    bool finish = false;
    std::vector<std::pair<int, int>> sizeAndNode;   // (size, node id); id order == creation order
    auto split_into_front = [&](int id, int& nToExpand) {   // shared body of :611-654 and :681-718
        Node c[4];
        divide(L.pool[id], c);
        for (int q = 0; q < 4; q++) {
            if (c[q].keys.empty()) continue;
            const int sz = (int)c[q].keys.size();
            int cid = L.push_front(std::move(c[q]));
            if (sz > 1) { nToExpand++; sizeAndNode.push_back({sz, cid}); }
        }
    };
    while (!finish) {   // :595
        int prevSize = L.count;
        int nToExpand = 0;
        sizeAndNode.clear();
        for (int it = L.head; it >= 0;) {   // :607-660
            if (L.pool[it].no_more) { it = L.pool[it].next; continue; }
            split_into_front(it, nToExpand);
            it = L.erase(it);
        }
        if (L.count >= N || L.count == prevSize) {   // :664
            finish = true;
        } else if (L.count + nToExpand * 3 > N) {   // :668
            while (!finish) {
                prevSize = L.count;
                std::vector<std::pair<int, int>> prev = sizeAndNode;
                sizeAndNode.clear();
                std::sort(prev.begin(), prev.end());   // ties: creation order (refinement 1)
                for (int j = (int)prev.size() - 1; j >= 0; j--) {
                    int dummy = 0;
                    split_into_front(prev[j].second, dummy);
                    L.erase(prev[j].second);
                    if (L.count >= N) break;
                }
                if (L.count >= N || L.count == prevSize) finish = true;
            }
        }
    }
    // :742-760 best response per node, first wins ties, in list order
    result.reserve(L.count);
    for (int it = L.head; it >= 0; it = L.pool[it].next) {
        const std::vector<Candidate>& v = L.pool[it].keys;
        const Candidate* best = &v[0];
        for (size_t k = 1; k < v.size(); k++)
            if (v[k].score > best->score) best = &v[k];
        result.push_back(*best);
    }
    return result;
}

// reference src/ORBextractor.cc:77-104
float OrbOracle::ic_angle(const Image& img, int x, int y) const {
    int m_01 = 0, m_10 = 0;
    const uint8_t* center = img.row(y) + x;
    for (int u = -HALF_PATCH_SIZE; u <= HALF_PATCH_SIZE; ++u) m_10 += u * center[u];
    const int step = img.w;
    for (int v = 1; v <= HALF_PATCH_SIZE; ++v) {
        int v_sum = 0;
        const int d = umax[v];
        for (int u = -d; u <= d; ++u) {
            int val_plus = center[u + v * step], val_minus = center[u - v * step];
            v_sum += (val_plus - val_minus);
            m_10 += u * (val_plus + val_minus);
        }
        m_01 += v * v_sum;
    }
    return fast_atan2((float)m_01, (float)m_10);
}

// reference src/ORBextractor.cc:107-147
void OrbOracle::brief(const Image& img, const KeyPoint& kp, uint8_t* desc) const {
    const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
    const float angle = kp.angle * factorPI;
    const float a = (float)std::cos((double)angle), b = (float)std::sin((double)angle);   // refinement 2
    const uint8_t* center = img.row(cv_round(kp.y)) + cv_round(kp.x);
    const int step = img.w;
    const int8_t* pat = kBriefPattern;
    for (int i = 0; i < 32; i++) {
        int val = 0;
        for (int bit = 0; bit < 8; bit++, pat += 4) {
            const float x0 = pat[0], y0 = pat[1], x1 = pat[2], y1 = pat[3];
            int t0 = center[cv_round(x0 * b + y0 * a) * step + cv_round(x0 * a - y0 * b)];
            int t1 = center[cv_round(x1 * b + y1 * a) * step + cv_round(x1 * a - y1 * b)];
            val |= (t0 < t1) << bit;
        }
        desc[i] = (uint8_t)val;
    }
}

// reference src/ORBextractor.cc:1043-1105 + :765-853
int OrbOracle::extract(const uint8_t* gray, int W, int H, int pitch, std::vector<KeyPoint>& kps,
                       std::vector<uint8_t>& desc) {
    kps.clear(); desc.clear();
    if (!gray || W <= 0 || H <= 0) return 0;
    compute_pyramid(gray, W, H, pitch);
    candidates.assign(nlevels, {});
    level_kps.assign(nlevels, {});
    blurred.assign(nlevels, Image());
    for (int l = 0; l < nlevels; l++) {
        const Image& im = pyramid[l];
        const int minBX = EDGE_THRESHOLD - 3, minBY = minBX;
        const int maxBX = im.w - EDGE_THRESHOLD + 3, maxBY = im.h - EDGE_THRESHOLD + 3;
        detect_level(l, candidates[l]);
        std::vector<Candidate> kept = distribute_octree(candidates[l], minBX, maxBX, minBY, maxBY, features_per_level[l]);
        const int scaledPatchSize = (int)(PATCH_SIZE * scale[l]);   // :835
        for (const Candidate& c : kept) {
            KeyPoint k;
            k.x = (float)(c.x + minBX); k.y = (float)(c.y + minBY);
            k.size = (float)scaledPatchSize; k.angle = -1; k.response = (float)c.score;
            k.octave = l; k.class_id = -1;
            level_kps[l].push_back(k);
        }
    }
    for (int l = 0; l < nlevels; l++)   // :850-852
        for (KeyPoint& k : level_kps[l]) k.angle = ic_angle(pyramid[l], cv_round(k.x), cv_round(k.y));

    for (int l = 0; l < nlevels; l++) {   // :1074-1104
        std::vector<KeyPoint>& v = level_kps[l];
        if (v.empty()) continue;
        Image& bl = blurred[l];
        bl.w = pyramid[l].w; bl.h = pyramid[l].h; bl.px.resize(pyramid[l].px.size());
        gaussian7_s2_u8(pyramid[l].px.data(), bl.w, bl.h, bl.w, bl.px.data(), bl.w);
        size_t off = desc.size();
        desc.resize(off + 32 * v.size());
        for (size_t i = 0; i < v.size(); i++) brief(bl, v[i], &desc[off + 32 * i]);
        for (const KeyPoint& k0 : v) {
            KeyPoint k = k0;
            if (l != 0) { k.x *= scale[l]; k.y *= scale[l]; }
            kps.push_back(k);
        }
    }
    return (int)kps.size();
}

}  // namespace orc
