// oracle/match_oracle.cpp — TEST INFRASTRUCTURE (CPU oracle), not product code.
// CPU restatement of the descriptor matchers on the hot path:
//   ORBmatcher::DescriptorDistance   reference src/ORBmatcher.cc:1712-1728 (same code in LSDmatcher.cpp:316-332)
//   cv::BFMatcher(NORM_HAMMING).match / knnMatch(k=2)  (OpenCV, un-vendored; SURVEY.md A7: per query the
//       smallest distances, lowest train index first on ties, ascending)
//   ORBmatcher::MatchORBPoints       reference src/ORBmatcher.cc:1332-1394
//   LSDmatcher::SearchByDescriptor   reference src/LSDmatcher.cpp:242-279
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>

namespace orc {

// src/ORBmatcher.cc:1712-1728 (SWAR popcount over 8 x 32 bit)
int descriptor_distance(const uint8_t* a, const uint8_t* b) {
    const int32_t* pa = (const int32_t*)a;
    const int32_t* pb = (const int32_t*)b;
    int dist = 0;
    for (int i = 0; i < 8; i++, pa++, pb++) {
        unsigned int v = *pa ^ *pb;
        v = v - ((v >> 1) & 0x55555555);
        v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
        dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
    }
    return dist;
}

// BFMatcher::knnMatch: idx/dist are [nq][k]; missing neighbours (nt < k) are -1 / 0x7fffffff
void bf_knn(const uint8_t* q, int nq, const uint8_t* t, int nt, int k, int32_t* idx, int32_t* dist) {
    for (int i = 0; i < nq; i++) {
        int bi[2] = {-1, -1}, bd[2] = {0x7fffffff, 0x7fffffff};
        for (int j = 0; j < nt; j++) {
            const int d = descriptor_distance(q + 32 * (size_t)i, t + 32 * (size_t)j);
            if (d < bd[0]) { bd[1] = bd[0]; bi[1] = bi[0]; bd[0] = d; bi[0] = j; }
            else if (d < bd[1]) { bd[1] = d; bi[1] = j; }
        }
        for (int n = 0; n < k; n++) { idx[(size_t)i * k + n] = bi[n]; dist[(size_t)i * k + n] = bd[n]; }
    }
}

// ORBmatcher::MatchORBPoints.  cur_match[q] = index of the last-frame keypoint whose MapPoint is copied to
// CurrentFrame.mvpMapPoints[q] (entries not assigned are left untouched).  Returns NPair.
int match_orb_points(const uint8_t* cur, int n_cur, const uint8_t* last, int n_last, const uint8_t* last_has_mp,
                     const uint8_t* last_outlier, int32_t* cur_match) {
    std::vector<int32_t> idx(n_cur), dist(n_cur);
    bf_knn(cur, n_cur, last, n_last, 1, idx.data(), dist.data());
    if (n_last == 0) return 0;   // BFMatcher returns no matches
    double min_dist = 1000;
    for (int i = 0; i < n_cur; i++) if ((float)dist[i] < min_dist) min_dist = (float)dist[i];
    const double DIST_THRESHOLD = 15;
    int npair = 0;
    for (int i = 0; i < n_cur; i++) {
        if ((float)dist[i] < std::max(2 * min_dist, DIST_THRESHOLD)) {
            // quirk (:1385): the outlier flag is indexed by the good-match counter, not by trainIdx
            if (last_has_mp[idx[i]] && !(npair < n_last && last_outlier[npair])) cur_match[i] = idx[i];
            npair++;
        }
    }
    return npair;
}

// LSDmatcher::SearchByDescriptor(KeyFrame*, Frame&, ...).  cur_match[t] = keyframe line index whose MapLine is
// written to vpMapLineMatches[t] (initialised to -1 here, as the reference re-creates the vector).
int lsd_search_by_descriptor(const uint8_t* kf, int n_kf, const uint8_t* cur, int n_cur, const uint8_t* kf_has_ml, int32_t* cur_match) {
    for (int i = 0; i < n_cur; i++) cur_match[i] = -1;
    if (n_cur < 2 || n_kf == 0) return 0;   // knnMatch(k=2) needs two train rows (the reference would read out of bounds)
    std::vector<int32_t> idx(2 * (size_t)n_kf), dist(2 * (size_t)n_kf);
    bf_knn(kf, n_kf, cur, n_cur, 2, idx.data(), dist.data());
    const float minRatio = 1.0f / 1.5f;
    int nmatches = 0;
    for (int i = 0; i < n_kf; i++) {
        const double dist_12 = (float)dist[2 * i] / (float)dist[2 * i + 1];
        if (dist_12 < minRatio && kf_has_ml[i]) { cur_match[idx[2 * i]] = i; nmatches++; }
    }
    return nmatches;
}

// MapPoint::ComputeDistinctiveDescriptors (reference src/MapPoint.cc:259-324): among the n observed descriptors, the one whose median Hamming
// distance to all of them (itself included: 0) is least; median = sorted[int(0.5 * (n - 1))]; the first such descriptor wins.  Returns its index
// (-1 for n == 0) and the median.
int distinctive_descriptor(const uint8_t* desc, int n, int* median_out) {
    int best_median = INT32_MAX, best = n > 0 ? 0 : -1;
    std::vector<int> row(n);
    for (int i = 0; i < n; i++) {
        for (int j = 0; j < n; j++) row[j] = i == j ? 0 : descriptor_distance(desc + 32 * (size_t)i, desc + 32 * (size_t)j);
        std::sort(row.begin(), row.end());
        const int median = row[(size_t)(0.5 * (n - 1))];
        if (median < best_median) { best_median = median; best = i; }
    }
    if (median_out) *median_out = n > 0 ? best_median : 0;
    return best;
}

}  // namespace orc

extern "C" {
int orc_distinctive_descriptor(const uint8_t* desc, int n, int* median) { return orc::distinctive_descriptor(desc, n, median); }
int orc_descriptor_distance(const uint8_t* a, const uint8_t* b) { return orc::descriptor_distance(a, b); }
void orc_bf_knn(const uint8_t* q, int nq, const uint8_t* t, int nt, int k, int32_t* idx, int32_t* dist) { orc::bf_knn(q, nq, t, nt, k, idx, dist); }
int orc_match_orb_points(const uint8_t* cur, int n_cur, const uint8_t* last, int n_last, const uint8_t* has_mp, const uint8_t* outl, int32_t* m) {
    return orc::match_orb_points(cur, n_cur, last, n_last, has_mp, outl, m);
}
int orc_lsd_search_by_descriptor(const uint8_t* kf, int n_kf, const uint8_t* cur, int n_cur, const uint8_t* has_ml, int32_t* m) {
    return orc::lsd_search_by_descriptor(kf, n_kf, cur, n_cur, has_ml, m);
}
}
