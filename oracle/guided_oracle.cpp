// oracle/guided_oracle.cpp — TEST INFRASTRUCTURE (CPU oracle), not product code.
// CPU restatement of the guided matchers on the hot path:
//   ORBmatcher::SearchByProjection(Frame&, const Frame&, th, bMono)      reference src/ORBmatcher.cc:1396-1535
//   ORBmatcher::SearchByProjection(Frame&, vector<MapPoint*>&, th)       reference src/ORBmatcher.cc:46-130
//   ORBmatcher::SearchByBoW(KeyFrame*, Frame&, vector<MapPoint*>&)       reference src/ORBmatcher.cc:160-292
//   ORBmatcher::ComputeThreeMaxima                                       reference src/ORBmatcher.cc:1666-1708
//   Frame::AssignFeaturesToGrid / PosInGrid / GetFeaturesInArea          reference src/Frame.cc:155-166, 526-535, 440-489
//   LSDmatcher::SearchByProjection + Frame::GetLinesInArea               reference src/LSDmatcher.cpp:141-211, src/Frame.cc:491-524
//   PlaneMatcher::SearchMapByCoefficients + PointDistanceFromPlane       reference src/PlaneMatcher.cpp:10-79
//   Frame::ComputePlaneWorldCoeff                                        reference src/Frame.cc:815-820
// The loops are the reference's loops (same order, same float32 expressions); the Frame / MapPoint objects
// are replaced by the flat views of include/planar_abi.h.
//
// PARITY UNPINNED for the two cv::Mat products these functions use (Rcw*x3Dw+tcw, -Rcw.t()*tcw, temp*coef):
// OpenCV is not in this container, so cv::gemm's float32 small-matrix path (core/src/matmul.cpp, len<=4:
// float products summed left to right, then (float)(t*alpha + c*beta) in double) and its general path
// (double accumulation, used when a transpose flag is set) are restated from the 3.4 sources by reading.
#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../include/planar_abi.h"

namespace orc {

int descriptor_distance(const uint8_t* a, const uint8_t* b);   // match_oracle.cpp

static const int TH_HIGH = 100, TH_LOW = 50, HISTO_LENGTH = 30;   // src/ORBmatcher.cc:38-40

// Frame::mGrid of one frame (src/Frame.cc:155-166, 526-535)
struct Grid {
    std::vector<int> cell[PLANAR_GRID_COLS][PLANAR_GRID_ROWS];
    void build(const planar_frame_view& f, const planar_keypoint* k, int n) {
        for (int i = 0; i < n; i++) {
            const int px = (int)std::round((k[i].x - f.min_x) * f.grid_w_inv);
            const int py = (int)std::round((k[i].y - f.min_y) * f.grid_h_inv);
            if (px < 0 || px >= PLANAR_GRID_COLS || py < 0 || py >= PLANAR_GRID_ROWS) continue;
            cell[px][py].push_back(i);
        }
    }
};

// Frame::GetFeaturesInArea (src/Frame.cc:440-489)
static void features_in_area(const planar_frame_view& f, const Grid& g, const planar_keypoint* k, float x, float y, float r, int minLevel,
                             int maxLevel, std::vector<int>& out) {
    out.clear();
    const int nMinCellX = std::max(0, (int)std::floor((x - f.min_x - r) * f.grid_w_inv));
    if (nMinCellX >= PLANAR_GRID_COLS) return;
    const int nMaxCellX = std::min(PLANAR_GRID_COLS - 1, (int)std::ceil((x - f.min_x + r) * f.grid_w_inv));
    if (nMaxCellX < 0) return;
    const int nMinCellY = std::max(0, (int)std::floor((y - f.min_y - r) * f.grid_h_inv));
    if (nMinCellY >= PLANAR_GRID_ROWS) return;
    const int nMaxCellY = std::min(PLANAR_GRID_ROWS - 1, (int)std::ceil((y - f.min_y + r) * f.grid_h_inv));
    if (nMaxCellY < 0) return;
    const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
    for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
        for (int iy = nMinCellY; iy <= nMaxCellY; iy++)
            for (int idx : g.cell[ix][iy]) {
                if (bCheckLevels) {
                    if (k[idx].octave < minLevel) continue;
                    if (maxLevel >= 0 && k[idx].octave > maxLevel) continue;
                }
                const float distx = k[idx].x - x, disty = k[idx].y - y;
                if (std::fabs(distx) < r && std::fabs(disty) < r) out.push_back(idx);
            }
}

// src/ORBmatcher.cc:1666-1708
static void three_maxima(const std::vector<int>* histo, int L, int& ind1, int& ind2, int& ind3) {
    int max1 = 0, max2 = 0, max3 = 0;
    for (int i = 0; i < L; i++) {
        const int s = (int)histo[i].size();
        if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
        else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
        else if (s > max3) { max3 = s; ind3 = i; }
    }
    if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
    else if (max3 < 0.1f * (float)max1) ind3 = -1;
}

static int rot_bin(float a_from, float a_to) {   // :1495-1501
    const float factor = 1.0f / HISTO_LENGTH;
    float rot = a_from - a_to;
    if (rot < 0.0) rot += 360.0f;
    int bin = (int)std::round(rot * factor);
    if (bin == HISTO_LENGTH) bin = 0;
    return bin;
}

// cv::gemm float32 small-matrix path: d = A(3x3) * x + c
static inline float gemm3_row(const float* a, const float* x, float c) {
    const float t = a[0] * x[0] + a[1] * x[1] + a[2] * x[2];
    return (float)((double)t * 1.0 + (double)c * 1.0);
}

// ---- a20 ---------------------------------------------------------------------------------------
int search_by_projection_frame(const planar_frame_view& cur, const planar_last_frame_view& last, int b, float th, bool mono,
                               bool check_orientation, int32_t* cur_match) {
    const int N = cur.n[b], NL = last.n[b];
    const planar_keypoint* keys = cur.keys_un + (size_t)b * cur.stride;
    const float* uR = cur.u_right + (size_t)b * cur.stride;
    const uint8_t* desc = cur.desc + (size_t)b * cur.stride * 32;
    const size_t lo = (size_t)b * last.stride;
    std::vector<uint8_t> blocked(N, 0);
    if (cur.blocked) for (int i = 0; i < N; i++) blocked[i] = cur.blocked[(size_t)b * cur.stride + i];
    Grid grid; grid.build(cur, keys, N);

    int nmatches = 0;
    std::vector<int> rotHist[HISTO_LENGTH];
    const float* Tc = cur.Tcw + (size_t)b * 16;
    const float* Tl = last.Tcw + (size_t)b * 16;
    const float Rcw[9] = {Tc[0], Tc[1], Tc[2], Tc[4], Tc[5], Tc[6], Tc[8], Tc[9], Tc[10]};
    const float tcw[3] = {Tc[3], Tc[7], Tc[11]};
    // twc = -Rcw.t()*tcw : general gemm path (transpose flag) -> double accumulation, alpha = -1
    float twc[3];
    for (int i = 0; i < 3; i++) {
        double s = 0;
        for (int k = 0; k < 3; k++) s += (double)Rcw[3 * k + i] * (double)tcw[k];
        twc[i] = (float)(s * -1.0);
    }
    const float Rlw2[3] = {Tl[8], Tl[9], Tl[10]};
    const float tlc2 = gemm3_row(Rlw2, twc, Tl[11]);
    const bool bForward = tlc2 > cur.b && !mono;
    const bool bBackward = -tlc2 > cur.b && !mono;

    std::vector<int> vIndices2;
    for (int i = 0; i < NL; i++) {
        if (!last.usable[lo + i]) continue;
        const float* xw = last.xw + (lo + i) * 3;
        const float xc = gemm3_row(Rcw, xw, tcw[0]);
        const float yc = gemm3_row(Rcw + 3, xw, tcw[1]);
        const float zc = gemm3_row(Rcw + 6, xw, tcw[2]);
        const float invzc = (float)(1.0 / zc);
        if (invzc < 0) continue;
        const float u = cur.fx * xc * invzc + cur.cx;
        const float v = cur.fy * yc * invzc + cur.cy;
        if (u < cur.min_x || u > cur.max_x) continue;
        if (v < cur.min_y || v > cur.max_y) continue;
        const int nLastOctave = last.octave[lo + i];
        const float radius = th * cur.scale_factors[nLastOctave];
        if (bForward) features_in_area(cur, grid, keys, u, v, radius, nLastOctave, -1, vIndices2);
        else if (bBackward) features_in_area(cur, grid, keys, u, v, radius, 0, nLastOctave, vIndices2);
        else features_in_area(cur, grid, keys, u, v, radius, nLastOctave - 1, nLastOctave + 1, vIndices2);
        if (vIndices2.empty()) continue;
        const uint8_t* dMP = last.mp_desc + (lo + i) * 32;
        int bestDist = 256, bestIdx2 = -1;
        for (int i2 : vIndices2) {
            if (blocked[i2]) continue;
            if (uR[i2] > 0) {
                const float ur = u - cur.bf * invzc;
                const float er = std::fabs(ur - uR[i2]);
                if (er > radius) continue;
            }
            const int dist = descriptor_distance(dMP, desc + (size_t)i2 * 32);
            if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
        }
        if (bestDist <= TH_HIGH) {
            cur_match[bestIdx2] = i;
            blocked[bestIdx2] = last.mp_observed[lo + i];   // what the next `->Observations()>0` test will see
            nmatches++;
            if (check_orientation) rotHist[rot_bin(last.angle[lo + i], keys[bestIdx2].angle)].push_back(bestIdx2);
        }
    }
    if (check_orientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++)
            if (i != ind1 && i != ind2 && i != ind3)
                for (int idx : rotHist[i]) { cur_match[idx] = -1; nmatches--; }
    }
    return nmatches;
}

// ---- a21 ---------------------------------------------------------------------------------------
int search_by_projection_map(const planar_frame_view& F, const planar_map_probes& mp, int b, float th, float nn_ratio, int32_t* match) {
    const int N = F.n[b], NP = mp.n[b];
    const planar_keypoint* keys = F.keys_un + (size_t)b * F.stride;
    const float* uR = F.u_right + (size_t)b * F.stride;
    const uint8_t* desc = F.desc + (size_t)b * F.stride * 32;
    const size_t po = (size_t)b * mp.stride;
    std::vector<uint8_t> blocked(N, 0);
    if (F.blocked) for (int i = 0; i < N; i++) blocked[i] = F.blocked[(size_t)b * F.stride + i];
    Grid grid; grid.build(F, keys, N);
    int nmatches = 0;
    const bool bFactor = th != 1.0;
    std::vector<int> vIndices;
    for (int iMP = 0; iMP < NP; iMP++) {
        if (!mp.in_view[po + iMP]) continue;
        const int nPredictedLevel = mp.level[po + iMP];
        float r = mp.view_cos[po + iMP] > 0.998 ? 2.5f : 4.0f;   // RadiusByViewingCos :132-138
        if (bFactor) r *= th;
        features_in_area(F, grid, keys, mp.proj_x[po + iMP], mp.proj_y[po + iMP], r * F.scale_factors[nPredictedLevel], nPredictedLevel - 1,
                         nPredictedLevel, vIndices);
        if (vIndices.empty()) continue;
        const uint8_t* dMP = mp.desc + (po + iMP) * 32;
        int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
        for (int idx : vIndices) {
            if (blocked[idx]) continue;
            if (uR[idx] > 0) {
                const float er = std::fabs(mp.proj_xr[po + iMP] - uR[idx]);
                if (er > r * F.scale_factors[nPredictedLevel]) continue;
            }
            const int dist = descriptor_distance(dMP, desc + (size_t)idx * 32);
            if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestLevel2 = bestLevel; bestLevel = keys[idx].octave; bestIdx = idx; }
            else if (dist < bestDist2) { bestLevel2 = keys[idx].octave; bestDist2 = dist; }
        }
        if (bestDist <= TH_HIGH) {
            if (bestLevel == bestLevel2 && bestDist > nn_ratio * bestDist2) continue;
            match[bestIdx] = iMP;
            blocked[bestIdx] = mp.observed[po + iMP];
            nmatches++;
        }
    }
    return nmatches;
}

// ---- a22 ---------------------------------------------------------------------------------------
// The FeatureVectors are std::map<node, vector<feature idx>> filled in feature order (DBoW2
// TemplatedVocabulary::transform -> FeatureVector::addFeature); rebuilt here from per-feature node ids.
int search_by_bow(int n_kf, const int32_t* kf_node, const uint8_t* kf_usable, const float* kf_angle, const uint8_t* kf_desc, int n_f,
                  const int32_t* f_node, const float* f_angle, const uint8_t* f_desc, float nn_ratio, bool check_orientation, int32_t* match) {
    std::vector<std::pair<int, std::vector<int>>> fvK, fvF;
    auto build = [](int n, const int32_t* node, std::vector<std::pair<int, std::vector<int>>>& fv) {
        std::vector<std::pair<int, int>> tmp;
        for (int i = 0; i < n; i++) if (node[i] >= 0) tmp.push_back({node[i], i});
        std::stable_sort(tmp.begin(), tmp.end(), [](const std::pair<int, int>& a, const std::pair<int, int>& b) { return a.first < b.first; });
        for (auto& t : tmp) {
            if (fv.empty() || fv.back().first != t.first) fv.push_back({t.first, {}});
            fv.back().second.push_back(t.second);
        }
    };
    build(n_kf, kf_node, fvK);
    build(n_f, f_node, fvF);
    for (int i = 0; i < n_f; i++) match[i] = -1;   // vpMapPointMatches = vector<MapPoint*>(F.N, NULL)
    int nmatches = 0;
    std::vector<int> rotHist[HISTO_LENGTH];
    size_t ik = 0, jf = 0;
    while (ik < fvK.size() && jf < fvF.size()) {
        if (fvK[ik].first == fvF[jf].first) {
            for (int realIdxKF : fvK[ik].second) {
                if (!kf_usable[realIdxKF]) continue;
                const uint8_t* dKF = kf_desc + (size_t)realIdxKF * 32;
                int bestDist1 = 256, bestIdxF = -1, bestDist2 = 256;
                for (int realIdxF : fvF[jf].second) {
                    if (match[realIdxF] >= 0) continue;
                    const int dist = descriptor_distance(dKF, f_desc + (size_t)realIdxF * 32);
                    if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdxF = realIdxF; }
                    else if (dist < bestDist2) bestDist2 = dist;
                }
                if (bestDist1 <= TH_LOW && (float)bestDist1 < nn_ratio * (float)bestDist2) {
                    match[bestIdxF] = realIdxKF;
                    if (check_orientation) rotHist[rot_bin(kf_angle[realIdxKF], f_angle[bestIdxF])].push_back(bestIdxF);
                    nmatches++;
                }
            }
            ik++; jf++;
        } else if (fvK[ik].first < fvF[jf].first) {
            while (ik < fvK.size() && fvK[ik].first < fvF[jf].first) ik++;   // lower_bound
        } else {
            while (jf < fvF.size() && fvF[jf].first < fvK[ik].first) jf++;
        }
    }
    if (check_orientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int idx : rotHist[i]) { match[idx] = -1; nmatches--; }
        }
    }
    return nmatches;
}

// ---- a24 ---------------------------------------------------------------------------------------
int lsd_search_by_projection(int n_lines, const planar_keyline* kl, const uint8_t* ldesc, const uint8_t* blocked_in, int n_ml,
                             const uint8_t* ml_in_view, const float* ml_proj, const int32_t* ml_level, const float* ml_view_cos,
                             const uint8_t* ml_desc, const uint8_t* ml_observed, const float* scale_factors, float th, float nn_ratio,
                             int32_t* match) {
    std::vector<uint8_t> blocked(n_lines, 0);
    if (blocked_in) std::copy(blocked_in, blocked_in + n_lines, blocked.begin());
    int nmatches = 0;
    const bool bFactor = th != 1.0;
    std::vector<int> vIndices;
    for (int j = 0; j < n_ml; j++) {
        if (!ml_in_view[j]) continue;
        const int nPredictLevel = ml_level[j];
        if (nPredictLevel < 0 || nPredictLevel >= PLANAR_MAX_LEVELS) continue;   // mvScaleFactors[level] out of bounds = UB in the reference; defined as "skip"
        float r = ml_view_cos[j] > 0.998 ? 5.0f : 8.0f;   // LSDmatcher::RadiusByViewingCos (src/LSDmatcher.cpp:369-375: 5 / 8, not ORBmatcher's 2.5 / 4)
        if (bFactor) r *= th;
        // Frame::GetLinesInArea(x1,y1,x2,y2, r*scale, level-1, level)   src/Frame.cc:491-524
        const float x1 = ml_proj[4 * j], y1 = ml_proj[4 * j + 1], x2 = ml_proj[4 * j + 2], y2 = ml_proj[4 * j + 3];
        const float rr = r * scale_factors[nPredictLevel];
        const int minLevel = nPredictLevel - 1, maxLevel = nPredictLevel;
        const bool bCheckLevels = (minLevel > 0) || (maxLevel > 0);
        vIndices.clear();
        for (int i = 0; i < n_lines; i++) {
            const float distance = (float)((0.5 * (x1 + x2) - kl[i].pt_x) * (0.5 * (x1 + x2) - kl[i].pt_x) +
                                           (0.5 * (y1 + y2) - kl[i].pt_y) * (0.5 * (y1 + y2) - kl[i].pt_y));
            if (distance > rr * rr) continue;
            const float slope = (y1 - y2) / (x1 - x2) - kl[i].angle;
            if (slope > rr * 0.01) continue;
            if (bCheckLevels) {
                if (kl[i].octave < minLevel) continue;
                if (maxLevel >= 0 && kl[i].octave > maxLevel) continue;
            }
            vIndices.push_back(i);
        }
        if (vIndices.empty()) continue;
        int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
        for (int idx : vIndices) {
            if (blocked[idx]) continue;
            const int dist = descriptor_distance(ml_desc + (size_t)j * 32, ldesc + (size_t)idx * 32);
            if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestLevel2 = bestLevel; bestLevel = kl[idx].octave; bestIdx = idx; }
            else if (dist < bestDist2) { bestLevel2 = kl[idx].octave; bestDist2 = dist; }
        }
        if (bestDist <= TH_HIGH) {
            if (bestLevel == bestLevel2 && bestDist > nn_ratio * bestDist2) continue;
            match[bestIdx] = j;
            blocked[bestIdx] = ml_observed[j];
            nmatches++;
        }
    }
    return nmatches;
}

// ---- a25 ---------------------------------------------------------------------------------------
int plane_search_by_coefficients(int n_planes, const float* pl_coef, const float* Tcw, int n_mp, const uint8_t* mp_valid,
                                 const float* mp_coef, const int32_t* mp_npts, int pts_stride, const float* mp_pts, const float* th,
                                 int32_t* match, int32_t* ver, int32_t* par) {
    const float dTh = th[0], aTh = th[1], verTh = th[2], parTh = th[3];
    int nmatches = 0;
    for (int i = 0; i < n_planes; i++) {
        // ComputePlaneWorldCoeff: transpose(mTcw) * coef, cv::gemm 4x4 small-matrix float path
        float pM[4];
        for (int r = 0; r < 4; r++) {
            const float t = Tcw[r] * pl_coef[4 * i] + Tcw[4 + r] * pl_coef[4 * i + 1] + Tcw[8 + r] * pl_coef[4 * i + 2] + Tcw[12 + r] * pl_coef[4 * i + 3];
            pM[r] = (float)((double)t * 1.0);
        }
        float ldTh = dTh, lverTh = verTh, lparTh = parTh;
        bool found = false;
        for (int j = 0; j < n_mp; j++) {
            if (!mp_valid[j]) continue;
            const float* pW = mp_coef + 4 * j;
            const float angle = pM[0] * pW[0] + pM[1] * pW[1] + pM[2] * pW[2];
            if (angle > aTh || angle < -aTh) {
                double res = 100;   // PointDistanceFromPlane :67-79
                const float* pts = mp_pts + (size_t)j * pts_stride * 3;
                for (int k = 0; k < mp_npts[j]; k++) {
                    const double dis = std::fabs(pM[0] * pts[3 * k] + pM[1] * pts[3 * k + 1] + pM[2] * pts[3 * k + 2] + pM[3]);
                    if (dis < res) res = dis;
                }
                if (res < ldTh) { ldTh = (float)res; match[i] = j; found = true; continue; }
            }
            if (angle < lverTh && angle > -lverTh) { lverTh = std::fabs(angle); ver[i] = j; continue; }
            if (angle > lparTh || angle < -lparTh) { lparTh = std::fabs(angle); par[i] = j; }
        }
        if (found) nmatches++;
    }
    return nmatches;
}

// ---- Frame::isInFrustum (src/Frame.cc:312-367 points, :369-438 lines); UNPINNED (Frame.cc cannot be built here) ----
struct FrustumPose { float Rcw[9], tcw[3], Ow[3]; };
static FrustumPose frustum_pose(const float* T) {
    FrustumPose p;
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) p.Rcw[3 * r + c] = T[4 * r + c]; p.tcw[r] = T[4 * r + 3]; }
    for (int i = 0; i < 3; i++) {   // mOw = -mRcw.t()*mtcw : general gemm path, double accumulation
        double s = 0;
        for (int k = 0; k < 3; k++) s += (double)p.Rcw[3 * k + i] * (double)p.tcw[k];
        p.Ow[i] = (float)(s * -1.0);
    }
    return p;
}
static inline float norm3(const float* v) { return (float)std::sqrt((double)v[0] * v[0] + (double)v[1] * v[1] + (double)v[2] * v[2]); }   // cv::norm: double accumulation
static inline double dot3(const float* a, const float* b) { return (double)a[0] * b[0] + (double)a[1] * b[1] + (double)a[2] * b[2]; }
static inline float logf_cr(float x) { return (float)std::log((double)x); }

void is_in_frustum_points(const planar_frame_view& F, int b, float log_scale_factor, int n_levels, int n, const uint8_t* valid, const float* xw,
                          const float* normal, const float* min_dist, const float* max_dist, float limit, uint8_t* in_view, float* proj_x,
                          float* proj_y, float* proj_xr, int32_t* level, float* view_cos) {
    const FrustumPose P = frustum_pose(F.Tcw + (size_t)b * 16);
    for (int j = 0; j < n; j++) {
        in_view[j] = 0;
        if (!valid[j]) continue;
        const float* X = xw + 3 * j;
        const float PcX = gemm3_row(P.Rcw, X, P.tcw[0]), PcY = gemm3_row(P.Rcw + 3, X, P.tcw[1]), PcZ = gemm3_row(P.Rcw + 6, X, P.tcw[2]);
        if (PcZ < 0.0f) continue;
        const float invz = 1.0f / PcZ;
        const float u = F.fx * PcX * invz + F.cx, v = F.fy * PcY * invz + F.cy;
        if (u < F.min_x || u > F.max_x) continue;
        if (v < F.min_y || v > F.max_y) continue;
        const float maxDistance = 1.2f * max_dist[j], minDistance = 0.8f * min_dist[j];
        const float PO[3] = {X[0] - P.Ow[0], X[1] - P.Ow[1], X[2] - P.Ow[2]};
        const float dist = norm3(PO);
        if (dist < minDistance || dist > maxDistance) continue;
        const float viewCos = (float)(dot3(PO, normal + 3 * j) / dist);
        if (viewCos < limit) continue;
        const float ratio = max_dist[j] / dist;                      // MapPoint::PredictScale
        int nScale = (int)std::ceil(logf_cr(ratio) / log_scale_factor);
        if (nScale < 0) nScale = 0; else if (nScale >= n_levels) nScale = n_levels - 1;
        in_view[j] = 1; proj_x[j] = u; proj_xr[j] = u - F.bf * invz; proj_y[j] = v; level[j] = nScale; view_cos[j] = viewCos;
    }
}

void is_in_frustum_lines(const planar_frame_view& F, int b, float log_scale_factor, int n, const uint8_t* valid, const double* xw6, const double* normal,
                         const float* min_dist, const float* max_dist, float limit, uint8_t* in_view, float* proj, int32_t* level, float* view_cos) {
    const FrustumPose P = frustum_pose(F.Tcw + (size_t)b * 16);
    for (int j = 0; j < n; j++) {
        in_view[j] = 0;
        if (!valid[j]) continue;
        float SP[3], EP[3];
        for (int k = 0; k < 3; k++) { SP[k] = (float)xw6[6 * j + k]; EP[k] = (float)xw6[6 * j + 3 + k]; }
        const float SPcX = gemm3_row(P.Rcw, SP, P.tcw[0]), SPcY = gemm3_row(P.Rcw + 3, SP, P.tcw[1]), SPcZ = gemm3_row(P.Rcw + 6, SP, P.tcw[2]);
        const float EPcX = gemm3_row(P.Rcw, EP, P.tcw[0]), EPcY = gemm3_row(P.Rcw + 3, EP, P.tcw[1]), EPcZ = gemm3_row(P.Rcw + 6, EP, P.tcw[2]);
        if (SPcZ < 0.0f || EPcZ < 0.0f) continue;
        const float invz1 = 1.0f / SPcZ;
        const float u1 = F.fx * SPcX * invz1 + F.cx, v1 = F.fy * SPcY * invz1 + F.cy;
        if (u1 < F.min_x || u1 > F.max_x) continue;
        if (v1 < F.min_y || v1 > F.max_y) continue;
        const float invz2 = 1.0f / EPcZ;
        const float u2 = F.fx * EPcX * invz2 + F.cx, v2 = F.fy * EPcY * invz2 + F.cy;
        if (u2 < F.min_x || u2 > F.max_x) continue;
        if (v2 < F.min_y || v2 > F.max_y) continue;
        const float maxDistance = 1.2f * max_dist[j], minDistance = 0.8f * min_dist[j];
        float OM[3];   // 0.5 * (SP + EP) - mOw : float sum, (float)(x * 0.5) in double, float difference
        for (int k = 0; k < 3; k++) OM[k] = (float)((double)(SP[k] + EP[k]) * 0.5) - P.Ow[k];
        const float dist = norm3(OM);
        if (dist < minDistance || dist > maxDistance) continue;
        const float pn[3] = {(float)normal[3 * j], (float)normal[3 * j + 1], (float)normal[3 * j + 2]};
        const float viewCos = (float)(dot3(OM, pn) / dist);
        if (viewCos < limit) continue;
        const float ratio = max_dist[j] / dist;                      // MapLine::PredictScale: no clamping
        in_view[j] = 1;
        proj[4 * j] = u1; proj[4 * j + 1] = v1; proj[4 * j + 2] = u2; proj[4 * j + 3] = v2;
        level[j] = (int)std::ceil(logf_cr(ratio) / log_scale_factor);
        view_cos[j] = viewCos;
    }
}

// ---- ORBmatcher::Fuse(KeyFrame* pKF, const vector<MapPoint*>&, th), the search half (src/ORBmatcher.cc:829-951); pinned through oracle/_ref/ref_match
//      mode `fuse` (the real function with KeyFrame::GetFeaturesInArea / IsInImage src/KeyFrame.cc:639-678, 715-718 and MapPoint::PredictScale
//      src/MapPoint.cc:402-417 compiled in).  fuse_idx[j] = bestIdx where bestDist <= TH_LOW, else -1; returns nFused.
int fuse_search(const planar_frame_view& K, int b, const float* inv_sigma2, float log_scale_factor, int n_levels, int n, const uint8_t* usable, const float* xw,
                const float* normal, const float* min_dist, const float* max_dist, const uint8_t* desc, float th, int32_t* fuse_idx, int32_t* fuse_dist) {
    const planar_keypoint* keys = K.keys_un + (size_t)b * K.stride;
    const float* uR = K.u_right + (size_t)b * K.stride;
    const uint8_t* kdesc = K.desc + (size_t)b * K.stride * 32;
    Grid g;
    g.build(K, keys, K.n[b]);                                              // KeyFrame::mGrid = the frame's grid (src/KeyFrame.cc:56-63)
    const FrustumPose P = frustum_pose(K.Tcw + (size_t)b * 16);           // GetRotation, GetTranslation, GetCameraCenter (Ow = -Rwc * tcw)
    std::vector<int> ind;
    int nFused = 0;
    for (int j = 0; j < n; j++) {
        fuse_idx[j] = -1;
        if (fuse_dist) fuse_dist[j] = 256;
        if (!usable[j]) continue;                                          // :849-853
        const float* X = xw + 3 * j;
        const float xc = gemm3_row(P.Rcw, X, P.tcw[0]), yc = gemm3_row(P.Rcw + 3, X, P.tcw[1]), zc = gemm3_row(P.Rcw + 6, X, P.tcw[2]);
        if (zc < 0.0f) continue;
        const float invz = 1 / zc;
        const float x = xc * invz, y = yc * invz;
        const float u = K.fx * x + K.cx, v = K.fy * y + K.cy;
        if (!(u >= K.min_x && u < K.max_x && v >= K.min_y && v < K.max_y)) continue;   // KeyFrame::IsInImage
        const float ur = u - K.bf * invz;
        const float maxDistance = 1.2f * max_dist[j], minDistance = 0.8f * min_dist[j];
        const float PO[3] = {X[0] - P.Ow[0], X[1] - P.Ow[1], X[2] - P.Ow[2]};
        const float dist3D = norm3(PO);
        if (dist3D < minDistance || dist3D > maxDistance) continue;
        if (dot3(PO, normal + 3 * j) < 0.5 * dist3D) continue;
        const float ratio = max_dist[j] / dist3D;                          // MapPoint::PredictScale(dist, pKF)
        int lvl = (int)std::ceil(logf_cr(ratio) / log_scale_factor);
        if (lvl < 0) lvl = 0; else if (lvl >= n_levels) lvl = n_levels - 1;
        const float radius = th * K.scale_factors[lvl];
        features_in_area(K, g, keys, u, v, radius, -1, -1, ind);           // KeyFrame::GetFeaturesInArea: no level bounds
        if (ind.empty()) continue;
        int bestDist = 256, bestIdx = -1;
        for (int idx : ind) {
            const planar_keypoint& kp = keys[idx];
            const int kl = kp.octave;
            if (kl < lvl - 1 || kl > lvl) continue;
            const float ex = u - kp.x, ey = v - kp.y;
            if (uR[idx] >= 0) {
                const float er = ur - uR[idx];
                const float e2 = ex * ex + ey * ey + er * er;
                if (e2 * inv_sigma2[kl] > 7.8) continue;
            } else {
                const float e2 = ex * ex + ey * ey;
                if (e2 * inv_sigma2[kl] > 5.99) continue;
            }
            const int dist = descriptor_distance(desc + (size_t)j * 32, kdesc + (size_t)idx * 32);
            if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
        }
        if (fuse_dist) fuse_dist[j] = bestDist;
        if (bestDist <= TH_LOW) { fuse_idx[j] = bestIdx; nFused++; }
    }
    return nFused;
}

// ---- LSDmatcher::Fuse(KeyFrame*, const vector<MapLine*>&, th), the search half (src/LSDmatcher.cpp:884-991); pinned through ref_match mode `lsd_fuse`
//      (the real function with KeyFrame::GetLinesInArea src/KeyFrame.cc:680-712 and MapLine::PredictScale src/MapLine.cpp:381-390 compiled in).
int lsd_fuse_search(const planar_frame_view& K, int b, float log_scale_factor, int n_levels, int n_lines, const planar_keyline* kl, const uint8_t* ldesc, int n,
                    const uint8_t* usable, const double* xw6, const double* normal, const float* min_dist, const float* max_dist, const uint8_t* desc, float th,
                    int32_t* fuse_idx, int32_t* fuse_dist) {
    const FrustumPose P = frustum_pose(K.Tcw + (size_t)b * 16);
    int nFused = 0;
    std::vector<int> vIndices;
    for (int j = 0; j < n; j++) {
        fuse_idx[j] = -1;
        if (fuse_dist) fuse_dist[j] = INT_MAX;
        if (!usable[j]) continue;
        float SP[3], EP[3];
        for (int k = 0; k < 3; k++) { SP[k] = (float)xw6[6 * j + k]; EP[k] = (float)xw6[6 * j + 3 + k]; }
        const float SPcX = gemm3_row(P.Rcw, SP, P.tcw[0]), SPcY = gemm3_row(P.Rcw + 3, SP, P.tcw[1]), SPcZ = gemm3_row(P.Rcw + 6, SP, P.tcw[2]);
        const float EPcX = gemm3_row(P.Rcw, EP, P.tcw[0]), EPcY = gemm3_row(P.Rcw + 3, EP, P.tcw[1]), EPcZ = gemm3_row(P.Rcw + 6, EP, P.tcw[2]);
        if (SPcZ < 0.0f || EPcZ < 0.0f) continue;
        const float invz1 = 1.0f / SPcZ;
        const float u1 = K.fx * SPcX * invz1 + K.cx, v1 = K.fy * SPcY * invz1 + K.cy;
        if (u1 < K.min_x || u1 > K.max_x) continue;
        if (v1 < K.min_y || v1 > K.max_y) continue;
        const float invz2 = 1.0f / EPcZ;
        const float u2 = K.fx * EPcX * invz2 + K.cx, v2 = K.fy * EPcY * invz2 + K.cy;
        if (u2 < K.min_x || u2 > K.max_x) continue;
        if (v2 < K.min_y || v2 > K.max_y) continue;
        const float maxDistance = 1.2f * max_dist[j], minDistance = 0.8f * min_dist[j];
        float OM[3];
        for (int k = 0; k < 3; k++) OM[k] = (float)((double)(SP[k] + EP[k]) * 0.5) - P.Ow[k];
        const float dist = norm3(OM);
        if (dist < minDistance || dist > maxDistance) continue;
        const float pn[3] = {(float)normal[3 * j], (float)normal[3 * j + 1], (float)normal[3 * j + 2]};
        if (dot3(OM, pn) < 0.5 * dist) continue;
        const float ratio = max_dist[j] / dist;
        const int lvl = (int)std::ceil(logf_cr(ratio) / log_scale_factor);   // MapLine::PredictScale: no clamping
        if (lvl < 0 || lvl >= n_levels) continue;                            // mvScaleFactors[lvl] out of bounds = UB in the reference; defined as "skip"
        const float radius = th * K.scale_factors[lvl];
        // KeyFrame::GetLinesInArea(u1, v1, u2, v2, radius) (src/KeyFrame.cc:680-712), minLevel = maxLevel = -1: no level bounds
        vIndices.clear();
        for (int i = 0; i < n_lines; i++) {
            const float distance = (float)((0.5 * (u1 + u2) - kl[i].pt_x) * (0.5 * (u1 + u2) - kl[i].pt_x) + (0.5 * (v1 + v2) - kl[i].pt_y) * (0.5 * (v1 + v2) - kl[i].pt_y));
            if (distance > radius * radius) continue;
            const float slope = (v1 - v2) / (u1 - u2) - kl[i].angle;
            if (slope > radius * 0.01) continue;
            vIndices.push_back(i);
        }
        if (vIndices.empty()) continue;
        int bestDist = INT_MAX, bestIdx = -1;
        for (int idx : vIndices) {
            const int klLevel = kl[idx].octave;
            if (klLevel < lvl - 1 || klLevel > lvl) continue;
            const int d = descriptor_distance(desc + (size_t)j * 32, ldesc + (size_t)idx * 32);
            if (d < bestDist) { bestDist = d; bestIdx = idx; }
        }
        if (fuse_dist) fuse_dist[j] = bestDist;
        if (bestDist <= TH_LOW) { fuse_idx[j] = bestIdx; nFused++; }
    }
    return nFused;
}

}  // namespace orc

extern "C" {
int orc_is_in_frustum_points(const planar_frame_view* F, float lsf, int n_levels, const int32_t* n, int stride, const uint8_t* valid, const float* xw,
                             const float* normal, const float* min_dist, const float* max_dist, float limit, uint8_t* in_view, float* px, float* py,
                             float* pxr, int32_t* level, float* vc) {
    for (int b = 0; b < F->B; b++) {
        const size_t o = (size_t)b * stride;
        orc::is_in_frustum_points(*F, b, lsf, n_levels, n[b], valid + o, xw + o * 3, normal + o * 3, min_dist + o, max_dist + o, limit, in_view + o, px + o,
                                  py + o, pxr + o, level + o, vc + o);
    }
    return 0;
}
int orc_is_in_frustum_lines(const planar_frame_view* F, float lsf, const int32_t* n, int stride, const uint8_t* valid, const double* xw6,
                            const double* normal, const float* min_dist, const float* max_dist, float limit, uint8_t* in_view, float* proj,
                            int32_t* level, float* vc) {
    for (int b = 0; b < F->B; b++) {
        const size_t o = (size_t)b * stride;
        orc::is_in_frustum_lines(*F, b, lsf, n[b], valid + o, xw6 + o * 6, normal + o * 3, min_dist + o, max_dist + o, limit, in_view + o, proj + o * 4,
                                 level + o, vc + o);
    }
    return 0;
}
int orc_fuse_search(const planar_frame_view* K, const float* inv_sigma2, float lsf, int n_levels, const int32_t* n, int stride, int points_shared,
                    const uint8_t* usable, const float* xw, const float* normal, const float* min_dist, const float* max_dist, const uint8_t* desc, float th,
                    int32_t* fuse_idx, int32_t* fuse_dist, int32_t* n_fused) {
    for (int b = 0; b < K->B; b++) {
        const size_t o = points_shared ? 0 : (size_t)b * stride, oo = (size_t)b * stride;
        n_fused[b] = orc::fuse_search(*K, b, inv_sigma2, lsf, n_levels, n[points_shared ? 0 : b], usable + o, xw + o * 3, normal + o * 3, min_dist + o, max_dist + o,
                                      desc + o * 32, th, fuse_idx + oo, fuse_dist ? fuse_dist + oo : nullptr);
    }
    return 0;
}
int orc_lsd_fuse_search(const planar_frame_view* K, float lsf, int n_levels, const int32_t* n_lines, int line_stride, const planar_keyline* keylines,
                        const uint8_t* ldesc, const int32_t* n_ml, int ml_stride, int lines_shared, const uint8_t* usable, const double* xw6, const double* normal,
                        const float* min_dist, const float* max_dist, const uint8_t* ml_desc, float th, int32_t* fuse_idx, int32_t* fuse_dist, int32_t* n_fused) {
    for (int b = 0; b < K->B; b++) {
        const size_t lo = (size_t)b * line_stride, o = lines_shared ? 0 : (size_t)b * ml_stride, oo = (size_t)b * ml_stride;
        n_fused[b] = orc::lsd_fuse_search(*K, b, lsf, n_levels, n_lines[b], keylines + lo, ldesc + lo * 32, n_ml[lines_shared ? 0 : b], usable + o, xw6 + o * 6,
                                          normal + o * 3, min_dist + o, max_dist + o, ml_desc + o * 32, th, fuse_idx + oo, fuse_dist ? fuse_dist + oo : nullptr);
    }
    return 0;
}
int orc_search_by_projection_frame(const planar_frame_view* cur, const planar_last_frame_view* last, float th, int mono,
                                   int check_orientation, int32_t* cur_match, int32_t* nmatches) {
    for (int b = 0; b < cur->B; b++)
        nmatches[b] = orc::search_by_projection_frame(*cur, *last, b, th, mono != 0, check_orientation != 0, cur_match + (size_t)b * cur->stride);
    return 0;
}
int orc_search_by_projection_map(const planar_frame_view* F, const planar_map_probes* mp, float th, float nn_ratio, int32_t* match,
                                 int32_t* nmatches) {
    for (int b = 0; b < F->B; b++) nmatches[b] = orc::search_by_projection_map(*F, *mp, b, th, nn_ratio, match + (size_t)b * F->stride);
    return 0;
}
int orc_search_by_bow(int B, const int32_t* n_kf, int kf_stride, const int32_t* kf_node, const uint8_t* kf_usable, const float* kf_angle,
                      const uint8_t* kf_desc, const int32_t* n_f, int f_stride, const int32_t* f_node, const float* f_angle,
                      const uint8_t* f_desc, float nn_ratio, int check_orientation, int32_t* match, int32_t* nmatches) {
    for (int b = 0; b < B; b++) {
        const size_t ko = (size_t)b * kf_stride, fo = (size_t)b * f_stride;
        nmatches[b] = orc::search_by_bow(n_kf[b], kf_node + ko, kf_usable + ko, kf_angle + ko, kf_desc + ko * 32, n_f[b], f_node + fo,
                                         f_angle + fo, f_desc + fo * 32, nn_ratio, check_orientation != 0, match + fo);
    }
    return 0;
}
int orc_lsd_search_by_projection(int B, const int32_t* n_lines, int line_stride, const planar_keyline* keylines, const uint8_t* ldesc,
                                 const uint8_t* blocked, const int32_t* n_ml, int ml_stride, const uint8_t* ml_in_view, const float* ml_proj,
                                 const int32_t* ml_level, const float* ml_view_cos, const uint8_t* ml_desc, const uint8_t* ml_observed,
                                 const float* scale_factors, int n_levels, float th, float nn_ratio, int32_t* match, int32_t* nmatches) {
    (void)n_levels;
    for (int b = 0; b < B; b++) {
        const size_t lo = (size_t)b * line_stride, mo = (size_t)b * ml_stride;
        nmatches[b] = orc::lsd_search_by_projection(n_lines[b], keylines + lo, ldesc + lo * 32, blocked ? blocked + lo : nullptr, n_ml[b],
                                                    ml_in_view + mo, ml_proj + mo * 4, ml_level + mo, ml_view_cos + mo, ml_desc + mo * 32,
                                                    ml_observed + mo, scale_factors, th, nn_ratio, match + lo);
    }
    return 0;
}
int orc_plane_search_by_coefficients(int B, const int32_t* n_planes, int pl_stride, const float* pl_coef, const float* Tcw, int map_shared,
                                     const int32_t* n_mp, int mp_stride, const uint8_t* mp_valid, const float* mp_coef, const int32_t* mp_npts,
                                     int pts_stride, const float* mp_pts, const float* th, int32_t* match, int32_t* ver, int32_t* par,
                                     int32_t* nmatches) {
    for (int b = 0; b < B; b++) {
        const int m = map_shared ? 0 : b;
        const size_t po = (size_t)b * pl_stride, mo = (size_t)m * mp_stride;
        nmatches[b] = orc::plane_search_by_coefficients(n_planes[b], pl_coef + po * 4, Tcw + (size_t)b * 16, n_mp[m], mp_valid + mo, mp_coef + mo * 4,
                                                        mp_npts + mo, pts_stride, mp_pts + mo * pts_stride * 3, th, match + po, ver + po, par + po);
    }
    return 0;
}
}

// ---- Frame::ComputeStereoFromRGBD (src/Frame.cc:603-621) + Frame::UnprojectStereo (:623-634) -----------------------------------------
// depth: the u16 image; imDepth = convertTo(CV_32F, factor) (src/Tracking.cc:174-175: cvtScale16u32f, float scale, float multiply).
// Pinned through oracle/_ref/ref_frame "stereo" (the two function bodies extracted from src/Frame.cc); tests/test_oracle_frame_ref.py.
namespace orc {
void stereo_from_rgbd(const planar_keypoint* keys, const planar_keypoint* keys_un, int n, const uint16_t* depth, int pitch_px, float factor,
                      float fx, float fy, float cx, float cy, float bf, const float* Tcw, float* u_right, float* z_out, float* xw, uint8_t* valid) {
    const FrustumPose P = frustum_pose(Tcw);
    const float invfx = 1.0f / fx, invfy = 1.0f / fy;        // src/Frame.cc:127-128
    for (int i = 0; i < n; i++) {
        const int v = (int)keys[i].y, u = (int)keys[i].x;      // imDepth.at<float>(v, u): float coordinates truncate
        const float d = (float)depth[(size_t)v * pitch_px + u] * factor;
        u_right[i] = -1.f; z_out[i] = -1.f; valid[i] = 0;
        xw[3 * i] = xw[3 * i + 1] = xw[3 * i + 2] = 0.f;
        if (d > 0) {
            z_out[i] = d;
            u_right[i] = keys_un[i].x - bf / d;
            const float z = d, x = (keys_un[i].x - cx) * z * invfx, y = (keys_un[i].y - cy) * z * invfy;
            // mRwc * x3Dc + mOw: mRwc = mRcw.t() materialised, plain 3x3 by 3x1 product = small-matrix gemm (float sums), + C in double
            const float xc[3] = {x, y, z};
            for (int r = 0; r < 3; r++) {
                float t = P.Rcw[r] * xc[0];                    // Rwc(r, k) = Rcw(k, r)
                t = t + P.Rcw[3 + r] * xc[1];
                t = t + P.Rcw[6 + r] * xc[2];
                xw[3 * i + r] = (float)((double)t * 1.0 + (double)P.Ow[r] * 1.0);
            }
            valid[i] = 1;
        }
    }
}
}  // namespace orc
// Frame::UndistortKeyPoints (src/Frame.cc:545-573) = cv::undistortPoints(mat, mat, mK, mDistCoef, cv::Mat(), mK).  PARITY UNPINNED: OpenCV is not in this image and
// the reference holds no test vector for it; this is the published loop of imgproc/src/undistort.cpp (3.4.x cvUndistortPointsInternal, default criteria = 5 iterations,
// no tilt: invMatTilt = I), kept in the library's literal form (k[14] with zeros past the five given, RR = P * I through a 3x3 double product) so that the device's
// simplified form is checked against every term the library evaluates.
extern "C" int orc_undistort_keypoints(const planar_keypoint* keys, int n, float fx_, float fy_, float cx_, float cy_, const float* dist5, planar_keypoint* keys_un) {
    for (int i = 0; i < n; i++) keys_un[i] = keys[i];
    if (dist5[0] == 0.0f) return 0;                                             // :546-549
    double A[3][3] = {{(double)fx_, 0, (double)cx_}, {0, (double)fy_, (double)cy_}, {0, 0, 1}}, RR[3][3], PP[3][3], k[14] = {0};
    for (int j = 0; j < 5; j++) k[j] = (double)dist5[j];
    double I3[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) PP[r][c] = A[r][c];
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) { double s = 0; for (int q = 0; q < 3; q++) s += PP[r][q] * I3[q][c]; RR[r][c] = s; }
    const double fx = A[0][0], fy = A[1][1], ifx = 1. / fx, ify = 1. / fy, cx = A[0][2], cy = A[1][2];
    for (int i = 0; i < n; i++) {
        double x = keys[i].x, y = keys[i].y, x0, y0;
        x = (x - cx) * ifx; y = (y - cy) * ify;
        const double vz = 0 * x + 0 * y + 1, invProj = vz ? 1. / vz : 1;          // invMatTilt * (x, y, 1)
        x0 = x = invProj * (1 * x + 0 * y + 0); y0 = y = invProj * (0 * x0 + 1 * y + 0);
        for (int j = 0; j < 5; j++) {
            const double r2 = x * x + y * y;
            const double icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
            const double deltaX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x) + k[8] * r2 + k[9] * r2 * r2;
            const double deltaY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y + k[10] * r2 + k[11] * r2 * r2;
            x = (x0 - deltaX) * icdist; y = (y0 - deltaY) * icdist;
        }
        const double xx = RR[0][0] * x + RR[0][1] * y + RR[0][2], yy = RR[1][0] * x + RR[1][1] * y + RR[1][2], ww = 1. / (RR[2][0] * x + RR[2][1] * y + RR[2][2]);
        keys_un[i].x = (float)(xx * ww); keys_un[i].y = (float)(yy * ww);
    }
    return 0;
}
extern "C" int orc_stereo_from_rgbd(const planar_keypoint* keys, const planar_keypoint* keys_un, int n, const uint16_t* depth, int pitch_px, float factor,
                                    float fx, float fy, float cx, float cy, float bf, const float* Tcw, float* u_right, float* z_out, float* xw, uint8_t* valid) {
    orc::stereo_from_rgbd(keys, keys_un, n, depth, pitch_px, factor, fx, fy, cx, cy, bf, Tcw, u_right, z_out, xw, valid);
    return 0;
}

// ---- MapPoint::UpdateNormalAndDepth (reference src/MapPoint.cc:347-388) with KeyFrame::SetPose's camera centre (src/KeyFrame.cc:79-86) ----
// Ow = -Rwc * tcw with Rwc = Rcw.t() materialised: cv::gemm's float small-matrix path (products summed in float, left to right), negated.
extern "C" void orc_keyframe_center(const float* Tcw, float* Ow) {
    for (int r = 0; r < 3; r++) {
        float t = Tcw[0 * 4 + r] * Tcw[3];
        t = t + Tcw[1 * 4 + r] * Tcw[7];
        t = t + Tcw[2 * 4 + r] * Tcw[11];
        Ow[r] = (float)((double)t * -1.0);
    }
}
// pos [3]; obs_ow [nobs][3]: camera centres of the observing key frames in observation-map order; ref_ow: mpRefKF's; level: octave of the point's keypoint there.
// normal: sum of normali * (float)(1 / |normali|) (cv::norm accumulates in double; Mat / double is a convertTo with the scale cast to float; the running sum is
// alpha * A + beta * C in float), times (float)(1.0 / n); dist = (float)|Pos - Ow_ref|; max = dist * sf[level]; min = max / sf[nlev - 1].
extern "C" void orc_update_normal_and_depth(const float* pos, const float* obs_ow, int nobs, const float* ref_ow, int level, const float* sf, int nlev, float* normal,
                                            float* min_d, float* max_d) {
    if (nobs <= 0) return;
    float nrm[3] = {0.f, 0.f, 0.f};
    for (int i = 0; i < nobs; i++) {
        float d[3];
        double s = 0;
        for (int k = 0; k < 3; k++) { d[k] = pos[k] - obs_ow[i * 3 + k]; s += (double)d[k] * (double)d[k]; }
        const float fa = (float)(1.0 / std::sqrt(s));
        for (int k = 0; k < 3; k++) nrm[k] = d[k] * fa + nrm[k] * 1.0f;
    }
    double s = 0;
    for (int k = 0; k < 3; k++) { const float pc = pos[k] - ref_ow[k]; s += (double)pc * (double)pc; }
    const float dist = (float)std::sqrt(s);
    *max_d = dist * sf[level];
    *min_d = *max_d / sf[nlev - 1];
    const float fn = (float)(1.0 / (double)nobs);
    for (int k = 0; k < 3; k++) normal[k] = nrm[k] * fn;
}
