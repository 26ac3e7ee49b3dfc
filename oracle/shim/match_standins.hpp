// oracle/shim/match_standins.hpp — TEST INFRASTRUCTURE.
// Data-holder stand-ins for the reference's map classes so that the REAL matcher translation units
// (src/ORBmatcher.cc, src/PlaneMatcher.cpp) compile where they lie and run in oracle/_ref/ref_match.  Force-included
// (-include) with the real headers' include guards pre-defined, so include/ORBmatcher.h and include/PlaneMatcher.h are
// the reference's own.  Only members those two files touch exist; methods return what the harness stored.
// Frame::AssignFeaturesToGrid / lineDescriptorMAD / GetFeaturesInArea / GetLinesInArea / PosInGrid / ComputePlaneWorldCoeff live in
// src/Frame.cc, which cannot be built as a whole here (PCL, threads, the extractors).  With -DSTANDINS_REAL_FRAME_FUNCS (how
// oracle/Makefile builds ref_match) they are only DECLARED here and their bodies are the reference's own: lines 155-168, 269-293,
// 440-535, 815-820 of src/Frame.cc extracted at build time into oracle/_ref/gen/frame_extract_match.cpp, together with src/MapPoint.cc:390-434
// (distance invariance, PredictScale), src/MapLine.cpp:369-390 (the same for lines) and src/KeyFrame.cc:79-93, 107-111, 120-130, 639-678, 680-713,
// 715-718 (pose getters, GetFeaturesInArea, GetLinesInArea, IsInImage) for ORBmatcher::Fuse / LSDmatcher::Fuse.  Without the macro the restated bodies / stubs below are used.  KeyFrame's line helpers (src/KeyFrame.cc, same text) stay restated.
#pragma once
#define MAPPOINT_H
#define KEYFRAME_H
#define FRAME_H
#define MAPPLANE_H
#define ORB_SLAM2_MAPLINE_H
#define MAP_H
#include <cmath>
#include <map>
#include <mutex>
#include <set>
#include <vector>

#include "cvshim.hpp"
#include "eigenshim.hpp"
#include "opencv2/line_descriptor/descriptor.hpp"
#include "pcl/point_types.h"
#ifndef STANDINS_NO_REFERENCE
#include "auxiliar.h"   // the reference's own (sort comparators, Vector6d)
#include "Thirdparty/DBoW2/DBoW2/BowVector.h"
#include "Thirdparty/DBoW2/DBoW2/FeatureVector.h"
#else
// -DSTANDINS_NO_REFERENCE: the GPU box has no /root/reference; tests/adapter_shim builds the harness against include/planar_adapters.hpp
// there, which needs only the data members below (the Frame-side search functions run on the device).
typedef Eigen::Matrix<double, 6, 1> Vector6d;
namespace DBoW2 {
typedef unsigned int NodeId;
class FeatureVector : public std::map<NodeId, std::vector<unsigned int>> {
public:
    void addFeature(NodeId id, unsigned int i_feature) { (*this)[id].push_back(i_feature); }
};
}  // namespace DBoW2
#endif

#define FRAME_GRID_ROWS 48
#define FRAME_GRID_COLS 64

namespace Planar_SLAM {
using namespace std;
class KeyFrame;
class Frame;
class Map;
class MapPlane;
class MapLine;

// a std::mutex that does not stop the holder classes from living in std::vector (the extracted reference bodies lock mMutexPos / mMutexPose)
struct CopyableMutex : std::mutex {
    CopyableMutex() {}
    CopyableMutex(const CopyableMutex&) {}
    CopyableMutex& operator=(const CopyableMutex&) { return *this; }
};
// ORBmatcher::Fuse does not return what it matched: the harness reads it off the calls it makes (MapPoint::GetDescriptor of the point being searched,
// then KeyFrame::GetMapPoint(bestIdx) exactly once per fused point)
inline int& fuse_current() { static int v = -1; return v; }
inline std::vector<std::pair<int, int>>& fuse_log() { static std::vector<std::pair<int, int>> v; return v; }

class MapPoint {
public:
    cv::Mat GetWorldPos() { return pos.clone(); }
    cv::Mat GetNormal() { return normal.clone(); }
    cv::Mat GetDescriptor() { fuse_current() = index; return desc.clone(); }
    bool isBad() { return bad; }
    int Observations() { return nobs; }
#ifdef STANDINS_REAL_FRAME_FUNCS
    // bodies: src/MapPoint.cc:390-434, extracted at build time
    float GetMinDistanceInvariance();
    float GetMaxDistanceInvariance();
    int PredictScale(const float& currentDist, KeyFrame* pKF);
    int PredictScale(const float& currentDist, Frame* pF);
    float mfMinDistance = 0, mfMaxDistance = 0;
    CopyableMutex mMutexPos;
#else
    float GetMinDistanceInvariance() { return 0.f; }
    float GetMaxDistanceInvariance() { return 1e9f; }
    int PredictScale(const float&, KeyFrame*) { return 0; }
    int PredictScale(const float&, Frame*) { return 0; }
#endif
    bool IsInKeyFrame(KeyFrame*) { return in_kf; }
    bool in_kf = false;
#ifndef STANDINS_REAL_FRAME_FUNCS
    float mfMinDistance = 0, mfMaxDistance = 0;
#endif
    // the one accessor the Fuse adapter needs and include/MapPoint.h lacks: the UNSCALED invariance distances
    void GetDistanceRange(float& mn, float& mx) { mn = mfMinDistance; mx = mfMaxDistance; }
    int fuse_idx = -1, kf_slot = -1;      // harness: which key-frame slot Fuse paired this point with, read off the edits it makes
    int GetIndexInKeyFrame(KeyFrame*) { return -1; }
    void AddObservation(KeyFrame*, size_t i) { fuse_idx = (int)i; }
    void Replace(MapPoint* o) { if (kf_slot >= 0) o->fuse_idx = kf_slot; else if (o->kf_slot >= 0) fuse_idx = o->kf_slot; }
    float mTrackProjX = 0, mTrackProjY = 0, mTrackProjXR = 0;
    bool mbTrackInView = false;
    int mnTrackScaleLevel = 0;
    float mTrackViewCos = 0;
    long unsigned int mnFuseCandidateForKF = 0, mnId = 0;
    // harness data
    cv::Mat pos, normal, desc;
    bool bad = false;
    int nobs = 0;
    int index = -1;
};

class MapLine {
public:
    Vector6d GetWorldPos() { return mWorldPos; }
    Eigen::Vector3d GetNormal() { return normal; }
    cv::Mat GetDescriptor() { fuse_current() = index; return mLDescriptor.clone(); }
    bool isBad() { return bad; }
    int Observations() { return nobs; }
#ifdef STANDINS_REAL_FRAME_FUNCS
    // bodies: src/MapLine.cpp:369-390, extracted at build time
    float GetMinDistanceInvariance();
    float GetMaxDistanceInvariance();
    int PredictScale(const float& currentDist, const float& logScaleFactor);
    float mfMinDistance = 0, mfMaxDistance = 0;
    CopyableMutex mMutexPos;
#else
    float GetMinDistanceInvariance() { return 0.f; }
    float GetMaxDistanceInvariance() { return 1e9f; }
    int PredictScale(const float&, const float&) { return 0; }
    float mfMinDistance = 0, mfMaxDistance = 0;
#endif
    void GetDistanceRange(float& mn, float& mx) { mn = mfMinDistance; mx = mfMaxDistance; }
    int fuse_idx = -1, kf_slot = -1;
    int PredictScale(const float&, KeyFrame*) { return 0; }
    int PredictScale(const float&, Frame*) { return 0; }
    bool IsInKeyFrame(KeyFrame*) { return false; }
    int GetIndexInKeyFrame(KeyFrame*) { return -1; }
    void AddObservation(KeyFrame*, size_t i) { fuse_idx = (int)i; }
    void Replace(MapLine* o) { if (kf_slot >= 0) o->fuse_idx = kf_slot; else if (o->kf_slot >= 0) fuse_idx = o->kf_slot; }
    float mTrackProjX1 = 0, mTrackProjY1 = 0, mTrackProjX2 = 0, mTrackProjY2 = 0;
    bool mbTrackInView = false;
    int mnTrackScaleLevel = 0;
    float mTrackViewCos = 0;
    long unsigned int mnFuseCandidateForKF = 0, mnId = 0;
    cv::Mat mLDescriptor;
    Vector6d mWorldPos;
    Eigen::Vector3d normal;
    bool bad = false;
    int nobs = 0;
    int index = -1;
};

#ifndef STANDINS_NO_REFERENCE
// restated: src/Frame.cc:269-293 (also KeyFrame::lineDescriptorMAD, same body)
static inline void line_descriptor_mad(vector<vector<cv::DMatch>> line_matches, double& nn_mad, double& nn12_mad) {
    vector<vector<cv::DMatch>> matches_nn = line_matches, matches_12 = line_matches;
    sort(matches_nn.begin(), matches_nn.end(), compare_descriptor_by_NN_dist());
    const double nn_dist_median = matches_nn[int(matches_nn.size() / 2)][0].distance;
    for (unsigned int i = 0; i < matches_nn.size(); i++) matches_nn[i][0].distance = fabsf(matches_nn[i][0].distance - nn_dist_median);
    sort(matches_nn.begin(), matches_nn.end(), compare_descriptor_by_NN_dist());
    nn_mad = 1.4826 * matches_nn[int(matches_nn.size() / 2)][0].distance;
    sort(matches_12.begin(), matches_12.end(), conpare_descriptor_by_NN12_dist());
    const double nn12_dist_median = matches_12[int(matches_12.size() / 2)][1].distance - matches_12[int(matches_12.size() / 2)][0].distance;
    for (unsigned int j = 0; j < matches_12.size(); j++)
        matches_12[j][0].distance = fabsf(matches_12[j][1].distance - matches_12[j][0].distance - nn12_dist_median);
    sort(matches_12.begin(), matches_12.end(), compare_descriptor_by_NN_dist());
    nn12_mad = 1.4826 * matches_12[int(matches_12.size() / 2)][0].distance;
}

// restated: src/Frame.cc:491-524 (KeyFrame::GetLinesInArea has the same body)
static inline vector<size_t> lines_in_area(const vector<cv::line_descriptor::KeyLine>& vkl, const float& x1, const float& y1, const float& x2, const float& y2,
                                           const float& r, const int minLevel, const int maxLevel) {
    vector<size_t> vIndices;
    const bool bCheckLevels = (minLevel > 0) || (maxLevel > 0);
    for (size_t i = 0; i < vkl.size(); i++) {
        cv::line_descriptor::KeyLine keyline = vkl[i];
        float distance = (0.5 * (x1 + x2) - keyline.pt.x) * (0.5 * (x1 + x2) - keyline.pt.x) + (0.5 * (y1 + y2) - keyline.pt.y) * (0.5 * (y1 + y2) - keyline.pt.y);
        if (distance > r * r) continue;
        float slope = (y1 - y2) / (x1 - x2) - keyline.angle;
        if (slope > r * 0.01) continue;
        if (bCheckLevels) {
            if (keyline.octave < minLevel) continue;
            if (maxLevel >= 0 && keyline.octave > maxLevel) continue;
        }
        vIndices.push_back(i);
    }
    return vIndices;
}

#endif   // STANDINS_NO_REFERENCE

class MapPlane {
public:
    typedef pcl::PointXYZRGB PointT;
    typedef pcl::PointCloud<PointT> PointCloud;
    cv::Mat GetWorldPos() { return pos.clone(); }
    bool isBad() { return bad; }
    PointCloud::Ptr mvPlanePoints;
    cv::Mat pos;
    bool bad = false;
    int index = -1;
};

class Frame {
public:
    int N = 0;
    std::vector<cv::KeyPoint> mvKeys, mvKeysUn;
    std::vector<float> mvuRight;
    cv::Mat mDescriptors;
    std::vector<MapPoint*> mvpMapPoints;
    std::vector<bool> mvbOutlier;
    cv::Mat mTcw;
    static float fx, fy, cx, cy;
    float mbf = 0, mb = 0;
    static float mnMinX, mnMaxX, mnMinY, mnMaxY;
    static float mfGridElementWidthInv, mfGridElementHeightInv;
    std::vector<float> mvScaleFactors;
    float mfLogScaleFactor = 0;
    int mnScaleLevels = 0;
    DBoW2::FeatureVector mFeatVec;
    std::vector<std::size_t> mGrid[FRAME_GRID_COLS][FRAME_GRID_ROWS];
    // lines
    int NL = 0;
    std::vector<cv::line_descriptor::KeyLine> mvKeylinesUn;
    cv::Mat mLdesc;
    std::vector<MapLine*> mvpMapLines;
    std::vector<bool> mvbLineOutlier;
#ifdef STANDINS_REAL_FRAME_FUNCS
    vector<size_t> GetLinesInArea(const float& x1, const float& y1, const float& x2, const float& y2, const float& r, const int minLevel = -1,
                                  const int maxLevel = -1) const;
    void lineDescriptorMAD(vector<vector<cv::DMatch>> line_matches, double& nn_mad, double& nn12_mad) const;
#elif !defined(STANDINS_NO_REFERENCE)
    vector<size_t> GetLinesInArea(const float& x1, const float& y1, const float& x2, const float& y2, const float& r, const int minLevel = -1,
                                  const int maxLevel = -1) const { return lines_in_area(mvKeylinesUn, x1, y1, x2, y2, r, minLevel, maxLevel); }
    void lineDescriptorMAD(vector<vector<cv::DMatch>> m, double& a, double& b) const { line_descriptor_mad(m, a, b); }
#endif
    // pose optimisation (only the adapter harness, tests/adapter_shim, uses these)
    std::vector<float> mvInvLevelSigma2;
    std::vector<Eigen::Vector3d> mvKeyLineFunctions;
    std::vector<bool> mvbPlaneOutlier, mvbParPlaneOutlier, mvbVerPlaneOutlier;
    void SetPose(cv::Mat Tcw) { mTcw = Tcw.clone(); }
    // planes
    std::vector<cv::Mat> mvPlaneCoefficients;
    std::vector<MapPlane*> mvpMapPlanes, mvpParallelPlanes, mvpVerticalPlanes;
    int mnPlaneNum = 0;
    bool mbNewPlane = false;

    cv::Mat mOw;   // read (unused result) by ComputePlaneWorldCoeff
#ifdef STANDINS_REAL_FRAME_FUNCS
    bool PosInGrid(const cv::KeyPoint& kp, int& posX, int& posY);
    void AssignFeaturesToGrid();
    vector<size_t> GetFeaturesInArea(const float& x, const float& y, const float& r, const int minLevel = -1, const int maxLevel = -1) const;
    cv::Mat ComputePlaneWorldCoeff(const int& idx);
#else
    // restated: src/Frame.cc:526-535, 155-166
    bool PosInGrid(const cv::KeyPoint& kp, int& posX, int& posY) {
        posX = round((kp.pt.x - mnMinX) * mfGridElementWidthInv);
        posY = round((kp.pt.y - mnMinY) * mfGridElementHeightInv);
        if (posX < 0 || posX >= FRAME_GRID_COLS || posY < 0 || posY >= FRAME_GRID_ROWS) return false;
        return true;
    }
    void AssignFeaturesToGrid() {
        for (int i = 0; i < N; i++) {
            int gx, gy;
            if (PosInGrid(mvKeysUn[i], gx, gy)) mGrid[gx][gy].push_back(i);
        }
    }
    // restated: src/Frame.cc:440-489
    vector<size_t> GetFeaturesInArea(const float& x, const float& y, const float& r, const int minLevel = -1, const int maxLevel = -1) const {
        vector<size_t> vIndices;
        vIndices.reserve(N);
        const int nMinCellX = max(0, (int)floor((x - mnMinX - r) * mfGridElementWidthInv));
        if (nMinCellX >= FRAME_GRID_COLS) return vIndices;
        const int nMaxCellX = min((int)FRAME_GRID_COLS - 1, (int)ceil((x - mnMinX + r) * mfGridElementWidthInv));
        if (nMaxCellX < 0) return vIndices;
        const int nMinCellY = max(0, (int)floor((y - mnMinY - r) * mfGridElementHeightInv));
        if (nMinCellY >= FRAME_GRID_ROWS) return vIndices;
        const int nMaxCellY = min((int)FRAME_GRID_ROWS - 1, (int)ceil((y - mnMinY + r) * mfGridElementHeightInv));
        if (nMaxCellY < 0) return vIndices;
        const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
        for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
            for (int iy = nMinCellY; iy <= nMaxCellY; iy++) {
                const vector<size_t> vCell = mGrid[ix][iy];
                for (size_t j = 0, jend = vCell.size(); j < jend; j++) {
                    const cv::KeyPoint& kpUn = mvKeysUn[vCell[j]];
                    if (bCheckLevels) {
                        if (kpUn.octave < minLevel) continue;
                        if (maxLevel >= 0) if (kpUn.octave > maxLevel) continue;
                    }
                    const float distx = kpUn.pt.x - x, disty = kpUn.pt.y - y;
                    if (fabs(distx) < r && fabs(disty) < r) vIndices.push_back(vCell[j]);
                }
            }
        return vIndices;
    }
    // restated: src/Frame.cc:815-820
    cv::Mat ComputePlaneWorldCoeff(const int& idx) {
        cv::Mat temp;
        cv::transpose(mTcw, temp);
        return temp * mvPlaneCoefficients[idx];
    }
#endif
};

class KeyFrame {
public:
    int N = 0;
    std::vector<cv::KeyPoint> mvKeys, mvKeysUn;
    std::vector<float> mvuRight, mvScaleFactors, mvLevelSigma2, mvInvLevelSigma2;
    cv::Mat mDescriptors;
    DBoW2::FeatureVector mFeatVec;
    float fx = 0, fy = 0, cx = 0, cy = 0, mbf = 0, mb = 0;
    long unsigned int mnId = 0;
    std::vector<MapPoint*> mps;
    // lines
    cv::Mat mLineDescriptors;
    std::vector<cv::line_descriptor::KeyLine> mvKeyLines;
    std::vector<MapLine*> mls;
    float mnMinX = 0, mnMaxX = 0, mnMinY = 0, mnMaxY = 0, mfLogScaleFactor = 0;
    std::vector<MapLine*> GetMapLineMatches() { return mls; }
    std::set<MapLine*> GetMapLines() { return std::set<MapLine*>(); }
    MapLine* GetMapLine(const size_t& i) { fuse_log().push_back(std::make_pair(fuse_current(), (int)i)); return mls[i]; }
    void AddMapLine(MapLine*, const size_t&) {}
#ifdef STANDINS_REAL_FRAME_FUNCS
    // body: src/KeyFrame.cc:680-713, extracted at build time
    vector<size_t> GetLinesInArea(const float& x1, const float& y1, const float& x2, const float& y2, const float& r, const int minLevel = -1,
                                  const int maxLevel = -1) const;
    void lineDescriptorMAD(vector<vector<cv::DMatch>> m, double& a, double& b) const { line_descriptor_mad(m, a, b); }
#elif !defined(STANDINS_NO_REFERENCE)
    vector<size_t> GetLinesInArea(const float& x1, const float& y1, const float& x2, const float& y2, const float& r, const int minLevel = -1,
                                  const int maxLevel = -1) const { return lines_in_area(mvKeyLines, x1, y1, x2, y2, r, minLevel, maxLevel); }
    void lineDescriptorMAD(vector<vector<cv::DMatch>> m, double& a, double& b) const { line_descriptor_mad(m, a, b); }
#endif
    std::vector<MapPoint*> GetMapPointMatches() { return mps; }
    std::set<MapPoint*> GetMapPoints() { return std::set<MapPoint*>(); }
    MapPoint* GetMapPoint(const size_t& i) { fuse_log().push_back(std::make_pair(fuse_current(), (int)i)); return mps[i]; }
    void AddMapPoint(MapPoint*, const size_t&) {}
    int mnScaleLevels = 0;
    cv::Mat GetPose() { return Tcw.clone(); }     // src/KeyFrame.cc:95-99
    cv::Mat Tcw;
    float mfGridElementWidthInv = 0, mfGridElementHeightInv = 0;
#ifdef STANDINS_REAL_FRAME_FUNCS
    // bodies: src/KeyFrame.cc:79-93 (SetPose), 107-111 (GetCameraCenter), 120-130 (GetRotation, GetTranslation), 639-678 (GetFeaturesInArea), 715-718
    // (IsInImage), extracted at build time; the members they touch (include/KeyFrame.h)
    void SetPose(const cv::Mat& Tcw);
    cv::Mat GetRotation();
    cv::Mat GetTranslation();
    cv::Mat GetCameraCenter();
    bool IsInImage(const float& x, const float& y) const;
    vector<size_t> GetFeaturesInArea(const float& x, const float& y, const float& r) const;
    cv::Mat Twc, Ow, Cw;
    float mHalfBaseline = 0;
    CopyableMutex mMutexPose;
    int mnGridCols = FRAME_GRID_COLS, mnGridRows = FRAME_GRID_ROWS;
    std::vector<std::vector<std::vector<size_t>>> mGrid;
#else
    cv::Mat GetRotation() { return cv::Mat(); }
    cv::Mat GetTranslation() { return cv::Mat(); }
    cv::Mat GetCameraCenter() { return cv::Mat(); }
    bool IsInImage(const float&, const float&) const { return true; }
    vector<size_t> GetFeaturesInArea(const float&, const float&, const float&) const { return vector<size_t>(); }
#endif
};

}  // namespace Planar_SLAM
