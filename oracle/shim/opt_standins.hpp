// oracle/shim/opt_standins.hpp — TEST INFRASTRUCTURE.
// Data-holder stand-ins for the reference's Frame / KeyFrame / MapPoint / MapLine / MapPlane / Map / LoopClosing / Config so
// that the REAL src/Optimizer.cc and src/Converter.cc (with the real vendored g2o, g2oAddition/*.h and include/EdgeLine.h)
// compile where they lie and run in oracle/_ref/ref_opt.  Force-included (-include) with the real headers' include
// guards pre-defined, so include/Optimizer.h and include/Converter.h are the reference's own.  Only the members
// Optimizer.cc touches exist; they carry the real declarations' names and types (include/Frame.h, KeyFrame.h,
// MapPoint.h, MapLine.h, MapPlane.h, Map.h) and return what the harness stored.  Nothing here computes.
#pragma once
#define MAPPOINT_H
#define KEYFRAME_H
#define FRAME_H
#define MAPPLANE_H
#define ORB_SLAM2_MAPLINE_H
#define MAP_H
#define LOOPCLOSING_H
#define CONFIG_H
#include <cmath>
#include <list>
#include <map>
#include <mutex>
#include <set>
#include <string>
#include <vector>

#include "cvshim.hpp"
#include <Eigen/Core>
#include <Eigen/Geometry>
#include "opencv2/line_descriptor/descriptor.hpp"
#include "pcl/point_types.h"
#ifndef STANDINS_NO_REFERENCE   // -DSTANDINS_NO_REFERENCE: the GPU box has no /root/reference; tests/adapter_shim builds the BA harness against include/planar_adapters.hpp there
#include "Thirdparty/g2o/g2o/types/types_seven_dof_expmap.h"
#endif

namespace Planar_SLAM {
using namespace std;
typedef Eigen::Matrix<double, 6, 1> Vector6d;   // include/auxiliar.h:44
class KeyFrame;
class Frame;
class Map;
class MapPlane;
class MapLine;
class MapPoint;

// include/Config.h: values of Examples/RGB-D/TUM3.yaml:103-110 unless the harness overrides them
class Config {
public:
    static std::map<std::string, double>& table() {
        static std::map<std::string, double> t = {{"Plane.AngleInfo", 0.5}, {"Plane.DistanceInfo", 50}, {"Plane.Chi", 100},
                                                  {"Plane.VPChi", 50}, {"Plane.ParallelInfo", 0.1}, {"Plane.VerticalInfo", 0.1}};
        return t;
    }
    template <typename T> static T Get(const std::string& key) { return T(table().at(key)); }
};

#ifndef STANDINS_NO_REFERENCE
class LoopClosing {
public:
    typedef map<KeyFrame*, g2o::Sim3, std::less<KeyFrame*>, Eigen::aligned_allocator<std::pair<KeyFrame* const, g2o::Sim3>>> KeyFrameAndPose;
};
#endif

class MapPoint {
public:
    cv::Mat GetWorldPos() { return mWorldPos.clone(); }
    void SetWorldPos(const cv::Mat& Pos) { Pos.copyTo(mWorldPos); }
    bool isBad() { return bad; }
    std::map<KeyFrame*, size_t> GetObservations() { return mObservations; }
    KeyFrame* GetReferenceKeyFrame() { return nullptr; }
    int GetIndexInKeyFrame(KeyFrame* pKF) { return mObservations.count(pKF) ? (int)mObservations[pKF] : -1; }
    void EraseObservation(KeyFrame* pKF) { erased.push_back(pKF); mObservations.erase(pKF); }
    void UpdateNormalAndDepth() {}
    static std::mutex mGlobalMutex;
    long unsigned int mnId = 0, mnBALocalForKF = ~0ul, mnBAGlobalForKF = 0, mnCorrectedByKF = 0, mnCorrectedReference = 0;
    cv::Mat mPosGBA, mWorldPos;
    // harness data
    bool bad = false;
    std::map<KeyFrame*, size_t> mObservations;
    std::vector<KeyFrame*> erased;
};

class MapLine {
public:
    Vector6d GetWorldPos() { return mWorldPos; }
    void SetWorldPos(const Vector6d& Pos) { mWorldPos = Pos; }
    bool isBad() { return bad; }
    map<KeyFrame*, size_t> GetObservations() { return mObservations; }
    KeyFrame* GetReferenceKeyFrame() { return nullptr; }
    int GetIndexInKeyFrame(KeyFrame* pKF) { return mObservations.count(pKF) ? (int)mObservations[pKF] : -1; }
    void EraseObservation(KeyFrame* pKF) { erased.push_back(pKF); mObservations.erase(pKF); }
    void UpdateAverageDir() {}
    void ComputeDistinctiveDescriptors() {}
    static std::mutex mGlobalMutex;
    long unsigned int mnId = 0, mnBALocalForKF = ~0ul, mnBAGlobalForKF = 0, mnCorrectedByKF = 0, mnCorrectedReference = 0;
    cv::Mat mPosGBA;
    Vector6d mWorldPos;
    bool bad = false;
    map<KeyFrame*, size_t> mObservations;
    std::vector<KeyFrame*> erased;
};

class MapPlane {
public:
    typedef pcl::PointXYZRGB PointT;
    typedef pcl::PointCloud<PointT> PointCloud;
    cv::Mat GetWorldPos() { return mWorldPos.clone(); }
    void SetWorldPos(const cv::Mat& Pos) { Pos.copyTo(mWorldPos); }
    bool isBad() { return bad; }
    std::map<KeyFrame*, size_t> GetObservations() { return mObservations; }
    std::map<KeyFrame*, size_t> GetParObservations() { return mParObservations; }
    std::map<KeyFrame*, size_t> GetVerObservations() { return mVerObservations; }
    KeyFrame* GetReferenceKeyFrame() { return nullptr; }
    int GetIndexInKeyFrame(KeyFrame* pKF) { return mObservations.count(pKF) ? (int)mObservations[pKF] : -1; }
    void EraseObservation(KeyFrame* pKF) { erased.push_back(pKF); mObservations.erase(pKF); }
    void EraseParObservation(KeyFrame* pKF) { erasedPar.push_back(pKF); mParObservations.erase(pKF); }
    void EraseVerObservation(KeyFrame* pKF) { erasedVer.push_back(pKF); mVerObservations.erase(pKF); }
    void UpdateCoefficientsAndPoints() {}
    static std::mutex mGlobalMutex;
    long unsigned int mnId = 0, mnBALocalForKF = ~0ul, mnBAGlobalForKF = 0, mnCorrectedByKF = 0, mnCorrectedReference = 0;
    cv::Mat mPosGBA, mWorldPos;
    PointCloud::Ptr mvPlanePoints;
    bool bad = false;
    std::map<KeyFrame*, size_t> mObservations, mParObservations, mVerObservations;
    std::vector<KeyFrame*> erased, erasedPar, erasedVer;
};

class Frame {
public:
    typedef pcl::PointXYZRGB PointT;
    typedef pcl::PointCloud<PointT> PointCloud;
    void SetPose(cv::Mat Tcw) { mTcw = Tcw.clone(); }
    int N = 0;
    std::vector<cv::KeyPoint> mvKeysUn;
    std::vector<float> mvuRight;
    std::vector<MapPoint*> mvpMapPoints;
    std::vector<bool> mvbOutlier;
    std::vector<float> mvInvLevelSigma2;
    cv::Mat mTcw;
    static float fx, fy, cx, cy;
    float mbf = 0;
    int NL = 0;
    std::vector<Eigen::Vector3d> mvKeyLineFunctions;
    std::vector<MapLine*> mvpMapLines;
    std::vector<bool> mvbLineOutlier;
    int mnPlaneNum = 0;
    std::vector<cv::Mat> mvPlaneCoefficients;
    std::vector<MapPlane*> mvpMapPlanes, mvpParallelPlanes, mvpVerticalPlanes;
    std::vector<bool> mvbPlaneOutlier, mvbParPlaneOutlier, mvbVerPlaneOutlier;
};

class KeyFrame {
public:
    typedef pcl::PointXYZRGB PointT;
    typedef pcl::PointCloud<PointT> PointCloud;
    void SetPose(const cv::Mat& Tcw) { Tcw.copyTo(pose); }
    cv::Mat GetPose() { return pose.clone(); }
    cv::Mat GetRotation() { return pose.rowRange(0, 3).colRange(0, 3).clone(); }
    cv::Mat GetTranslation() { return pose.rowRange(0, 3).col(3).clone(); }
    std::vector<KeyFrame*> GetVectorCovisibleKeyFrames() { return covisible; }
    std::vector<KeyFrame*> GetCovisiblesByWeight(const int&) { return covisible; }
    int GetWeight(KeyFrame*) { return 0; }
    KeyFrame* GetParent() { return nullptr; }
    bool hasChild(KeyFrame*) { return false; }
    std::set<KeyFrame*> GetLoopEdges() { return std::set<KeyFrame*>(); }
    void EraseMapPointMatch(const size_t& idx) { erasedPoints.push_back(idx); mps[idx] = nullptr; }
    void EraseMapPointMatch(MapPoint* pMP) { for (size_t i = 0; i < mps.size(); i++) if (mps[i] == pMP) EraseMapPointMatch(i); }
    std::vector<MapPoint*> GetMapPointMatches() { return mps; }
    void EraseMapLineMatch(const size_t& idx) { erasedLines.push_back(idx); mls[idx] = nullptr; }
    void EraseMapLineMatch(MapLine* pML) { for (size_t i = 0; i < mls.size(); i++) if (mls[i] == pML) EraseMapLineMatch(i); }
    std::vector<MapLine*> GetMapLineMatches() { return mls; }
    void EraseMapPlaneMatch(const int& idx) { erasedPlanes.push_back(idx); mpls[idx] = nullptr; }
    void EraseMapVerticalPlaneMatch(const int& idx) { erasedVer.push_back(idx); }
    void EraseMapParallelPlaneMatch(const int& idx) { erasedPar.push_back(idx); }
    void EraseMapPlaneMatch(MapPlane* pMP) { for (size_t i = 0; i < mpls.size(); i++) if (mpls[i] == pMP) EraseMapPlaneMatch((int)i); }
    void EraseMapVerticalPlaneMatch(MapPlane*) {}
    void EraseMapParallelPlaneMatch(MapPlane*) {}
    std::vector<MapPlane*> GetMapPlaneMatches() { return mpls; }
    bool isBad() { return bad; }
    long unsigned int mnId = 0, mnBALocalForKF = ~0ul, mnBAFixedForKF = ~0ul, mnBAGlobalForKF = 0;
    cv::Mat mTcwGBA, mK;
    float fx = 0, fy = 0, cx = 0, cy = 0, mbf = 0;
    std::vector<cv::KeyPoint> mvKeysUn;
    std::vector<float> mvuRight, mvInvLevelSigma2;
    std::vector<Eigen::Vector3d> mvKeyLineFunctions;
    std::vector<cv::Mat> mvPlaneCoefficients;
    int mnPlaneNum = 0;
    // harness data
    cv::Mat pose;
    bool bad = false;
    std::vector<KeyFrame*> covisible;
    std::vector<MapPoint*> mps;
    std::vector<MapLine*> mls;
    std::vector<MapPlane*> mpls;
    std::vector<size_t> erasedPoints, erasedLines;
    std::vector<int> erasedPlanes, erasedVer, erasedPar;
};

class Map {
public:
    std::vector<KeyFrame*> GetAllKeyFrames() { return kfs; }
    std::vector<MapPoint*> GetAllMapPoints() { return mps; }
    std::vector<MapLine*> GetAllMapLines() { return mls; }
    std::vector<MapPlane*> GetAllMapPlanes() { return mpls; }
    long unsigned int GetMaxKFid() { long unsigned int m = 0; for (KeyFrame* k : kfs) if (k->mnId > m) m = k->mnId; return m; }
    std::mutex mMutexMapUpdate;
    std::vector<KeyFrame*> kfs;
    std::vector<MapPoint*> mps;
    std::vector<MapLine*> mls;
    std::vector<MapPlane*> mpls;
};

}  // namespace Planar_SLAM
