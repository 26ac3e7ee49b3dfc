// oracle/shim/frame_standins.hpp — TEST INFRASTRUCTURE.
// Declarations (data members + the prototypes of the member functions under test) for Frame / MapPoint / MapLine / KeyFrame / Tracking,
// so that LINE RANGES of the real src/Frame.cc, src/MapPoint.cc, src/MapLine.cpp and src/Tracking.cc — extracted at build time by
// oracle/Makefile into oracle/_ref/gen/ (git-ignored, never committed) — compile and run in oracle/_ref/ref_frame:
//   src/Frame.cc:155-168    Frame::AssignFeaturesToGrid            src/Frame.cc:296-310   SetPose / UpdatePoseMatrices
//   src/Frame.cc:312-438    Frame::isInFrustum (MapPoint*, MapLine*)
//   src/Frame.cc:440-535    GetFeaturesInArea / GetLinesInArea / PosInGrid               src/Frame.cc:815-820   ComputePlaneWorldCoeff
//   src/Frame.cc:603-634    ComputeStereoFromRGBD / UnprojectStereo
//   src/MapPoint.cc:390-434 Get{Min,Max}DistanceInvariance, PredictScale x2             src/MapLine.cpp:369-390 the same for lines
//   src/MapPoint.cc:259-324 MapPoint::ComputeDistinctiveDescriptors                       src/ORBmatcher.cc:1710-1728 ORBmatcher::DescriptorDistance
//   src/MapPoint.cc:347-388 MapPoint::UpdateNormalAndDepth                                src/KeyFrame.cc:79-93, 107-111 KeyFrame::SetPose, GetCameraCenter
//   src/Tracking.cc:763-1157 ProjectSN2MF (5 arguments), ProjectSN2Conic, TrackManhattanFrame, MeanShift
// The member names and types are the real headers' (include/Frame.h, MapPoint.h, MapLine.h, Tracking.h, LSDextractor.h:33-57,141-199); the
// function bodies are the reference's own.  The whole files cannot be built here: they need PCL, the extractors, the viewer stack.
#pragma once
#include <algorithm>
#include <climits>
#include <cmath>
#include <iostream>
#include <map>
#include <mutex>
#include <vector>

#include "cvshim.hpp"
#include <Eigen/Core>
#include "opencv2/line_descriptor/descriptor.hpp"

#define FRAME_GRID_ROWS 48
#define FRAME_GRID_COLS 64

typedef Eigen::Matrix<double, 6, 1> Vector6d;   // include/auxiliar.h:44

// include/LSDextractor.h:33-57
class SurfaceNormal {
public:
    cv::Point3f normal;
    cv::Point3f cameraPosition;
    cv::Point2i FramePosition;
    SurfaceNormal() {}
};
typedef struct meanshiftResult { cv::Mat R_cm_Rec; float s_j_density; int axis; } ResultOfMS;
typedef struct meanshift2d { cv::Point2d centerOfShift; float density; } sMS;
typedef struct RandomPoint3ds { cv::Point3d pos; } RandomPoint3d;                       // :59-...
typedef struct FrameLines {                                                             // :141-185 (data members the tracker reads)
    cv::Point2d p, q;
    cv::Point3d direction;
    std::vector<RandomPoint3d> rndpts3d;
} FrameLine;
typedef struct axiSNVector { std::vector<SurfaceNormal> SNVector; std::vector<FrameLine> Linesvector; int axis; } axiSNV;   // :195-199

namespace Planar_SLAM {
using std::vector;
class Frame;

class KeyFrame {
public:
    float mfLogScaleFactor = 0;
    int mnScaleLevels = 0;
    cv::Mat mDescriptors;                        // include/KeyFrame.h: one row per keypoint
    cv::Mat mLineDescriptors;                    // one row per key line
    bool mbBad = false;
    bool isBad() { return mbBad; }
    void SetPose(const cv::Mat& Tcw);
    cv::Mat GetCameraCenter();
    std::vector<cv::KeyPoint> mvKeysUn;
    std::vector<float> mvScaleFactors;
    cv::Mat Tcw, Twc, Ow, Cw;
    float mHalfBaseline = 0;
    std::mutex mMutexPose;
};

class ORBmatcher {
public:
    static int DescriptorDistance(const cv::Mat& a, const cv::Mat& b);     // include/ORBmatcher.h:44
};

class MapPoint {
public:
    cv::Mat GetWorldPos() { return mWorldPos.clone(); }
    cv::Mat GetNormal() { return mNormalVector.clone(); }
    float GetMinDistanceInvariance();
    float GetMaxDistanceInvariance();
    int PredictScale(const float& currentDist, KeyFrame* pKF);
    int PredictScale(const float& currentDist, Frame* pF);
    float mTrackProjX = 0, mTrackProjY = 0, mTrackProjXR = 0;
    bool mbTrackInView = false;
    int mnTrackScaleLevel = 0;
    float mTrackViewCos = 0;
    cv::Mat mWorldPos, mNormalVector;
    float mfMinDistance = 0, mfMaxDistance = 0;
    std::mutex mMutexPos;
    void ComputeDistinctiveDescriptors();
    void UpdateNormalAndDepth();
    KeyFrame* mpRefKF = nullptr;
    std::map<KeyFrame*, size_t> mObservations;   // include/MapPoint.h: keyframe -> index of the observing keypoint
    cv::Mat mDescriptor;
    bool mbBad = false;
    std::mutex mMutexFeatures;
};

class MapLine {
public:
    Vector6d GetWorldPos() { return mWorldPos; }
    Eigen::Vector3d GetNormal() { return mNormalVector; }
    float GetMinDistanceInvariance();
    float GetMaxDistanceInvariance();
    int PredictScale(const float& currentDist, const float& logScaleFactor);
    float mTrackProjX1 = 0, mTrackProjY1 = 0, mTrackProjX2 = 0, mTrackProjY2 = 0;
    bool mbTrackInView = false;
    int mnTrackScaleLevel = 0;
    float mTrackViewCos = 0;
    Vector6d mWorldPos;
    Eigen::Vector3d mNormalVector;
    float mfMinDistance = 0, mfMaxDistance = 0;
    std::mutex mMutexPos;
    void ComputeDistinctiveDescriptors();
    std::map<KeyFrame*, size_t> mObservations;   // include/MapLine.h: keyframe -> index of the observing key line
    cv::Mat mLDescriptor;
    bool mbBad = false;
    std::mutex mMutexFeatures;
};

class Frame {
public:
    void AssignFeaturesToGrid();
    void SetPose(cv::Mat Tcw);
    void UpdatePoseMatrices();
    bool isInFrustum(MapPoint* pMP, float viewingCosLimit);
    bool isInFrustum(MapLine* pML, float viewingCosLimit);
    vector<size_t> GetFeaturesInArea(const float& x, const float& y, const float& r, const int minLevel = -1, const int maxLevel = -1) const;
    vector<size_t> GetLinesInArea(const float& x1, const float& y1, const float& x2, const float& y2, const float& r, const int minLevel = -1,
                                  const int maxLevel = -1) const;
    bool PosInGrid(const cv::KeyPoint& kp, int& posX, int& posY);
    cv::Mat ComputePlaneWorldCoeff(const int& idx);
    void ComputeStereoFromRGBD(const cv::Mat& imDepth);
    cv::Mat UnprojectStereo(const int& i);

    static float fx, fy, cx, cy, invfx, invfy;
    static float mnMinX, mnMaxX, mnMinY, mnMaxY;
    static float mfGridElementWidthInv, mfGridElementHeightInv;
    float mbf = 0, mb = 0;
    int N = 0;
    std::vector<cv::KeyPoint> mvKeys, mvKeysUn;
    std::vector<float> mvuRight, mvDepth;
    std::vector<cv::line_descriptor::KeyLine> mvKeylinesUn;
    std::vector<std::size_t> mGrid[FRAME_GRID_COLS][FRAME_GRID_ROWS];
    cv::Mat mTcw, mRcw, mtcw, mRwc, mOw, mTwc;
    int mnScaleLevels = 0;
    float mfLogScaleFactor = 0;
    std::vector<cv::Mat> mvPlaneCoefficients;
    // filled by Tracking::ProjectSN2MF (include/Frame.h:180-186)
    std::vector<cv::Point2i> vSurfaceNormalx, vSurfaceNormaly, vSurfaceNormalz;
    std::vector<cv::Point3f> vSurfacePointx, vSurfacePointy, vSurfacePointz;
    std::vector<vector<cv::Point2d>> vVanishingLinex, vVanishingLiney, vVanishingLinez;
    std::vector<RandomPoint3d> vVaishingLinePCx, vVaishingLinePCy, vVaishingLinePCz;
};

class Tracking {   // include/Tracking.h:75-78
public:
    sMS MeanShift(vector<cv::Point2d>& v2D);
    ResultOfMS ProjectSN2MF(int a, const cv::Mat& R_cm, const vector<SurfaceNormal>& vTempSurfaceNormal, vector<FrameLine>& vVanishingDirection, const int numOfSN);
    axiSNV ProjectSN2Conic(int a, const cv::Mat& R_cm, const vector<SurfaceNormal>& vTempSurfaceNormal, vector<FrameLine>& vVanishingDirection);
    cv::Mat TrackManhattanFrame(cv::Mat& mLastRcm, vector<SurfaceNormal>& vSurfaceNormal, vector<FrameLine>& vVanishingDirection);
    void ManhattanPoseStatements();   // src/Tracking.cc:251-253 (MF_can_T, mRotation_wc) and :1778 (copied into mCurrentFrame.mTcw), wrapped by oracle/Makefile
    cv::Mat MF_can, MF_can_T, mRotation_wc, Rotation_cm;   // include/Tracking.h
    Frame mCurrentFrame;
};

}  // namespace Planar_SLAM
