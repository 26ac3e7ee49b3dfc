// tests/adapter_shim/adapter_decls.hpp — TEST INFRASTRUCTURE.
// Declarations of the reference classes whose member functions include/planar_adapters.hpp DEFINES (PLANAR_ADAPTERS_WITH_TRACKING):
// the GPU box has no /root/reference, so the harnesses (oracle/ref_match_main.cpp, tests/adapter_shim/adapter_pose_main.cpp) are built
// there against these instead of include/ORBmatcher.h:38-106, include/LSDmatcher.h:14-42, include/PlaneMatcher.h:11-31,
// include/Optimizer.h:37-43 and include/Config.h.  Only what the adapters cover is declared.  The map classes come from
// oracle/shim/match_standins.hpp (-DSTANDINS_NO_REFERENCE), force-included before this file.
#pragma once
#include <map>
#include <string>
#include <vector>

namespace Planar_SLAM {

class ORBmatcher {
public:
    ORBmatcher(float nnratio = 0.6, bool checkOri = true) : mfNNratio(nnratio), mbCheckOrientation(checkOri) {}
    int SearchByProjection(Frame& F, const std::vector<MapPoint*>& vpMapPoints, const float th = 3);
    int SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, const float th, const bool bMono);
    int MatchORBPoints(Frame& CurrentFrame, const Frame& LastFrame);
    int SearchByBoW(KeyFrame* pKF, Frame& F, std::vector<MapPoint*>& vpMapPointMatches);
    int Fuse(KeyFrame* pKF, const std::vector<MapPoint*>& vpMapPoints, const float th = 3.0);
protected:
    float mfNNratio;
    bool mbCheckOrientation;
};

class LSDmatcher {
public:
    LSDmatcher(float nnratio = 0.6, bool checkOri = true) : mfNNratio(nnratio), mbCheckOrientation(checkOri) {}
    int SearchByDescriptor(KeyFrame* pKF, Frame& currentF, std::vector<MapLine*>& vpMapLineMatches);
    int SearchByProjection(Frame& F, const std::vector<MapLine*>& vpMapLines, const float th = 3);
    int Fuse(KeyFrame* pKF, const std::vector<MapLine*>& vpMapLines, const float th = 3.0);
protected:
    float mfNNratio;
    bool mbCheckOrientation;
};

class PlaneMatcher {
public:
    PlaneMatcher(float dTh = 0.1, float aTh = 0.86, float verTh = 0.08716, float parTh = 0.9962) : dTh(dTh), aTh(aTh), verTh(verTh), parTh(parTh) {}
    int SearchMapByCoefficients(Frame& pF, const std::vector<MapPlane*>& vpMapPlanes);
protected:
    float dTh, aTh, verTh, parTh;
};

class Config {
public:
    static std::map<std::string, double>& table() { static std::map<std::string, double> t; return t; }
    template <typename T> static T Get(const std::string& key) { return T(table().at(key)); }
};

class Optimizer {
public:
    int static PoseOptimization(Frame* pFrame);
    int static TranslationOptimization(Frame* pFrame);
};

}  // namespace Planar_SLAM

#define PLANAR_ADAPTERS_WITH_TRACKING
#define PLANAR_ADAPTERS_WITH_FUSE      // the stand-in MapPoint / MapLine carry GetDistanceRange (oracle/shim/match_standins.hpp)
#include "planar_adapters.hpp"
