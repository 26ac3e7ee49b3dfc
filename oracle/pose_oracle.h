// oracle/pose_oracle.h — TEST INFRASTRUCTURE (CPU oracle), not product code.
// CPU restatement of Optimizer::PoseOptimization / TranslationOptimization
// (reference src/Optimizer.cc:550-1275, :2995-3738) including the g2o pieces they use.
#pragma once
#include <cstdint>

namespace orc {

struct PoseParams {
    float fx, fy, cx, cy, bf;
    // raw values of Config keys Plane.AngleInfo / DistanceInfo / ParallelInfo / VerticalInfo / Chi / VPChi
    double angle_info, distance_info, parallel_info, vertical_info, plane_chi, vp_chi;
};

// One frame's problem, plain arrays.
struct PoseProblem {
    int n_points, n_lines, n_planes;
    const uint8_t* pt_valid;      // [N]     mvpMapPoints[i] != NULL
    const float* pt_xw;           // [N][3]  MapPoint::GetWorldPos() (float32)
    const float* pt_obs;          // [N][3]  mvKeysUn[i].pt.x, .pt.y, mvuRight[i] (<0 => monocular)
    const float* pt_inv_sigma2;   // [N]     mvInvLevelSigma2[mvKeysUn[i].octave]
    const uint8_t* ln_valid;      // [NL]    mvpMapLines[i] != NULL
    const double* ln_obs;         // [NL][3] mvKeyLineFunctions[i]
    const double* ln_xw;          // [NL][6] MapLine::mWorldPos (start xyz, end xyz)
    const float* pl_meas;         // [M][4]  mvPlaneCoefficients[i]
    const uint8_t* pl_valid;      // [M][3]  mvpMapPlanes[i], mvpParallelPlanes[i], mvpVerticalPlanes[i] != NULL
    const float* pl_world;        // [M][3][4] the three map planes' GetWorldPos()
    const float* Tcw;             // [16]    pFrame->mTcw row-major 4x4 float32
};

struct PoseResult {
    float Tcw[16];
    uint8_t* pt_outlier;   // [N]   mvbOutlier (only entries with pt_valid are written)
    uint8_t* ln_outlier;   // [NL]  mvbLineOutlier
    uint8_t* pl_outlier;   // [M][3] mvbPlaneOutlier, mvbParPlaneOutlier, mvbVerPlaneOutlier
    int n_inliers;         // return value of the reference function
    int lm_iterations;     // total LM iterations run (diagnostic)
    double final_chi2;     // robust chi2 after the last round (diagnostic)
};

enum PoseMode { MODE_POSE = 0, MODE_TRANSLATION = 1 };

// max_rounds = 4 and its = 10 reproduce the reference; other values are for the bench's
// "one optimize(10) round" shape.
void pose_optimize(const PoseProblem& p, const PoseParams& prm, int mode, int rounds, int its, PoseResult& out);

// edge-level check against oracle/_ref/ref_opt "edges" (see pose_oracle.cpp)
void pose_edge_eval(int cls, const float* Tcw, const double* X, const double* obs, const float* pw, const float* pm, const PoseParams& prm,
                    double* err, double* chi2_out, double* J);

}  // namespace orc
