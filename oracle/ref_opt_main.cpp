// oracle/ref_opt_main.cpp — TEST INFRASTRUCTURE: harness around the REAL reference optimiser.
//
// Linked with the reference's own src/Optimizer.cc, src/Converter.cc, Thirdparty/g2o/g2o/{core,types,stuff}/*.cpp and (through
// their headers) g2oAddition/*.h, include/EdgeLine.h, compiled where they lie against oracle/shim (mini-Eigen, cv::Mat stand-in,
// data-holder Frame / KeyFrame / MapPoint / MapLine / MapPlane).  Output: oracle/_ref/ref_opt (git-ignored).  Modes:
//   ref_opt pose  <in> <out>   Optimizer::PoseOptimization / TranslationOptimization on a batch of frames (synth.pose_batch layout)
//   ref_opt ba    <in> <out>   Optimizer::LocalBundleAdjustment on one local map (synth.ba_problem layout)
//   ref_opt edges <in> <out>   computeError() + linearizeOplus() of every pose-only edge type at given poses (analytic and the real
//                              base_unary_edge.hpp numeric path), and of the binary BA edges
// This file only moves data between flat arrays and the stand-in objects and calls the reference; it computes nothing itself.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "Optimizer.h"
#include "Converter.h"
#include "Thirdparty/g2o/g2o/core/sparse_optimizer.h"
#include "Thirdparty/g2o/g2o/core/block_solver.h"
#include "Thirdparty/g2o/g2o/core/optimization_algorithm_levenberg.h"
#include "Thirdparty/g2o/g2o/solvers/linear_solver_dense.h"
#include "Thirdparty/g2o/g2o/types/types_six_dof_expmap.h"
#include "g2oAddition/EdgePlane.h"
#include "g2oAddition/EdgeParallelPlane.h"
#include "g2oAddition/EdgeVerticalPlane.h"

using namespace Planar_SLAM;

// static members the stand-in classes declare (defined in src/Frame.cc, MapPoint.cc, ... in the reference)
float Frame::fx, Frame::fy, Frame::cx, Frame::cy;
std::mutex MapPoint::mGlobalMutex, MapLine::mGlobalMutex, MapPlane::mGlobalMutex;

#include "ref_opt_harness.hpp"

namespace {

// ---------------------------------------------------------------------------------------------------------------------
int run_pose(const char* in, const char* out) {
    Reader r(in);
    const int B = r.get<int>(), MP = r.get<int>(), ML = r.get<int>(), MM = r.get<int>(), mode = r.get<int>();
    const float* cam = r.arr<float>(5);
    const double* cfg = r.arr<double>(6);
    set_config(cfg);
    const int* n_points = r.arr<int>(B); const int* n_lines = r.arr<int>(B); const int* n_planes = r.arr<int>(B);
    const unsigned char* pt_valid = r.arr<unsigned char>((size_t)B * MP);
    const float* pt_xw = r.arr<float>((size_t)B * MP * 3);
    const float* pt_obs = r.arr<float>((size_t)B * MP * 3);
    const float* pt_is2 = r.arr<float>((size_t)B * MP);
    const unsigned char* ln_valid = r.arr<unsigned char>((size_t)B * ML);
    const double* ln_obs = r.arr<double>((size_t)B * ML * 3);
    const double* ln_xw = r.arr<double>((size_t)B * ML * 6);
    const float* pl_meas = r.arr<float>((size_t)B * MM * 4);
    const unsigned char* pl_valid = r.arr<unsigned char>((size_t)B * MM * 3);
    const float* pl_world = r.arr<float>((size_t)B * MM * 12);
    const float* Tcw = r.arr<float>((size_t)B * 16);
    Writer w(out);
    for (int b = 0; b < B; b++) {
        Frame F;
        Frame::fx = cam[0]; Frame::fy = cam[1]; Frame::cx = cam[2]; Frame::cy = cam[3]; F.mbf = cam[4];
        F.mTcw = mat_f32(4, 4, Tcw + (size_t)b * 16);
        const int N = n_points[b], NL = n_lines[b], M = n_planes[b];
        std::vector<MapPoint> mps(N);
        F.N = N;
        F.mvKeysUn.resize(N); F.mvuRight.resize(N); F.mvpMapPoints.assign(N, nullptr); F.mvbOutlier.assign(N, false); F.mvInvLevelSigma2.resize(N);
        for (int i = 0; i < N; i++) {
            const size_t k = (size_t)b * MP + i;
            F.mvKeysUn[i].pt.x = pt_obs[k * 3]; F.mvKeysUn[i].pt.y = pt_obs[k * 3 + 1]; F.mvKeysUn[i].octave = i;   // one "level" per point
            F.mvuRight[i] = pt_obs[k * 3 + 2];
            F.mvInvLevelSigma2[i] = pt_is2[k];
            if (pt_valid[k]) { mps[i].mWorldPos = mat_f32(3, 1, pt_xw + k * 3); F.mvpMapPoints[i] = &mps[i]; }
        }
        std::vector<MapLine> mls(NL);
        F.NL = NL;
        F.mvKeyLineFunctions.resize(NL); F.mvpMapLines.assign(NL, nullptr); F.mvbLineOutlier.assign(NL, false);
        for (int i = 0; i < NL; i++) {
            const size_t k = (size_t)b * ML + i;
            F.mvKeyLineFunctions[i] = Eigen::Vector3d(ln_obs[k * 3], ln_obs[k * 3 + 1], ln_obs[k * 3 + 2]);
            if (ln_valid[k]) { for (int j = 0; j < 6; j++) mls[i].mWorldPos[j] = ln_xw[k * 6 + j]; F.mvpMapLines[i] = &mls[i]; }
        }
        std::vector<MapPlane> mpl((size_t)M * 3);
        F.mnPlaneNum = M;
        F.mvPlaneCoefficients.resize(M); F.mvpMapPlanes.assign(M, nullptr); F.mvpParallelPlanes.assign(M, nullptr); F.mvpVerticalPlanes.assign(M, nullptr);
        F.mvbPlaneOutlier.assign(M, false); F.mvbParPlaneOutlier.assign(M, false); F.mvbVerPlaneOutlier.assign(M, false);
        for (int i = 0; i < M; i++) {
            const size_t k = (size_t)b * MM + i;
            F.mvPlaneCoefficients[i] = mat_f32(4, 1, pl_meas + k * 4);
            for (int j = 0; j < 3; j++)
                if (pl_valid[k * 3 + j]) {
                    MapPlane* p = &mpl[(size_t)i * 3 + j];
                    p->mWorldPos = mat_f32(4, 1, pl_world + (k * 3 + j) * 4);
                    (j == 0 ? F.mvpMapPlanes : j == 1 ? F.mvpParallelPlanes : F.mvpVerticalPlanes)[i] = p;
                }
        }
        const int ret = mode == 0 ? Optimizer::PoseOptimization(&F) : Optimizer::TranslationOptimization(&F);
        w.put<int>(ret);
        w.arr((const float*)F.mTcw.data, 16);
        std::vector<unsigned char> fl;
        fl.assign((size_t)MP, 0); for (int i = 0; i < N; i++) fl[i] = F.mvbOutlier[i]; w.arr(fl.data(), fl.size());
        fl.assign((size_t)ML, 0); for (int i = 0; i < NL; i++) fl[i] = F.mvbLineOutlier[i]; w.arr(fl.data(), fl.size());
        fl.assign((size_t)MM * 3, 0);
        for (int i = 0; i < M; i++) { fl[i * 3] = F.mvbPlaneOutlier[i]; fl[i * 3 + 1] = F.mvbParPlaneOutlier[i]; fl[i * 3 + 2] = F.mvbVerPlaneOutlier[i]; }
        w.arr(fl.data(), fl.size());
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// Edge level: for n cases (pose Tcw float[16], mode) evaluate one edge of each pose-only class on given data and dump error, chi2 and
// the 6 Jacobian columns as linearizeOplus() leaves them (numeric classes go through the REAL BaseUnaryEdge::linearizeOplus).
template <class E> void dump_unary(E* e, g2o::VertexSE3Expmap* v, Writer& w, int dim) {
    g2o::SparseOptimizer opt;   // owns nothing here; the edge only needs its vertex
    e->setVertex(0, v);
    g2o::JacobianWorkspace jw;
    jw.updateSize(e);
    jw.allocate();
    e->computeError();
    double err[3] = {0, 0, 0};
    for (int i = 0; i < dim; i++) err[i] = e->errorData()[i];
    const double chi2 = e->chi2();
    static_cast<g2o::OptimizableGraph::Edge*>(e)->linearizeOplus(jw);   // BaseUnaryEdge::linearizeOplus(JacobianWorkspace&) -> virtual linearizeOplus()
    e->computeError();
    double J[18];
    for (int i = 0; i < 18; i++) J[i] = 0;
    const double* jp = jw.workspaceForVertex(0);   // column major D x 6
    for (int c = 0; c < 6; c++) for (int i = 0; i < dim; i++) J[i * 6 + c] = jp[c * dim + i];
    w.arr(err, 3); w.put(chi2); w.arr(J, 18);
}

int run_edges(const char* in, const char* out) {
    Reader r(in);
    const int n = r.get<int>();
    const float* cam = r.arr<float>(5);
    Writer w(out);
    for (int c = 0; c < n; c++) {
        const float* T = r.arr<float>(16);
        const double* X = r.arr<double>(3);        // world point
        const double* obs = r.arr<double>(3);      // u, v, ur
        const double* lobs = r.arr<double>(3);     // line function
        const float* pw = r.arr<float>(4);         // map plane (world), as MapPlane::GetWorldPos holds it
        const float* pm = r.arr<float>(4);         // measured plane (camera), as Frame::mvPlaneCoefficients holds it
        g2o::VertexSE3Expmap* v = new g2o::VertexSE3Expmap();
        v->setEstimate(Converter::toSE3Quat(mat_f32(4, 4, T)));
        v->setId(0);
        Eigen::Vector3d Xw(X[0], X[1], X[2]);
#define CAM(e) e.fx = cam[0]; e.fy = cam[1]; e.cx = cam[2]; e.cy = cam[3];
        { g2o::EdgeSE3ProjectXYZOnlyPose e; CAM(e) e.Xw = Xw; e.setMeasurement(Eigen::Vector2d(obs[0], obs[1])); e.setInformation(Eigen::Matrix2d::Identity()); dump_unary(&e, v, w, 2); }
        { g2o::EdgeStereoSE3ProjectXYZOnlyPose e; CAM(e) e.bf = cam[4]; e.Xw = Xw; e.setMeasurement(Eigen::Vector3d(obs[0], obs[1], obs[2])); e.setInformation(Eigen::Matrix3d::Identity()); dump_unary(&e, v, w, 3); }
        { g2o::EdgeSE3ProjectXYZOnlyTranslation e; CAM(e) e.Xc = Xw; e.setMeasurement(Eigen::Vector2d(obs[0], obs[1])); e.setInformation(Eigen::Matrix2d::Identity()); dump_unary(&e, v, w, 2); }
        { g2o::EdgeStereoSE3ProjectXYZOnlyTranslation e; CAM(e) e.bf = cam[4]; e.Xc = Xw; e.setMeasurement(Eigen::Vector3d(obs[0], obs[1], obs[2])); e.setInformation(Eigen::Matrix3d::Identity()); dump_unary(&e, v, w, 3); }
        { EdgeLineProjectXYZOnlyPose e; CAM(e) e.Xw = Xw; e.setMeasurement(Eigen::Vector3d(lobs[0], lobs[1], lobs[2])); e.setInformation(Eigen::Matrix3d::Identity()); dump_unary(&e, v, w, 3); }
        { EdgeLineProjectXYZOnlyTranslation e; CAM(e) e.Xc = Xw; e.setMeasurement(Eigen::Vector3d(lobs[0], lobs[1], lobs[2])); e.setInformation(Eigen::Matrix3d::Identity()); dump_unary(&e, v, w, 3); }
        const g2o::Plane3D PW = Converter::toPlane3D(mat_f32(4, 1, pw)), PM = Converter::toPlane3D(mat_f32(4, 1, pm));
        { g2o::EdgePlaneOnlyPose e; e.Xw = PW; e.setMeasurement(PM); e.setInformation(Eigen::Matrix3d::Identity()); dump_unary(&e, v, w, 3); }
        { g2o::EdgePlaneOnlyTranslation e; e.Xc = PW; e.setMeasurement(PM); e.setInformation(Eigen::Matrix3d::Identity()); dump_unary(&e, v, w, 3); }
        { g2o::EdgeParallelPlaneOnlyPose e; e.Xw = PW; e.setMeasurement(PM); e.setInformation(Eigen::Matrix2d::Identity()); dump_unary(&e, v, w, 2); }
        { g2o::EdgeParallelPlaneOnlyTranslation e; e.Xc = PW; e.setMeasurement(PM); e.setInformation(Eigen::Matrix2d::Identity()); dump_unary(&e, v, w, 2); }
        { g2o::EdgeVerticalPlaneOnlyPose e; e.Xw = PW; e.setMeasurement(PM); e.setInformation(Eigen::Matrix2d::Identity()); dump_unary(&e, v, w, 2); }
        { g2o::EdgeVerticalPlaneOnlyTranslation e; e.Xc = PW; e.setMeasurement(PM); e.setInformation(Eigen::Matrix2d::Identity()); dump_unary(&e, v, w, 2); }
#undef CAM
        // SE3Quat::exp of the first six numbers of (X, obs) as an update, applied on the left (VertexSE3Expmap::oplusImpl)
        double upd[6] = {X[0] * 0.01, X[1] * 0.01, X[2] * 0.01, obs[0] * 1e-4, obs[1] * 1e-4, obs[2] * 1e-4};
        v->oplus(upd);
        Eigen::Matrix<double, 4, 4> H = v->estimate().to_homogeneous_matrix();
        double Hd[16];
        for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) Hd[i * 4 + j] = H(i, j);
        w.arr(Hd, 16);
        delete v;
    }
    return 0;
}
}  // namespace

int main(int argc, char** argv) {
    if (argc != 4) { fprintf(stderr, "usage: ref_opt pose|ba|edges <in> <out>\n"); return 2; }
    const std::string m = argv[1];
    if (m == "pose") return run_pose(argv[2], argv[3]);
    if (m == "ba") return run_ba(argv[2], argv[3]);
    if (m == "edges") return run_edges(argv[2], argv[3]);
    return 2;
}
