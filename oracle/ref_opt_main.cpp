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

namespace {
struct Reader {
    std::vector<unsigned char> buf;
    size_t off = 0;
    explicit Reader(const char* path) {
        FILE* f = fopen(path, "rb");
        if (!f) { perror(path); exit(2); }
        fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
        buf.resize((size_t)n);
        if (fread(buf.data(), 1, (size_t)n, f) != (size_t)n) exit(2);
        fclose(f);
    }
    template <class T> T get() { T v; memcpy(&v, buf.data() + off, sizeof(T)); off += sizeof(T); return v; }
    template <class T> const T* arr(size_t n) { const T* p = (const T*)(buf.data() + off); off += n * sizeof(T); if (off > buf.size()) { fprintf(stderr, "short input\n"); exit(2); } return p; }
};
struct Writer {
    FILE* f;
    explicit Writer(const char* path) { f = fopen(path, "wb"); if (!f) { perror(path); exit(2); } }
    ~Writer() { fclose(f); }
    template <class T> void put(const T& v) { fwrite(&v, sizeof(T), 1, f); }
    template <class T> void arr(const T* p, size_t n) { fwrite(p, sizeof(T), n, f); }
};

cv::Mat mat_f32(int r, int c, const float* p) { cv::Mat m(r, c, CV_32F); memcpy(m.data, p, sizeof(float) * r * c); return m; }

void set_config(const double* cfg) {
    auto& t = Config::table();
    t["Plane.AngleInfo"] = cfg[0]; t["Plane.DistanceInfo"] = cfg[1]; t["Plane.ParallelInfo"] = cfg[2];
    t["Plane.VerticalInfo"] = cfg[3]; t["Plane.Chi"] = cfg[4]; t["Plane.VPChi"] = cfg[5];
}

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
// Local BA.  Input = synth.ba_problem arrays.  Edge types: 0 mono, 1 stereo, 2 line end-point, 3 plane, 4 vertical, 5 parallel.
// Landmarks of type 0 that carry line edges come in (start, end) pairs; cur_kf is the keyframe LocalBundleAdjustment is
// called for.  The reference attaches every line edge to cur_kf's vertex and reads cur_kf's line function at the OBSERVER's
// slot index (src/Optimizer.cc:2170-2176, 2192-2194), so a line edge has e_kf == cur_kf and e_obs_kf = the keyframe whose
// observation it came from; slot j of cur_kf->mvKeyLineFunctions holds the j-th line observation's function.
int run_ba(const char* in, const char* out) {
    Reader r(in);
    const int K = r.get<int>(), NLM = r.get<int>(), NE = r.get<int>(), cur_kf = r.get<int>();
    const float* cam = r.arr<float>(5);
    const double* cfg = r.arr<double>(6);
    set_config(cfg);
    const float* kf_Tcw = r.arr<float>((size_t)K * 16);
    const unsigned char* kf_fixed = r.arr<unsigned char>(K);
    const unsigned char* lm_type = r.arr<unsigned char>(NLM);
    const double* lm_init = r.arr<double>((size_t)NLM * 4);
    const int* e_kf = r.arr<int>(NE); const int* e_obs_kf = r.arr<int>(NE); const int* e_lm = r.arr<int>(NE);
    const unsigned char* e_type = r.arr<unsigned char>(NE);
    const double* e_meas = r.arr<double>((size_t)NE * 4);
    const float* e_is2 = r.arr<float>(NE);

    // keyframes in one array => std::map<KeyFrame*, size_t> iterates in keyframe order
    std::vector<KeyFrame> kfs(K);
    // Which keyframes are optimised is decided by the reference from the covisibility list (local) versus "sees a local
    // landmark" (fixed); mnId == 0 is fixed as well.  kf_fixed[k] => not in the covisibility list of cur_kf.
    for (int k = 0; k < K; k++) {
        KeyFrame& kf = kfs[k];
        kf.mnId = (unsigned long)k;
        kf.fx = cam[0]; kf.fy = cam[1]; kf.cx = cam[2]; kf.cy = cam[3]; kf.mbf = cam[4];
        kf.pose = mat_f32(4, 4, kf_Tcw + (size_t)k * 16);
    }
    for (int k = 0; k < K; k++) if (k != cur_kf && !kf_fixed[k]) kfs[cur_kf].covisible.push_back(&kfs[k]);
    // landmark objects
    std::vector<int> lm_obj(NLM, -1);   // index into mps / mls / mpls
    std::vector<char> is_line_pt(NLM, 0);
    for (int e = 0; e < NE; e++) if (e_type[e] == 2) is_line_pt[e_lm[e]] = 1;
    int n_pt = 0, n_ln = 0, n_pl = 0;
    for (int l = 0; l < NLM; l++) {
        if (lm_type[l] == 1) lm_obj[l] = n_pl++;
        else if (is_line_pt[l]) { if (l > 0 && is_line_pt[l - 1] && lm_obj[l - 1] >= 0 && (l < 2 || !(is_line_pt[l - 2] && lm_obj[l - 2] == lm_obj[l - 1]))) lm_obj[l] = lm_obj[l - 1]; else lm_obj[l] = n_ln++; }
        else lm_obj[l] = n_pt++;
    }
    std::vector<MapPoint> mps(n_pt);
    std::vector<MapLine> mls(n_ln);
    std::vector<MapPlane> mpls(n_pl);
    std::vector<int> line_first(n_ln, -1);
    {
        int ip = 0, ipl = 0;
        for (int l = 0; l < NLM; l++) {
            if (lm_type[l] == 1) {
                MapPlane& p = mpls[lm_obj[l]]; p.mnId = (unsigned long)ipl++;
                float c[4] = {(float)lm_init[l * 4], (float)lm_init[l * 4 + 1], (float)lm_init[l * 4 + 2], (float)lm_init[l * 4 + 3]};
                p.mWorldPos = mat_f32(4, 1, c);
            } else if (is_line_pt[l]) {
                MapLine& m = mls[lm_obj[l]];
                if (line_first[lm_obj[l]] < 0) { line_first[lm_obj[l]] = l; m.mnId = (unsigned long)lm_obj[l]; for (int j = 0; j < 3; j++) m.mWorldPos[j] = lm_init[l * 4 + j]; }
                else for (int j = 0; j < 3; j++) m.mWorldPos[3 + j] = lm_init[l * 4 + j];
            } else {
                MapPoint& p = mps[lm_obj[l]]; p.mnId = (unsigned long)ip++;
                float c[3] = {(float)lm_init[l * 4], (float)lm_init[l * 4 + 1], (float)lm_init[l * 4 + 2]};
                p.mWorldPos = mat_f32(3, 1, c);
            }
        }
    }
    // observations: one feature slot per edge in its keyframe
    for (int e = 0; e < NE; e++) {
        KeyFrame& kf = kfs[e_obs_kf[e]];
        const int l = e_lm[e];
        if (e_type[e] == 2 ? e_kf[e] != cur_kf : e_kf[e] != e_obs_kf[e]) { fprintf(stderr, "edge %d: not expressible in the reference's LocalBundleAdjustment\n", e); return 3; }
        switch (e_type[e]) {
            case 0: case 1: {
                cv::KeyPoint kp; kp.pt.x = (float)e_meas[e * 4]; kp.pt.y = (float)e_meas[e * 4 + 1]; kp.octave = (int)kf.mvKeysUn.size();
                kf.mvKeysUn.push_back(kp);
                kf.mvuRight.push_back(e_type[e] == 1 ? (float)e_meas[e * 4 + 2] : -1.f);
                kf.mvInvLevelSigma2.push_back(e_is2[e]);
                kf.mps.push_back(&mps[lm_obj[l]]);
                mps[lm_obj[l]].mObservations[&kf] = kf.mvKeysUn.size() - 1;
                break;
            }
            case 2: {
                if (l != line_first[lm_obj[l]]) break;   // the end-point edge shares the observation of the start-point edge
                KeyFrame& cur = kfs[cur_kf];
                cur.mvKeyLineFunctions.push_back(Eigen::Vector3d(e_meas[e * 4], e_meas[e * 4 + 1], e_meas[e * 4 + 2]));
                kf.mls.push_back(&mls[lm_obj[l]]);
                mls[lm_obj[l]].mObservations[&kf] = cur.mvKeyLineFunctions.size() - 1;
                break;
            }
            default: {
                float c[4] = {(float)e_meas[e * 4], (float)e_meas[e * 4 + 1], (float)e_meas[e * 4 + 2], (float)e_meas[e * 4 + 3]};
                kf.mvPlaneCoefficients.push_back(mat_f32(4, 1, c));
                MapPlane& p = mpls[lm_obj[l]];
                const size_t idx = kf.mvPlaneCoefficients.size() - 1;
                if (e_type[e] == 3) { p.mObservations[&kf] = idx; kf.mpls.push_back(&p); }
                else if (e_type[e] == 4) p.mVerObservations[&kf] = idx;
                else p.mParObservations[&kf] = idx;
                break;
            }
        }
    }
    Map map;
    bool stop = false;
    Optimizer::LocalBundleAdjustment(&kfs[cur_kf], &stop, &map);
    Writer w(out);
    for (int k = 0; k < K; k++) w.arr((const float*)kfs[k].pose.data, 16);
    for (int l = 0; l < NLM; l++) {
        double v[4] = {0, 0, 0, 0};
        if (lm_type[l] == 1) { cv::Mat c = mpls[lm_obj[l]].mWorldPos; for (int j = 0; j < 4; j++) v[j] = c.at<float>(j); }
        else if (is_line_pt[l]) { const int o = (l == line_first[lm_obj[l]]) ? 0 : 3; for (int j = 0; j < 3; j++) v[j] = mls[lm_obj[l]].mWorldPos[o + j]; }
        else { cv::Mat c = mps[lm_obj[l]].mWorldPos; for (int j = 0; j < 3; j++) v[j] = c.at<float>(j); }
        w.arr(v, 4);
    }
    // erased associations per edge (the reference's outlier verdict after the second round)
    std::vector<unsigned char> er(NE, 0);
    for (int e = 0; e < NE; e++) {
        KeyFrame* kf = &kfs[e_obs_kf[e]];
        const int l = e_lm[e];
        auto has = [&](const std::vector<KeyFrame*>& v) { for (KeyFrame* k : v) if (k == kf) return true; return false; };
        if (e_type[e] <= 1) er[e] = has(mps[lm_obj[l]].erased);
        else if (e_type[e] == 2) er[e] = has(mls[lm_obj[l]].erased);
        else if (e_type[e] == 3) er[e] = has(mpls[lm_obj[l]].erased);
        else if (e_type[e] == 4) er[e] = has(mpls[lm_obj[l]].erasedVer);
        else er[e] = has(mpls[lm_obj[l]].erasedPar);
    }
    w.arr(er.data(), er.size());
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
