// oracle/pose_oracle.cpp — TEST INFRASTRUCTURE (CPU oracle), not product code.
//
// CPU restatement of Optimizer::PoseOptimization (reference src/Optimizer.cc:550-1275) and
// Optimizer::TranslationOptimization (:2995-3738), with the vendored-g2o pieces they drive:
//   LM driver        Thirdparty/g2o/g2o/core/optimization_algorithm_levenberg.cpp:61-190
//   optimize loop    Thirdparty/g2o/g2o/core/sparse_optimizer.cpp:354-420
//   quadratic form   Thirdparty/g2o/g2o/core/base_unary_edge.hpp:43-72, numeric Jacobian :82-122
//   Huber            Thirdparty/g2o/g2o/core/robust_kernel_impl.cpp:65-91
//   dense solve      Thirdparty/g2o/g2o/solvers/linear_solver_dense.h:65-114 (Eigen::LDLT)
//   SE3Quat          Thirdparty/g2o/g2o/types/se3quat.h
//   point edges      Thirdparty/g2o/g2o/types/types_six_dof_expmap.{h,cpp}
//   line edges       include/EdgeLine.h:155-337
//   plane edges      g2oAddition/{Plane3D,EdgePlane,EdgeParallelPlane,EdgeVerticalPlane}.h
//
// PARITY UNPINNED: g2o needs Eigen, which is not in the container, so this restatement cannot
// be checked against a build of the reference; the Eigen kernels it relies on (quaternion <->
// matrix, AngleAxis, pivoted LDLT) are restated from Eigen 3.3's published algorithms.  The
// stated tolerance for this path is 1e-5 on the SE3 pose (BASELINE.json), not bit-exactness.
#include "pose_oracle.h"

#include "geom.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <vector>

namespace orc {
namespace {

using namespace geom;

// ---- edges ----
enum EdgeType { E_MONO, E_STEREO, E_LINE, E_PLANE, E_PAR, E_VER };
struct Edge {
    int type, dim, level = 0, idx = 0;
    bool robust = true;
    double delta = 0, dsqr = 0;   // RobustKernelHuber::setDelta (robust_kernel_impl.cpp:65-69)
    double info[3] = {0, 0, 0};   // diagonal information
    V3 X{0, 0, 0};                // Xw (pose mode) or Xc (translation mode)
    double obs[3] = {0, 0, 0};
    Plane pw{}, pm{};             // map plane (Xw / Xc) and measurement
    double err[3] = {0, 0, 0};
};

struct Ctx {
    double fx, fy, cx, cy, bf;
    int mode;
};

void compute_error(Edge& e, const SE3& T, const Ctx& c) {
    switch (e.type) {
        case E_MONO: {   // types_six_dof_expmap.h:152-156, cam_project .cpp:290-296
            V3 p = c.mode == MODE_POSE ? se3_map(T, e.X) : se3_map_trans(T, e.X);
            const double u = p.x / p.z * c.fx + c.cx, v = p.y / p.z * c.fy + c.cy;
            e.err[0] = e.obs[0] - u; e.err[1] = e.obs[1] - v;
            break;
        }
        case E_STEREO: {   // .h:212-216, cam_project .cpp:299-307 (float invz quirk)
            V3 p = c.mode == MODE_POSE ? se3_map(T, e.X) : se3_map_trans(T, e.X);
            const float invz = (float)(1.0f / p.z);
            const double u = p.x * invz * c.fx + c.cx, v = p.y * invz * c.fy + c.cy;
            e.err[0] = e.obs[0] - u; e.err[1] = e.obs[1] - v; e.err[2] = e.obs[2] - (u - c.bf * invz);
            break;
        }
        case E_LINE: {   // EdgeLine.h:162-170
            V3 p = c.mode == MODE_POSE ? se3_map(T, e.X) : se3_map_trans(T, e.X);
            const double u = p.x / p.z * c.fx + c.cx, v = p.y / p.z * c.fy + c.cy;
            e.err[0] = e.obs[0] * u + e.obs[1] * v + e.obs[2]; e.err[1] = 0; e.err[2] = 0;
            break;
        }
        case E_PLANE: {   // EdgePlane.h:137-142 / :233-238
            Plane local = c.mode == MODE_POSE ? plane_transform(T, e.pw) : plane_translate(T, e.pw);
            ominus(local, e.pm, e.err);
            break;
        }
        case E_PAR: {   // EdgeParallelPlane.h:117-122 / :199-206 (the translation variant exists but Optimizer.cc never instantiates it)
            Plane local = c.mode == MODE_POSE ? plane_transform(T, e.pw) : plane_translate(T, e.pw);
            ominus_par(local, e.pm, e.err);
            break;
        }
        case E_VER: {   // EdgeVerticalPlane.h:118-123 / :200-207
            Plane local = c.mode == MODE_POSE ? plane_transform(T, e.pw) : plane_translate(T, e.pw);
            ominus_ver(local, e.pm, e.err);
            break;
        }
    }
}

inline double chi2(const Edge& e) {   // base_edge.h:58-61
    double s = 0;
    for (int i = 0; i < e.dim; i++) s += e.err[i] * (e.info[i] * e.err[i]);
    return s;
}

void linearize(Edge& e, const SE3& T, const Ctx& c, double J[3][6]) {
    for (int i = 0; i < 3; i++) for (int j = 0; j < 6; j++) J[i][j] = 0;
    if (e.type == E_MONO || e.type == E_STEREO || e.type == E_LINE) {
        V3 p = c.mode == MODE_POSE ? se3_map(T, e.X) : se3_map_trans(T, e.X);
        const double x = p.x, y = p.y, invz = 1.0 / p.z, invz_2 = invz * invz;
        const double fx = c.fx, fy = c.fy;
        if (e.type == E_LINE) {   // EdgeLine.h:176-192 / :268-284
            const double lx = e.obs[0], ly = e.obs[1];
            if (c.mode == MODE_POSE) {
                J[0][0] = -fy * ly - fx * lx * x * y * invz_2 - fy * ly * y * y * invz_2;
                J[0][1] = fx * lx + fx * lx * x * x * invz_2 + fy * ly * x * y * invz_2;
                J[0][2] = -fx * lx * y * invz + fy * ly * x * invz;
            }
            J[0][3] = fx * lx * invz;
            J[0][4] = fy * ly * invz;
            J[0][5] = -(fx * lx * x + fy * ly * y) * invz_2;
            return;
        }
        // types_six_dof_expmap.cpp:266-288 / :335-364 / :404-434 / :463-485
        if (c.mode == MODE_POSE) {
            J[0][0] = x * y * invz_2 * fx; J[0][1] = -(1 + (x * x * invz_2)) * fx; J[0][2] = y * invz * fx;
            J[1][0] = (1 + y * y * invz_2) * fy; J[1][1] = -x * y * invz_2 * fy; J[1][2] = -x * invz * fy;
        }
        J[0][3] = -invz * fx; J[0][4] = 0; J[0][5] = x * invz_2 * fx;
        J[1][3] = 0; J[1][4] = -invz * fy; J[1][5] = y * invz_2 * fy;
        if (e.type == E_STEREO) {
            if (c.mode == MODE_POSE) {
                J[2][0] = J[0][0] - c.bf * y * invz_2; J[2][1] = J[0][1] + c.bf * x * invz_2; J[2][2] = J[0][2];
            }
            J[2][3] = J[0][3]; J[2][4] = 0; J[2][5] = J[0][5] - c.bf * invz_2;
        }
        return;
    }
    // numeric central differences through oplus (base_unary_edge.hpp:82-122)
    const double delta = 1e-9, scalar = 1.0 / (2 * delta);
    double before[3] = {e.err[0], e.err[1], e.err[2]};
    double add[6] = {0, 0, 0, 0, 0, 0};
    for (int d = 0; d < 6; d++) {
        add[d] = delta;
        SE3 Tp = se3_mul(se3_exp(add), T);
        compute_error(e, Tp, c);
        double e1[3] = {e.err[0], e.err[1], e.err[2]};
        add[d] = -delta;
        SE3 Tm = se3_mul(se3_exp(add), T);
        compute_error(e, Tm, c);
        add[d] = 0.0;
        for (int i = 0; i < e.dim; i++) J[i][d] = scalar * (e1[i] - e.err[i]);
    }
    for (int i = 0; i < 3; i++) e.err[i] = before[i];
    if (c.mode == MODE_TRANSLATION)   // EdgePlane.h:276-287
        for (int i = 0; i < 3; i++) J[i][0] = J[i][1] = J[i][2] = 0;
}

inline void robustify(const Edge& e, double c2, double rho[3]) {   // robust_kernel_impl.cpp:78-91
    if (c2 <= e.dsqr) { rho[0] = c2; rho[1] = 1.; rho[2] = 0.; }
    else {
        const double sq = std::sqrt(c2);
        rho[0] = 2 * sq * e.delta - e.dsqr; rho[1] = e.delta / sq; rho[2] = -0.5 * rho[1] / c2;
    }
}

// Eigen::LDLT<MatrixXd> (pivoted, lower) + isPositive + solve; returns false if not positive
bool ldlt_solve6(const double Hin[6][6], const double b[6], double x[6]) {
    const int n = 6;
    double A[6][6];
    for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) A[i][j] = Hin[i][j];
    int tr[6];
    enum { Zero, PosSemi, NegSemi, Indef } sign = Zero;
    for (int k = 0; k < n; k++) {
        int big = k; double bv = std::fabs(A[k][k]);
        for (int i = k + 1; i < n; i++) if (std::fabs(A[i][i]) > bv) { bv = std::fabs(A[i][i]); big = i; }
        tr[k] = big;
        if (k != big) {   // symmetric row/col swap on the lower triangle
            const int s = n - big - 1;
            for (int j = 0; j < k; j++) std::swap(A[k][j], A[big][j]);
            for (int i = 0; i < s; i++) std::swap(A[big + 1 + i][k], A[big + 1 + i][big]);
            std::swap(A[k][k], A[big][big]);
            for (int i = k + 1; i < big; i++) std::swap(A[i][k], A[big][i]);
        }
        const int rs = n - k - 1;
        if (k > 0) {
            double temp[6];
            for (int j = 0; j < k; j++) temp[j] = A[j][j] * A[k][j];
            double s = 0;
            for (int j = 0; j < k; j++) s += A[k][j] * temp[j];
            A[k][k] -= s;
            for (int i = 0; i < rs; i++) {
                double t = 0;
                for (int j = 0; j < k; j++) t += A[k + 1 + i][j] * temp[j];
                A[k + 1 + i][k] -= t;
            }
        }
        const double akk = A[k][k];
        const bool valid = std::fabs(akk) > 0;
        if (k == 0 && !valid) { sign = Zero; for (int j = 0; j < n; j++) tr[j] = j; break; }
        if (rs > 0 && valid) for (int i = 0; i < rs; i++) A[k + 1 + i][k] /= akk;
        if (sign == PosSemi) { if (akk < 0) sign = Indef; }
        else if (sign == NegSemi) { if (akk > 0) sign = Indef; }
        else if (sign == Zero) { if (akk > 0) sign = PosSemi; else if (akk < 0) sign = NegSemi; }
    }
    if (!(sign == PosSemi || sign == Zero)) return false;
    double y[6];
    for (int i = 0; i < n; i++) y[i] = b[i];
    for (int k = 0; k < n; k++) std::swap(y[k], y[tr[k]]);                        // P b
    for (int i = 0; i < n; i++) for (int j = 0; j < i; j++) y[i] -= A[i][j] * y[j];   // L^-1
    const double tol = 1.0 / std::numeric_limits<double>::max();
    for (int i = 0; i < n; i++) y[i] = std::fabs(A[i][i]) > tol ? y[i] / A[i][i] : 0;   // D^-1 (pseudo-inverse)
    for (int i = n - 1; i >= 0; i--) for (int j = i + 1; j < n; j++) y[i] -= A[j][i] * y[j];   // L^-T
    for (int k = n - 1; k >= 0; k--) std::swap(y[k], y[tr[k]]);                   // P^T
    for (int i = 0; i < n; i++) x[i] = y[i];
    return true;
}

struct Problem {
    std::vector<Edge> edges;
    std::vector<int> active;
    Ctx c;
    SE3 T;
    int lm_iters = 0;
};

void compute_active_errors(Problem& P) { for (int k : P.active) compute_error(P.edges[k], P.T, P.c); }
double active_robust_chi2(const Problem& P) {   // sparse_optimizer.cpp:100-113
    double chi = 0;
    for (int k : P.active) {
        const Edge& e = P.edges[k];
        if (e.robust) { double rho[3]; robustify(e, chi2(e), rho); chi += rho[0]; }
        else chi += chi2(e);
    }
    return chi;
}

// SparseOptimizer::optimize(iterations) with OptimizationAlgorithmLevenberg
void optimize(Problem& P, int iterations) {
    if (P.active.empty()) return;   // _ivMap empty -> optimize() returns -1 without touching the vertex
    double lambda = -1, ni = 2;
    int nBad = 0;
    for (int it = 0; it < iterations; it++) {
        P.lm_iters++;
        compute_active_errors(P);
        double currentChi = active_robust_chi2(P), tempChi = currentChi;
        const double iniChi = currentChi;
        // buildSystem (block_solver.hpp:502-560)
        double H[6][6] = {{0}}, b[6] = {0};
        for (int k : P.active) {
            Edge& e = P.edges[k];
            double J[3][6];
            linearize(e, P.T, P.c, J);
            double w = 1.0;
            if (e.robust) { double rho[3]; robustify(e, chi2(e), rho); w = rho[1]; }
            for (int i = 0; i < e.dim; i++) {
                const double oi = e.info[i];
                for (int a = 0; a < 6; a++) {
                    b[a] -= (w * J[i][a]) * oi * e.err[i];
                    const double wa = J[i][a] * (w * oi);
                    for (int bb = 0; bb < 6; bb++) H[a][bb] += wa * J[i][bb];
                }
            }
        }
        if (it == 0) {   // computeLambdaInit :166-180
            double maxDiag = 0;
            for (int j = 0; j < 6; j++) maxDiag = std::max(std::fabs(H[j][j]), maxDiag);
            lambda = 1e-5 * maxDiag; ni = 2; nBad = 0;
        }
        double rho = 0;
        int qmax = 0;
        double x[6];
        do {
            const SE3 backup = P.T;   // push
            double Hl[6][6];
            for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) Hl[i][j] = H[i][j] + (i == j ? lambda : 0.0);
            const bool ok2 = ldlt_solve6(Hl, b, x);
            // (on failure g2o leaves x untouched from the previous solve and still applies it)
            if (!ok2 && qmax == 0 && it == 0) for (int i = 0; i < 6; i++) x[i] = 0;
            P.T = se3_mul(se3_exp(x), P.T);   // VertexSE3Expmap::oplusImpl
            compute_active_errors(P);
            tempChi = active_robust_chi2(P);
            if (!ok2) tempChi = std::numeric_limits<double>::max();
            rho = currentChi - tempChi;
            double scale = 0;
            for (int j = 0; j < 6; j++) scale += x[j] * (lambda * x[j] + b[j]);
            scale += 1e-3;
            rho /= scale;
            if (rho > 0 && std::isfinite(tempChi)) {
                double alpha = 1. - std::pow((2 * rho - 1), 3);
                alpha = std::min(alpha, 2. / 3.);
                const double scaleFactor = std::max(1. / 3., alpha);
                lambda *= scaleFactor; ni = 2; currentChi = tempChi;
            } else {
                lambda *= ni; ni *= 2; P.T = backup;   // pop
            }
            qmax++;
        } while (rho < 0 && qmax < 10);
        if (std::getenv("ORC_POSE_TRACE")) std::fprintf(stderr, "oracle it %d chi2 %.6f lambda %.6f trials %d edges %zu\n", it, currentChi, lambda, qmax, P.active.size());
        if (qmax == 10 || rho == 0) break;                       // Terminate
        if ((iniChi - currentChi) * 1e3 < iniChi) nBad++; else nBad = 0;
        if (nBad >= 3) break;                                    // Terminate
    }
}

// Converter::toSE3Quat (src/Converter.cc:37-47)
SE3 to_se3(const float* T) {
    M3 R;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R.m[i][j] = (double)T[4 * i + j];
    SE3 s;
    s.r = qnormalize(qfrom(R));
    s.t = {(double)T[3], (double)T[7], (double)T[11]};
    return s;
}

}  // namespace

// One pose-only edge at one pose: error, chi2 (identity information) and the Jacobian linearizeOplus() leaves (row-major [3][6]).
// cls: 0 mono pose, 1 stereo pose, 2 mono translation, 3 stereo translation, 4 line pose, 5 line translation, 6 plane pose,
// 7 plane translation, 8 parallel pose, 9 parallel translation, 10 vertical pose, 11 vertical translation — the order
// oracle/ref_opt_main.cpp (mode "edges") dumps the reference's own classes in.
void pose_edge_eval(int cls, const float* Tcw, const double* X, const double* obs, const float* pw, const float* pm, const PoseParams& prm,
                    double* err, double* chi2_out, double* J) {
    static const int type_of[12] = {E_MONO, E_STEREO, E_MONO, E_STEREO, E_LINE, E_LINE, E_PLANE, E_PLANE, E_PAR, E_PAR, E_VER, E_VER};
    static const int mode_of[12] = {0, 0, 1, 1, 0, 1, 0, 1, 0, 1, 0, 1};
    static const int dim_of[12] = {2, 3, 2, 3, 3, 3, 3, 3, 2, 2, 2, 2};
    Ctx c{(double)prm.fx, (double)prm.fy, (double)prm.cx, (double)prm.cy, (double)prm.bf, mode_of[cls]};
    Edge e;
    e.type = type_of[cls]; e.dim = dim_of[cls];
    e.info[0] = e.info[1] = 1; e.info[2] = e.dim == 3 ? 1 : 0;
    e.X = {X[0], X[1], X[2]};
    for (int k = 0; k < 3; k++) e.obs[k] = obs[k];
    e.pw = plane_from_float(pw); e.pm = plane_from_float(pm);
    const SE3 T = to_se3(Tcw);
    compute_error(e, T, c);
    for (int i = 0; i < 3; i++) err[i] = i < e.dim ? e.err[i] : 0.0;
    *chi2_out = chi2(e);
    double Jm[3][6];
    linearize(e, T, c, Jm);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 6; j++) J[i * 6 + j] = i < e.dim ? Jm[i][j] : 0.0;
}

void pose_optimize(const PoseProblem& p, const PoseParams& prm, int mode, int rounds, int its, PoseResult& out) {
    Problem P;
    P.c = Ctx{(double)prm.fx, (double)prm.fy, (double)prm.cx, (double)prm.cy, (double)prm.bf, mode};
    const float deltaMono = std::sqrt(5.991), deltaStereo = std::sqrt(7.815);   // float, as in the reference (:583-584)
    int nInitial = 0;
    // float32 rotation for the translation-only variant (:3021, :3067 `R_cw * Xw`): a 3x3 by 3x1 CV_32F product takes cv::gemm's
    // small-matrix path (inner length 3 == output height): the three products are summed in float, left to right (the same rule
    // oracle/shim/cvalgebra.hpp gives the real Optimizer.cc in oracle/_ref/ref_opt).
    float Rcw[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Rcw[3 * i + j] = p.Tcw[4 * i + j];
    auto rot_f32 = [&](const float X[3]) {
        V3 r;
        double v[3];
        for (int i = 0; i < 3; i++) {
            float t = Rcw[3 * i] * X[0];
            t = t + Rcw[3 * i + 1] * X[1];
            t = t + Rcw[3 * i + 2] * X[2];
            v[i] = (double)t;
        }
        r = {v[0], v[1], v[2]};
        return r;
    };
    auto set_huber = [](Edge& e, float delta) { e.robust = true; e.delta = delta; e.dsqr = (double)delta * (double)delta; };
    // ---- points (:593-668 / :3034-3125) ----
    for (int i = 0; i < p.n_points; i++) {
        if (!p.pt_valid[i]) continue;
        nInitial++;
        out.pt_outlier[i] = 0;
        Edge e;
        const bool mono = p.pt_obs[3 * i + 2] < 0;
        e.type = mono ? E_MONO : E_STEREO; e.dim = mono ? 2 : 3; e.idx = i;
        const double is2 = (double)p.pt_inv_sigma2[i];
        e.info[0] = e.info[1] = is2; e.info[2] = mono ? 0 : is2;
        e.obs[0] = p.pt_obs[3 * i]; e.obs[1] = p.pt_obs[3 * i + 1]; e.obs[2] = mono ? 0 : p.pt_obs[3 * i + 2];
        set_huber(e, mono ? deltaMono : deltaStereo);
        if (mode == MODE_POSE) e.X = {(double)p.pt_xw[3 * i], (double)p.pt_xw[3 * i + 1], (double)p.pt_xw[3 * i + 2]};
        else e.X = rot_f32(&p.pt_xw[3 * i]);
        P.edges.push_back(e);
    }
    // ---- lines: start edge then end edge (:689-745 / :3135-3194) ----
    for (int i = 0; i < p.n_lines; i++) {
        if (!p.ln_valid[i]) continue;
        if (mode == MODE_POSE) nInitial++;   // the translation variant does not count lines (quirk, SURVEY addendum 16)
        out.ln_outlier[i] = 0;
        for (int s = 0; s < 2; s++) {
            Edge e;
            e.type = E_LINE; e.dim = 3; e.idx = i;
            e.info[0] = e.info[1] = e.info[2] = 1.0;
            for (int k = 0; k < 3; k++) e.obs[k] = p.ln_obs[3 * i + k];
            set_huber(e, deltaStereo);
            const double* X = &p.ln_xw[6 * i + 3 * s];
            if (mode == MODE_POSE) e.X = {X[0], X[1], X[2]};
            else { float Xf[3] = {(float)X[0], (float)X[1], (float)X[2]}; e.X = rot_f32(Xf); }   // Converter::toCvVec -> float
            P.edges.push_back(e);
        }
    }
    if (mode == MODE_TRANSLATION && nInitial < 3) { out.n_inliers = 0; std::memcpy(out.Tcw, p.Tcw, sizeof(out.Tcw)); return; }   // :3199-3201
    // ---- planes (:767-981 / :3203-3270) ----
    const double angleInfo = 3282.8 / (prm.angle_info * prm.angle_info), disInfo = prm.distance_info * prm.distance_info;
    const double parInfo = 3282.8 / (prm.parallel_info * prm.parallel_info), verInfo = 3282.8 / (prm.vertical_info * prm.vertical_info);
    const float deltaPlane = std::sqrt(prm.plane_chi), VPdeltaPlane = std::sqrt(prm.vp_chi);
    const int kinds = mode == MODE_POSE ? 3 : 1;
    for (int kind = 0; kind < kinds; kind++) {
        for (int i = 0; i < p.n_planes; i++) {
            if (!p.pl_valid[3 * i + kind]) continue;
            if (mode == MODE_POSE) nInitial++;
            out.pl_outlier[3 * i + kind] = 0;
            Edge e;
            e.idx = i;
            e.pm = plane_from_float(&p.pl_meas[4 * i]);
            e.pw = plane_from_float(&p.pl_world[(3 * i + kind) * 4]);
            if (kind == 0) {
                e.type = E_PLANE; e.dim = 3; e.info[0] = e.info[1] = angleInfo; e.info[2] = disInfo; set_huber(e, deltaPlane);
                if (mode == MODE_TRANSLATION) {   // Xw.rotateNormal(R_cw) (:3258-3260), float R widened to double, no re-normalisation
                    V3 n = pnormal(e.pw);
                    double r[3];
                    for (int a = 0; a < 3; a++) r[a] = (double)Rcw[3 * a] * n.x + (double)Rcw[3 * a + 1] * n.y + (double)Rcw[3 * a + 2] * n.z;
                    e.pw.c[0] = r[0]; e.pw.c[1] = r[1]; e.pw.c[2] = r[2];
                }
            } else if (kind == 1) {
                e.type = E_PAR; e.dim = 2; e.info[0] = e.info[1] = parInfo; set_huber(e, VPdeltaPlane);
            } else {
                e.type = E_VER; e.dim = 2; e.info[0] = e.info[1] = verInfo; set_huber(e, VPdeltaPlane);
            }
            P.edges.push_back(e);
        }
    }
    if (mode == MODE_POSE && nInitial < 3) { out.n_inliers = 0; std::memcpy(out.Tcw, p.Tcw, sizeof(out.Tcw)); return; }   // :985

    const float chi2Mono = 5.991f, chi2Stereo = 7.815f;
    int nBad = 0;
    P.T = to_se3(p.Tcw);
    for (int it = 0; it < rounds; it++) {
        P.T = to_se3(p.Tcw);   // restart from the initial pose every round (:998)
        P.active.clear();
        for (size_t k = 0; k < P.edges.size(); k++) if (P.edges[k].level == 0) P.active.push_back((int)k);
        optimize(P, its);
        nBad = 0;
        for (size_t k = 0; k < P.edges.size(); k++) {
            Edge& e = P.edges[k];
            const bool strip = (it == 2);
            if (e.type == E_MONO || e.type == E_STEREO) {   // :1004-1065
                if (out.pt_outlier[e.idx]) compute_error(e, P.T, P.c);
                const float c2 = (float)chi2(e);
                if (c2 > (e.type == E_MONO ? chi2Mono : chi2Stereo)) { out.pt_outlier[e.idx] = 1; e.level = 1; nBad++; }
                else { out.pt_outlier[e.idx] = 0; e.level = 0; }
                if (strip) e.robust = false;
            } else if (e.type == E_LINE) {   // :1071-1117 (pairs: start edge at k, end edge at k+1)
                Edge& e2 = P.edges[k + 1];
                if (mode == MODE_POSE || out.ln_outlier[e.idx]) { compute_error(e, P.T, P.c); compute_error(e2, P.T, P.c); }
                const float cs = (float)(e.err[0] * e.err[0]), ce = (float)(e2.err[0] * e2.err[0]);
                if (cs > 2 * chi2Mono || ce > 2 * chi2Mono) { out.ln_outlier[e.idx] = 1; e.level = e2.level = 1; if (mode == MODE_POSE) nBad++; }
                else { out.ln_outlier[e.idx] = 0; e.level = e2.level = 0; }
                if (strip) e.robust = e2.robust = false;
                k++;
            } else {   // planes :1123-1260
                const int kind = e.type == E_PLANE ? 0 : (e.type == E_PAR ? 1 : 2);
                uint8_t& flag = out.pl_outlier[3 * e.idx + kind];
                if (flag) compute_error(e, P.T, P.c);
                const float c2 = (float)chi2(e);
                const double th = kind == 0 ? prm.plane_chi : prm.vp_chi;
                if (c2 > th) { flag = 1; e.level = 1; nBad++; }
                else { flag = 0; e.level = 0; }
                if (strip) e.robust = false;
            }
        }
        if (P.edges.size() < 10) break;   // :1265
    }
    // recover pose: SE3Quat -> 4x4 double -> float32 (Converter::toCvMat, src/Converter.cc:49-70)
    M3 R = qmat(P.T.r);
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) out.Tcw[4 * i + j] = (float)R.m[i][j];
    }
    out.Tcw[3] = (float)P.T.t.x; out.Tcw[7] = (float)P.T.t.y; out.Tcw[11] = (float)P.T.t.z;
    out.Tcw[12] = out.Tcw[13] = out.Tcw[14] = 0; out.Tcw[15] = 1;
    out.n_inliers = nInitial - nBad;
    out.lm_iterations = P.lm_iters;
    P.active.clear();
    for (size_t k = 0; k < P.edges.size(); k++) if (P.edges[k].level == 0) P.active.push_back((int)k);
    compute_active_errors(P);
    out.final_chi2 = active_robust_chi2(P);
}

}  // namespace orc
