// oracle/ba_oracle.cpp — TEST INFRASTRUCTURE (CPU oracle), not product code.
//
// CPU restatement of the numerical core of Optimizer::LocalBundleAdjustment (reference src/Optimizer.cc:1853-2680):
// the graph the reference assembles from KeyFrame/MapPoint/MapLine/MapPlane objects is taken as plain arrays
// (poses, 3-dof landmarks, binary edges); everything from optimizer.initializeOptimization() (:2354) to the outlier
// lists (:2471-2575) is restated:
//   BlockSolver_6_3::buildSystem / solve (Schur)   Thirdparty/g2o/g2o/core/block_solver.hpp:354-489,502-560
//   BaseBinaryEdge::constructQuadraticForm / numeric linearizeOplus  Thirdparty/g2o/g2o/core/base_binary_edge.hpp:55-198
//   LM driver                                       Thirdparty/g2o/g2o/core/optimization_algorithm_levenberg.cpp:61-190
//   EdgeSE3ProjectXYZ / EdgeStereoSE3ProjectXYZ     Thirdparty/g2o/g2o/types/types_six_dof_expmap.cpp:103-235
//   EdgeLineProjectXYZ                              include/EdgeLine.h:53-153
//   EdgePlane / EdgeVerticalPlane / EdgeParallelPlane, VertexPlane::oplusImpl -> Plane3D::oplus   g2oAddition/*.h
// The reduced camera system is solved densely (the reference uses Eigen SimplicialLDLT on the same matrix).
// PARITY UNPINNED (g2o/Eigen cannot be built here); tolerance-based tests only (1e-5 on poses).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <vector>

#include "geom.h"
#include "pose_oracle.h"

namespace orc {
namespace {
using namespace geom;

enum { BE_MONO = 0, BE_STEREO = 1, BE_LINE = 2, BE_PLANE = 3, BE_VER = 4, BE_PAR = 5 };

struct Lm { int type; V3 X; Plane P; };          // type 0: VertexSBAPointXYZ, 1: VertexPlane
struct Edge {
    int kf, lm, type, dim, level = 0;
    bool robust = true;
    double delta = 0, info[3] = {0, 0, 0}, meas[4] = {0, 0, 0, 0};
    Plane pm{};
    double err[3] = {0, 0, 0};                    // g2o's stored _error (stale for inactive edges)
};
struct State { std::vector<SE3> T; std::vector<Lm> lm; };

struct BA {
    double fx, fy, cx, cy, bf;
    std::vector<uint8_t> fixed;
    std::vector<int> pidx;                        // hessian block index of each keyframe (-1 if fixed)
    int np = 0;
    State st;
    std::vector<Edge> edges;
    std::vector<int> active;
    int lm_iters = 0;

    void error(Edge& e, const State& s) const {
        const SE3& T = s.T[e.kf];
        const Lm& L = s.lm[e.lm];
        if (e.type <= BE_LINE) {
            V3 p = se3_map(T, L.X);
            if (e.type == BE_MONO) {
                e.err[0] = e.meas[0] - (p.x / p.z * fx + cx); e.err[1] = e.meas[1] - (p.y / p.z * fy + cy); e.err[2] = 0;
            } else if (e.type == BE_STEREO) {
                const float invz = (float)(1.0f / p.z);
                const double u = p.x * invz * fx + cx, v = p.y * invz * fy + cy;
                e.err[0] = e.meas[0] - u; e.err[1] = e.meas[1] - v; e.err[2] = e.meas[2] - (u - bf * invz);
            } else {
                const double u = p.x / p.z * fx + cx, v = p.y / p.z * fy + cy;
                e.err[0] = e.meas[0] * u + e.meas[1] * v + e.meas[2]; e.err[1] = 0; e.err[2] = 0;
            }
        } else {
            Plane local = plane_transform(T, L.P);
            double r[3] = {0, 0, 0};
            if (e.type == BE_PLANE) ominus(local, e.pm, r);
            else if (e.type == BE_VER) ominus_ver(local, e.pm, r);
            else ominus_par(local, e.pm, r);
            e.err[0] = r[0]; e.err[1] = r[1]; e.err[2] = e.type == BE_PLANE ? r[2] : 0;
        }
    }
    static double chi2(const Edge& e) { double s = 0; for (int i = 0; i < e.dim; i++) s += e.err[i] * (e.info[i] * e.err[i]); return s; }
    static void robustify(const Edge& e, double c2, double& r0, double& r1) {
        const double dsqr = e.delta * e.delta;
        if (c2 <= dsqr) { r0 = c2; r1 = 1; } else { const double sq = std::sqrt(c2); r0 = 2 * sq * e.delta - dsqr; r1 = e.delta / sq; }
    }
    void oplus_lm(Lm& L, const double* u) const { if (L.type == 0) { L.X.x += u[0]; L.X.y += u[1]; L.X.z += u[2]; } else plane_oplus(L.P, u); }

    // A = d err / d landmark (dim x 3), B = d err / d pose (dim x 6)
    void linearize(Edge& e, double A[3][3], double B[3][6]) {
        for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) A[i][j] = 0; for (int j = 0; j < 6; j++) B[i][j] = 0; }
        const SE3& T = st.T[e.kf];
        if (e.type <= BE_LINE) {
            const V3 p = se3_map(T, st.lm[e.lm].X);
            const M3 R = qmat(T.r);
            const double x = p.x, y = p.y, z = p.z, z_2 = z * z;
            if (e.type == BE_LINE) {   // EdgeLine.h:71-114
                const double invz = 1.0 / z, invz_2 = invz * invz, lx = e.meas[0], ly = e.meas[1];
                B[0][0] = -fy * ly - fx * lx * x * y * invz_2 - fy * ly * y * y * invz_2;
                B[0][1] = fx * lx + fx * lx * x * x * invz_2 + fy * ly * x * y * invz_2;
                B[0][2] = -fx * lx * y * invz + fy * ly * x * invz;
                B[0][3] = fx * lx * invz; B[0][4] = fy * ly * invz; B[0][5] = -(fx * lx * x + fy * ly * y) * invz_2;
                const double t[3] = {fx * lx, fy * ly, -(fx * lx * x + fy * ly * y) * invz};
                for (int j = 0; j < 3; j++) A[0][j] = invz * (t[0] * R.m[0][j] + t[1] * R.m[1][j] + t[2] * R.m[2][j]);
                return;
            }
            // types_six_dof_expmap.cpp:103-139 (mono) / :188-235 (stereo)
            B[0][0] = x * y / z_2 * fx; B[0][1] = -(1 + (x * x / z_2)) * fx; B[0][2] = y / z * fx; B[0][3] = -1. / z * fx; B[0][4] = 0; B[0][5] = x / z_2 * fx;
            B[1][0] = (1 + y * y / z_2) * fy; B[1][1] = -x * y / z_2 * fy; B[1][2] = -x / z * fy; B[1][3] = 0; B[1][4] = -1. / z * fy; B[1][5] = y / z_2 * fy;
            if (e.type == BE_MONO) {
                const double t0[3] = {fx, 0, -x / z * fx}, t1[3] = {0, fy, -y / z * fy};
                for (int j = 0; j < 3; j++) {
                    A[0][j] = -1. / z * (t0[0] * R.m[0][j] + t0[1] * R.m[1][j] + t0[2] * R.m[2][j]);
                    A[1][j] = -1. / z * (t1[0] * R.m[0][j] + t1[1] * R.m[1][j] + t1[2] * R.m[2][j]);
                }
            } else {
                for (int j = 0; j < 3; j++) {
                    A[0][j] = -fx * R.m[0][j] / z + fx * x * R.m[2][j] / z_2;
                    A[1][j] = -fy * R.m[1][j] / z + fy * y * R.m[2][j] / z_2;
                    A[2][j] = A[0][j] - bf * R.m[2][j] / z_2;
                }
                B[2][0] = B[0][0] - bf * y / z_2; B[2][1] = B[0][1] + bf * x / z_2; B[2][2] = B[0][2]; B[2][3] = B[0][3]; B[2][4] = 0; B[2][5] = B[0][5] - bf / z_2;
            }
            return;
        }
        // numeric, both vertices (base_binary_edge.hpp:131-198)
        const double delta = 1e-9, scalar = 1.0 / (2 * delta);
        const double before[3] = {e.err[0], e.err[1], e.err[2]};
        State tmp = st;   // (only the two touched vertices matter)
        for (int d = 0; d < 3; d++) {
            double add[3] = {0, 0, 0};
            add[d] = delta; tmp.lm[e.lm] = st.lm[e.lm]; oplus_lm(tmp.lm[e.lm], add); error(e, tmp);
            const double e1[3] = {e.err[0], e.err[1], e.err[2]};
            add[d] = -delta; tmp.lm[e.lm] = st.lm[e.lm]; oplus_lm(tmp.lm[e.lm], add); error(e, tmp);
            for (int i = 0; i < e.dim; i++) A[i][d] = scalar * (e1[i] - e.err[i]);
        }
        tmp.lm[e.lm] = st.lm[e.lm];
        if (!fixed[e.kf]) {
            for (int d = 0; d < 6; d++) {
                double add[6] = {0, 0, 0, 0, 0, 0};
                add[d] = delta; tmp.T[e.kf] = se3_mul(se3_exp(add), st.T[e.kf]); error(e, tmp);
                const double e1[3] = {e.err[0], e.err[1], e.err[2]};
                add[d] = -delta; tmp.T[e.kf] = se3_mul(se3_exp(add), st.T[e.kf]); error(e, tmp);
                for (int i = 0; i < e.dim; i++) B[i][d] = scalar * (e1[i] - e.err[i]);
            }
        }
        for (int i = 0; i < 3; i++) e.err[i] = before[i];
    }

    void compute_active_errors() { for (int k : active) error(edges[k], st); }
    double active_robust_chi2() const {
        double c = 0;
        for (int k : active) { const Edge& e = edges[k]; const double c2 = chi2(e); if (e.robust) { double r0, r1; robustify(e, c2, r0, r1); c += r0; } else c += c2; }
        return c;
    }

    // Dense Cholesky solve of the symmetric reduced system; false if not positive definite.
    static bool chol_solve(std::vector<double>& S, int n, const std::vector<double>& b, std::vector<double>& x) {
        for (int j = 0; j < n; j++) {
            double d = S[(size_t)j * n + j];
            for (int k = 0; k < j; k++) d -= S[(size_t)j * n + k] * S[(size_t)j * n + k];
            if (!(d > 0)) return false;
            d = std::sqrt(d);
            S[(size_t)j * n + j] = d;
            for (int i = j + 1; i < n; i++) {
                double v = S[(size_t)i * n + j];
                for (int k = 0; k < j; k++) v -= S[(size_t)i * n + k] * S[(size_t)j * n + k];
                S[(size_t)i * n + j] = v / d;
            }
        }
        x = b;
        for (int i = 0; i < n; i++) { double v = x[i]; for (int k = 0; k < i; k++) v -= S[(size_t)i * n + k] * x[k]; x[i] = v / S[(size_t)i * n + i]; }
        for (int i = n - 1; i >= 0; i--) { double v = x[i]; for (int k = i + 1; k < n; k++) v -= S[(size_t)k * n + i] * x[k]; x[i] = v / S[(size_t)i * n + i]; }
        return true;
    }

    // ---- linear system pieces (block_solver.hpp) ----
    struct Cpl { int lm, p; double W[6][3]; };
    std::vector<std::vector<int>> lm_edges, lm_cpl;
    std::vector<Cpl> cpl;
    std::vector<double> Hpp, bp, Hll, bl, Dinv;

    void build_system() {   // buildSystem :502-560 (errors must be current)
        const int L = (int)st.lm.size(), NP = 6 * np;
        lm_edges.assign(L, {});
        for (int k : active) lm_edges[edges[k].lm].push_back(k);
        Hpp.assign((size_t)np * 36, 0.0); bp.assign(NP, 0.0); Hll.assign((size_t)L * 9, 0.0); bl.assign((size_t)L * 3, 0.0);
        cpl.clear(); lm_cpl.assign(L, {});
        for (int l = 0; l < L; l++) {
            for (int k : lm_edges[l]) {
                Edge& e = edges[k];
                double A[3][3], B[3][6];
                linearize(e, A, B);
                double w = 1;
                if (e.robust) { double r0; robustify(e, chi2(e), r0, w); }
                const int p = pidx[e.kf];
                int ci = -1;
                if (p >= 0) {
                    for (int c : lm_cpl[l]) if (cpl[c].p == p) { ci = c; break; }
                    if (ci < 0) { Cpl c; c.lm = l; c.p = p; std::memset(c.W, 0, sizeof(c.W)); cpl.push_back(c); ci = (int)cpl.size() - 1; lm_cpl[l].push_back(ci); }
                }
                for (int i = 0; i < e.dim; i++) {
                    const double wo = w * e.info[i], r = -e.info[i] * e.err[i] * w;      // omega_r (scaled by rho')
                    for (int a = 0; a < 3; a++) {
                        bl[(size_t)l * 3 + a] += A[i][a] * r;
                        for (int c = 0; c < 3; c++) Hll[(size_t)l * 9 + a * 3 + c] += A[i][a] * wo * A[i][c];
                    }
                    if (p >= 0) {
                        for (int a = 0; a < 6; a++) {
                            bp[p * 6 + a] += B[i][a] * r;
                            for (int c = 0; c < 6; c++) Hpp[(size_t)p * 36 + a * 6 + c] += B[i][a] * wo * B[i][c];
                            for (int c = 0; c < 3; c++) cpl[ci].W[a][c] += B[i][a] * wo * A[i][c];     // Hpl block (pose x landmark)
                        }
                    }
                }
            }
        }
    }
    // Schur complement (block_solver.hpp:368-430): S = Hpp + lambda I - sum_l W Dinv W^T, bs = bp - sum_l W Dinv bl
    void schur(double lambda, std::vector<double>& S, std::vector<double>& bs) {
        const int L = (int)st.lm.size(), NP = 6 * np;
        S.assign((size_t)NP * NP, 0.0); bs = bp; Dinv.assign((size_t)L * 9, 0.0);
        for (int p = 0; p < np; p++)
            for (int a = 0; a < 6; a++) for (int c = 0; c < 6; c++) S[(size_t)(p * 6 + a) * NP + p * 6 + c] = Hpp[(size_t)p * 36 + a * 6 + c] + (a == c ? lambda : 0.0);
        for (int l = 0; l < L; l++) {
            if (lm_edges[l].empty()) continue;
            double D[3][3], Di[3][3];
            for (int a = 0; a < 3; a++) for (int c = 0; c < 3; c++) D[a][c] = Hll[(size_t)l * 9 + a * 3 + c] + (a == c ? lambda : 0.0);
            const double det = D[0][0] * (D[1][1] * D[2][2] - D[1][2] * D[2][1]) - D[0][1] * (D[1][0] * D[2][2] - D[1][2] * D[2][0]) + D[0][2] * (D[1][0] * D[2][1] - D[1][1] * D[2][0]);
            const double id = 1.0 / det;
            Di[0][0] = (D[1][1] * D[2][2] - D[1][2] * D[2][1]) * id; Di[0][1] = (D[0][2] * D[2][1] - D[0][1] * D[2][2]) * id; Di[0][2] = (D[0][1] * D[1][2] - D[0][2] * D[1][1]) * id;
            Di[1][0] = (D[1][2] * D[2][0] - D[1][0] * D[2][2]) * id; Di[1][1] = (D[0][0] * D[2][2] - D[0][2] * D[2][0]) * id; Di[1][2] = (D[0][2] * D[1][0] - D[0][0] * D[1][2]) * id;
            Di[2][0] = (D[1][0] * D[2][1] - D[1][1] * D[2][0]) * id; Di[2][1] = (D[0][1] * D[2][0] - D[0][0] * D[2][1]) * id; Di[2][2] = (D[0][0] * D[1][1] - D[0][1] * D[1][0]) * id;
            for (int a = 0; a < 3; a++) for (int c = 0; c < 3; c++) Dinv[(size_t)l * 9 + a * 3 + c] = Di[a][c];
            double db[3];
            for (int a = 0; a < 3; a++) db[a] = Di[a][0] * bl[(size_t)l * 3] + Di[a][1] * bl[(size_t)l * 3 + 1] + Di[a][2] * bl[(size_t)l * 3 + 2];
            for (int ci : lm_cpl[l]) {
                const Cpl& Ci = cpl[ci];
                double BD[6][3];
                for (int a = 0; a < 6; a++) for (int c = 0; c < 3; c++) BD[a][c] = Ci.W[a][0] * Di[0][c] + Ci.W[a][1] * Di[1][c] + Ci.W[a][2] * Di[2][c];
                for (int a = 0; a < 6; a++) bs[Ci.p * 6 + a] -= Ci.W[a][0] * db[0] + Ci.W[a][1] * db[1] + Ci.W[a][2] * db[2];
                for (int cj : lm_cpl[l]) {
                    const Cpl& Cj = cpl[cj];
                    for (int a = 0; a < 6; a++) for (int c = 0; c < 6; c++)
                        S[(size_t)(Ci.p * 6 + a) * NP + Cj.p * 6 + c] -= BD[a][0] * Cj.W[c][0] + BD[a][1] * Cj.W[c][1] + BD[a][2] * Cj.W[c][2];
                }
            }
        }
    }

    void optimize(int iterations) {
        const int L = (int)st.lm.size(), NP = 6 * np;
        if (active.empty()) return;
        double lambda = -1, ni = 2;
        int nBad = 0;
        for (int it = 0; it < iterations; it++) {
            lm_iters++;
            compute_active_errors();
            double currentChi = active_robust_chi2(), tempChi = currentChi;
            const double iniChi = currentChi;
            build_system();
            if (it == 0) {
                double mx = 0;
                for (int p = 0; p < np; p++) for (int a = 0; a < 6; a++) mx = std::max(mx, std::fabs(Hpp[(size_t)p * 36 + a * 7]));
                for (int l = 0; l < L; l++) if (!lm_edges[l].empty()) for (int a = 0; a < 3; a++) mx = std::max(mx, std::fabs(Hll[(size_t)l * 9 + a * 4]));
                lambda = 1e-5 * mx; ni = 2; nBad = 0;
            }
            double rho = 0;
            int qmax = 0;
            std::vector<double> xp(NP, 0.0), xl((size_t)L * 3, 0.0), S, bs;
            do {
                const State backup = st;
                schur(lambda, S, bs);
                bool ok2 = NP == 0 ? true : chol_solve(S, NP, bs, xp);
                if (ok2) {
                    for (int l = 0; l < L; l++) {   // xl = Dinv (bl - Hpl^T xp)
                        if (lm_edges[l].empty()) continue;
                        double cl[3] = {bl[(size_t)l * 3], bl[(size_t)l * 3 + 1], bl[(size_t)l * 3 + 2]};
                        for (int ci : lm_cpl[l]) for (int c = 0; c < 3; c++) for (int a = 0; a < 6; a++) cl[c] -= cpl[ci].W[a][c] * xp[cpl[ci].p * 6 + a];
                        for (int a = 0; a < 3; a++) xl[(size_t)l * 3 + a] = Dinv[(size_t)l * 9 + a * 3] * cl[0] + Dinv[(size_t)l * 9 + a * 3 + 1] * cl[1] + Dinv[(size_t)l * 9 + a * 3 + 2] * cl[2];
                    }
                    // update (SparseOptimizer::update): poses exp(x)*T, points +=, planes oplus
                    for (size_t k = 0; k < st.T.size(); k++) if (pidx[k] >= 0) st.T[k] = se3_mul(se3_exp(&xp[pidx[k] * 6]), st.T[k]);
                    for (int l = 0; l < L; l++) if (!lm_edges[l].empty()) oplus_lm(st.lm[l], &xl[(size_t)l * 3]);
                }
                compute_active_errors();
                tempChi = active_robust_chi2();
                if (!ok2) tempChi = std::numeric_limits<double>::max();
                rho = currentChi - tempChi;
                double scale = 0;
                if (ok2) {
                    for (int j = 0; j < NP; j++) scale += xp[j] * (lambda * xp[j] + bp[j]);
                    for (int l = 0; l < L; l++) if (!lm_edges[l].empty()) for (int a = 0; a < 3; a++) scale += xl[(size_t)l * 3 + a] * (lambda * xl[(size_t)l * 3 + a] + bl[(size_t)l * 3 + a]);
                }
                scale += 1e-3;
                rho /= scale;
                if (rho > 0 && std::isfinite(tempChi)) {
                    double alpha = 1. - std::pow((2 * rho - 1), 3);
                    alpha = std::min(alpha, 2. / 3.);
                    lambda *= std::max(1. / 3., alpha); ni = 2; currentChi = tempChi;
                } else { lambda *= ni; ni *= 2; st = backup; }
                qmax++;
            } while (rho < 0 && qmax < 10);
            if (std::getenv("ORC_BA_TRACE")) std::fprintf(stderr, "oracle it %d chi2 %.6f lambda %.6f trials %d edges %zu\n", it, currentChi, lambda, qmax, active.size());
            if (qmax == 10 || rho == 0) break;
            if ((iniChi - currentChi) * 1e3 < iniChi) nBad++; else nBad = 0;
            if (nBad >= 3) break;
        }
    }
};

SE3 to_se3f(const float* T) {
    M3 R;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R.m[i][j] = (double)T[4 * i + j];
    SE3 s; s.r = qnormalize(qfrom(R)); s.t = {(double)T[3], (double)T[7], (double)T[11]};
    return s;
}
}  // namespace
}  // namespace orc

namespace orc { namespace {
void ba_setup(BA& ba, int n_kf, const float* kf_Tcw, const uint8_t* kf_fixed, int n_lm, const uint8_t* lm_type, const double* lm_init,
              int n_edges, const int32_t* e_kf, const int32_t* e_lm, const uint8_t* e_type, const double* e_meas, const float* e_inv_sigma2,
              const PoseParams* prm) {
    using namespace geom;
    ba.fx = prm->fx; ba.fy = prm->fy; ba.cx = prm->cx; ba.cy = prm->cy; ba.bf = prm->bf;
    ba.fixed.assign(kf_fixed, kf_fixed + n_kf);
    ba.pidx.assign(n_kf, -1);
    for (int k = 0; k < n_kf; k++) { ba.st.T.push_back(to_se3f(kf_Tcw + 16 * k)); if (!kf_fixed[k]) ba.pidx[k] = ba.np++; }
    for (int l = 0; l < n_lm; l++) {
        Lm L; L.type = lm_type[l]; L.X = {lm_init[4 * l], lm_init[4 * l + 1], lm_init[4 * l + 2]};
        L.P = Plane{{lm_init[4 * l], lm_init[4 * l + 1], lm_init[4 * l + 2], lm_init[4 * l + 3]}};
        if (L.type == 1) { if (L.P.c[3] < 0) for (int i = 0; i < 4; i++) L.P.c[i] = -L.P.c[i]; plane_normalize(L.P); }   // Converter::toPlane3D
        ba.st.lm.push_back(L);
    }
    const double angleInfo = 3282.8 / (prm->angle_info * prm->angle_info), disInfo = prm->distance_info * prm->distance_info;
    const float dMono = std::sqrt(5.991), dStereo = std::sqrt(7.815), dPlane = std::sqrt(prm->plane_chi), dVP = std::sqrt(prm->vp_chi);
    for (int k = 0; k < n_edges; k++) {
        Edge e; e.kf = e_kf[k]; e.lm = e_lm[k]; e.type = e_type[k];
        for (int i = 0; i < 4; i++) e.meas[i] = e_meas[4 * k + i];
        const double is2 = (double)e_inv_sigma2[k];
        switch (e.type) {
            case BE_MONO: e.dim = 2; e.info[0] = e.info[1] = is2; e.delta = dMono; break;
            case BE_STEREO: e.dim = 3; e.info[0] = e.info[1] = e.info[2] = is2; e.delta = dStereo; break;
            case BE_LINE: e.dim = 3; e.info[0] = e.info[1] = e.info[2] = 1; e.delta = dStereo; break;
            case BE_PLANE: e.dim = 3; e.info[0] = e.info[1] = angleInfo; e.info[2] = disInfo; e.delta = dPlane; break;
            default: e.dim = 2; e.info[0] = e.info[1] = angleInfo; e.delta = dVP; break;   // both VP edges use angleInfo (:2274-2276)
        }
        if (e.type >= BE_PLANE) {
            e.pm = Plane{{e.meas[0], e.meas[1], e.meas[2], e.meas[3]}};
            if (e.pm.c[3] < 0) for (int i = 0; i < 4; i++) e.pm.c[i] = -e.pm.c[i];
            plane_normalize(e.pm);
        }
        ba.edges.push_back(e);
    }
    for (int k = 0; k < n_edges; k++) ba.active.push_back(k);   // initializeOptimization(): all edges are level 0
}
} }

extern "C" {
// Reduced camera system of the first LM iteration WITHOUT the lambda on the pose diagonal and without Hpp's own lambda:
// S = Hpp - sum_l W (Hll + lambda I)^-1 W^T, b = bp - sum_l W (Hll + lambda I)^-1 bl, robust chi2 — every term is a sum over
// landmarks/edges, so per-shard results add up (what the RCCL all-reduce exchanges).  S: [6np x 6np], b: [6np].
int orc_ba_reduced_system(int n_kf, const float* kf_Tcw, const uint8_t* kf_fixed, int n_lm, const uint8_t* lm_type, const double* lm_init,
                          int n_edges, const int32_t* e_kf, const int32_t* e_lm, const uint8_t* e_type, const double* e_meas,
                          const float* e_inv_sigma2, const orc::PoseParams* prm, double lambda, double* S_out, double* b_out, double* chi_out) {
    using namespace orc;
    BA ba;
    ba_setup(ba, n_kf, kf_Tcw, kf_fixed, n_lm, lm_type, lm_init, n_edges, e_kf, e_lm, e_type, e_meas, e_inv_sigma2, prm);
    ba.compute_active_errors();
    *chi_out = ba.active_robust_chi2();
    ba.build_system();
    std::vector<double> S, bs;
    ba.schur(0.0, S, bs);            // pose diagonal without lambda ...
    std::vector<double> S2, bs2;
    // ... but the landmark blocks need it: redo with lambda and remove it from the pose diagonal
    ba.schur(lambda, S2, bs2);
    const int NP = 6 * ba.np;
    for (int i = 0; i < NP; i++) S2[(size_t)i * NP + i] -= lambda;
    std::memcpy(S_out, S2.data(), S2.size() * sizeof(double));
    std::memcpy(b_out, bs2.data(), bs2.size() * sizeof(double));
    return ba.np;
}

// Flat layout == include/planar_abi.h planar_ba_problem.  e_meas: [n_edges][4] = (u, v, ur, -) | line (a, b, c, -) | plane coefficients.
// Outputs: kf_out [n_kf][16] float32, lm_out [n_lm][4] double (xyz,0 | plane coefficients), e_outlier [n_edges] = the reference's
// "to erase" lists (:2471-2575); returns total LM iterations.
int orc_local_ba(int n_kf, const float* kf_Tcw, const uint8_t* kf_fixed, int n_lm, const uint8_t* lm_type, const double* lm_init,
                 int n_edges, const int32_t* e_kf, const int32_t* e_lm, const uint8_t* e_type, const double* e_meas,
                 const float* e_inv_sigma2, const orc::PoseParams* prm, int its1, int its2, float* kf_out, double* lm_out,
                 uint8_t* e_outlier, double* chi2_out) {
    using namespace orc;
    using namespace orc::geom;
    BA ba;
    ba_setup(ba, n_kf, kf_Tcw, kf_fixed, n_lm, lm_type, lm_init, n_edges, e_kf, e_lm, e_type, e_meas, e_inv_sigma2, prm);
    auto thr = [&](const Edge& e) { return e.type == BE_MONO ? 5.991 : (e.type <= BE_LINE ? 7.815 : (e.type == BE_PLANE ? prm->plane_chi : prm->vp_chi)); };
    auto depth_ok = [&](const Edge& e) { return e.type > BE_STEREO || se3_map(ba.st.T[e.kf], ba.st.lm[e.lm].X).z > 0.0; };
    ba.optimize(its1);                                            // :2355
    for (size_t k = 0; k < ba.edges.size(); k++) {                // :2363-2462
        Edge& e = ba.edges[k];
        bool bad = BA::chi2(e) > thr(e) || !depth_ok(e);
        if (e.type == BE_LINE) {   // start/end edges of a line are consecutive and share their fate (:2399-2412)
            Edge& e2 = ba.edges[k + 1];
            bad = BA::chi2(e) > 7.815 || BA::chi2(e2) > 7.815;
            if (bad) e.level = e2.level = 1;
            e.robust = e2.robust = false;
            k++;
            continue;
        }
        if (bad) e.level = 1;
        e.robust = false;
    }
    ba.active.clear();
    for (int k = 0; k < n_edges; k++) if (ba.edges[k].level == 0) ba.active.push_back(k);
    ba.optimize(its2);                                            // :2467
    for (size_t k = 0; k < ba.edges.size(); k++) {                // :2471-2575 (stored errors: stale for level-1 edges)
        Edge& e = ba.edges[k];
        if (e.type == BE_LINE) {
            Edge& e2 = ba.edges[k + 1];
            const bool bad = BA::chi2(e) > 7.815 || BA::chi2(e2) > 7.815;
            e_outlier[k] = e_outlier[k + 1] = bad;
            k++;
            continue;
        }
        e_outlier[k] = BA::chi2(e) > thr(e) || !depth_ok(e);
    }
    for (int k = 0; k < n_kf; k++) {
        const M3 R = qmat(ba.st.T[k].r);
        float* o = kf_out + 16 * k;
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) o[4 * i + j] = (float)R.m[i][j];
        o[3] = (float)ba.st.T[k].t.x; o[7] = (float)ba.st.T[k].t.y; o[11] = (float)ba.st.T[k].t.z; o[12] = o[13] = o[14] = 0; o[15] = 1;
    }
    for (int l = 0; l < n_lm; l++) {
        const Lm& L = ba.st.lm[l];
        if (L.type == 0) { lm_out[4 * l] = L.X.x; lm_out[4 * l + 1] = L.X.y; lm_out[4 * l + 2] = L.X.z; lm_out[4 * l + 3] = 0; }
        else for (int i = 0; i < 4; i++) lm_out[4 * l + i] = L.P.c[i];
    }
    if (chi2_out) { ba.compute_active_errors(); *chi2_out = ba.active_robust_chi2(); }
    return ba.lm_iters;
}
}
