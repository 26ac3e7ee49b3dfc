// oracle/line3d_oracle.cpp — TEST INFRASTRUCTURE (CPU oracle), not product code.
//
// 3-D line back-projection: Frame::isLineGood (reference src/Frame.cc:189-267) with compPt3dCov (src/LineExtractor.cpp:1196-1250), extract3dline_mahdist
// (:1265-1359, RANSAC over rand()), verify3dLine (:1361-1416), mah_dist3d_pt_line (:1418-1470), computeLine3d_svd (:1157-1178), projectPt3d2Ln3d (:278-286)
// and random_unique (include/LSDextractor.h:241-251).  Produces mvDepthLine, mvLines3D and the FrameLine directions (mVF3DLines) the Manhattan tracker reads.
//
// Reproducibility: the reference draws from the process-global rand(); here every line draws from its own glibc-rand() stream seeded with
// (frame seed + line index), i.e. what `srand(seed + i)` before line i would give (glibc TYPE_3 generator restated below and checked against libc in
// tests/test_oracle_line3d.py).
// Pinned against the reference's own code: oracle/_ref/ref_line3d compiles src/Frame.cc:189-267 and the src/LineExtractor.cpp / include/LSDextractor.h
// functions above where they lie (extracted by line range) and tests/test_oracle_line3d.py requires every output bit to agree (fixtures in
// tests/golden/line3d_ref.npz).  What stays unpinned is OpenCV underneath: cv::SVD (JacobiSVDImpl_<double>, core/src/lapack.cpp) and the cv::gemm
// summation order are restated as read (oracle/cvprim.cpp, oracle/shim/cvalgebra.hpp) for both sides of that comparison.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../include/planar_abi.h"
#include "cvprim.h"

namespace orc {

// ---- glibc random_r TYPE_3 (x^31 + x^3 + 1), as seeded by srand(seed) ----
struct GlibcRand {
    uint32_t r[34];
    int k = 0;          // next output index: o_k = r[k + 344] >> 1 with r[i] = r[i - 31] + r[i - 3]
    std::vector<uint32_t> hist;
    explicit GlibcRand(uint32_t seed) {
        if (seed == 0) seed = 1;
        hist.resize(344);
        int32_t w = (int32_t)seed;
        hist[0] = (uint32_t)w;
        for (int i = 1; i < 31; i++) {
            const int32_t hi = w / 127773, lo = w % 127773;
            w = 16807 * lo - 2836 * hi;
            if (w < 0) w += 2147483647;
            hist[i] = (uint32_t)w;
        }
        for (int i = 31; i < 34; i++) hist[i] = hist[i - 31];
        for (int i = 34; i < 344; i++) hist[i] = hist[i - 31] + hist[i - 3];
    }
    int next() {
        const size_t i = hist.size();
        const uint32_t v = hist[i - 31] + hist[i - 3];
        hist.push_back(v);
        return (int)(v >> 1);
    }
};

struct P3 { double x = 0, y = 0, z = 0; };
static inline P3 operator-(const P3& a, const P3& b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline P3 operator+(const P3& a, const P3& b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
static inline P3 operator*(const P3& a, double s) { return {a.x * s, a.y * s, a.z * s}; }
static inline P3 operator*(double s, const P3& a) { return {a.x * s, a.y * s, a.z * s}; }
static inline P3 operator/(const P3& a, double s) { return {a.x / s, a.y / s, a.z / s}; }
static inline double dot(const P3& a, const P3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline double norm(const P3& a) { return std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }

struct RandomPoint3d { P3 pos; double DU[9]; };

static double depthStdDev(double d) { const double c1 = 0.00273, c2 = 0.00074, c3 = -0.00058; return c1 * d * d + c2 * d + c3; }

// 3x3 CV_64F products as cv::gemm evaluates them: every element a left-to-right sum of three products (both the small-matrix path of A * B and the general
// path of (A * B) * C.t() / D * U.t() accumulate k = 0, 1, 2 in order)
static void mul33(const double A[9], const double B[9], bool transB, double D[9]) {
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += A[i * 3 + k] * (transB ? B[j * 3 + k] : B[k * 3 + j]);
            D[i * 3 + j] = s;
        }
}

static RandomPoint3d compPt3dCov(const P3& pt, double f) {
    RandomPoint3d rp;
    const double J0[9] = {pt.z / f, 0, pt.x / pt.z, 0, pt.z / f, pt.y / pt.z, 0, 0, 1};
    const double cov_g_d0[9] = {1, 0, 0, 0, 1, 0, 0, 0, depthStdDev(pt.z) * depthStdDev(pt.z)};
    double t[9], cov0[9];
    mul33(J0, cov_g_d0, false, t);
    mul33(t, J0, true, cov0);
    rp.pos = pt;
    // cv::SVD svd(cov0): At = cov0^T, u(:, i) = normalised row i of At
    double At[9], W[3], Vt[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) At[i * 3 + j] = cov0[j * 3 + i];
    jacobi_svd_f64(At, 3, W, Vt, 3, 3, 3);
    double U[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) U[i * 3 + j] = At[j * 3 + i];
    const double ws[3] = {std::sqrt(W[0]), std::sqrt(W[1]), std::sqrt(W[2])};
    const double D[9] = {1 / ws[0], 0, 0, 0, 1 / ws[1], 0, 0, 0, 1 / ws[2]};
    mul33(D, U, true, rp.DU);      // D * U.t()
    return rp;
}

static double mah_dist3d_pt_line(const RandomPoint3d& pt, const P3& q1, const P3& q2) {
    // u = DU (x - q1), w = DU (x - q2) (each component a left-to-right three-term sum, as the reference writes them out); the distance is
    // |u x w| / |u - w| with u - w accumulated term by term in the reference's order
    const double ax = pt.pos.x - q1.x, ay = pt.pos.y - q1.y, az = pt.pos.z - q1.z, bx = pt.pos.x - q2.x, by = pt.pos.y - q2.y, bz = pt.pos.z - q2.z;
    const double* D = pt.DU;
    double u[3], w[3], dif[3];
    for (int r = 0; r < 3; r++) {
        const double pa0 = D[3 * r] * ax, pa1 = D[3 * r + 1] * ay, pa2 = D[3 * r + 2] * az, pb0 = D[3 * r] * bx, pb1 = D[3 * r + 1] * by, pb2 = D[3 * r + 2] * bz;
        u[r] = pa0 + pa1 + pa2;
        w[r] = pb0 + pb1 + pb2;
        dif[r] = pa0 - pb0 + pa1 - pb1 + pa2 - pb2;
    }
    const double n01 = u[0] * w[1] - u[1] * w[0], n02 = u[0] * w[2] - u[2] * w[0], n12 = u[1] * w[2] - u[2] * w[1];
    return std::sqrt((n01 * n01 + n02 * n02 + n12 * n12) / (dif[0] * dif[0] + dif[1] * dif[1] + dif[2] * dif[2]));
}

static P3 projectPt3d2Ln3d(const P3& P, const P3& mid, const P3& drct) {
    const P3 A = mid, B = mid + drct, AB = B - A, AP = P - A;
    return A + (dot(AB, AP) / (dot(AB, AB))) * AB;
}

static const double EPS_ = 1e-10;

static bool verify3dLine(const std::vector<RandomPoint3d>& pts, const P3& A, const P3& B) {
    const int nCells = 10;
    int cells[10] = {0};
    const double ratio = 0.7;
    const int nPts = (int)pts.size();
    double minv = 100, maxv = -100;
    int idx1 = 0, idx2 = 0;
    for (int i = 0; i < nPts; i++) {
        if (dot(pts[i].pos - A, B - A) < minv) { minv = dot(pts[i].pos - A, B - A); idx1 = i; }
        if (dot(pts[i].pos - A, B - A) > maxv) { maxv = dot(pts[i].pos - A, B - A); idx2 = i; }
    }
    const P3 C = projectPt3d2Ln3d(pts[idx1].pos, (A + B) * 0.5, B - A), D = projectPt3d2Ln3d(pts[idx2].pos, (A + B) * 0.5, B - A);
    const double cd = norm(D - C);
    if (cd < EPS_) return false;
    for (int i = 0; i < nPts; i++) {
        const P3 X = pts[i].pos;
        const double lambda = std::abs(dot(X - C, D - C) / cd / cd);
        if (lambda >= 1) cells[nCells - 1] += 1; else cells[(unsigned int)std::floor(lambda * 10)] += 1;
    }
    double sum = 0;
    for (int i = 0; i < nCells; i++) if (cells[i] > 0) sum = sum + 1;
    return sum / nCells > ratio;
}

static void computeLine3d_svd(const std::vector<RandomPoint3d>& pts, const std::vector<int>& idx, P3& mean, P3& drct) {
    const int n = (int)idx.size();
    mean = P3();
    for (int i = 0; i < n; i++) mean = mean + pts[idx[i]].pos;
    mean = mean * (1.0 / n);
    // P (3 x n), cv::SVD(P.t(), MODIFY_A): the n x 3 matrix has m = n >= 3 rows; At (3 x n) = P
    std::vector<double> At((size_t)3 * n);
    for (int i = 0; i < n; i++) { At[i] = pts[idx[i]].pos.x - mean.x; At[n + i] = pts[idx[i]].pos.y - mean.y; At[2 * n + i] = pts[idx[i]].pos.z - mean.z; }
    double W[3], Vt[9];
    jacobi_svd_f64(At.data(), n, W, Vt, 3, n, 3);
    drct = {Vt[0], Vt[1], Vt[2]};
}

struct Line3dOut { P3 A, B, director; int n_pts = 0; };

static Line3dOut extract3dline_mahdist(const std::vector<RandomPoint3d>& pts, GlibcRand& rng) {
    const int maxIterNo = std::min(10, int(pts.size() * (pts.size() - 1) * 0.5));
    const double distThresh = 1.5;
    std::vector<int> indexes(pts.size());
    for (size_t i = 0; i < indexes.size(); ++i) indexes[i] = (int)i;
    std::vector<int> maxInlierSet;
    RandomPoint3d bestA, bestB;
    for (int iter = 0; iter < maxIterNo; iter++) {
        std::vector<int> inlierSet;
        {   // random_unique(indexes.begin(), indexes.end(), 2)
            size_t left = indexes.size(), begin = 0;
            for (int q = 0; q < 2; q++) { const size_t r = begin + (size_t)(rng.next() % (int)left); std::swap(indexes[begin], indexes[r]); ++begin; --left; }
        }
        const RandomPoint3d& A = pts[indexes[0]];
        const RandomPoint3d& B = pts[indexes[1]];
        if (norm(B.pos - A.pos) < EPS_) continue;
        for (size_t i = 0; i < pts.size(); ++i)
            if (mah_dist3d_pt_line(pts[i], A.pos, B.pos) < distThresh) inlierSet.push_back((int)i);
        if (inlierSet.size() > maxInlierSet.size()) {
            std::vector<RandomPoint3d> inlierPts(inlierSet.size());
            for (size_t ii = 0; ii < inlierSet.size(); ++ii) inlierPts[ii] = pts[inlierSet[ii]];
            if (verify3dLine(inlierPts, A.pos, B.pos)) { maxInlierSet = inlierSet; bestA = pts[indexes[0]]; bestB = pts[indexes[1]]; }
        }
        if (maxInlierSet.size() > pts.size() * 0.6) break;
    }
    Line3dOut rl;
    if (maxInlierSet.size() >= 2) {
        P3 m = (bestA.pos + bestB.pos) * 0.5, d = bestB.pos - bestA.pos;
        while (true) {
            std::vector<int> tmpInlierSet;
            P3 tmp_m, tmp_d;
            computeLine3d_svd(pts, maxInlierSet, tmp_m, tmp_d);
            for (size_t i = 0; i < pts.size(); ++i)
                if (mah_dist3d_pt_line(pts[i], tmp_m, tmp_m + tmp_d) < distThresh) tmpInlierSet.push_back((int)i);
            if (tmpInlierSet.size() > maxInlierSet.size()) { maxInlierSet = tmpInlierSet; m = tmp_m; d = tmp_d; } else break;
        }
        double minv = 100, maxv = -100;
        int idx_end1 = 0, idx_end2 = 0;
        for (size_t i = 0; i < maxInlierSet.size(); ++i) {
            const double dproduct = dot(pts[maxInlierSet[i]].pos - m, d);
            if (dproduct < minv) { minv = dproduct; idx_end1 = (int)i; }
            if (dproduct > maxv) { maxv = dproduct; idx_end2 = (int)i; }
        }
        rl.A = pts[maxInlierSet[idx_end1]].pos;
        rl.B = pts[maxInlierSet[idx_end2]].pos;
    }
    rl.director = (rl.A - rl.B) / std::sqrt(dot(rl.A - rl.B, rl.A - rl.B));
    rl.n_pts = (int)maxInlierSet.size();
    return rl;
}

}  // namespace orc

extern "C" {

int orc_glibc_rand(uint32_t seed, int count, int32_t* out) {
    orc::GlibcRand g(seed);
    for (int i = 0; i < count; i++) out[i] = g.next();
    return 0;
}

// Frame::isLineGood for one frame.  depth_line [n] (mvDepthLine), lines3d [n][6] (mvLines3D), good [n] (the line went into mVF3DLines),
// direction [n][3] (FrameLine::direction; NaN-free only where good), n_inliers [n] (tmpLine.pts.size()), n_samples [n] (pts3d.size()).  Returns the number of good lines.
int orc_is_line_good(const planar_keyline* kl, int n_lines, const uint16_t* depth, int W, int H, int pitch_px, float factor, float fx, float fy, float cx, float cy,
                     uint32_t seed, float* depth_line, double* lines3d, uint8_t* good, double* direction, int32_t* n_inliers, int32_t* n_samples) {
    using namespace orc;
    const float invfx = 1.0f / fx, invfy = 1.0f / fy;
    auto imDepth = [&](int row, int col) -> float { return (float)depth[(size_t)row * pitch_px + col] * factor; };
    int ngood = 0;
    for (int i = 0; i < n_lines; i++) {
        depth_line[i] = -1.0f; good[i] = 0; n_inliers[i] = 0; n_samples[i] = 0;
        for (int k = 0; k < 6; k++) lines3d[(size_t)i * 6 + k] = 0;
        for (int k = 0; k < 3; k++) direction[(size_t)i * 3 + k] = 0;
        const float sx = kl[i].start_x, sy = kl[i].start_y, ex = kl[i].end_x, ey = kl[i].end_y;
        const float dxf = sx - ex, dyf = sy - ey;                                       // Point2f difference
        const double len = std::sqrt((double)dxf * dxf + (double)dyf * dyf);            // cv::norm(Point2f)
        std::vector<P3> pts3d;
        if ((int)len < 1) continue;      // shorter than one pixel: the reference divides 0 / 0 and indexes the image with int(NaN) (undefined); skipped on both sides
        const double numSmp = (double)std::min((int)len, 50);
        for (int j = 0; j <= numSmp; ++j) {
            const double w1 = 1 - j / numSmp, w2 = j / numSmp;
            const float ax = (float)(sx * w1), ay = (float)(sy * w1), bx = (float)(ex * w2), by = (float)(ey * w2);   // Point2f * double -> Point2f
            const double ptx = (double)(ax + bx), pty = (double)(ay + by);              // Point2f + Point2f, then Point2d
            if (ptx < 0 || pty < 0 || ptx >= W || pty >= H) continue;
            int row, col;
            if ((std::floor(ptx) == ptx) && (std::floor(pty) == pty)) { col = std::max(int(ptx - 1), 0); row = std::max(int(pty - 1), 0); }
            else { col = int(ptx); row = int(pty); }
            if (imDepth(row, col) <= 0.01) continue;
            const float d = imDepth(row, col);
            P3 p;
            p.z = d;
            p.x = (col - cx) * p.z * invfx;
            p.y = (row - cy) * p.z * invfy;
            pts3d.push_back(p);
        }
        n_samples[i] = (int)pts3d.size();
        if (pts3d.size() < 10.0) continue;
        std::vector<RandomPoint3d> rnd;
        rnd.reserve(pts3d.size());
        for (size_t j = 0; j < pts3d.size(); ++j) rnd.push_back(compPt3dCov(pts3d[j], (double)fx));
        GlibcRand rng(seed + (uint32_t)i);
        const Line3dOut tl = extract3dline_mahdist(rnd, rng);
        n_inliers[i] = tl.n_pts;
        if (tl.n_pts / len > 0.4 && norm(tl.A - tl.B) > 0.02) {
            depth_line[i] = std::min(imDepth((int)ey, (int)ex), imDepth((int)sy, (int)sx));
            good[i] = 1; ngood++;
            direction[(size_t)i * 3] = tl.director.x; direction[(size_t)i * 3 + 1] = tl.director.y; direction[(size_t)i * 3 + 2] = tl.director.z;
            double* o = lines3d + (size_t)i * 6;
            o[0] = tl.A.x; o[1] = tl.A.y; o[2] = tl.A.z; o[3] = tl.B.x; o[4] = tl.B.y; o[5] = tl.B.z;
        }
    }
    return ngood;
}

}  // extern "C"
