// manhattan_oracle.cpp — TEST INFRASTRUCTURE (not product code; see oracle/README in DESIGN.md §2).
// CPU restatement of Tracking::TrackManhattanFrame and its helpers (the per-frame Manhattan-frame rotation update that
// Tracking::Track runs between extraction and TranslationOptimization, src/Tracking.cc:248):
//   ProjectSN2Conic      src/Tracking.cc:886-953   which surface normals / vanishing directions lie in the cone around axis a
//   ProjectSN2MF (5-arg) src/Tracking.cc:757-884   tangent-plane projection of the cone members, MeanShift, new axis column
//   MeanShift            src/Tracking.cc:1140-1157 one Gaussian-kernel mean (k = exp(-20 |m|^2)), sequential double sums
//   TrackManhattanFrame  src/Tracking.cc:963-1138  three axes in turn (the matrix is updated IN PLACE: `cv::Mat R_cm = R_cm_update`
//                                                   shares the buffer, so axis 2 / 3 see the columns axis 1 / 2 wrote), completion
//                                                   of a missing axis by a cross product, SVD re-orthogonalisation R = U * Vt
// OpenCV pieces restated from the library (3.4.x): Mat::cross in float, cv::determinant (double inner products), cv::sum / cv::norm (double
// accumulation), JacobiSVDImpl_<float> (modules/core/src/lapack.cpp, shared through oracle/cvprim.cpp), plain small CV_32F products summed in float.
// PINNED against the reference's own function bodies: src/Tracking.cc:763-1157 is extracted at build time and compiled against the cv::Mat
// stand-in into oracle/_ref/ref_frame (recipe: oracle/Makefile); tests/test_oracle_frame_ref.py compares rotation (bit-exact) and cone
// membership.  What stays unpinned are the OpenCV primitives named above.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>
#include <algorithm>

#include "cvprim.h"

namespace {

struct M3 { float m[3][3]; };

// SVD::compute(R, W, U, Vt) for a 3x3 CV_32F matrix followed by R = U * Vt
void svd_orthogonalise(M3& R) {
    float At[3][3], W[3], Vt[3][3];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) At[i][j] = R.m[j][i];      // transpose(src, temp_a)
    orc::jacobi_svd3_f32(At, W, Vt);
    // u = temp_a^T: U(r, c) = At[c][r]; result(r, c) = sum_k U(r, k) * Vt[k][c]: a plain 3x3 by 3x3 CV_32F product is cv::gemm's small-matrix
    // case (inner length 3 == output width): float products summed in float, left to right
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) {
            float t = At[0][r] * Vt[0][c];
            t = t + At[1][r] * Vt[1][c];
            t = t + At[2][r] * Vt[2][c];
            R.m[r][c] = t;
        }
}

double det3(const M3& A) { return orc::det3_f32(A.m); }   // cv::determinant: double inner products (cvprim.cpp)

inline void axis_cols(int a, int& c1, int& c2, int& c3) { c1 = (a + 3) % 3; c2 = (a + 4) % 3; c3 = (a + 5) % 3; }

// n_ini = R_mc_new * v with R_mc_new = (columns c1, c2, c3 of R)^T, float arithmetic, left to right
template <typename T>
inline void rotate_into_axis(const M3& R, int c1, int c2, int c3, T vx, T vy, T vz, float& x, float& y, float& z) {
    x = (float)(R.m[0][c1] * vx + R.m[1][c1] * vy + R.m[2][c1] * vz);
    y = (float)(R.m[0][c2] * vx + R.m[1][c2] * vy + R.m[2][c2] * vz);
    z = (float)(R.m[0][c3] * vx + R.m[1][c3] * vy + R.m[2][c3] * vz);
}

}  // namespace

extern "C" {

// normals: [n][3] float (SurfaceNormal::normal); lines: [nl][3] double (FrameLine::direction).
// R_last / R_out: row-major 3x3 float.  member[i] (i < n + nl): bit a-1 set iff element i was handed to MeanShift of axis a
// (= pushed to Frame::vSurfaceNormalx/y/z resp. vVanishingLinex/y/z).  info[8]: numDirectionFound, found flags (bit a-1),
// numInCone[0..2] (surface normals only, as the reference counts), number of points given to MeanShift per axis [3].
// density[3]: s_j_density of the axes that were found (0 otherwise).  Returns numDirectionFound.
int orc_track_manhattan(const float* R_last, const float* normals, int n, const double* lines, int nl, float* R_out, uint8_t* member,
                        int32_t* info, float* density) {
    M3 R;
    std::memcpy(R.m, R_last, sizeof(R.m));                     // R_cm_update = mLastRcm.clone(); R_cm shares its buffer
    const int tot = n + nl;
    std::vector<uint8_t> cone(tot, 0);                          // first pass (ProjectSN2Conic)
    int numInCone[3] = {0, 0, 0};
    const double th_sn = std::sin(0.2018), th_ln = std::sin(0.1018), th_mf = std::sin(0.2518);
    for (int a = 1; a <= 3; a++) {
        int c1, c2, c3; axis_cols(a, c1, c2, c3);
        for (int i = 0; i < tot; i++) {
            float x, y, z;
            if (i < n) rotate_into_axis<float>(R, c1, c2, c3, normals[3 * i], normals[3 * i + 1], normals[3 * i + 2], x, y, z);
            else { const double* d = lines + 3 * (i - n); rotate_into_axis<double>(R, c1, c2, c3, d[0], d[1], d[2], x, y, z); }
            const double lambda = std::sqrt(x * x + y * y);   // sqrt(float) -> float, widened
            if (lambda < (i < n ? th_sn : th_ln)) { cone[i] |= (uint8_t)(1u << (a - 1)); if (i < n) numInCone[a - 1]++; }
        }
    }
    int minNumOfSN = n / 20;
    {
        int a = numInCone[0], b = numInCone[1], c = numInCone[2], t;
        if (a > b) t = a, a = b, b = t;
        if (b > c) t = b, b = c, c = t;
        if (a > b) t = a, a = b, b = t;
        if (b < minNumOfSN) minNumOfSN = (b + a) / 2;
    }
    if (member) std::memset(member, 0, tot);
    int found_mask = 0, nfound = 0, npts[3] = {0, 0, 0};
    float dens[3] = {0, 0, 0};
    for (int a = 1; a <= 3; a++) {
        int c1, c2, c3; axis_cols(a, c1, c2, c3);              // R already carries the columns of the axes found before
        double nom_x = 0, nom_y = 0, den = 0;
        int cnt = 0;
        for (int i = 0; i < tot; i++) {
            if (!(cone[i] & (1u << (a - 1)))) continue;
            float x, y, z;
            if (i < n) rotate_into_axis<float>(R, c1, c2, c3, normals[3 * i], normals[3 * i + 1], normals[3 * i + 2], x, y, z);
            else { const double* d = lines + 3 * (i - n); rotate_into_axis<double>(R, c1, c2, c3, d[0], d[1], d[2], x, y, z); }
            const double lambda = std::sqrt(x * x + y * y);
            if (!(lambda < th_mf)) continue;
            const double tan_alfa = lambda / std::abs(z);
            const double alfa = std::asin(lambda);
            const double mx = alfa / tan_alfa * x / z, my = alfa / tan_alfa * y / z;
            if (member) member[i] |= (uint8_t)(1u << (a - 1));
            if (std::isnan(mx) || std::isnan(my)) continue;
            const double nrm = std::sqrt(mx * mx + my * my);   // cv::norm(Point2d)
            const double k = std::exp(-20 * nrm * nrm);
            nom_x += k * mx; nom_y += k * my; den += k;
            cnt++;
        }
        npts[a - 1] = cnt;
        if ((size_t)cnt > (size_t)minNumOfSN) {
            const double sx = nom_x / den, sy = nom_y / den;
            const float s_j_density = (float)(den / cnt);
            const float alfa = (float)std::sqrt(sx * sx + sy * sy);
            const float ma_x = (float)(std::tan(alfa) / alfa * sx), ma_y = (float)(std::tan(alfa) / alfa * sy);   // tan(float) / float * double
            float col[3];
            for (int r = 0; r < 3; r++) {                      // rtemp * temp1 (:877-878): 3x3 by 3x1, cv::gemm's small-matrix case, float sums
                float t = R.m[r][c1] * ma_x;
                t = t + R.m[r][c2] * ma_y;
                t = t + R.m[r][c3] * 1.0f;
                col[r] = t;
            }
            double nn = 0;
            for (int r = 0; r < 3; r++) nn += (double)col[r] * col[r];
            nn = std::sqrt(nn);
            const float inv = (float)(1.0 / nn);               // Mat / double -> scale by 1/s, float multiply
            for (int r = 0; r < 3; r++) col[r] = col[r] * inv;
            if ((double)col[0] + (double)col[1] + (double)col[2] != 0) {   // sum(R_cm_Rec)[0] != 0 (cv::sum accumulates in double)
                nfound++; found_mask |= 1 << (a - 1);
                for (int r = 0; r < 3; r++) R.m[r][a - 1] = col[r];
                dens[a - 1] = s_j_density;
            }
        }
    }
    if (nfound >= 2) {
        if (nfound == 2) {
            auto cross_into = [&](int ca, int cb, int cdst) {  // dst = col(ca) x col(cb) in float; flipped if det is near -1
                const float a0 = R.m[0][ca], a1 = R.m[1][ca], a2 = R.m[2][ca], b0 = R.m[0][cb], b1 = R.m[1][cb], b2 = R.m[2][cb];
                const float v0 = a1 * b2 - a2 * b1, v1 = a2 * b0 - a0 * b2, v2 = a0 * b1 - a1 * b0;
                R.m[0][cdst] = v0; R.m[1][cdst] = v1; R.m[2][cdst] = v2;
                if (std::abs(det3(R) + 1) < 0.5) { R.m[0][cdst] = -v0; R.m[1][cdst] = -v1; R.m[2][cdst] = -v2; }
            };
            if ((found_mask & 3) == 3) cross_into(0, 1, 2);            // v3 = v1 x v2
            else if ((found_mask & 6) == 6) cross_into(2, 1, 0);       // v1 = v3 x v2
            else cross_into(0, 2, 1);                                  // v2 = v1 x v3
        }
        svd_orthogonalise(R);
    }
    std::memcpy(R_out, R.m, sizeof(R.m));
    if (info) { info[0] = nfound; info[1] = found_mask; for (int t = 0; t < 3; t++) { info[2 + t] = numInCone[t]; info[5 + t] = npts[t]; } }
    if (density) for (int t = 0; t < 3; t++) density[t] = dens[t];
    return nfound;
}


// mRotation_wc = (Rotation_cm * MF_can^T)^T copied into mTcw's rotation block before TranslationOptimization (src/Tracking.cc:251-253, 1778).  MF_can_T is a
// materialised cv::Mat, so the product carries no transpose flag and takes cv::gemm's small-matrix path (3x3 CV_32F): products and sums in float, then
// (float)(t * alpha) with alpha = 1.0 a double.  Pinned to the real statements: oracle/_ref/ref_frame manhattan_pose (tests/test_oracle_frame_ref.py).
void orc_manhattan_pose(const float* Rcm0, const float* MF_can, const float* Tcw_in, float* Tcw_out, int n) {
    for (int b = 0; b < n; b++) {
        const float* R0 = Rcm0 + 9 * b; const float* A = MF_can + 9 * b;
        float T[16];
        std::memcpy(T, Tcw_in + 16 * b, 64);
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) {
                const float t = R0[3 * i] * A[3 * j] + R0[3 * i + 1] * A[3 * j + 1] + R0[3 * i + 2] * A[3 * j + 2];   // (Rotation_cm * MF_can_T)(i, j)
                T[4 * j + i] = (float)((double)t * 1.0);                                                            // transposed into mTcw(j, i)
            }
        std::memcpy(Tcw_out + 16 * b, T, 64);
    }
}

}  // extern "C"
