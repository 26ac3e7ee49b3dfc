// oracle/cvprim.h — TEST INFRASTRUCTURE (CPU oracle), not product code.
//
// Plain-C++ restatements of the un-vendored OpenCV 3.4.x primitives the
// PlanarSLAM hot path calls.  The real OpenCV is not
// available in the authoring container, so these are written from the
// published algorithms of the pinned version (3.4.1, reference README.md:62)
// and parity against the real library is UNPINNED.
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
// link or call anything in oracle/.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

namespace orc {

// cvRound(): SSE cvtss2si / cvtsd2si => round-half-to-even (A6).
int cv_round(double v);
int cv_round(float v);
inline int cv_floor(double v) { int i = (int)v; return i - (i > v); }
inline int cv_ceil(double v) { int i = (int)v; return i + (i < v); }

// cv::fastAtan2(y, x) in degrees, [0,360) (A5).
float fast_atan2(float y, float x);

// reflect-101 border index map (A3).
inline int reflect101(int p, int n) {
    if (n == 1) return 0;
    while (p < 0 || p >= n) { p = p < 0 ? -p : 2 * n - 2 - p; }
    return p;
}

// cv::resize(src, dst, dsize, 0, 0, INTER_LINEAR) for CV_8UC1 (A2).
void resize_linear_u8(const uint8_t* src, int sw, int sh, int sstep,
                      uint8_t* dst, int dw, int dh, int dstep);

// cv::GaussianBlur(src, dst, Size(7,7), 2, 2, BORDER_REFLECT_101) for CV_8UC1 (A4).
// dst may alias src (the reference blurs a clone in place).
void gaussian7_s2_u8(const uint8_t* src, int w, int h, int sstep, uint8_t* dst, int dstep);
void gaussian7_s2_u8_fixedpoint341(const uint8_t* src, int w, int h, int sstep, uint8_t* dst, int dstep);   // the OpenCV 3.4.1 ufixedpoint16 variant, written out separately

struct FastKp { int x, y, score; };
// cv::FAST(img, kps, threshold, nonmaxSuppression) TYPE_9_16 on the ROI it is given (A1).
void fast9_16(const uint8_t* img, int w, int h, int step, int threshold, bool nms,
              std::vector<FastKp>& out);
// cornerScore<16>: largest threshold for which the pixel stays a 9/16 corner
// (>= threshold-1 by construction).
int fast_corner_score(const uint8_t* ptr, const int pixel[25], int threshold);

}  // namespace orc

namespace orc {
// Q8 taps of cv::getGaussianKernel(ksize, sigma) as the 8U fixed-point GaussianBlur uses them
// (modules/imgproc/src/smooth.cpp, 3.4.1 ufixedpoint16 path): cvRound(k[i] * 256).
void gaussian_taps_q8(int ksize, double sigma, int* taps);
// cv::GaussianBlur(src, dst, Size(ksize,ksize), sigma, sigma, BORDER_REFLECT_101) for CV_8UC1, ksize odd <= 31.
void gaussian_u8(const uint8_t* src, int w, int h, int sstep, int ksize, double sigma, uint8_t* dst, int dstep);
// cv::resize(src, dst, Size(), fx, fy, INTER_LINEAR_EXACT) for CV_8UC1 (resize.cpp resize_bitExact:
// interpolationLinear<uint8_t> with ufixedpoint16 coefficients, vertical pass in ufixedpoint32, round half up).
void resize_linear_exact_u8(const uint8_t* src, int sw, int sh, int sstep, double inv_scale_x, double inv_scale_y, uint8_t* dst, int dw,
                            int dh, int dstep);
// cv::Sobel(src, dst, CV_16S, dx, dy, 3) with BORDER_REFLECT_101 (dx,dy) in {(1,0),(0,1)}
void sobel3_s16(const uint8_t* src, int w, int h, int sstep, int dx, int dy, int16_t* dst);
}  // namespace orc

namespace orc {
// JacobiSVDImpl_<float>(At, W, Vt, m = n = n1 = 3, minval = FLT_MIN, eps = 2 FLT_EPSILON) (modules/core/src/lapack.cpp): one-sided Jacobi on
// the rows of At (= columns of A); on return the rows of At are the left singular vectors, W descending, Vt the right ones.
void jacobi_svd3_f32(float At[3][3], float W[3], float Vt[3][3]);
// cv::determinant of a 3x3 CV_32F matrix: the det3 macro of lapack.cpp (inner 2x2 products in double, result double)
double det3_f32(const float m[3][3]);
// cv::SVD for CV_64F (JacobiSVDImpl_<double>, OpenCV 3.4 core/src/lapack.cpp): At = the n columns of A stored as n rows of length m (m >= n), orthogonalised in
// place and normalised (the left singular vectors), W descending, Vt (n x n) the right singular vectors as rows
void jacobi_svd_f64(double* At, int astep, double* W, double* Vt, int vstep, int m, int n);
}  // namespace orc
