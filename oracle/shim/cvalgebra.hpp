// oracle/shim/cvalgebra.hpp — TEST INFRASTRUCTURE.  The slice of cv::Mat matrix algebra the reference's matchers use
// (src/ORBmatcher.cc, src/PlaneMatcher.cpp), so that those translation units compile and run unmodified in oracle/_ref.
// Restated from OpenCV 3.4 (core/src/matmul.cpp, matop.cpp) by reading — UNPINNED against the library:
//   * A*B [+ C] is one lazy cv::gemm.  For CV_32F with no transpose flag and inner length 2..4 equal to the output's width or
//     height it takes the small-matrix path: products summed in float, left to right, then (float)(t*alpha + c*beta) in
//     double.  Otherwise (e.g. -A.t()*B) GEMMSingleMul<float,double>: products accumulated in double, (float)(s*alpha + c*beta).
//   * alpha*A, A/s, -A: convertTo with a scale = cvtScale32f, the scale cast to float and multiplied in float; alpha*A + beta*C without a
//     product: addWeighted in float.
//   * A+B, A-B: float.   norm(), dot(): double accumulation.
//   * BFMatcher(NORM_HAMMING).match: smallest distance, lowest train index on ties.
#pragma once
#include <algorithm>
#include <climits>
#include <cmath>
#include <vector>

namespace cv {

struct MatExpr {
    Mat a, b, c;
    bool tr = false, has_b = false, has_c = false;
    double alpha = 1, beta = 1;
    // CV_64F operands (the 3-D line code, src/LineExtractor.cpp): every product element is the k = 0, 1, 2 ... sum of double products, in order, on
    // both the small-matrix path and the general path (GEMMSingleMul<double, double>; its 4-way unrolled block only starts at inner length 4, which the
    // 3 x 3 products never reach); alpha * A = cvtScale64f.
    Mat eval64() const {
        const int ar = tr ? a.cols : a.rows, ac = tr ? a.rows : a.cols;
        auto A = [&](int i, int k) -> double { return tr ? a.at<double>(k, i) : a.at<double>(i, k); };
        if (!has_b) {
            Mat d(ar, ac, CV_64F);
            for (int i = 0; i < ar; i++)
                for (int j = 0; j < ac; j++) d.at<double>(i, j) = has_c ? (A(i, j) * alpha + c.at<double>(i, j) * beta) : (alpha == 1 ? A(i, j) : A(i, j) * alpha);
            return d;
        }
        assert(ac == b.rows && b.type() == CV_64F);
        Mat d(ar, b.cols, CV_64F);
        for (int i = 0; i < ar; i++)
            for (int j = 0; j < b.cols; j++) {
                double s0 = 0;
                for (int k = 0; k < ac; k++) s0 += A(i, k) * b.at<double>(k, j);
                d.at<double>(i, j) = has_c ? s0 * alpha + c.at<double>(i, j) * beta : s0 * alpha;
            }
        return d;
    }
    Mat eval() const {
        if (a.type() == CV_64F) return eval64();
        const int ar = tr ? a.cols : a.rows, ac = tr ? a.rows : a.cols;
        auto A = [&](int i, int k) -> float { return tr ? a.at<float>(k, i) : a.at<float>(i, k); };
        if (!has_b) {
            // alpha*A: convertTo(CV_32F, alpha) = cvtScale32f, scale cast to float, float multiply.  alpha*A + beta*C: addWeighted on
            // CV_32F, weights cast to float, a*alpha + c*beta in float.
            Mat d(ar, ac, CV_32F);
            const float fa = (float)alpha, fb = (float)beta;
            for (int i = 0; i < ar; i++)
                for (int j = 0; j < ac; j++) d.at<float>(i, j) = has_c ? (A(i, j) * fa + c.at<float>(i, j) * fb) : (A(i, j) * fa);
            return d;
        }
        assert(ac == b.rows);
        const int len = ac, dw = b.cols, dh = ar;
        Mat d(dh, dw, CV_32F);
        const bool small = !tr && len >= 2 && len <= 4 && (len == dw || len == dh);
        for (int i = 0; i < dh; i++)
            for (int j = 0; j < dw; j++) {
                const double cv_ = has_c ? (double)c.at<float>(i, j) * beta : 0.0;
                if (small) {
                    float t = A(i, 0) * b.at<float>(0, j);
                    for (int k = 1; k < len; k++) t = t + A(i, k) * b.at<float>(k, j);
                    d.at<float>(i, j) = (float)((double)t * alpha + cv_);
                } else {
                    double s = 0;
                    for (int k = 0; k < len; k++) s += (double)A(i, k) * (double)b.at<float>(k, j);
                    d.at<float>(i, j) = (float)(s * alpha + cv_);
                }
            }
        return d;
    }
    operator Mat() const { return eval(); }
    template <typename T> T at(int i) const { return eval().at<T>(i); }
    template <typename T> T at(int y, int x) const { return eval().at<T>(y, x); }
    MatExpr t() const { assert(!has_b && !has_c); MatExpr e = *this; e.tr = !e.tr; return e; }
    double dot(const Mat& m) const { return eval().dot(m); }
};

inline MatExpr Mat::t() const { MatExpr e; e.a = *this; e.tr = true; return e; }
inline Mat::Mat(const MatExpr& e) { *this = e.eval(); }
inline Mat& Mat::operator=(const MatExpr& e) { return *this = e.eval(); }
inline double Mat::dot(const Mat& m) const {
    double s = 0;
    for (int y = 0; y < rows; y++) for (int x = 0; x < cols; x++) s += (double)at<float>(y, x) * (double)m.at<float>(y, x);
    return s;
}

static inline MatExpr operator*(const Mat& a, const Mat& b) { MatExpr e; e.a = a; e.b = b; e.has_b = true; return e; }
static inline MatExpr operator*(const MatExpr& x, const Mat& b) {
    if (x.has_b || x.has_c) return x.eval() * b;
    MatExpr e = x; e.b = b; e.has_b = true; return e;
}
static inline MatExpr operator*(const Mat& a, const MatExpr& y) { return a * y.eval(); }
static inline MatExpr operator*(const MatExpr& x, const MatExpr& y) { return x.eval() * y.eval(); }
static inline MatExpr operator*(double s, const Mat& a) { MatExpr e; e.a = a; e.alpha = s; return e; }
static inline MatExpr operator*(const Mat& a, double s) { return s * a; }
static inline MatExpr operator*(double s, const MatExpr& x) { if (x.has_c) return s * x.eval(); MatExpr e = x; e.alpha *= s; return e; }
static inline MatExpr operator/(const Mat& a, double s) { return (1.0 / s) * a; }
static inline MatExpr operator-(const Mat& a) { return -1.0 * a; }
static inline MatExpr operator-(const MatExpr& x) { return -1.0 * x; }
static inline MatExpr operator+(const MatExpr& x, const Mat& c) {
    if (x.has_c) { MatExpr e; e.a = x.eval(); e.c = c; e.has_c = true; return e; }
    MatExpr e = x; e.c = c; e.has_c = true; e.beta = 1; return e;
}
static inline MatExpr operator-(const MatExpr& x, const Mat& c) {
    if (x.has_c) { MatExpr e; e.a = x.eval(); e.c = c; e.has_c = true; e.beta = -1; return e; }
    MatExpr e = x; e.c = c; e.has_c = true; e.beta = -1; return e;
}
static inline Mat operator+(const Mat& a, const Mat& b) {
    Mat d(a.rows, a.cols, CV_32F);
    for (int y = 0; y < a.rows; y++) for (int x = 0; x < a.cols; x++) d.at<float>(y, x) = a.at<float>(y, x) + b.at<float>(y, x);
    return d;
}
static inline Mat operator-(const Mat& a, const Mat& b) {
    Mat d(a.rows, a.cols, CV_32F);
    for (int y = 0; y < a.rows; y++) for (int x = 0; x < a.cols; x++) d.at<float>(y, x) = a.at<float>(y, x) - b.at<float>(y, x);
    return d;
}
static inline double norm(const Mat& m) {
    double s = 0;
    for (int y = 0; y < m.rows; y++) for (int x = 0; x < m.cols; x++) s += (double)m.at<float>(y, x) * (double)m.at<float>(y, x);
    return std::sqrt(s);
}
static inline double norm(const MatExpr& e) { return norm(e.eval()); }
// Mat::cross for two CV_32F 3-vectors: float products and differences
inline Mat Mat::cross(const Mat& m) const {
    Mat d(rows, cols, CV_32F);
    const float a0 = at<float>(0), a1 = at<float>(1), a2 = at<float>(2), b0 = m.at<float>(0), b1 = m.at<float>(1), b2 = m.at<float>(2);
    d.at<float>(0) = a1 * b2 - a2 * b1; d.at<float>(1) = a2 * b0 - a0 * b2; d.at<float>(2) = a0 * b1 - a1 * b0;
    return d;
}
// cv::sum / cv::trace: double accumulation, element order
static inline Scalar sum(const Mat& m) {
    double s = 0;
    for (int y = 0; y < m.rows; y++) for (int x = 0; x < m.cols; x++) s += (double)m.at<float>(y, x);
    return Scalar(s);
}
static inline Scalar sum(const MatExpr& e) { return sum(e.eval()); }
static inline Scalar trace(const Mat& m) {
    double s = 0;
    for (int i = 0; i < std::min(m.rows, m.cols); i++) s += (double)m.at<float>(i, i);
    return Scalar(s);
}
static inline Scalar trace(const MatExpr& e) { return trace(e.eval()); }
static inline double determinant(const Mat& m) {
    assert(m.rows == 3 && m.cols == 3 && m.type() == CV_32F);
    float a[3][3];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) a[i][j] = m.at<float>(i, j);
    return orc::det3_f32(a);
}
// cv::SVD::compute(src, w, u, vt) for a 3x3 CV_32F matrix: transpose, JacobiSVDImpl_<float>, u = transposed rows, vt as computed
class SVD {
public:
    enum { MODIFY_A = 1, NO_UV = 2, FULL_UV = 4 };
    Mat u, w, vt;
    SVD() {}
    SVD(const Mat& src, int flags = 0) { compute(src, w, u, vt, flags); }
    SVD(const MatExpr& src, int flags = 0) { compute(src.eval(), w, u, vt, flags); }
    static void compute(const Mat& src, Mat& w, Mat& u, Mat& vt, int = 0) {
        if (src.type() == CV_64F) {      // m x n with m >= n: temp_a = src^T (n rows of length m), JacobiSVDImpl_<double>, u = transposed rows, vt as computed
            const int m = src.rows, n = src.cols;
            assert(m >= n);
            std::vector<double> At((size_t)n * m), W(n), Vt((size_t)n * n);
            for (int i = 0; i < n; i++) for (int k = 0; k < m; k++) At[(size_t)i * m + k] = src.at<double>(k, i);
            orc::jacobi_svd_f64(At.data(), m, W.data(), Vt.data(), n, m, n);
            w.create(n, 1, CV_64F); u.create(m, n, CV_64F); vt.create(n, n, CV_64F);
            for (int i = 0; i < n; i++) { w.at<double>(i) = W[i]; for (int k = 0; k < m; k++) u.at<double>(k, i) = At[(size_t)i * m + k]; for (int j = 0; j < n; j++) vt.at<double>(i, j) = Vt[(size_t)i * n + j]; }
            return;
        }
        assert(src.rows == 3 && src.cols == 3 && src.type() == CV_32F);
        float At[3][3], W[3], Vt[3][3];
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) At[i][j] = src.at<float>(j, i);
        orc::jacobi_svd3_f32(At, W, Vt);
        w.create(3, 1, CV_32F); u.create(3, 3, CV_32F); vt.create(3, 3, CV_32F);
        for (int i = 0; i < 3; i++) { w.at<float>(i) = W[i]; for (int j = 0; j < 3; j++) { u.at<float>(i, j) = At[j][i]; vt.at<float>(i, j) = Vt[i][j]; } }
    }
};
static inline void transpose(const Mat& src, Mat& dst) {
    Mat d(src.cols, src.rows, CV_32F);
    for (int y = 0; y < src.rows; y++) for (int x = 0; x < src.cols; x++) d.at<float>(x, y) = src.at<float>(y, x);
    dst = d;
}

// debug prints only (std::cout << mat.t()): the text is never compared
template <class OS> static inline OS& operator<<(OS& os, const Mat& m) {
    os << "[";
    for (int y = 0; y < m.rows; y++) { for (int x = 0; x < m.cols; x++) os << (x ? ", " : "") << m.at<float>(y, x); os << (y + 1 < m.rows ? ";\n " : ""); }
    os << "]";
    return os;
}
template <class OS> static inline OS& operator<<(OS& os, const MatExpr& e) { return os << e.eval(); }

// (Mat_<float>(r, c) << a, b, ...) as used by the reference
template <typename T> struct MatCommaInit {
    Mat m; int i = 0;
    template <typename U> MatCommaInit& operator,(U v) { ((T*)m.data)[i++] = (T)v; return *this; }
    operator Mat() const { return m; }
};
template <typename T> struct Mat_ : public Mat {
    Mat_(int r, int c) : Mat(r, c, sizeof(T) == 4 ? CV_32F : CV_64F) {}
    template <typename U> MatCommaInit<T> operator<<(U v) { MatCommaInit<T> ci; ci.m = *this; ((T*)ci.m.data)[ci.i++] = (T)v; return ci; }
};

struct DMatch { int queryIdx = -1, trainIdx = -1, imgIdx = -1; float distance = 0; };
enum { NORM_HAMMING = 6 };
class BFMatcher {
public:
    explicit BFMatcher(int normType = NORM_HAMMING, bool crossCheck = false) { assert(normType == NORM_HAMMING && !crossCheck); }
    void match(const Mat& q, const Mat& t, std::vector<DMatch>& out) const {
        out.clear();
        if (t.rows == 0) return;
        for (int i = 0; i < q.rows; i++) {
            int best = -1, bd = 1 << 30;
            for (int j = 0; j < t.rows; j++) {
                int d = 0;
                for (int k = 0; k < 32; k++) d += __builtin_popcount(q.ptr(i)[k] ^ t.ptr(j)[k]);
                if (d < bd) { bd = d; best = j; }
            }
            DMatch m; m.queryIdx = i; m.trainIdx = best; m.imgIdx = 0; m.distance = (float)bd;
            out.push_back(m);
        }
    }
    // knnMatch: per query the k smallest distances, ascending, lowest train index first on ties
    void knnMatch(const Mat& q, const Mat& t, std::vector<std::vector<DMatch>>& out, int k) const {
        out.clear();
        for (int i = 0; i < q.rows; i++) {
            std::vector<DMatch> all;
            for (int j = 0; j < t.rows; j++) {
                int d = 0;
                for (int kk = 0; kk < 32; kk++) d += __builtin_popcount(q.ptr(i)[kk] ^ t.ptr(j)[kk]);
                DMatch m; m.queryIdx = i; m.trainIdx = j; m.imgIdx = 0; m.distance = (float)d;
                all.push_back(m);
            }
            std::stable_sort(all.begin(), all.end(), [](const DMatch& a, const DMatch& b) { return a.distance < b.distance; });
            if ((int)all.size() > k) all.resize(k);
            out.push_back(all);
        }
    }
};

}  // namespace cv
