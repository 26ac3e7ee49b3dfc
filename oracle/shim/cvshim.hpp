// oracle/shim/cvshim.hpp — TEST INFRASTRUCTURE, not product code.
//
// A minimal stand-in for the handful of OpenCV 3.4 types and functions that the
// reference's hot-path translation units use, so that the REAL reference sources
// (compiled from /root/reference where they lie, never copied) can be built in a
// container that has no OpenCV.  Container types (Mat, Point_, KeyPoint, ...) carry
// only the semantics those files rely on; the image-processing functions forward to
// the restatements in oracle/cvprim.cpp.  Output goes to oracle/_ref/ only.
#pragma once
#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <vector>

#include "../cvprim.h"

typedef unsigned char uchar;
#define CV_PI 3.1415926535897932384626433832795
#define CV_8U 0
#define CV_8UC1 0
#define CV_8UC3 16
#define CV_32S 4
#define CV_32SC1 4
#define CV_32F 5
#define CV_32FC1 5
#define CV_64F 6
#define CV_16U 2

static inline int cvRound(double v) { return orc::cv_round(v); }
static inline int cvRound(float v) { return orc::cv_round(v); }
static inline int cvRound(int v) { return v; }
static inline int cvFloor(double v) { return orc::cv_floor(v); }
static inline int cvCeil(double v) { return orc::cv_ceil(v); }

namespace cv {

template <typename T> static inline T saturate_cast(float v) { return (T)v; }

template <typename T> struct Point_ {
    T x, y;
    Point_() : x(0), y(0) {}
    Point_(T _x, T _y) : x(_x), y(_y) {}
    template <typename U> Point_(const Point_<U>& p) : x((T)p.x), y((T)p.y) {}
};
typedef Point_<int> Point2i;
typedef Point_<int> Point;
typedef Point_<float> Point2f;
typedef Point_<double> Point2d;
template <typename T> static inline Point_<T>& operator*=(Point_<T>& a, float b) {
    a.x = (T)(a.x * b); a.y = (T)(a.y * b); return a;
}
template <typename T> static inline bool operator==(const Point_<T>& a, const Point_<T>& b) { return a.x == b.x && a.y == b.y; }
template <typename T> static inline Point_<T> operator/(const Point_<T>& a, double b) { return Point_<T>((T)(a.x / b), (T)(a.y / b)); }
template <typename T> static inline double norm(const Point_<T>& p) { return std::sqrt((double)p.x * p.x + (double)p.y * p.y); }

template <typename T> static inline Point_<T> operator+(const Point_<T>& a, const Point_<T>& b) { return Point_<T>((T)(a.x + b.x), (T)(a.y + b.y)); }
template <typename T> static inline Point_<T> operator-(const Point_<T>& a, const Point_<T>& b) { return Point_<T>((T)(a.x - b.x), (T)(a.y - b.y)); }
template <typename T> static inline Point_<T> operator*(const Point_<T>& a, double b) { return Point_<T>((T)(a.x * b), (T)(a.y * b)); }   // saturate_cast<T>(a.x * b)
template <typename T> struct Point3_ {
    T x, y, z;
    Point3_() : x(0), y(0), z(0) {}
    Point3_(T a, T b, T c) : x(a), y(b), z(c) {}
    T dot(const Point3_& o) const { return (T)(x * o.x + y * o.y + z * o.z); }
    Point3_ cross(const Point3_& o) const { return Point3_(y * o.z - z * o.y, z * o.x - x * o.z, x * o.y - y * o.x); }
};
template <typename T> static inline Point3_<T> operator+(const Point3_<T>& a, const Point3_<T>& b) { return Point3_<T>((T)(a.x + b.x), (T)(a.y + b.y), (T)(a.z + b.z)); }
template <typename T> static inline Point3_<T> operator-(const Point3_<T>& a, const Point3_<T>& b) { return Point3_<T>((T)(a.x - b.x), (T)(a.y - b.y), (T)(a.z - b.z)); }
template <typename T> static inline Point3_<T> operator*(const Point3_<T>& a, double b) { return Point3_<T>((T)(a.x * b), (T)(a.y * b), (T)(a.z * b)); }
template <typename T> static inline Point3_<T> operator*(double b, const Point3_<T>& a) { return Point3_<T>((T)(a.x * b), (T)(a.y * b), (T)(a.z * b)); }
template <typename T> static inline Point3_<T> operator/(const Point3_<T>& a, double b) { return Point3_<T>((T)(a.x / b), (T)(a.y / b), (T)(a.z / b)); }
template <typename T> static inline double norm(const Point3_<T>& p) { return std::sqrt((double)p.x * p.x + (double)p.y * p.y + (double)p.z * p.z); }
typedef Point3_<float> Point3f;
typedef Point3_<double> Point3d;
template <typename T, int N> struct Vec {
    T val[N];
    Vec() { for (int i = 0; i < N; i++) val[i] = T(0); }
    Vec(T a, T b, T c) { static_assert(N == 3, "Vec3 ctor"); val[0] = a; val[1] = b; val[2] = c; }
    explicit Vec(const T* p) { for (int i = 0; i < N; i++) val[i] = p[i]; }
    T& operator[](int i) { return val[i]; }
    const T& operator[](int i) const { return val[i]; }
};
typedef Vec<uchar, 3> Vec3b;
typedef Vec<double, 2> Vec2d;
struct Range { int start, end; Range(int s, int e) : start(s), end(e) {} };

struct Size { int width, height; Size() : width(0), height(0) {} Size(int w, int h) : width(w), height(h) {}
              int area() const { return width * height; } };
struct Rect { int x, y, width, height; Rect() : x(0), y(0), width(0), height(0) {}
              Rect(int _x, int _y, int w, int h) : x(_x), y(_y), width(w), height(h) {} };
struct Scalar { double v[4]; Scalar(double a = 0, double b = 0, double c = 0, double d = 0) { v[0] = a; v[1] = b; v[2] = c; v[3] = d; }
                double operator[](int i) const { return v[i]; } };

struct KeyPoint {
    Point2f pt; float size, angle, response; int octave, class_id;
    KeyPoint() : pt(0, 0), size(0), angle(-1), response(0), octave(0), class_id(-1) {}
    KeyPoint(float x, float y, float _size, float _angle = -1, float _response = 0, int _octave = 0, int _class_id = -1)
        : pt(x, y), size(_size), angle(_angle), response(_response), octave(_octave), class_id(_class_id) {}
};
static_assert(sizeof(KeyPoint) == 28, "cv::KeyPoint layout");

static inline int cvElemSize(int type) {
    int depth = type & 7, cn = (type >> 3) + 1;
    static const int sz[8] = {1, 1, 2, 2, 4, 4, 8, 0};
    return sz[depth] * cn;
}

struct MatZerosExpr { int rows, cols, type; };
struct MatOnesExpr { int rows, cols, type; };

struct MatStep {
    size_t v = 0;
    operator size_t() const { return v; }
};

struct MatExpr;   // cvalgebra.hpp (only the matcher harness uses matrix algebra)
class Mat {
public:
    int rows = 0, cols = 0;
    uchar* data = nullptr;
    MatStep step;
    Mat() {}
    Mat(int r, int c, int t) { create(r, c, t); }
    Mat(Size s, int t) { create(s.height, s.width, t); }
    Mat(int r, int c, int t, void* ext, size_t stp = 0) : rows(r), cols(c), data((uchar*)ext), type_(t) {
        step.v = stp ? stp : (size_t)c * cvElemSize(t);
    }
    void create(int r, int c, int t) {
        if (data && r == rows && c == cols && t == type_) return;
        type_ = t; rows = r; cols = c;
        step.v = (size_t)c * cvElemSize(t);
        size_t n = step.v * (size_t)r;
        owner_ = std::shared_ptr<uchar>((uchar*)std::malloc(n ? n : 1), std::free);
        data = owner_.get();
        parent_w_ = c; parent_h_ = r; off_x_ = off_y_ = 0;
    }
    void release() { *this = Mat(); }
    int type() const { return type_; }
    int depth() const { return type_ & 7; }
    int channels() const { return (type_ >> 3) + 1; }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    size_t step1() const { return step.v / (size_t)cvElemSize(type_ & 7); }
    size_t elemSize() const { return cvElemSize(type_); }
    bool isContinuous() const { return step.v == (size_t)cols * cvElemSize(type_); }
    bool isSubmatrix() const { return cols != parent_w_ || rows != parent_h_; }
    Size size() const { return Size(cols, rows); }
    Mat operator()(const Rect& r) const {
        Mat m = *this;
        m.data = data + (size_t)r.y * step.v + (size_t)r.x * cvElemSize(type_);
        m.rows = r.height; m.cols = r.width;
        m.off_x_ = off_x_ + r.x; m.off_y_ = off_y_ + r.y;
        return m;
    }
    Mat operator()(const Range& rr, const Range& cr) const { return (*this)(Rect(cr.start, rr.start, cr.end - cr.start, rr.end - rr.start)); }
    Mat rowRange(int a, int b) const { return (*this)(Rect(0, a, cols, b - a)); }
    Mat colRange(int a, int b) const { return (*this)(Rect(a, 0, b - a, rows)); }
    Mat row(int y) const { return rowRange(y, y + 1); }
    Mat col(int x) const { return colRange(x, x + 1); }
    inline MatExpr t() const;                 // cvalgebra.hpp
    inline double dot(const Mat& m) const;    // cvalgebra.hpp
    inline Mat cross(const Mat& m) const;     // cvalgebra.hpp
    inline Mat(const MatExpr& e);             // cvalgebra.hpp
    inline Mat& operator=(const MatExpr& e);  // cvalgebra.hpp
    Mat clone() const {
        Mat m(rows, cols, type_);
        for (int y = 0; y < rows; y++) std::memcpy(m.data + (size_t)y * m.step.v, data + (size_t)y * step.v, (size_t)cols * cvElemSize(type_));
        return m;
    }
    void copyTo(Mat& dst) const {
        dst.create(rows, cols, type_);
        for (int y = 0; y < rows; y++) std::memmove(dst.data + (size_t)y * dst.step.v, data + (size_t)y * step.v, (size_t)cols * cvElemSize(type_));
    }
    void copyTo(Mat&& dst) const { Mat d = dst; copyTo(d); }   // destination is a view expression (m.rowRange(..)): same shape, written in place
    template <typename T> T& at(int y, int x) { return *(T*)(data + (size_t)y * step.v + (size_t)x * sizeof(T)); }
    template <typename T> const T& at(int y, int x) const { return *(const T*)(data + (size_t)y * step.v + (size_t)x * sizeof(T)); }
    template <typename T> T& at(int i) { assert(isContinuous()); return ((T*)data)[i]; }
    template <typename T> const T& at(int i) const { assert(isContinuous()); return ((const T*)data)[i]; }
    uchar* ptr(int y = 0) { return data + (size_t)y * step.v; }
    const uchar* ptr(int y = 0) const { return data + (size_t)y * step.v; }
    template <typename T> T* ptr(int y = 0) { return (T*)(data + (size_t)y * step.v); }
    template <typename T> const T* ptr(int y = 0) const { return (const T*)(data + (size_t)y * step.v); }
    static MatZerosExpr zeros(int r, int c, int t) { return MatZerosExpr{r, c, t}; }
    static MatZerosExpr zeros(Size s, int t) { return MatZerosExpr{s.height, s.width, t}; }
    // cv::Mat::operator=(const MatExpr&) for zeros: create() (a no-op when shape/type match, so a
    // view stays a view) followed by setTo(0).
    Mat& operator=(const MatZerosExpr& e) {
        create(e.rows, e.cols, e.type);
        for (int y = 0; y < rows; y++) std::memset(data + (size_t)y * step.v, 0, (size_t)cols * cvElemSize(type_));
        return *this;
    }
    Mat(const MatZerosExpr& e) { *this = e; }
    // setTo for the element types the reference uses: CV_8U, CV_32S scalars and CV_8UC3 colours
    Mat& setTo(int v) {
        for (int y = 0; y < rows; y++)
            for (int x = 0; x < cols; x++) {
                if (type_ == CV_32SC1) at<int>(y, x) = v;
                else if (type_ == CV_8UC1) at<uchar>(y, x) = (uchar)v;
                else { assert(type_ == CV_8UC3); uchar* p = data + (size_t)y * step.v + 3 * (size_t)x; p[0] = (uchar)v; p[1] = p[2] = 0; }
            }
        return *this;
    }
    Mat& setTo(const Vec3b& c) {
        assert(type_ == CV_8UC3);
        for (int y = 0; y < rows; y++)
            for (int x = 0; x < cols; x++) { uchar* p = data + (size_t)y * step.v + 3 * (size_t)x; p[0] = c[0]; p[1] = c[1]; p[2] = c[2]; }
        return *this;
    }
    static MatOnesExpr ones(int r, int c, int t) { return MatOnesExpr{r, c, t}; }
    static Mat eye(int r, int c, int t) {   // CV_32F only (what src/Optimizer.cc / Converter.cc build)
        assert(t == CV_32F);
        Mat m(r, c, t);
        for (int y = 0; y < r; y++) for (int x = 0; x < c; x++) m.at<float>(y, x) = (x == y) ? 1.f : 0.f;
        return m;
    }
    Mat& operator=(const MatOnesExpr& e) { create(e.rows, e.cols, e.type); return setTo(1); }
    // geometry of the view inside its allocation (for copyMakeBorder without BORDER_ISOLATED)
    int parent_w_ = 0, parent_h_ = 0, off_x_ = 0, off_y_ = 0;
private:
    int type_ = 0;
    std::shared_ptr<uchar> owner_;
};

class _InputArray {
public:
    _InputArray(const Mat& m) : m_(&m) {}
    bool empty() const { return m_->empty(); }
    Mat getMat() const { return *m_; }
private:
    const Mat* m_;
};
class _OutputArray {
public:
    _OutputArray(Mat& m) : m_(&m) {}
    void release() const { m_->release(); }
    void create(int r, int c, int t) const { m_->create(r, c, t); }
    void create(Size s, int t) const { m_->create(s.height, s.width, t); }
    Mat getMat() const { return *m_; }
    Mat& getMatRef() const { return *m_; }
private:
    Mat* m_;
};
typedef const _InputArray& InputArray;
typedef const _OutputArray& OutputArray;
typedef const _OutputArray& InputOutputArray;

enum { BORDER_CONSTANT = 0, BORDER_REPLICATE = 1, BORDER_REFLECT = 2, BORDER_WRAP = 3, BORDER_REFLECT_101 = 4,
       BORDER_REFLECT101 = 4, BORDER_DEFAULT = 4, BORDER_ISOLATED = 16 };
enum { INTER_NEAREST = 0, INTER_LINEAR = 1, INTER_CUBIC = 2, INTER_AREA = 3 };

static inline int64_t getTickCount() { return 0; }
static inline double getTickFrequency() { return 1.0; }

// Only referenced from dead code (ORBextractor::ComputeKeyPointsOld, never called).
struct KeyPointsFilter {
    static void retainBest(std::vector<KeyPoint>&, int) { std::abort(); }
};

static inline float fastAtan2(float y, float x) { return orc::fast_atan2(y, x); }

static inline void FAST(InputArray image, std::vector<KeyPoint>& kps, int threshold, bool nms = true) {
    Mat m = image.getMat();
    assert(m.type() == CV_8UC1);
    std::vector<orc::FastKp> out;
    orc::fast9_16(m.data, m.cols, m.rows, (int)(size_t)m.step, threshold, nms, out);
    kps.clear();
    for (const orc::FastKp& k : out) kps.push_back(KeyPoint((float)k.x, (float)k.y, 7.f, -1, (float)k.score));
}

static inline void resize(InputArray src, OutputArray dst, Size dsize, double fx = 0, double fy = 0, int interp = INTER_LINEAR) {
    assert(fx == 0 && fy == 0 && interp == INTER_LINEAR);
    Mat s = src.getMat();
    assert(s.type() == CV_8UC1);
    dst.create(dsize, s.type());
    Mat d = dst.getMat();
    orc::resize_linear_u8(s.data, s.cols, s.rows, (int)(size_t)s.step, d.data, d.cols, d.rows, (int)(size_t)d.step);
}

static inline void copyMakeBorder(InputArray src, OutputArray dst, int top, int bottom, int left, int right, int borderType) {
    Mat s = src.getMat();
    assert(s.type() == CV_8UC1 && (borderType & ~BORDER_ISOLATED) == BORDER_REFLECT_101);
    dst.create(s.rows + top + bottom, s.cols + left + right, s.type());
    Mat d = dst.getMat();
    // Without BORDER_ISOLATED a sub-matrix source borrows real pixels from its parent first.
    // The reference only passes whole images there (src/ORBextractor.cc:1127), asserted here.
    if (!(borderType & BORDER_ISOLATED)) assert(!s.isSubmatrix());
    Mat tmp = s.clone();
    for (int y = 0; y < d.rows; y++) {
        int sy = orc::reflect101(y - top, tmp.rows);
        for (int x = 0; x < d.cols; x++) d.data[(size_t)y * d.step + x] = tmp.data[(size_t)sy * tmp.step + orc::reflect101(x - left, tmp.cols)];
    }
}

static inline void GaussianBlur(InputArray src, OutputArray dst, Size ksize, double sx, double sy = 0, int borderType = BORDER_DEFAULT) {
    Mat s = src.getMat();
    assert(s.type() == CV_8UC1 && ksize.width == 7 && ksize.height == 7 && sx == 2 && sy == 2 && borderType == BORDER_REFLECT_101);
    dst.create(s.size(), s.type());
    Mat d = dst.getMat();
    orc::gaussian7_s2_u8(s.data, s.cols, s.rows, (int)(size_t)s.step, d.data, (int)(size_t)d.step);
}

// cv::norm(a, b, NORM_HAMMING) as MapLine::ComputeDistinctiveDescriptors calls it on two descriptor rows: the number of differing bits
#ifndef CVSHIM_ALGEBRA   // (cvalgebra.hpp declares NORM_HAMMING for its BFMatcher)
enum { NORM_HAMMING = 6 };
#endif
static inline double norm(const Mat& a, const Mat& b, int normType) {
    (void)normType;
    const size_t n = (size_t)a.rows * a.cols * a.elemSize();
    int d = 0;
    for (size_t i = 0; i < n; i++) d += __builtin_popcount((unsigned)(a.data[i] ^ b.data[i]));
    return (double)d;
}

}  // namespace cv
#ifdef CVSHIM_ALGEBRA
#include "cvalgebra.hpp"
#endif
