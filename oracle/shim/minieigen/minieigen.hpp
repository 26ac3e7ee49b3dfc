// oracle/shim/minieigen/minieigen.hpp — TEST INFRASTRUCTURE, not product code.
//
// A small, eagerly evaluated stand-in for the part of Eigen 3.3's API that the reference's g2o side uses, so that the REAL
// reference sources — Thirdparty/g2o/g2o/{core,types,solvers}, g2oAddition/*.h, include/EdgeLine.h, src/Optimizer.cc,
// src/Converter.cc — compile where they lie under /root/reference and run in oracle/_ref/ref_opt (Eigen itself is not in
// this image and cannot be installed).  Nothing here is derived from Eigen's sources; what must agree with Eigen for the
// 1e-5 pose tolerance are the published algorithms, restated: products sum k = 0..n-1, Quaternion <-> rotation matrix,
// AngleAxis -> matrix, quaternion product / vector rotation, pivoted LDLT (largest |diagonal| pivot), closed-form 2x2 / 3x3
// inverses.  Expression templates do not exist: every operator returns a plain Matrix, views (block / head / col /
// transpose / Map) are (pointer, strides) windows onto the object they were taken from.
#pragma once
#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstddef>
#include <cstring>
#include <iostream>
#include <limits>
#include <memory>
#include <type_traits>
#include <vector>

#define EIGEN_MAKE_ALIGNED_OPERATOR_NEW
#define EIGEN_MAKE_ALIGNED_OPERATOR_NEW_IF(x)
#define EIGEN_DEFINE_STL_VECTOR_SPECIALIZATION(...)
#define EIGEN_STRONG_INLINE inline
#define EIGEN_WORLD_VERSION 3
#define EIGEN_MAJOR_VERSION 3
#define EIGEN_MINOR_VERSION 4
#define EIGEN_VERSION_AT_LEAST(x, y, z) (EIGEN_WORLD_VERSION > x || (EIGEN_WORLD_VERSION >= x && (EIGEN_MAJOR_VERSION > y || (EIGEN_MAJOR_VERSION >= y && EIGEN_MINOR_VERSION >= z))))
#define MINIEIGEN 1

namespace Eigen {

typedef std::ptrdiff_t Index;
typedef std::ptrdiff_t DenseIndex;
const int Dynamic = -1;
const int Infinity = -1;
enum StorageOptions { ColMajor = 0, RowMajor = 1, AutoAlign = 0, DontAlign = 2 };
enum AlignmentType { Unaligned = 0, Aligned = 16 };
const unsigned int AlignedBit = 0x80;
enum UpLoType { Lower = 1, Upper = 2, UnitDiag = 4, ZeroDiag = 8, UnitLower = 5, UnitUpper = 6, StrictlyLower = 9, StrictlyUpper = 10, SelfAdjoint = 16, Symmetric = 16 };
enum ComputationInfo { Success = 0, NumericalIssue = 1, NoConvergence = 2, InvalidInput = 3 };
enum TransformTraits { Isometry = 0x1, Affine = 0x2, AffineCompact = 0x10 | Affine, Projective = 0x20 };
enum DecompositionOptions { ComputeFullU = 0x04, ComputeThinU = 0x08, ComputeFullV = 0x10, ComputeThinV = 0x20 };

template <class T> using aligned_allocator = std::allocator<T>;
inline void initParallel() {}

template <class T> struct NumTraits {
    static T epsilon() { return std::numeric_limits<T>::epsilon(); }
    static T dummy_precision() { return std::is_same<T, float>::value ? T(1e-5) : T(1e-12); }
    static T highest() { return (std::numeric_limits<T>::max)(); }
    static T lowest() { return std::numeric_limits<T>::lowest(); }
    typedef T Real;
};

template <class D> class MatrixBase;
template <class T, int R, int C, int Opt = 0, int MR = R, int MC = C> class Matrix;
template <class S, int R, int C> class View;
template <class M, int MapOpt = Unaligned, class Stride = void> class Map;
template <class T> class Quaternion;
template <class T> class AngleAxis;
template <class T, int Dim, int Mode, int Opt = 0> class Transform;

namespace internal {
template <class D> struct traits;
template <class T, int R, int C, int Opt, int MR, int MC> struct traits<Matrix<T, R, C, Opt, MR, MC>> {
    typedef T Scalar;
    enum { Rows = R, Cols = C };
};
template <class S, int R, int C> struct traits<View<S, R, C>> {
    typedef typename std::remove_const<S>::type Scalar;
    enum { Rows = R, Cols = C };
};
template <class M, int O, class St> struct traits<Map<M, O, St>> {
    typedef typename traits<typename std::remove_const<M>::type>::Scalar Scalar;
    enum { Rows = traits<typename std::remove_const<M>::type>::Rows, Cols = traits<typename std::remove_const<M>::type>::Cols };
};
template <int A, int B> struct pick { enum { value = (A == Dynamic ? B : A) }; };
template <class U> struct is_scalar : std::integral_constant<bool, std::is_arithmetic<U>::value> {};
}  // namespace internal

// ---- comma initialiser (row by row, blocks allowed) ----
template <class D> struct CommaInitializer {
    typedef typename internal::traits<D>::Scalar Scalar;
    D& m;
    Index row, col, blockRows;
    template <class U, typename std::enable_if<internal::is_scalar<U>::value, int>::type = 0> CommaInitializer(D& m_, const U& s) : m(m_), row(0), col(1), blockRows(1) { m.coeffRef(0, 0) = Scalar(s); }
    template <class O> CommaInitializer(D& m_, const MatrixBase<O>& o) : m(m_), row(0), col(o.cols()), blockRows(o.rows()) {
        for (Index i = 0; i < o.rows(); i++) for (Index j = 0; j < o.cols(); j++) m.coeffRef(i, j) = o.coeff(i, j);
    }
    template <class U> typename std::enable_if<internal::is_scalar<U>::value, CommaInitializer&>::type operator,(const U& s) {
        if (col == m.cols()) { row += blockRows; col = 0; blockRows = 1; }
        m.coeffRef(row, col++) = Scalar(s);
        return *this;
    }
    template <class O> CommaInitializer& operator,(const MatrixBase<O>& o) {
        if (col == m.cols()) { row += blockRows; col = 0; blockRows = o.rows(); }
        for (Index i = 0; i < o.rows(); i++) for (Index j = 0; j < o.cols(); j++) m.coeffRef(row + i, col + j) = o.coeff(i, j);
        col += o.cols();
        return *this;
    }
    D& finished() { return m; }
};

template <class D> struct ArrayProxy {
    typedef typename internal::traits<D>::Scalar Scalar;
    D& m;
    explicit ArrayProxy(D& m_) : m(m_) {}
    ArrayProxy& operator+=(const Scalar& s) { for (Index j = 0; j < m.cols(); j++) for (Index i = 0; i < m.rows(); i++) m.coeffRef(i, j) += s; return *this; }
    ArrayProxy& operator-=(const Scalar& s) { for (Index j = 0; j < m.cols(); j++) for (Index i = 0; i < m.rows(); i++) m.coeffRef(i, j) -= s; return *this; }
    ArrayProxy& operator*=(const Scalar& s) { for (Index j = 0; j < m.cols(); j++) for (Index i = 0; i < m.rows(); i++) m.coeffRef(i, j) *= s; return *this; }
};

template <class MatrixType> class LDLT;
template <class MatrixType> class LLT;
template <class MatrixType> class PartialPivLU;

// ---------------------------------------------------------------------------------------------------------------------
template <class D> class MatrixBase {
public:
    typedef typename internal::traits<D>::Scalar Scalar;
    typedef Scalar RealScalar;
    typedef Eigen::Index Index;
    enum { RowsAtCompileTime = internal::traits<D>::Rows, ColsAtCompileTime = internal::traits<D>::Cols,
           SizeAtCompileTime = (RowsAtCompileTime == Dynamic || ColsAtCompileTime == Dynamic) ? Dynamic : RowsAtCompileTime * ColsAtCompileTime,
           IsVectorAtCompileTime = (RowsAtCompileTime == 1 || ColsAtCompileTime == 1) };
    typedef Matrix<Scalar, RowsAtCompileTime, ColsAtCompileTime> PlainObject;
    typedef Matrix<Scalar, ColsAtCompileTime, RowsAtCompileTime> TransposedPlain;

    D& derived() { return *static_cast<D*>(this); }
    const D& derived() const { return *static_cast<const D*>(this); }
    Index size() const { return derived().rows() * derived().cols(); }
    Index rows() const { return derived().rows(); }
    Index cols() const { return derived().cols(); }
    Scalar coeff(Index i, Index j) const { return derived().coeff(i, j); }
    Scalar& coeffRef(Index i, Index j) { return derived().coeffRef(i, j); }

    // ---- coefficient access ----
    Scalar operator()(Index i, Index j) const { return derived().coeff(i, j); }
    Scalar& operator()(Index i, Index j) { return derived().coeffRef(i, j); }
    Scalar coeffv(Index i) const { return derived().cols() == 1 ? derived().coeff(i, 0) : derived().coeff(0, i); }
    Scalar& coeffRefv(Index i) { return derived().cols() == 1 ? derived().coeffRef(i, 0) : derived().coeffRef(0, i); }
    Scalar operator()(Index i) const { return coeffv(i); }
    Scalar& operator()(Index i) { return coeffRefv(i); }
    Scalar operator[](Index i) const { return coeffv(i); }
    Scalar& operator[](Index i) { return coeffRefv(i); }
    Scalar x() const { return coeffv(0); }
    Scalar y() const { return coeffv(1); }
    Scalar z() const { return coeffv(2); }
    Scalar w() const { return coeffv(3); }
    Scalar& x() { return coeffRefv(0); }
    Scalar& y() { return coeffRefv(1); }
    Scalar& z() { return coeffRefv(2); }
    Scalar& w() { return coeffRefv(3); }
    Scalar value() const { return derived().coeff(0, 0); }

    // ---- views ----
    template <int BR, int BC> View<Scalar, BR, BC> mkview(Index i, Index j, Index r, Index c) {
        return View<Scalar, BR, BC>(derived().data() + i * derived().rowStride() + j * derived().colStride(), r, c, derived().rowStride(), derived().colStride());
    }
    template <int BR, int BC> View<const Scalar, BR, BC> mkview(Index i, Index j, Index r, Index c) const {
        return View<const Scalar, BR, BC>(derived().data() + i * derived().rowStride() + j * derived().colStride(), r, c, derived().rowStride(), derived().colStride());
    }
    View<Scalar, Dynamic, Dynamic> block(Index i, Index j, Index r, Index c) { return mkview<Dynamic, Dynamic>(i, j, r, c); }
    View<const Scalar, Dynamic, Dynamic> block(Index i, Index j, Index r, Index c) const { return mkview<Dynamic, Dynamic>(i, j, r, c); }
    template <int BR, int BC> View<Scalar, BR, BC> block(Index i, Index j) { return mkview<BR, BC>(i, j, BR, BC); }
    template <int BR, int BC> View<const Scalar, BR, BC> block(Index i, Index j) const { return mkview<BR, BC>(i, j, BR, BC); }
    template <int BR, int BC> View<Scalar, BR, BC> topLeftCorner() { return mkview<BR, BC>(0, 0, BR, BC); }
    template <int BR, int BC> View<const Scalar, BR, BC> topLeftCorner() const { return mkview<BR, BC>(0, 0, BR, BC); }
    template <int BR, int BC> View<Scalar, BR, BC> topRightCorner() { return mkview<BR, BC>(0, derived().cols() - BC, BR, BC); }
    template <int BR, int BC> View<const Scalar, BR, BC> topRightCorner() const { return mkview<BR, BC>(0, derived().cols() - BC, BR, BC); }
    View<Scalar, Dynamic, Dynamic> topLeftCorner(Index r, Index c) { return mkview<Dynamic, Dynamic>(0, 0, r, c); }
    View<const Scalar, Dynamic, Dynamic> topLeftCorner(Index r, Index c) const { return mkview<Dynamic, Dynamic>(0, 0, r, c); }
    View<Scalar, RowsAtCompileTime, 1> col(Index j) { return mkview<RowsAtCompileTime, 1>(0, j, derived().rows(), 1); }
    View<const Scalar, RowsAtCompileTime, 1> col(Index j) const { return mkview<RowsAtCompileTime, 1>(0, j, derived().rows(), 1); }
    View<Scalar, 1, ColsAtCompileTime> row(Index i) { return mkview<1, ColsAtCompileTime>(i, 0, 1, derived().cols()); }
    View<const Scalar, 1, ColsAtCompileTime> row(Index i) const { return mkview<1, ColsAtCompileTime>(i, 0, 1, derived().cols()); }
    // vector segments: column vectors unless the object is a row vector at compile time
    enum { IsRowVec = (RowsAtCompileTime == 1 && ColsAtCompileTime != 1) };
    template <int N> View<Scalar, IsRowVec ? 1 : N, IsRowVec ? N : 1> segment(Index i) {
        return IsRowVec ? mkview<IsRowVec ? 1 : N, IsRowVec ? N : 1>(0, i, 1, N) : mkview<IsRowVec ? 1 : N, IsRowVec ? N : 1>(i, 0, N, 1);
    }
    template <int N> View<const Scalar, IsRowVec ? 1 : N, IsRowVec ? N : 1> segment(Index i) const {
        return IsRowVec ? mkview<IsRowVec ? 1 : N, IsRowVec ? N : 1>(0, i, 1, N) : mkview<IsRowVec ? 1 : N, IsRowVec ? N : 1>(i, 0, N, 1);
    }
    View<Scalar, IsRowVec ? 1 : Dynamic, IsRowVec ? Dynamic : 1> segment(Index i, Index n) {
        return IsRowVec ? mkview<IsRowVec ? 1 : Dynamic, IsRowVec ? Dynamic : 1>(0, i, 1, n) : mkview<IsRowVec ? 1 : Dynamic, IsRowVec ? Dynamic : 1>(i, 0, n, 1);
    }
    View<const Scalar, IsRowVec ? 1 : Dynamic, IsRowVec ? Dynamic : 1> segment(Index i, Index n) const {
        return IsRowVec ? mkview<IsRowVec ? 1 : Dynamic, IsRowVec ? Dynamic : 1>(0, i, 1, n) : mkview<IsRowVec ? 1 : Dynamic, IsRowVec ? Dynamic : 1>(i, 0, n, 1);
    }
    template <int N> auto head() -> decltype(this->template segment<N>(0)) { return segment<N>(0); }
    template <int N> auto head() const -> decltype(this->template segment<N>(0)) { return segment<N>(0); }
    template <int N> auto tail() -> decltype(this->template segment<N>(0)) { return segment<N>(size() - N); }
    template <int N> auto tail() const -> decltype(this->template segment<N>(0)) { return segment<N>(size() - N); }
    auto head(Index n) -> decltype(this->segment(0, n)) { return segment(0, n); }
    auto head(Index n) const -> decltype(this->segment(0, n)) { return segment(0, n); }
    auto tail(Index n) -> decltype(this->segment(0, n)) { return segment(size() - n, n); }
    auto tail(Index n) const -> decltype(this->segment(0, n)) { return segment(size() - n, n); }
    View<Scalar, Dynamic, 1> diagonal() {
        return View<Scalar, Dynamic, 1>(derived().data(), std::min(derived().rows(), derived().cols()), 1, derived().rowStride() + derived().colStride(), 0);
    }
    View<const Scalar, Dynamic, 1> diagonal() const {
        return View<const Scalar, Dynamic, 1>(derived().data(), std::min(derived().rows(), derived().cols()), 1, derived().rowStride() + derived().colStride(), 0);
    }
    View<Scalar, ColsAtCompileTime, RowsAtCompileTime> transpose() {
        return View<Scalar, ColsAtCompileTime, RowsAtCompileTime>(derived().data(), derived().cols(), derived().rows(), derived().colStride(), derived().rowStride());
    }
    View<const Scalar, ColsAtCompileTime, RowsAtCompileTime> transpose() const {
        return View<const Scalar, ColsAtCompileTime, RowsAtCompileTime>(derived().data(), derived().cols(), derived().rows(), derived().colStride(), derived().rowStride());
    }
    View<const Scalar, ColsAtCompileTime, RowsAtCompileTime> adjoint() const { return transpose(); }
    D& noalias() { return derived(); }
    PlainObject eval() const { return PlainObject(*this); }
    ArrayProxy<D> array() { return ArrayProxy<D>(derived()); }
    PlainObject array() const { return PlainObject(*this); }
    PlainObject matrix() const { return PlainObject(*this); }
    template <class T2> Matrix<T2, RowsAtCompileTime, ColsAtCompileTime> cast() const {
        Matrix<T2, RowsAtCompileTime, ColsAtCompileTime> r(derived().rows(), derived().cols(), 0);
        for (Index j = 0; j < derived().cols(); j++) for (Index i = 0; i < derived().rows(); i++) r.coeffRef(i, j) = T2(derived().coeff(i, j));
        return r;
    }

    // ---- in-place ----
    template <class O> D& assign(const MatrixBase<O>& o) {
        assert(o.derived().rows() == derived().rows() && o.derived().cols() == derived().cols());
        // the right-hand side may be a view of this object (x = x.transpose() is the caller's problem, as in Eigen)
        for (Index j = 0; j < derived().cols(); j++) for (Index i = 0; i < derived().rows(); i++) derived().coeffRef(i, j) = o.derived().coeff(i, j);
        return derived();
    }
    template <class O> D& operator+=(const MatrixBase<O>& o) {
        for (Index j = 0; j < derived().cols(); j++) for (Index i = 0; i < derived().rows(); i++) derived().coeffRef(i, j) += o.derived().coeff(i, j);
        return derived();
    }
    template <class O> D& operator-=(const MatrixBase<O>& o) {
        for (Index j = 0; j < derived().cols(); j++) for (Index i = 0; i < derived().rows(); i++) derived().coeffRef(i, j) -= o.derived().coeff(i, j);
        return derived();
    }
    D& operator*=(const Scalar& s) { for (Index j = 0; j < derived().cols(); j++) for (Index i = 0; i < derived().rows(); i++) derived().coeffRef(i, j) *= s; return derived(); }
    D& operator/=(const Scalar& s) { for (Index j = 0; j < derived().cols(); j++) for (Index i = 0; i < derived().rows(); i++) derived().coeffRef(i, j) /= s; return derived(); }
    template <class O> D& operator*=(const MatrixBase<O>& o) { PlainObject t = (*this) * o; return assign(t); }
    D& setZero() { return fill(Scalar(0)); }
    D& setOnes() { return fill(Scalar(1)); }
    D& setConstant(const Scalar& s) { return fill(s); }
    D& fill(const Scalar& s) { for (Index j = 0; j < derived().cols(); j++) for (Index i = 0; i < derived().rows(); i++) derived().coeffRef(i, j) = s; return derived(); }
    D& setIdentity() { for (Index j = 0; j < derived().cols(); j++) for (Index i = 0; i < derived().rows(); i++) derived().coeffRef(i, j) = (i == j) ? Scalar(1) : Scalar(0); return derived(); }
    void normalize() { const Scalar n = norm(); if (n > Scalar(0)) *this /= n; }
    template <class U> typename std::enable_if<internal::is_scalar<U>::value, CommaInitializer<D>>::type operator<<(const U& s) { return CommaInitializer<D>(derived(), s); }
    template <class O> CommaInitializer<D> operator<<(const MatrixBase<O>& o) { return CommaInitializer<D>(derived(), o); }

    // ---- statics ----
    static PlainObject Zero() { PlainObject r; r.setZero(); return r; }
    static PlainObject Zero(Index r_, Index c_) { PlainObject r(r_, c_, 0); r.setZero(); return r; }
    static PlainObject Zero(Index n) { PlainObject r(ColsAtCompileTime == 1 ? n : 1, ColsAtCompileTime == 1 ? 1 : n, 0); r.setZero(); return r; }
    static PlainObject Ones() { PlainObject r; r.setOnes(); return r; }
    static PlainObject Ones(Index r_, Index c_) { PlainObject r(r_, c_, 0); r.setOnes(); return r; }
    static PlainObject Constant(const Scalar& s) { PlainObject r; r.fill(s); return r; }
    static PlainObject Constant(Index r_, Index c_, const Scalar& s) { PlainObject r(r_, c_, 0); r.fill(s); return r; }
    static PlainObject Identity() { PlainObject r; r.setIdentity(); return r; }
    static PlainObject Identity(Index r_, Index c_) { PlainObject r(r_, c_, 0); r.setIdentity(); return r; }
    static PlainObject Unit(Index k) { PlainObject r; r.setZero(); r.coeffRefv(k) = Scalar(1); return r; }
    static PlainObject UnitX() { return Unit(0); }
    static PlainObject UnitY() { return Unit(1); }
    static PlainObject UnitZ() { return Unit(2); }
    static PlainObject UnitW() { return Unit(3); }

    // ---- arithmetic (eager) ----
    template <class O> Matrix<Scalar, internal::pick<RowsAtCompileTime, internal::traits<O>::Rows>::value, internal::pick<ColsAtCompileTime, internal::traits<O>::Cols>::value>
    operator+(const MatrixBase<O>& o) const {
        Matrix<Scalar, internal::pick<RowsAtCompileTime, internal::traits<O>::Rows>::value, internal::pick<ColsAtCompileTime, internal::traits<O>::Cols>::value> r(derived().rows(), derived().cols(), 0);
        for (Index j = 0; j < derived().cols(); j++) for (Index i = 0; i < derived().rows(); i++) r.coeffRef(i, j) = derived().coeff(i, j) + o.derived().coeff(i, j);
        return r;
    }
    template <class O> Matrix<Scalar, internal::pick<RowsAtCompileTime, internal::traits<O>::Rows>::value, internal::pick<ColsAtCompileTime, internal::traits<O>::Cols>::value>
    operator-(const MatrixBase<O>& o) const {
        Matrix<Scalar, internal::pick<RowsAtCompileTime, internal::traits<O>::Rows>::value, internal::pick<ColsAtCompileTime, internal::traits<O>::Cols>::value> r(derived().rows(), derived().cols(), 0);
        for (Index j = 0; j < derived().cols(); j++) for (Index i = 0; i < derived().rows(); i++) r.coeffRef(i, j) = derived().coeff(i, j) - o.derived().coeff(i, j);
        return r;
    }
    PlainObject operator-() const {
        PlainObject r(derived().rows(), derived().cols(), 0);
        for (Index j = 0; j < derived().cols(); j++) for (Index i = 0; i < derived().rows(); i++) r.coeffRef(i, j) = -derived().coeff(i, j);
        return r;
    }
    PlainObject operator*(const Scalar& s) const {
        PlainObject r(derived().rows(), derived().cols(), 0);
        for (Index j = 0; j < derived().cols(); j++) for (Index i = 0; i < derived().rows(); i++) r.coeffRef(i, j) = derived().coeff(i, j) * s;
        return r;
    }
    friend PlainObject operator*(const Scalar& s, const MatrixBase& m) {
        PlainObject r(m.derived().rows(), m.derived().cols(), 0);
        for (Index j = 0; j < m.derived().cols(); j++) for (Index i = 0; i < m.derived().rows(); i++) r.coeffRef(i, j) = s * m.derived().coeff(i, j);
        return r;
    }
    PlainObject operator/(const Scalar& s) const {
        PlainObject r(derived().rows(), derived().cols(), 0);
        for (Index j = 0; j < derived().cols(); j++) for (Index i = 0; i < derived().rows(); i++) r.coeffRef(i, j) = derived().coeff(i, j) / s;
        return r;
    }
    template <class O> Matrix<Scalar, RowsAtCompileTime, internal::traits<O>::Cols> operator*(const MatrixBase<O>& o) const {
        const Index n = derived().rows(), m = o.derived().cols(), kk = derived().cols();
        assert(kk == o.derived().rows());
        Matrix<Scalar, RowsAtCompileTime, internal::traits<O>::Cols> r(n, m, 0);
        for (Index j = 0; j < m; j++)
            for (Index i = 0; i < n; i++) {
                Scalar s = kk > 0 ? derived().coeff(i, 0) * o.derived().coeff(0, j) : Scalar(0);
                for (Index k = 1; k < kk; k++) s += derived().coeff(i, k) * o.derived().coeff(k, j);
                r.coeffRef(i, j) = s;
            }
        return r;
    }
    template <class O> bool operator==(const MatrixBase<O>& o) const {
        for (Index j = 0; j < derived().cols(); j++) for (Index i = 0; i < derived().rows(); i++) if (!(derived().coeff(i, j) == o.derived().coeff(i, j))) return false;
        return true;
    }
    template <class O> bool operator!=(const MatrixBase<O>& o) const { return !(*this == o); }

    // ---- reductions / coefficient-wise ----
    template <class O> Scalar dot(const MatrixBase<O>& o) const {
        const Index n = size();
        Scalar s = n > 0 ? coeffv(0) * o.coeffv(0) : Scalar(0);
        for (Index k = 1; k < n; k++) s += coeffv(k) * o.coeffv(k);
        return s;
    }
    template <class O> PlainObject cross(const MatrixBase<O>& o) const {
        PlainObject r;
        r.coeffRefv(0) = coeffv(1) * o.coeffv(2) - coeffv(2) * o.coeffv(1);
        r.coeffRefv(1) = coeffv(2) * o.coeffv(0) - coeffv(0) * o.coeffv(2);
        r.coeffRefv(2) = coeffv(0) * o.coeffv(1) - coeffv(1) * o.coeffv(0);
        return r;
    }
    Scalar squaredNorm() const {
        Scalar s = Scalar(0);
        bool first = true;
        for (Index j = 0; j < derived().cols(); j++) for (Index i = 0; i < derived().rows(); i++) {
            const Scalar v = derived().coeff(i, j);
            if (first) { s = v * v; first = false; } else s += v * v;
        }
        return s;
    }
    Scalar norm() const { return std::sqrt(squaredNorm()); }
    PlainObject normalized() const { PlainObject r(*this); r.normalize(); return r; }
    Scalar sum() const { Scalar s = Scalar(0); for (Index j = 0; j < derived().cols(); j++) for (Index i = 0; i < derived().rows(); i++) s += derived().coeff(i, j); return s; }
    Scalar prod() const { Scalar s = Scalar(1); for (Index j = 0; j < derived().cols(); j++) for (Index i = 0; i < derived().rows(); i++) s *= derived().coeff(i, j); return s; }
    Scalar mean() const { return sum() / Scalar(size()); }
    Scalar trace() const { Scalar s = Scalar(0); for (Index i = 0; i < std::min(derived().rows(), derived().cols()); i++) s += derived().coeff(i, i); return s; }
    Scalar maxCoeff() const { Scalar s = derived().coeff(0, 0); for (Index j = 0; j < derived().cols(); j++) for (Index i = 0; i < derived().rows(); i++) if (derived().coeff(i, j) > s) s = derived().coeff(i, j); return s; }
    Scalar minCoeff() const { Scalar s = derived().coeff(0, 0); for (Index j = 0; j < derived().cols(); j++) for (Index i = 0; i < derived().rows(); i++) if (derived().coeff(i, j) < s) s = derived().coeff(i, j); return s; }
    template <class I> Scalar maxCoeff(I* idx) const { Scalar s = coeffv(0); *idx = 0; for (Index k = 1; k < size(); k++) if (coeffv(k) > s) { s = coeffv(k); *idx = I(k); } return s; }
    template <class I> Scalar minCoeff(I* idx) const { Scalar s = coeffv(0); *idx = 0; for (Index k = 1; k < size(); k++) if (coeffv(k) < s) { s = coeffv(k); *idx = I(k); } return s; }
    PlainObject cwiseAbs() const { PlainObject r(*this); for (Index j = 0; j < r.cols(); j++) for (Index i = 0; i < r.rows(); i++) r.coeffRef(i, j) = std::abs(r.coeff(i, j)); return r; }
    PlainObject cwiseSqrt() const { PlainObject r(*this); for (Index j = 0; j < r.cols(); j++) for (Index i = 0; i < r.rows(); i++) r.coeffRef(i, j) = std::sqrt(r.coeff(i, j)); return r; }
    PlainObject cwiseInverse() const { PlainObject r(*this); for (Index j = 0; j < r.cols(); j++) for (Index i = 0; i < r.rows(); i++) r.coeffRef(i, j) = Scalar(1) / r.coeff(i, j); return r; }
    template <class O> PlainObject cwiseProduct(const MatrixBase<O>& o) const { PlainObject r(*this); for (Index j = 0; j < r.cols(); j++) for (Index i = 0; i < r.rows(); i++) r.coeffRef(i, j) *= o.derived().coeff(i, j); return r; }
    template <class O> PlainObject cwiseQuotient(const MatrixBase<O>& o) const { PlainObject r(*this); for (Index j = 0; j < r.cols(); j++) for (Index i = 0; i < r.rows(); i++) r.coeffRef(i, j) /= o.derived().coeff(i, j); return r; }
    bool allFinite() const { for (Index j = 0; j < derived().cols(); j++) for (Index i = 0; i < derived().rows(); i++) if (!std::isfinite(derived().coeff(i, j))) return false; return true; }
    bool hasNaN() const { for (Index j = 0; j < derived().cols(); j++) for (Index i = 0; i < derived().rows(); i++) if (std::isnan(derived().coeff(i, j))) return true; return false; }
    template <class O> bool isApprox(const MatrixBase<O>& o, Scalar prec = NumTraits<Scalar>::dummy_precision()) const {
        return (*this - o).squaredNorm() <= prec * prec * std::min(squaredNorm(), o.squaredNorm());
    }

    // ---- small dense algebra ----
    Scalar determinant() const;
    PlainObject inverse() const;
    LDLT<PlainObject> ldlt() const;
    LLT<PlainObject> llt() const;
    PartialPivLU<PlainObject> lu() const;
    PartialPivLU<PlainObject> partialPivLu() const;
};

template <class D> std::ostream& operator<<(std::ostream& os, const MatrixBase<D>& m) {
    for (Index i = 0; i < m.derived().rows(); i++) {
        for (Index j = 0; j < m.derived().cols(); j++) os << (j ? " " : "") << m.derived().coeff(i, j);
        if (i + 1 < m.derived().rows()) os << "\n";
    }
    return os;
}

// ---------------------------------------------------------------------------------------------------------------------
namespace internal {
template <class T, int R, int C, bool Dyn = (R == Dynamic || C == Dynamic)> struct storage;
template <class T, int R, int C> struct storage<T, R, C, false> {
    T d[R * C > 0 ? R * C : 1];
    storage() { for (int i = 0; i < R * C; i++) d[i] = T(0); }
    Index rows() const { return R; }
    Index cols() const { return C; }
    T* data() { return d; }
    const T* data() const { return d; }
    void resize(Index r, Index c) { assert(r == R && c == C); (void)r; (void)c; }
};
template <class T, int R, int C> struct storage<T, R, C, true> {
    std::vector<T> d;
    Index r_, c_;
    storage() : r_(R == Dynamic ? 0 : R), c_(C == Dynamic ? 0 : C) {}
    Index rows() const { return r_; }
    Index cols() const { return c_; }
    T* data() { return d.data(); }
    const T* data() const { return d.data(); }
    void resize(Index r, Index c) { r_ = r; c_ = c; d.assign(size_t(r * c), T(0)); }
};
}  // namespace internal

template <class T, int R, int C, int Opt, int MR, int MC> class Matrix : public MatrixBase<Matrix<T, R, C, Opt, MR, MC>> {
    internal::storage<T, R, C> s_;
public:
    typedef MatrixBase<Matrix> Base;
    typedef T Scalar;
    enum { Options = Opt, Flags = 0, IsRowMajor = ((Opt & RowMajor) && R != 1 && C != 1) || (R == 1 && C != 1) ? 1 : 0, IsDyn = (R == Dynamic || C == Dynamic) };
    typedef Map<Matrix, Unaligned> MapType;
    typedef Map<const Matrix, Unaligned> ConstMapType;
    typedef Map<Matrix, Aligned> AlignedMapType;
    typedef Map<const Matrix, Aligned> ConstAlignedMapType;

    Matrix() {}
    Matrix(const Matrix& o) : Base(), s_(o.s_) {}
    Matrix(Index r, Index c, int /*dims tag*/) { s_.resize(r, c); }
    template <class O> Matrix(const MatrixBase<O>& o) { s_.resize(o.derived().rows(), o.derived().cols()); this->assign(o); }
    explicit Matrix(const T* p) { for (Index k = 0; k < this->size(); k++) s_.data()[k] = p[k]; }
    // one argument: size of a dynamic vector, or the coefficient of a 1x1
    template <class A, typename std::enable_if<internal::is_scalar<A>::value, int>::type = 0> explicit Matrix(const A& a) {
        if (IsDyn) s_.resize(C == 1 ? Index(a) : (R == Dynamic ? Index(a) : R), C == 1 ? 1 : (R == 1 ? Index(a) : C));
        else s_.data()[0] = T(a);
    }
    // two arguments: (rows, cols) of a dynamic matrix, or the coefficients of a 2-vector
    template <class A, class B, typename std::enable_if<internal::is_scalar<A>::value && internal::is_scalar<B>::value, int>::type = 0> Matrix(const A& a, const B& b) {
        if (IsDyn) s_.resize(Index(a), Index(b));
        else { s_.data()[0] = T(a); s_.data()[1] = T(b); }
    }
    Matrix(const T& x, const T& y, const T& z) { static_assert(R * C == 3, "3-vector constructor"); s_.d[0] = x; s_.d[1] = y; s_.d[2] = z; }
    Matrix(const T& x, const T& y, const T& z, const T& w) { static_assert(R * C == 4, "4-vector constructor"); s_.d[0] = x; s_.d[1] = y; s_.d[2] = z; s_.d[3] = w; }

    Index rows() const { return s_.rows(); }
    Index cols() const { return s_.cols(); }
    Index rowStride() const { return IsRowMajor ? cols() : 1; }
    Index colStride() const { return IsRowMajor ? 1 : rows(); }
    Index innerSize() const { return IsRowMajor ? cols() : rows(); }
    Index outerSize() const { return IsRowMajor ? rows() : cols(); }
    T* data() { return s_.data(); }
    const T* data() const { return s_.data(); }
    T coeff(Index i, Index j) const { return s_.data()[i * rowStride() + j * colStride()]; }
    T& coeffRef(Index i, Index j) { return s_.data()[i * rowStride() + j * colStride()]; }
    void resize(Index r, Index c) { if (r != rows() || c != cols()) s_.resize(r, c); }
    void resize(Index n) { if (C == 1) resize(n, 1); else resize(1, n); }
    void conservativeResize(Index r, Index c) {
        Matrix t(r, c, 0);
        for (Index j = 0; j < std::min(c, cols()); j++) for (Index i = 0; i < std::min(r, rows()); i++) t.coeffRef(i, j) = coeff(i, j);
        *this = t;
    }
    void conservativeResize(Index n) { if (C == 1) conservativeResize(n, 1); else conservativeResize(1, n); }
    Matrix& setZero() { Base::setZero(); return *this; }
    Matrix& setZero(Index n) { resize(n); Base::setZero(); return *this; }
    Matrix& setZero(Index r, Index c) { resize(r, c); Base::setZero(); return *this; }
    Matrix& setIdentity() { Base::setIdentity(); return *this; }
    Matrix& setIdentity(Index r, Index c) { resize(r, c); Base::setIdentity(); return *this; }
    Matrix& operator=(const Matrix& o) { s_ = o.s_; return *this; }
    template <class O> Matrix& operator=(const MatrixBase<O>& o) {
        if (IsDyn && (o.derived().rows() != rows() || o.derived().cols() != cols())) { Matrix t(o); s_ = t.s_; return *this; }
        // evaluate first: the source may be a view of *this
        Matrix t(o.derived().rows(), o.derived().cols(), 0);
        t.assign(o);
        s_ = t.s_;
        return *this;
    }
    template <class QT> Matrix& operator=(const Quaternion<QT>& q) { return *this = q.toRotationMatrix(); }
    template <class QT> Matrix& operator=(const AngleAxis<QT>& q) { return *this = q.toRotationMatrix(); }
    void swap(Matrix& o) { std::swap(s_, o.s_); }
};

// a (pointer, strides) window; S may be const-qualified
template <class S, int R, int C> class View : public MatrixBase<View<S, R, C>> {
    S* p_;
    Index r_, c_, rs_, cs_;
public:
    typedef typename std::remove_const<S>::type Scalar;
    View(S* p, Index r, Index c, Index rs, Index cs) : p_(p), r_(r), c_(c), rs_(rs), cs_(cs) {}
    View(const View& o) : MatrixBase<View>(), p_(o.p_), r_(o.r_), c_(o.c_), rs_(o.rs_), cs_(o.cs_) {}
    Index rows() const { return r_; }
    Index cols() const { return c_; }
    Index rowStride() const { return rs_; }
    Index colStride() const { return cs_; }
    S* data() const { return p_; }
    Scalar coeff(Index i, Index j) const { return p_[i * rs_ + j * cs_]; }
    S& coeffRef(Index i, Index j) const { return p_[i * rs_ + j * cs_]; }
    View& operator=(const View& o) { Matrix<Scalar, R, C> t(o); this->assign(t); return *this; }
    template <class O> View& operator=(const MatrixBase<O>& o) { Matrix<Scalar, R, C> t(o); this->assign(t); return *this; }
};

template <class M, int MapOpt, class Stride> class Map : public MatrixBase<Map<M, MapOpt, Stride>> {
    typedef typename std::remove_const<M>::type Plain;
public:
    typedef typename internal::traits<Plain>::Scalar Scalar;
    typedef typename std::conditional<std::is_const<M>::value, const Scalar, Scalar>::type S;
private:
    S* p_;
    Index r_, c_;
public:
    enum { R = internal::traits<Plain>::Rows, C = internal::traits<Plain>::Cols };
    Map(S* p) : p_(p), r_(R), c_(C) {}
    Map(S* p, Index n) : p_(p), r_(C == 1 ? n : (R == Dynamic ? n : R)), c_(C == 1 ? 1 : (R == 1 ? n : C)) {}
    Map(S* p, Index r, Index c) : p_(p), r_(r), c_(c) {}
    Map(const Map& o) : MatrixBase<Map>(), p_(o.p_), r_(o.r_), c_(o.c_) {}
    Index rows() const { return r_; }
    Index cols() const { return c_; }
    Index rowStride() const { return Plain::IsRowMajor ? c_ : 1; }
    Index colStride() const { return Plain::IsRowMajor ? 1 : r_; }
    S* data() const { return p_; }
    Scalar coeff(Index i, Index j) const { return p_[i * rowStride() + j * colStride()]; }
    S& coeffRef(Index i, Index j) const { return p_[i * rowStride() + j * colStride()]; }
    Map& operator=(const Map& o) { Plain t(o); this->assign(t); return *this; }
    template <class O> Map& operator=(const MatrixBase<O>& o) { Plain t(o); this->assign(t); return *this; }
};

#define MINIEIGEN_TYPEDEFS(T, sfx)                               \
    typedef Matrix<T, 2, 2> Matrix2##sfx;                        \
    typedef Matrix<T, 3, 3> Matrix3##sfx;                        \
    typedef Matrix<T, 4, 4> Matrix4##sfx;                        \
    typedef Matrix<T, Dynamic, Dynamic> MatrixX##sfx;            \
    typedef Matrix<T, 2, 1> Vector2##sfx;                        \
    typedef Matrix<T, 3, 1> Vector3##sfx;                        \
    typedef Matrix<T, 4, 1> Vector4##sfx;                        \
    typedef Matrix<T, Dynamic, 1> VectorX##sfx;                  \
    typedef Matrix<T, 1, 2> RowVector2##sfx;                     \
    typedef Matrix<T, 1, 3> RowVector3##sfx;                     \
    typedef Matrix<T, 1, 4> RowVector4##sfx;                     \
    typedef Matrix<T, 1, Dynamic> RowVectorX##sfx;
MINIEIGEN_TYPEDEFS(double, d)
MINIEIGEN_TYPEDEFS(float, f)
MINIEIGEN_TYPEDEFS(int, i)
#undef MINIEIGEN_TYPEDEFS

// ---------------------------------------------------------------------------------------------------------------------
// LU with partial pivoting (rows), determinant, inverse.  2x2 / 3x3 inverses and determinants use the closed forms
// (cofactors / determinant), as Eigen does for fixed sizes up to 4; 4x4 and larger go through the LU.
template <class MatrixType> class PartialPivLU {
public:
    typedef typename MatrixType::Scalar Scalar;
    typedef Matrix<Scalar, Dynamic, Dynamic> Dyn;
    Dyn lu_;
    std::vector<Index> perm_;
    int sign_ = 1;
    PartialPivLU() {}
    template <class O> explicit PartialPivLU(const MatrixBase<O>& a) { compute(a); }
    template <class O> PartialPivLU& compute(const MatrixBase<O>& a) {
        lu_ = a;
        const Index n = lu_.rows();
        perm_.resize(n);
        for (Index i = 0; i < n; i++) perm_[i] = i;
        sign_ = 1;
        for (Index k = 0; k < n; k++) {
            Index piv = k;
            Scalar best = std::abs(lu_(k, k));
            for (Index i = k + 1; i < n; i++) if (std::abs(lu_(i, k)) > best) { best = std::abs(lu_(i, k)); piv = i; }
            if (piv != k) {
                for (Index j = 0; j < n; j++) std::swap(lu_(k, j), lu_(piv, j));
                std::swap(perm_[k], perm_[piv]);
                sign_ = -sign_;
            }
            if (lu_(k, k) != Scalar(0)) {
                for (Index i = k + 1; i < n; i++) lu_(i, k) /= lu_(k, k);
                for (Index j = k + 1; j < n; j++) for (Index i = k + 1; i < n; i++) lu_(i, j) -= lu_(i, k) * lu_(k, j);
            }
        }
        return *this;
    }
    Scalar determinant() const { Scalar d = Scalar(sign_); for (Index i = 0; i < lu_.rows(); i++) d *= lu_(i, i); return d; }
    template <class O> Matrix<Scalar, MatrixType::RowsAtCompileTime, internal::traits<O>::Cols> solve(const MatrixBase<O>& b) const {
        const Index n = lu_.rows(), m = b.derived().cols();
        Matrix<Scalar, MatrixType::RowsAtCompileTime, internal::traits<O>::Cols> x(n, m, 0);
        for (Index c = 0; c < m; c++) {
            for (Index i = 0; i < n; i++) x(i, c) = b.derived().coeff(perm_[i], c);
            for (Index i = 0; i < n; i++) for (Index k = 0; k < i; k++) x(i, c) -= lu_(i, k) * x(k, c);
            for (Index i = n - 1; i >= 0; i--) { for (Index k = i + 1; k < n; k++) x(i, c) -= lu_(i, k) * x(k, c); x(i, c) /= lu_(i, i); }
        }
        return x;
    }
    MatrixType inverse() const { Dyn I = Dyn::Identity(lu_.rows(), lu_.rows()); return MatrixType(solve(I)); }
};

template <class D> typename MatrixBase<D>::Scalar MatrixBase<D>::determinant() const {
    const D& a = derived();
    const Index n = a.rows();
    if (n == 1) return a.coeff(0, 0);
    if (n == 2) return a.coeff(0, 0) * a.coeff(1, 1) - a.coeff(1, 0) * a.coeff(0, 1);
    if (n == 3)
        return a.coeff(0, 0) * (a.coeff(1, 1) * a.coeff(2, 2) - a.coeff(1, 2) * a.coeff(2, 1)) - a.coeff(0, 1) * (a.coeff(1, 0) * a.coeff(2, 2) - a.coeff(1, 2) * a.coeff(2, 0)) +
               a.coeff(0, 2) * (a.coeff(1, 0) * a.coeff(2, 1) - a.coeff(1, 1) * a.coeff(2, 0));
    return PartialPivLU<PlainObject>(*this).determinant();
}
template <class D> typename MatrixBase<D>::PlainObject MatrixBase<D>::inverse() const {
    const D& a = derived();
    const Index n = a.rows();
    PlainObject r(n, n, 0);
    if (n == 1) { r.coeffRef(0, 0) = Scalar(1) / a.coeff(0, 0); return r; }
    if (n == 2) {
        const Scalar invdet = Scalar(1) / determinant();
        r.coeffRef(0, 0) = a.coeff(1, 1) * invdet; r.coeffRef(1, 0) = -a.coeff(1, 0) * invdet;
        r.coeffRef(0, 1) = -a.coeff(0, 1) * invdet; r.coeffRef(1, 1) = a.coeff(0, 0) * invdet;
        return r;
    }
    if (n == 3) {
        // cofactor(i,j) = m((i+1)%3,(j+1)%3) m((i+2)%3,(j+2)%3) - m((i+1)%3,(j+2)%3) m((i+2)%3,(j+1)%3); det = col 0 of the matrix . row 0 of the
        // transposed cofactors; result = transposed cofactors * (1/det)
        Scalar cof[3][3];
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
            const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
            cof[i][j] = a.coeff(i1, j1) * a.coeff(i2, j2) - a.coeff(i1, j2) * a.coeff(i2, j1);
        }
        const Scalar det = cof[0][0] * a.coeff(0, 0) + cof[1][0] * a.coeff(1, 0) + cof[2][0] * a.coeff(2, 0);
        const Scalar invdet = Scalar(1) / det;
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.coeffRef(i, j) = cof[j][i] * invdet;
        return r;
    }
    return PartialPivLU<PlainObject>(*this).inverse();
}
template <class D> PartialPivLU<typename MatrixBase<D>::PlainObject> MatrixBase<D>::lu() const { return PartialPivLU<PlainObject>(*this); }
template <class D> PartialPivLU<typename MatrixBase<D>::PlainObject> MatrixBase<D>::partialPivLu() const { return PartialPivLU<PlainObject>(*this); }

// Cholesky LL^T (lower), no pivoting
template <class MatrixType> class LLT {
public:
    typedef typename MatrixType::Scalar Scalar;
    Matrix<Scalar, Dynamic, Dynamic> l_;
    ComputationInfo info_ = Success;
    LLT() {}
    template <class O> explicit LLT(const MatrixBase<O>& a) { compute(a); }
    template <class O> LLT& compute(const MatrixBase<O>& a) {
        l_ = a;
        const Index n = l_.rows();
        info_ = Success;
        for (Index k = 0; k < n; k++) {
            Scalar x = l_(k, k);
            for (Index j = 0; j < k; j++) x -= l_(k, j) * l_(k, j);
            if (!(x > Scalar(0))) { info_ = NumericalIssue; return *this; }
            x = std::sqrt(x);
            l_(k, k) = x;
            for (Index i = k + 1; i < n; i++) {
                Scalar s = l_(i, k);
                for (Index j = 0; j < k; j++) s -= l_(i, j) * l_(k, j);
                l_(i, k) = s / x;
            }
        }
        return *this;
    }
    ComputationInfo info() const { return info_; }
    template <class O> Matrix<Scalar, MatrixType::RowsAtCompileTime, internal::traits<O>::Cols> solve(const MatrixBase<O>& b) const {
        const Index n = l_.rows(), m = b.derived().cols();
        Matrix<Scalar, MatrixType::RowsAtCompileTime, internal::traits<O>::Cols> x(n, m, 0);
        for (Index c = 0; c < m; c++) {
            for (Index i = 0; i < n; i++) { Scalar s = b.derived().coeff(i, c); for (Index k = 0; k < i; k++) s -= l_(i, k) * x(k, c); x(i, c) = s / l_(i, i); }
            for (Index i = n - 1; i >= 0; i--) { Scalar s = x(i, c); for (Index k = i + 1; k < n; k++) s -= l_(k, i) * x(k, c); x(i, c) = s / l_(i, i); }
        }
        return x;
    }
    Matrix<Scalar, Dynamic, Dynamic> matrixL() const { Matrix<Scalar, Dynamic, Dynamic> L = l_; for (Index j = 0; j < L.cols(); j++) for (Index i = 0; i < j; i++) L(i, j) = 0; return L; }
};

// Robust Cholesky LDL^T with symmetric pivoting on the largest |diagonal| entry (the algorithm Eigen::LDLT documents:
// unblocked, lower storage, transpositions P, sign tracking for isPositive / isNegative).
template <class MatrixType> class LDLT {
public:
    typedef typename MatrixType::Scalar Scalar;
    Matrix<Scalar, Dynamic, Dynamic> m_;
    std::vector<Index> tr_;
    int sign_ = 0;   // 0 zero, 1 positive semi-definite, -1 negative semi-definite, 2 indefinite
    ComputationInfo info_ = Success;
    bool init_ = false;
    LDLT() {}
    template <class O> explicit LDLT(const MatrixBase<O>& a) { compute(a); }
    template <class O> LDLT& compute(const MatrixBase<O>& a) {
        m_ = a;
        const Index n = m_.rows();
        tr_.assign(size_t(n), 0);
        sign_ = 0;
        info_ = Success;
        init_ = true;
        if (n <= 1) {
            if (n == 1) { tr_[0] = 0; const Scalar v = m_(0, 0); sign_ = v > 0 ? 1 : (v < 0 ? -1 : 0); }
            return *this;
        }
        std::vector<Scalar> temp(size_t(n), Scalar(0));
        bool found_zero_pivot = false;
        for (Index k = 0; k < n; k++) {
            Index piv = k;
            Scalar best = std::abs(m_(k, k));
            for (Index i = k + 1; i < n; i++) if (std::abs(m_(i, i)) > best) { best = std::abs(m_(i, i)); piv = i; }
            tr_[k] = piv;
            if (k != piv) {
                // symmetric row/column exchange on the lower triangle
                const Index s = n - piv - 1;
                for (Index j = 0; j < k; j++) std::swap(m_(k, j), m_(piv, j));
                for (Index i = 0; i < s; i++) std::swap(m_(piv + 1 + i, k), m_(piv + 1 + i, piv));
                std::swap(m_(k, k), m_(piv, piv));
                for (Index i = k + 1; i < piv; i++) std::swap(m_(i, k), m_(piv, i));
            }
            const Index rs = n - k - 1;
            if (k > 0) {
                for (Index j = 0; j < k; j++) temp[j] = m_(j, j) * m_(k, j);
                Scalar acc = m_(k, 0) * temp[0];
                for (Index j = 1; j < k; j++) acc += m_(k, j) * temp[j];
                m_(k, k) -= acc;
                for (Index i = 0; i < rs; i++) {
                    Scalar a2 = m_(k + 1 + i, 0) * temp[0];
                    for (Index j = 1; j < k; j++) a2 += m_(k + 1 + i, j) * temp[j];
                    m_(k + 1 + i, k) -= a2;
                }
            }
            const Scalar akk = m_(k, k);
            const bool pivot_is_valid = std::abs(akk) > Scalar(0);
            if (k == 0 && !pivot_is_valid) {
                sign_ = 0;
                for (Index j = 0; j < n; j++) tr_[j] = j;
                return *this;
            }
            if (rs > 0 && pivot_is_valid) for (Index i = 0; i < rs; i++) m_(k + 1 + i, k) /= akk;
            else if (rs > 0) { for (Index i = 0; i < rs; i++) if (m_(k + 1 + i, k) != Scalar(0)) info_ = NumericalIssue; }
            if (found_zero_pivot && pivot_is_valid) info_ = NumericalIssue;
            else if (!pivot_is_valid) found_zero_pivot = true;
            if (sign_ == 1) { if (akk < Scalar(0)) sign_ = 2; }
            else if (sign_ == -1) { if (akk > Scalar(0)) sign_ = 2; }
            else if (sign_ == 0) { if (akk > Scalar(0)) sign_ = 1; else if (akk < Scalar(0)) sign_ = -1; }
        }
        return *this;
    }
    bool isPositive() const { return sign_ == 1 || sign_ == 0; }
    bool isNegative() const { return sign_ == -1 || sign_ == 0; }
    ComputationInfo info() const { return info_; }
    Matrix<Scalar, Dynamic, 1> vectorD() const { return Matrix<Scalar, Dynamic, 1>(m_.diagonal()); }
    template <class O> Matrix<Scalar, MatrixType::RowsAtCompileTime, internal::traits<O>::Cols> solve(const MatrixBase<O>& b) const {
        const Index n = m_.rows(), mcols = b.derived().cols();
        Matrix<Scalar, MatrixType::RowsAtCompileTime, internal::traits<O>::Cols> x(n, mcols, 0);
        const Scalar tol = Scalar(1) / NumTraits<Scalar>::highest();
        for (Index c = 0; c < mcols; c++) {
            for (Index i = 0; i < n; i++) x(i, c) = b.derived().coeff(i, c);
            for (Index i = 0; i < n; i++) if (tr_[i] != i) std::swap(x(i, c), x(tr_[i], c));              // P b
            for (Index i = 0; i < n; i++) for (Index k = 0; k < i; k++) x(i, c) -= m_(i, k) * x(k, c);     // L^-1 (unit lower)
            for (Index i = 0; i < n; i++) { if (std::abs(m_(i, i)) > tol) x(i, c) /= m_(i, i); else x(i, c) = Scalar(0); }
            for (Index i = n - 1; i >= 0; i--) for (Index k = i + 1; k < n; k++) x(i, c) -= m_(k, i) * x(k, c);   // L^-T
            for (Index i = n - 1; i >= 0; i--) if (tr_[i] != i) std::swap(x(i, c), x(tr_[i], c));          // P^T
        }
        return x;
    }
};
template <class D> LDLT<typename MatrixBase<D>::PlainObject> MatrixBase<D>::ldlt() const { return LDLT<PlainObject>(*this); }
template <class D> LLT<typename MatrixBase<D>::PlainObject> MatrixBase<D>::llt() const { return LLT<PlainObject>(*this); }

// Symmetric eigenvalues by cyclic Jacobi rotations (only g2o's verifyInformationMatrices diagnostic uses it; not on the pose path)
enum { ComputeEigenvectors = 0x80, EigenvaluesOnly = 0x40 };
template <class MatrixType> class SelfAdjointEigenSolver {
public:
    typedef typename MatrixType::Scalar Scalar;
    Matrix<Scalar, MatrixType::RowsAtCompileTime, 1> vals_;
    MatrixType vecs_;
    SelfAdjointEigenSolver() {}
    template <class O> explicit SelfAdjointEigenSolver(const MatrixBase<O>& a, int options = ComputeEigenvectors) { compute(a, options); }
    template <class O> SelfAdjointEigenSolver& compute(const MatrixBase<O>& a, int = ComputeEigenvectors) {
        Matrix<Scalar, Dynamic, Dynamic> A = a;
        const Index n = A.rows();
        Matrix<Scalar, Dynamic, Dynamic> V = Matrix<Scalar, Dynamic, Dynamic>::Identity(n, n);
        for (int sweep = 0; sweep < 64; sweep++) {
            Scalar off = 0;
            for (Index p = 0; p < n; p++) for (Index q = p + 1; q < n; q++) off += A(p, q) * A(p, q);
            if (off < std::numeric_limits<Scalar>::min()) break;
            for (Index p = 0; p < n; p++) for (Index q = p + 1; q < n; q++) {
                if (A(p, q) == Scalar(0)) continue;
                const Scalar theta = (A(q, q) - A(p, p)) / (Scalar(2) * A(p, q));
                const Scalar t = (theta >= 0 ? Scalar(1) : Scalar(-1)) / (std::abs(theta) + std::sqrt(theta * theta + Scalar(1)));
                const Scalar c = Scalar(1) / std::sqrt(t * t + Scalar(1)), s = t * c;
                for (Index k = 0; k < n; k++) { const Scalar akp = A(k, p), akq = A(k, q); A(k, p) = c * akp - s * akq; A(k, q) = s * akp + c * akq; }
                for (Index k = 0; k < n; k++) { const Scalar apk = A(p, k), aqk = A(q, k); A(p, k) = c * apk - s * aqk; A(q, k) = s * apk + c * aqk; }
                for (Index k = 0; k < n; k++) { const Scalar vkp = V(k, p), vkq = V(k, q); V(k, p) = c * vkp - s * vkq; V(k, q) = s * vkp + c * vkq; }
            }
        }
        std::vector<Index> ord(size_t(n), 0);
        for (Index i = 0; i < n; i++) ord[i] = i;
        std::sort(ord.begin(), ord.end(), [&](Index x, Index y) { return A(x, x) < A(y, y); });
        vals_ = Matrix<Scalar, MatrixType::RowsAtCompileTime, 1>::Zero(n);
        vecs_ = MatrixType::Zero(n, n);
        for (Index i = 0; i < n; i++) { vals_[i] = A(ord[i], ord[i]); for (Index k = 0; k < n; k++) vecs_(k, i) = V(k, ord[i]); }
        return *this;
    }
    const Matrix<Scalar, MatrixType::RowsAtCompileTime, 1>& eigenvalues() const { return vals_; }
    const MatrixType& eigenvectors() const { return vecs_; }
    ComputationInfo info() const { return Success; }
};

// ---------------------------------------------------------------------------------------------------------------------
// Geometry
template <class T> class AngleAxis {
    Matrix<T, 3, 1> axis_;
    T angle_;
public:
    typedef Matrix<T, 3, 3> Matrix3;
    typedef Matrix<T, 3, 1> Vector3;
    AngleAxis() : angle_(0) { axis_ << T(1), T(0), T(0); }
    template <class O> AngleAxis(const T& angle, const MatrixBase<O>& axis) : axis_(axis), angle_(angle) {}
    explicit AngleAxis(const Quaternion<T>& q);
    template <class O> explicit AngleAxis(const MatrixBase<O>& R) { *this = AngleAxis(Quaternion<T>(R)); }
    T angle() const { return angle_; }
    T& angle() { return angle_; }
    const Vector3& axis() const { return axis_; }
    Vector3& axis() { return axis_; }
    Matrix3 toRotationMatrix() const {
        Matrix3 res;
        const Vector3 sin_axis = std::sin(angle_) * axis_;
        const T c = std::cos(angle_);
        const Vector3 cos1_axis = (T(1) - c) * axis_;
        T tmp;
        tmp = cos1_axis.x() * axis_.y(); res.coeffRef(0, 1) = tmp - sin_axis.z(); res.coeffRef(1, 0) = tmp + sin_axis.z();
        tmp = cos1_axis.x() * axis_.z(); res.coeffRef(0, 2) = tmp + sin_axis.y(); res.coeffRef(2, 0) = tmp - sin_axis.y();
        tmp = cos1_axis.y() * axis_.z(); res.coeffRef(1, 2) = tmp - sin_axis.x(); res.coeffRef(2, 1) = tmp + sin_axis.x();
        for (int i = 0; i < 3; i++) res.coeffRef(i, i) = cos1_axis[i] * axis_[i] + c;
        return res;
    }
    Matrix3 matrix() const { return toRotationMatrix(); }
    AngleAxis inverse() const { return AngleAxis(-angle_, axis_); }
    Quaternion<T> operator*(const AngleAxis& o) const;
    Quaternion<T> operator*(const Quaternion<T>& o) const;
    template <class O> Vector3 operator*(const MatrixBase<O>& v) const { return toRotationMatrix() * v; }
};
typedef AngleAxis<double> AngleAxisd;
typedef AngleAxis<float> AngleAxisf;

template <class T> class Quaternion {
    Matrix<T, 4, 1> c_;   // x, y, z, w
public:
    typedef T Scalar;
    typedef Matrix<T, 3, 1> Vector3;
    typedef Matrix<T, 3, 3> Matrix3;
    typedef Matrix<T, 4, 1> Coefficients;
    Quaternion() {}
    Quaternion(const T& w, const T& x, const T& y, const T& z) { c_ << x, y, z, w; }
    explicit Quaternion(const T* d) { c_ << d[0], d[1], d[2], d[3]; }
    Quaternion(const Quaternion& o) : c_(o.c_) {}
    Quaternion(const AngleAxis<T>& aa) { *this = aa; }
    template <class O> explicit Quaternion(const MatrixBase<O>& m) { *this = m; }
    Quaternion& operator=(const Quaternion& o) { c_ = o.c_; return *this; }
    Quaternion& operator=(const AngleAxis<T>& aa) {
        const T ha = T(0.5) * aa.angle();
        w() = std::cos(ha);
        const Vector3 v = std::sin(ha) * aa.axis();
        x() = v.x(); y() = v.y(); z() = v.z();
        return *this;
    }
    template <class O> Quaternion& operator=(const MatrixBase<O>& mb) {
        const O& m = mb.derived();
        if (m.rows() == 4 && m.cols() == 1) { for (int i = 0; i < 4; i++) c_[i] = m.coeff(i, 0); return *this; }
        // rotation matrix -> quaternion (Shepperd's branches: trace, else the largest diagonal entry)
        T t = m.coeff(0, 0) + m.coeff(1, 1) + m.coeff(2, 2);
        if (t > T(0)) {
            t = std::sqrt(t + T(1.0));
            w() = T(0.5) * t;
            t = T(0.5) / t;
            x() = (m.coeff(2, 1) - m.coeff(1, 2)) * t;
            y() = (m.coeff(0, 2) - m.coeff(2, 0)) * t;
            z() = (m.coeff(1, 0) - m.coeff(0, 1)) * t;
        } else {
            Index i = 0;
            if (m.coeff(1, 1) > m.coeff(0, 0)) i = 1;
            if (m.coeff(2, 2) > m.coeff(i, i)) i = 2;
            const Index j = (i + 1) % 3, k = (j + 1) % 3;
            t = std::sqrt(m.coeff(i, i) - m.coeff(j, j) - m.coeff(k, k) + T(1.0));
            c_[i] = T(0.5) * t;
            t = T(0.5) / t;
            w() = (m.coeff(k, j) - m.coeff(j, k)) * t;
            c_[j] = (m.coeff(j, i) + m.coeff(i, j)) * t;
            c_[k] = (m.coeff(k, i) + m.coeff(i, k)) * t;
        }
        return *this;
    }
    static Quaternion Identity() { return Quaternion(T(1), T(0), T(0), T(0)); }
    Quaternion& setIdentity() { c_ << T(0), T(0), T(0), T(1); return *this; }
    T x() const { return c_[0]; } T y() const { return c_[1]; } T z() const { return c_[2]; } T w() const { return c_[3]; }
    T& x() { return c_[0]; } T& y() { return c_[1]; } T& z() { return c_[2]; } T& w() { return c_[3]; }
    const Coefficients& coeffs() const { return c_; }
    Coefficients& coeffs() { return c_; }
    Vector3 vec() const { return Vector3(c_[0], c_[1], c_[2]); }
    T squaredNorm() const { return c_.squaredNorm(); }
    T norm() const { return c_.norm(); }
    void normalize() { c_.normalize(); }
    Quaternion normalized() const { Quaternion q(*this); q.normalize(); return q; }
    Quaternion conjugate() const { return Quaternion(w(), -x(), -y(), -z()); }
    Quaternion inverse() const {
        const T n2 = squaredNorm();
        if (n2 > T(0)) { Quaternion q = conjugate(); q.c_ /= n2; return q; }
        Quaternion q; q.c_.setZero(); return q;
    }
    T dot(const Quaternion& o) const { return c_.dot(o.c_); }
    Quaternion operator*(const Quaternion& b) const {
        const Quaternion& a = *this;
        return Quaternion(a.w() * b.w() - a.x() * b.x() - a.y() * b.y() - a.z() * b.z(),
                          a.w() * b.x() + a.x() * b.w() + a.y() * b.z() - a.z() * b.y(),
                          a.w() * b.y() + a.y() * b.w() + a.z() * b.x() - a.x() * b.z(),
                          a.w() * b.z() + a.z() * b.w() + a.x() * b.y() - a.y() * b.x());
    }
    Quaternion& operator*=(const Quaternion& o) { *this = *this * o; return *this; }
    // rotate a vector: v + w*uv + q.vec x uv with uv = 2 (q.vec x v)
    template <class O> Vector3 operator*(const MatrixBase<O>& vb) const { return _transformVector(Vector3(vb)); }
    Vector3 _transformVector(const Vector3& v) const {
        Vector3 uv = vec().cross(v);
        uv += uv;
        return v + w() * uv + vec().cross(uv);
    }
    Matrix3 toRotationMatrix() const {
        Matrix3 res;
        const T tx = T(2) * x(), ty = T(2) * y(), tz = T(2) * z();
        const T twx = tx * w(), twy = ty * w(), twz = tz * w();
        const T txx = tx * x(), txy = ty * x(), txz = tz * x();
        const T tyy = ty * y(), tyz = tz * y(), tzz = tz * z();
        res.coeffRef(0, 0) = T(1) - (tyy + tzz); res.coeffRef(0, 1) = txy - twz; res.coeffRef(0, 2) = txz + twy;
        res.coeffRef(1, 0) = txy + twz; res.coeffRef(1, 1) = T(1) - (txx + tzz); res.coeffRef(1, 2) = tyz - twx;
        res.coeffRef(2, 0) = txz - twy; res.coeffRef(2, 1) = tyz + twx; res.coeffRef(2, 2) = T(1) - (txx + tyy);
        return res;
    }
    Matrix3 matrix() const { return toRotationMatrix(); }
    template <class T2> Quaternion<T2> cast() const { return Quaternion<T2>(T2(w()), T2(x()), T2(y()), T2(z())); }
};
typedef Quaternion<double> Quaterniond;
typedef Quaternion<float> Quaternionf;

template <class T> AngleAxis<T>::AngleAxis(const Quaternion<T>& q) {
    T n = q.vec().norm();
    if (n < std::numeric_limits<T>::epsilon()) n = q.vec().norm();
    if (n != T(0)) { angle_ = T(2) * std::atan2(n, std::abs(q.w())); if (q.w() < T(0)) n = -n; axis_ = q.vec() / n; }
    else { angle_ = T(0); axis_ << T(1), T(0), T(0); }
}
template <class T> Quaternion<T> AngleAxis<T>::operator*(const AngleAxis& o) const { return Quaternion<T>(*this) * Quaternion<T>(o); }
template <class T> Quaternion<T> AngleAxis<T>::operator*(const Quaternion<T>& o) const { return Quaternion<T>(*this) * o; }
template <class T> Quaternion<T> operator*(const Quaternion<T>& a, const AngleAxis<T>& b) { return a * Quaternion<T>(b); }

template <class T, int Dim, int Mode, int Opt> class Transform {
    Matrix<T, Dim + 1, Dim + 1> m_;
public:
    typedef T Scalar;
    typedef Matrix<T, Dim + 1, Dim + 1> MatrixType;
    typedef Matrix<T, Dim, Dim> LinearMatrixType;
    typedef Matrix<T, Dim, 1> VectorType;
    EIGEN_MAKE_ALIGNED_OPERATOR_NEW
    Transform() { m_.setIdentity(); }
    Transform(const Transform& o) : m_(o.m_) {}
    template <int M2, int O2> Transform(const Transform<T, Dim, M2, O2>& o) : m_(o.matrix()) {}
    explicit Transform(const Quaternion<T>& q) { m_.setIdentity(); linear() = q.toRotationMatrix(); }
    explicit Transform(const AngleAxis<T>& q) { m_.setIdentity(); linear() = q.toRotationMatrix(); }
    template <class O> explicit Transform(const MatrixBase<O>& o) { *this = o; }
    Transform& operator=(const Transform& o) { m_ = o.m_; return *this; }
    template <class O> Transform& operator=(const MatrixBase<O>& o) {
        if (o.derived().rows() == Dim) { m_.setIdentity(); linear() = o; } else m_ = o;
        return *this;
    }
    Transform& operator=(const Quaternion<T>& q) { m_.setIdentity(); linear() = q.toRotationMatrix(); return *this; }
    static Transform Identity() { return Transform(); }
    void setIdentity() { m_.setIdentity(); }
    const MatrixType& matrix() const { return m_; }
    MatrixType& matrix() { return m_; }
    View<T, Dim, Dim> linear() { return m_.template block<Dim, Dim>(0, 0); }
    View<const T, Dim, Dim> linear() const { return m_.template block<Dim, Dim>(0, 0); }
    View<T, Dim, Dim + 1> affine() { return m_.template block<Dim, Dim + 1>(0, 0); }
    View<const T, Dim, Dim + 1> affine() const { return m_.template block<Dim, Dim + 1>(0, 0); }
    View<T, Dim, 1> translation() { return m_.template block<Dim, 1>(0, Dim); }
    View<const T, Dim, 1> translation() const { return m_.template block<Dim, 1>(0, Dim); }
    // Eigen 3.3: an Isometry's rotation() is its linear part (no polar decomposition)
    LinearMatrixType rotation() const { return LinearMatrixType(linear()); }
    T operator()(Index i, Index j) const { return m_(i, j); }
    T& operator()(Index i, Index j) { return m_(i, j); }
    T* data() { return m_.data(); }
    const T* data() const { return m_.data(); }
    Transform operator*(const Transform& o) const { Transform r; r.m_ = m_ * o.m_; return r; }
    Transform& operator*=(const Transform& o) { m_ = m_ * o.m_; return *this; }
    template <class O> Matrix<T, internal::traits<O>::Rows, internal::traits<O>::Cols> operator*(const MatrixBase<O>& v) const {
        if (v.derived().rows() == Dim + 1) return Matrix<T, internal::traits<O>::Rows, internal::traits<O>::Cols>(m_ * v);
        Matrix<T, Dim, internal::traits<O>::Cols> r = linear() * v;
        for (Index j = 0; j < r.cols(); j++) for (Index i = 0; i < Dim; i++) r(i, j) += m_(i, Dim);
        return Matrix<T, internal::traits<O>::Rows, internal::traits<O>::Cols>(r);
    }
    Transform operator*(const Quaternion<T>& q) const { Transform r(*this); r.linear() = LinearMatrixType(linear()) * q.toRotationMatrix(); return r; }
    Transform inverse(TransformTraits = (TransformTraits)Mode) const {
        Transform r;
        if (Mode == Isometry) {
            LinearMatrixType Rt = LinearMatrixType(linear().transpose());
            r.linear() = Rt;
            r.translation() = -(Rt * VectorType(translation()));
        } else r.m_ = m_.inverse();
        return r;
    }
    template <class O> Transform& translate(const MatrixBase<O>& t) { translation() += LinearMatrixType(linear()) * t; return *this; }
    template <class O> Transform& pretranslate(const MatrixBase<O>& t) { translation() += t; return *this; }
    template <class RT> Transform& rotate(const RT& r) { linear() = LinearMatrixType(linear()) * LinearMatrixType(r.toRotationMatrix()); return *this; }
    template <class RT> Transform& prerotate(const RT& r) { m_.template block<Dim, Dim + 1>(0, 0) = LinearMatrixType(r.toRotationMatrix()) * Matrix<T, Dim, Dim + 1>(affine()); return *this; }
    template <class T2> Transform<T2, Dim, Mode, Opt> cast() const { Transform<T2, Dim, Mode, Opt> r; r.matrix() = m_.template cast<T2>(); return r; }
};
typedef Transform<double, 3, Isometry> Isometry3d;
typedef Transform<double, 2, Isometry> Isometry2d;
typedef Transform<double, 3, Affine> Affine3d;
typedef Transform<double, 2, Affine> Affine2d;
typedef Transform<float, 3, Isometry> Isometry3f;
typedef Transform<float, 3, Affine> Affine3f;

template <class T> class Rotation2D {
    T a_;
public:
    explicit Rotation2D(const T& a = T(0)) : a_(a) {}
    T angle() const { return a_; }
    T& angle() { return a_; }
    Matrix<T, 2, 2> toRotationMatrix() const { Matrix<T, 2, 2> r; const T s = std::sin(a_), c = std::cos(a_); r << c, -s, s, c; return r; }
    Matrix<T, 2, 2> matrix() const { return toRotationMatrix(); }
    template <class O> Matrix<T, 2, 1> operator*(const MatrixBase<O>& v) const { return toRotationMatrix() * v; }
    Rotation2D inverse() const { return Rotation2D(-a_); }
    Rotation2D operator*(const Rotation2D& o) const { return Rotation2D(a_ + o.a_); }
    template <class O> Rotation2D& fromRotationMatrix(const MatrixBase<O>& m) { a_ = std::atan2(m.derived().coeff(1, 0), m.derived().coeff(0, 0)); return *this; }
};
typedef Rotation2D<double> Rotation2Dd;

}  // namespace Eigen
