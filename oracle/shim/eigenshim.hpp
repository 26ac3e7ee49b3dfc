// oracle/shim/eigenshim.hpp — TEST INFRASTRUCTURE, not product code.
// The few Eigen expressions used by the reference's PEAC plane extractor (include/peac/eig33sym.hpp:70-75,
// include/PlaneExtractor.h: Eigen::Vector3d vertices), so those REAL reference sources compile without
// Eigen.  SelfAdjointEigenSolver<Matrix3d> forwards to the restatement in oracle/eigprim.cpp.
#pragma once
#include "../eigprim.h"

namespace Eigen {
enum { ColMajor = 0, RowMajor = 1 };

template <typename T, int R, int C, int Opt = ColMajor> struct Matrix {
    T d[R * C];
    Matrix() { for (int i = 0; i < R * C; i++) d[i] = T(0); }
    Matrix(T x, T y, T z) { static_assert(R * C == 3, "vector3 ctor"); d[0] = x; d[1] = y; d[2] = z; }
    static Matrix Zero() { return Matrix(); }
    T& operator[](int i) { return d[i]; }
    const T& operator[](int i) const { return d[i]; }
    T& operator()(int i) { return d[i]; }
    const T& operator()(int i) const { return d[i]; }
    // (v << a, b, c), v.cross(w), v / s as the reference's LSDextractor.cpp uses them (Eigen evaluates them coefficient-wise)
    struct CommaInit { Matrix* m; int i; template <typename U> CommaInit& operator,(U v) { m->d[i++] = (T)v; return *this; }
                       CommaInit& operator,(const Matrix& o) { for (int k = 0; k < R * C; k++) m->d[i++] = o.d[k]; return *this; } };
    template <typename U> CommaInit operator<<(U v) { d[0] = (T)v; return CommaInit{this, 1}; }
    CommaInit operator<<(const Matrix& o) { for (int k = 0; k < R * C; k++) d[k] = o.d[k]; return CommaInit{this, R * C}; }
    Matrix cross(const Matrix& o) const { static_assert(R * C == 3, "cross"); Matrix r; r.d[0] = d[1] * o.d[2] - d[2] * o.d[1]; r.d[1] = d[2] * o.d[0] - d[0] * o.d[2]; r.d[2] = d[0] * o.d[1] - d[1] * o.d[0]; return r; }
    Matrix operator/(T s) const { Matrix r; for (int k = 0; k < R * C; k++) r.d[k] = d[k] / s; return r; }
    T* data() { return d; }
    T& operator()(int r, int c) { return Opt == RowMajor ? d[r * C + c] : d[c * R + r]; }
    const T& operator()(int r, int c) const { return Opt == RowMajor ? d[r * C + c] : d[c * R + r]; }
};
typedef Matrix<double, 3, 1> Vector3d;
typedef Matrix<double, 3, 3> Matrix3d;

template <typename M> struct Map;
template <typename T, int R, int C, int Opt> struct Map<Matrix<T, R, C, Opt>> {
    T* p;
    Map(T* ptr, int = R, int = C) : p(ptr) {}
    T& coeff(int r, int c) const { return Opt == RowMajor ? p[r * C + c] : p[c * R + r]; }
    template <int O2> Map& operator=(const Matrix<T, R, C, O2>& m) {
        for (int r = 0; r < R; r++) for (int c = 0; c < C; c++) coeff(r, c) = m(r, c);
        return *this;
    }
};

template <typename M> class SelfAdjointEigenSolver;
template <> class SelfAdjointEigenSolver<Matrix3d> {
public:
    explicit SelfAdjointEigenSolver(const Map<Matrix3d>& m) {
        double A[3][3], ev[3], Q[3][3];
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) A[r][c] = m.coeff(r, c);
        orc::eig33_selfadjoint(A, ev, Q);
        for (int i = 0; i < 3; i++) vals_[i] = ev[i];
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) vecs_(r, c) = Q[r][c];
    }
    const Vector3d& eigenvalues() const { return vals_; }
    const Matrix3d& eigenvectors() const { return vecs_; }
private:
    Vector3d vals_;
    Matrix3d vecs_;
};
}  // namespace Eigen
