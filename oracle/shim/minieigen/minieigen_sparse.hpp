// oracle/shim/minieigen/minieigen_sparse.hpp — TEST INFRASTRUCTURE, not product code.
// Just enough of Eigen's sparse module for the REAL Thirdparty/g2o/g2o/solvers/linear_solver_eigen.h to compile and run:
// a compressed-column matrix filled from triplets, and SimplicialLDLT as an un-pivoted LDL^T of the (upper-stored)
// symmetric matrix.  Eigen applies a fill-reducing AMD permutation first; that only changes the rounding of the solve,
// so the natural order is used here (and minimum_degree_ordering returns the identity).
#pragma once
#include <map>
#include "minieigen.hpp"

namespace Eigen {

template <class T, class I = int> class Triplet {
    I r_, c_; T v_;
public:
    Triplet() : r_(0), c_(0), v_(0) {}
    Triplet(const I& r, const I& c, const T& v = T(0)) : r_(r), c_(c), v_(v) {}
    const I& row() const { return r_; }
    const I& col() const { return c_; }
    const T& value() const { return v_; }
};

template <int SR = Dynamic, int MR = SR, class I = int> class PermutationMatrix {
    Matrix<I, Dynamic, 1> idx_;
public:
    PermutationMatrix() {}
    explicit PermutationMatrix(Index n) { resize(n); }
    void resize(Index n) { idx_.resize(n); }
    Index size() const { return idx_.size(); }
    Index rows() const { return idx_.size(); }
    Index cols() const { return idx_.size(); }
    Matrix<I, Dynamic, 1>& indices() { return idx_; }
    const Matrix<I, Dynamic, 1>& indices() const { return idx_; }
    void setIdentity(Index n) { resize(n); for (Index i = 0; i < n; i++) idx_[i] = I(i); }
    PermutationMatrix inverse() const { PermutationMatrix r(size()); for (Index i = 0; i < size(); i++) r.idx_[idx_[i]] = I(i); return r; }
};

template <class SM, unsigned UpLo> struct SparseSelfAdjointView;
template <class SM, unsigned UpLo, class Perm> struct SparseSymmetricPermutationProduct { const SM& m; const Perm& p; };

template <class T, int Opt = ColMajor, class I = int> class SparseMatrix {
    Index r_ = 0, c_ = 0;
    std::vector<T> val_;
    std::vector<I> inner_, outer_;
public:
    typedef T Scalar;
    typedef I StorageIndex;
    SparseMatrix() { outer_.assign(1, 0); }
    SparseMatrix(Index r, Index c) { resize(r, c); }
    void resize(Index r, Index c) { r_ = r; c_ = c; val_.clear(); inner_.clear(); outer_.assign(size_t(c + 1), 0); }
    Index rows() const { return r_; }
    Index cols() const { return c_; }
    Index nonZeros() const { return Index(val_.size()); }
    T* valuePtr() { return val_.data(); }
    const T* valuePtr() const { return val_.data(); }
    I* innerIndexPtr() { return inner_.data(); }
    const I* innerIndexPtr() const { return inner_.data(); }
    I* outerIndexPtr() { return outer_.data(); }
    const I* outerIndexPtr() const { return outer_.data(); }
    template <class It> void setFromTriplets(It b, It e) {
        std::vector<std::map<I, T>> cols;
        cols.resize(size_t(c_));
        for (It it = b; it != e; ++it) cols[size_t(it->col())][it->row()] += it->value();
        val_.clear(); inner_.clear(); outer_.assign(size_t(c_ + 1), 0);
        for (Index c = 0; c < c_; c++) {
            for (auto& kv : cols[size_t(c)]) { inner_.push_back(kv.first); val_.push_back(kv.second); }
            outer_[size_t(c + 1)] = I(val_.size());
        }
    }
    T coeff(Index r, Index c) const { for (I k = outer_[c]; k < outer_[c + 1]; k++) if (inner_[k] == r) return val_[k]; return T(0); }
    template <unsigned UpLo> SparseSelfAdjointView<SparseMatrix, UpLo> selfadjointView() { return SparseSelfAdjointView<SparseMatrix, UpLo>{*this}; }
    template <unsigned UpLo> SparseSelfAdjointView<const SparseMatrix, UpLo> selfadjointView() const { return SparseSelfAdjointView<const SparseMatrix, UpLo>{*this}; }
    template <class SM, unsigned UpLo> SparseMatrix& operator=(const SparseSelfAdjointView<SM, UpLo>& v) {   // full symmetric matrix from one triangle
        std::vector<Triplet<T, I>> t;
        for (Index c = 0; c < v.m.cols(); c++)
            for (I k = v.m.outerIndexPtr()[c]; k < v.m.outerIndexPtr()[c + 1]; k++) {
                const I r = v.m.innerIndexPtr()[k];
                if ((UpLo == Upper && r > c) || (UpLo == Lower && r < c)) continue;
                t.push_back(Triplet<T, I>(r, I(c), v.m.valuePtr()[k]));
                if (r != c) t.push_back(Triplet<T, I>(I(c), r, v.m.valuePtr()[k]));
            }
        resize(v.m.rows(), v.m.cols());
        setFromTriplets(t.begin(), t.end());
        return *this;
    }
    const SparseMatrix& nestedExpression() const { return *this; }
};

template <class SM, unsigned UpLo> struct SparseSelfAdjointView {
    SM& m;
    template <class Perm> SparseSymmetricPermutationProduct<SM, UpLo, Perm> twistedBy(const Perm& p) const { return SparseSymmetricPermutationProduct<SM, UpLo, Perm>{m, p}; }
    // natural order is kept (see the header comment): the permuted copy is the matrix itself
    template <class SM2, unsigned U2, class Perm> SparseSelfAdjointView& operator=(const SparseSymmetricPermutationProduct<SM2, U2, Perm>& o) {
        typedef typename std::remove_const<SM2>::type Plain;
        const_cast<Plain&>(static_cast<const Plain&>(m)) = static_cast<const Plain&>(o.m);
        return *this;
    }
};

namespace internal {
template <class SM, class Perm> void minimum_degree_ordering(SM& C, Perm& p) { p.setIdentity(C.cols()); }
}

template <class SM, int UpLo_ = Lower> class SimplicialLDLT {
public:
    typedef typename SM::Scalar Scalar;
    typedef SM CholMatrixType;
    typedef SM MatrixType;
    enum { UpLo = UpLo_ };
protected:
    Matrix<Scalar, Dynamic, Dynamic> m_;   // unit-lower L below the diagonal, D on it
    ComputationInfo info_ = Success;
    SM l_;
    PermutationMatrix<Dynamic, Dynamic> m_P, m_Pinv;
public:
    SimplicialLDLT() {}
    void analyzePattern(const SM&) {}
    void analyzePattern_preordered(const SM&, bool) {}
    void factorize(const SM& a) {
        const Index n = a.cols();
        m_ = Matrix<Scalar, Dynamic, Dynamic>::Zero(n, n);
        for (Index c = 0; c < n; c++)
            for (auto k = a.outerIndexPtr()[c]; k < a.outerIndexPtr()[c + 1]; k++) {
                const Index r = a.innerIndexPtr()[k];
                if ((UpLo_ == Upper && r > c) || (UpLo_ == Lower && r < c)) continue;
                m_(r, c) = a.valuePtr()[k];
                m_(c, r) = a.valuePtr()[k];
            }
        info_ = Success;
        for (Index k = 0; k < n; k++) {
            Scalar d = m_(k, k);
            for (Index j = 0; j < k; j++) d -= m_(k, j) * m_(k, j) * m_(j, j);
            m_(k, k) = d;
            if (d == Scalar(0)) { info_ = NumericalIssue; return; }
            for (Index i = k + 1; i < n; i++) {
                Scalar s = m_(i, k);
                for (Index j = 0; j < k; j++) s -= m_(i, j) * m_(k, j) * m_(j, j);
                m_(i, k) = s / d;
            }
        }
        std::vector<Triplet<Scalar>> t;
        for (Index c = 0; c < n; c++) for (Index r = c + 1; r < n; r++) if (m_(r, c) != Scalar(0)) t.push_back(Triplet<Scalar>(int(r), int(c), m_(r, c)));
        l_.resize(n, n);
        l_.setFromTriplets(t.begin(), t.end());
    }
    void compute(const SM& a) { analyzePattern(a); factorize(a); }
    ComputationInfo info() const { return info_; }
    const SM& matrixL() const { return l_; }
    template <class O> Matrix<Scalar, Dynamic, internal::traits<O>::Cols> solve(const MatrixBase<O>& b) const {
        const Index n = m_.rows(), mc = b.derived().cols();
        Matrix<Scalar, Dynamic, internal::traits<O>::Cols> x(n, mc, 0);
        for (Index c = 0; c < mc; c++) {
            for (Index i = 0; i < n; i++) { Scalar s = b.derived().coeff(i, c); for (Index k = 0; k < i; k++) s -= m_(i, k) * x(k, c); x(i, c) = s; }
            for (Index i = 0; i < n; i++) x(i, c) /= m_(i, i);
            for (Index i = n - 1; i >= 0; i--) { Scalar s = x(i, c); for (Index k = i + 1; k < n; k++) s -= m_(k, i) * x(k, c); x(i, c) = s; }
        }
        return x;
    }
};

}  // namespace Eigen
