// oracle/eigprim.h — TEST INFRASTRUCTURE (CPU oracle), not product code.
// Restatement of Eigen 3.3's SelfAdjointEigenSolver<Matrix3d>::compute() (the iterative path the
// reference uses through include/peac/eig33sym.hpp:70-75): scale to [-1,1], closed-form 3x3
// Householder tridiagonalisation, implicit symmetric QR steps with Wilkinson shift, ascending sort.
// Eigen is not in the container: PARITY against the real library is UNPINNED.
#pragma once
namespace orc {
// A: symmetric 3x3 (only the lower triangle A[r][c], r>=c, is read).  evals ascending,
// evecs[r][c] = component r of eigenvector c.  Returns false if QR did not converge.
bool eig33_selfadjoint(const double A[3][3], double evals[3], double evecs[3][3]);
}  // namespace orc
