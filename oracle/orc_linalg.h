// ORACLE — TEST INFRASTRUCTURE ONLY.  Small dense helpers shared by the oracle sources.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

namespace orc {

// column-major dense matrix
struct Mat {
  int64_t r = 0, c = 0;
  std::vector<double> a;
  Mat() {}
  Mat(int64_t r_, int64_t c_) : r(r_), c(c_), a((std::size_t)r_ * c_, 0.0) {}
  double& operator()(int64_t i, int64_t j) { return a[(std::size_t)i + (std::size_t)j * r]; }
  double operator()(int64_t i, int64_t j) const { return a[(std::size_t)i + (std::size_t)j * r]; }
  double* col(int64_t j) { return a.data() + (std::size_t)j * r; }
  const double* col(int64_t j) const { return a.data() + (std::size_t)j * r; }
};

// C = A' * B   (A: n x p, B: n x q) -> p x q ; optional per-row weight w[n]
Mat AtB(const Mat& A, const Mat& B, const double* w = nullptr);
// C = A * B
Mat mul(const Mat& A, const Mat& B);
Mat transpose(const Mat& A);
// Cholesky A = L L' (lower); returns false if not positive definite
bool cholesky(const Mat& A, Mat* L);
// solve A X = B for SPD A through Cholesky; false if the factorisation fails
bool chol_solve(const Mat& A, const Mat& B, Mat* X);
// symmetric solve with a pivoted LDL'-class elimination (used where the reference calls .ldlt())
bool sym_solve(const Mat& A, const Mat& B, Mat* X);
// eigenvalues of symmetric A, ascending (cyclic Jacobi)
std::vector<double> sym_eigvals(const Mat& A);

}  // namespace orc
