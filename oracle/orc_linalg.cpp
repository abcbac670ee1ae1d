// ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the shipped product path.
//
// Small dense linear algebra standing in for the Eigen 3.3.9 calls on the hot path (Eigen is
// pinned by third/Makefile:22-26 but not vendored, so it cannot be compiled here):
//   LLT            regression/LinearRegression.cpp:33-35, LinearRegressionScoreTest.cpp:227,243,
//                  LogisticRegression.cpp:299,330, SkatO.cpp:165-166
//   LDLT solve     regression/SkatO.cpp:85,151,156-158
//   SelfAdjointEigenSolver (eigenvalues only)  Skat.cpp:75-76, SkatO.cpp:351
// Eigenvalues use the cyclic Jacobi method (Golub & Van Loan §8.5) — deliberately a different
// algorithm from the device's tridiagonalisation + Sturm bisection so the two check each other.
#include "orc_linalg.h"
#include <algorithm>
#include <cmath>
#include "orc_api.h"

namespace orc {

Mat AtB(const Mat& A, const Mat& B, const double* w) {
  Mat C(A.c, B.c);
  const int64_t n = A.r;
  const int64_t CH = 2048;  // row chunk keeps both panels cache resident
  for (int64_t i0 = 0; i0 < n; i0 += CH) {
    const int64_t i1 = std::min(n, i0 + CH);
    for (int64_t q = 0; q < B.c; ++q) {
      const double* b = B.col(q);
      for (int64_t p = 0; p < A.c; ++p) {
        const double* a = A.col(p);
        double s = 0.0;
        if (w) {
          for (int64_t i = i0; i < i1; ++i) s += a[i] * w[i] * b[i];
        } else {
          for (int64_t i = i0; i < i1; ++i) s += a[i] * b[i];
        }
        C(p, q) += s;
      }
    }
  }
  return C;
}

Mat mul(const Mat& A, const Mat& B) {
  Mat C(A.r, B.c);
  const int64_t CH = 2048;
  for (int64_t i0 = 0; i0 < A.r; i0 += CH) {
    const int64_t i1 = std::min(A.r, i0 + CH);
    for (int64_t j = 0; j < B.c; ++j) {
      double* c = C.col(j);
      for (int64_t k = 0; k < A.c; ++k) {
        const double b = B(k, j);
        if (b == 0.0) continue;
        const double* a = A.col(k);
        for (int64_t i = i0; i < i1; ++i) c[i] += a[i] * b;
      }
    }
  }
  return C;
}

Mat transpose(const Mat& A) {
  Mat T(A.c, A.r);
  for (int64_t j = 0; j < A.c; ++j)
    for (int64_t i = 0; i < A.r; ++i) T(j, i) = A(i, j);
  return T;
}

bool cholesky(const Mat& A, Mat* Lout) {
  const int64_t n = A.r;
  Mat L(n, n);
  for (int64_t j = 0; j < n; ++j) {
    double s = A(j, j);
    for (int64_t k = 0; k < j; ++k) s -= L(j, k) * L(j, k);
    if (!(s > 0.0)) return false;
    const double d = std::sqrt(s);
    L(j, j) = d;
    for (int64_t i = j + 1; i < n; ++i) {
      double t = A(i, j);
      for (int64_t k = 0; k < j; ++k) t -= L(i, k) * L(j, k);
      L(i, j) = t / d;
    }
  }
  *Lout = L;
  return true;
}

bool chol_solve(const Mat& A, const Mat& B, Mat* Xout) {
  Mat L;
  if (!cholesky(A, &L)) return false;
  const int64_t n = A.r;
  Mat X = B;
  for (int64_t c = 0; c < B.c; ++c) {
    for (int64_t i = 0; i < n; ++i) {
      double t = X(i, c);
      for (int64_t k = 0; k < i; ++k) t -= L(i, k) * X(k, c);
      X(i, c) = t / L(i, i);
    }
    for (int64_t i = n - 1; i >= 0; --i) {
      double t = X(i, c);
      for (int64_t k = i + 1; k < n; ++k) t -= L(k, i) * X(k, c);
      X(i, c) = t / L(i, i);
    }
  }
  *Xout = X;
  return true;
}

// Gaussian elimination with partial pivoting: for the small, well-conditioned X'VX systems on
// the hot path it agrees with Eigen's pivoted LDLT to rounding.
bool sym_solve(const Mat& A, const Mat& B, Mat* Xout) {
  const int64_t n = A.r;
  Mat M = A, X = B;
  for (int64_t k = 0; k < n; ++k) {
    int64_t piv = k;
    double best = std::fabs(M(k, k));
    for (int64_t i = k + 1; i < n; ++i)
      if (std::fabs(M(i, k)) > best) {
        best = std::fabs(M(i, k));
        piv = i;
      }
    if (best == 0.0) return false;
    if (piv != k) {
      for (int64_t j = 0; j < n; ++j) std::swap(M(k, j), M(piv, j));
      for (int64_t j = 0; j < X.c; ++j) std::swap(X(k, j), X(piv, j));
    }
    for (int64_t i = k + 1; i < n; ++i) {
      const double f = M(i, k) / M(k, k);
      if (f == 0.0) continue;
      for (int64_t j = k; j < n; ++j) M(i, j) -= f * M(k, j);
      for (int64_t j = 0; j < X.c; ++j) X(i, j) -= f * X(k, j);
    }
  }
  for (int64_t c = 0; c < X.c; ++c)
    for (int64_t i = n - 1; i >= 0; --i) {
      double t = X(i, c);
      for (int64_t k = i + 1; k < n; ++k) t -= M(i, k) * X(k, c);
      X(i, c) = t / M(i, i);
    }
  *Xout = X;
  return true;
}

std::vector<double> sym_eigvals(const Mat& Ain) {
  const int n = (int)Ain.r;
  Mat A = Ain;
  // symmetrise defensively
  for (int i = 0; i < n; ++i)
    for (int j = i + 1; j < n; ++j) {
      const double s = 0.5 * (A(i, j) + A(j, i));
      A(i, j) = A(j, i) = s;
    }
  for (int sweep = 0; sweep < 100; ++sweep) {
    double off = 0.0, diag = 0.0;
    for (int i = 0; i < n; ++i) {
      diag += A(i, i) * A(i, i);
      for (int j = i + 1; j < n; ++j) off += A(i, j) * A(i, j);
    }
    if (off == 0.0 || off <= 1e-60 * diag) break;
    for (int p = 0; p < n - 1; ++p)
      for (int q = p + 1; q < n; ++q) {
        const double apq = A(p, q);
        if (apq == 0.0) continue;
        const double app = A(p, p), aqq = A(q, q);
        const double theta = (aqq - app) / (2.0 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < n; ++k) {
          const double akp = A(k, p), akq = A(k, q);
          A(k, p) = c * akp - s * akq;
          A(k, q) = s * akp + c * akq;
        }
        for (int k = 0; k < n; ++k) {
          const double apk = A(p, k), aqk = A(q, k);
          A(p, k) = c * apk - s * aqk;
          A(q, k) = s * apk + c * aqk;
        }
        A(p, q) = A(q, p) = 0.0;
      }
  }
  std::vector<double> w(n);
  for (int i = 0; i < n; ++i) w[i] = A(i, i);
  std::sort(w.begin(), w.end());
  return w;
}

}  // namespace orc

extern "C" void orc_sym_eigvals(const double* A, int n, double* w) {
  orc::Mat M(n, n);
  for (int i = 0; i < n * n; ++i) M.a[i] = A[i];
  std::vector<double> v = orc::sym_eigvals(M);
  for (int i = 0; i < n; ++i) w[i] = v[i];
}
