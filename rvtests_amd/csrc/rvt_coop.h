// rvtests_amd — workgroup-cooperative primitives for the per-gene statistics kernel.
//
// Code written against `Coop` runs as one 256-thread workgroup per gene on the GPU and as a single
// thread in the host test harness (tid = 0, nt = 1, sync() a no-op), so the CPU test-suite can walk
// the very same flip-algebra / eigenvalue code the kernel runs.
#pragma once
#include "rvt_special.h"

namespace rvt {

struct Coop {
  int tid, nt;
  double* red;  // scratch for reductions: >= 64 doubles (LDS on the device)

  RVT_HD void sync() const {
#if defined(__HIP_DEVICE_COMPILE__)
    __syncthreads();
#endif
  }

  // Block-wide sum, same value returned to every thread.  Fixed reduction tree => deterministic.
  RVT_HD double sum(double v) const {
#if defined(__HIP_DEVICE_COMPILE__)
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    const int wave = tid >> 6, nw = (nt + 63) >> 6;
    __syncthreads();  // protect `red` against the previous reduction's readers
    if ((tid & 63) == 0) red[wave] = v;
    __syncthreads();
    double s = 0.0;
    for (int w = 0; w < nw; ++w) s += red[w];
    return s;
#else
    return v;
#endif
  }
};

// ---- eigenvalues of a real symmetric matrix --------------------------------------------------------
// Replaces Eigen::SelfAdjointEigenSolver (eigenvalues only) at regression/Skat.cpp:75-76 and
// regression/SkatO.cpp:351.  Householder reduction to tridiagonal form with the rank-2 update spread
// over the workgroup, then one Sturm-sequence bisection per eigenvalue (one thread each): no
// sequential QL sweep, which would leave 255 of 256 threads idle.

// A: n x n, column-major, full symmetric storage (destroyed).  d[n], e[n-1] receive the tridiagonal.
// v, w: length-n work vectors.
RVT_HD void coop_tridiagonalize(const Coop& co, double* A, int n, double* d, double* e, double* v, double* w) {
  for (int k = 0; k < n - 1; ++k) {
    const int m = n - k - 1;           // order of the trailing block
    double* col = A + (size_t)k * n;   // column k
    // squared norm of the part of column k below the sub-diagonal
    double part = 0.0;
    for (int i = k + 2 + co.tid; i < n; i += co.nt) part += col[i] * col[i];
    const double xnorm2 = co.sum(part);
    const double alpha = col[k + 1];
    double beta, tau;
    if (xnorm2 == 0.0) {
      beta = alpha;
      tau = 0.0;
    } else {
      const double nrm = sqrt(alpha * alpha + xnorm2);
      beta = (alpha >= 0.0) ? -nrm : nrm;
      tau = (beta - alpha) / beta;
    }
    if (co.tid == 0) {
      d[k] = col[k];
      e[k] = beta;
    }
    if (tau != 0.0) {
      const double scal = 1.0 / (alpha - beta);
      for (int i = co.tid; i < m; i += co.nt) v[i] = (i == 0) ? 1.0 : col[k + 1 + i] * scal;
      co.sync();
      // p = tau * A22 v ;  A22 = A[k+1.., k+1..]
      const double* A22 = A + (size_t)(k + 1) * n + (k + 1);
      double dotpart = 0.0;
      for (int i = co.tid; i < m; i += co.nt) {
        double s = 0.0;
        for (int j = 0; j < m; ++j) s += A22[(size_t)j * n + i] * v[j];
        s *= tau;
        w[i] = s;
        dotpart += s * v[i];
      }
      const double pv = co.sum(dotpart);
      const double a2 = -0.5 * tau * pv;
      for (int i = co.tid; i < m; i += co.nt) w[i] = w[i] + a2 * v[i];
      co.sync();
      // A22 -= v w' + w v'
      double* A22w = A + (size_t)(k + 1) * n + (k + 1);
      for (int idx = co.tid; idx < m * m; idx += co.nt) {
        const int i = idx % m, j = idx / m;
        A22w[(size_t)j * n + i] -= v[i] * w[j] + w[i] * v[j];
      }
    }
    co.sync();
  }
  if (co.tid == 0) d[n - 1] = A[(size_t)(n - 1) * n + (n - 1)];
  co.sync();
}

// number of eigenvalues of the tridiagonal (d, e) that are < x
// 1/q for the Sturm recurrence: on the device the hardware reciprocal refined by one Newton step (error ~1e-15,
// a third of the latency of the IEEE division sequence; the count is a sign pattern and insensitive to it)
RVT_HD double sturm_recip(double q) {
#if defined(__HIP_DEVICE_COMPILE__)
  const double r = __builtin_amdgcn_rcp(q);
  return r * (2.0 - q * r);
#else
  return 1.0 / q;
#endif
}

RVT_HD int sturm_count(const double* d, const double* e, int n, double x, double pivmin) {
  int cnt = 0;
  double q = d[0] - x;
  if (fabs(q) < pivmin) q = -pivmin;
  if (q < 0.0) ++cnt;
  for (int j = 1; j < n; ++j) {
    q = d[j] - x - (e[j - 1] * e[j - 1]) * sturm_recip(q);
    if (fabs(q) < pivmin) q = -pivmin;
    if (q < 0.0) ++cnt;
  }
  return cnt;
}

// all eigenvalues, ascending, into out[n]
RVT_HD void coop_tridiag_eigvals(const Coop& co, const double* d, const double* e, int n, double* out) {
  // Gershgorin interval and pivot floor (every thread computes the same numbers)
  double lo = d[0], hi = d[0], emax2 = 0.0;
  for (int j = 0; j < n; ++j) {
    const double r = (j > 0 ? fabs(e[j - 1]) : 0.0) + (j < n - 1 ? fabs(e[j]) : 0.0);
    lo = fmin(lo, d[j] - r);
    hi = fmax(hi, d[j] + r);
    if (j < n - 1) emax2 = fmax(emax2, e[j] * e[j]);
  }
  const double span = fmax(fabs(lo), fabs(hi));
  const double pivmin = fmax(DBL_MIN * fmax(1.0, emax2) * 4.0, DBL_MIN * 1024.0);
  lo -= 2.0 * kDblEps * span * n + 2.0 * pivmin;
  hi += 2.0 * kDblEps * span * n + 2.0 * pivmin;
  for (int idx = co.tid; idx < n; idx += co.nt) {
    // eigenvalue number idx (0 = smallest): largest x with count(x) <= idx
    double a = lo, b = hi;
    for (int it = 0; it < 200; ++it) {
      const double mid = 0.5 * (a + b);
      if (mid <= a || mid >= b) break;
      if (sturm_count(d, e, n, mid, pivmin) <= idx)
        a = mid;
      else
        b = mid;
      if (b - a <= 2.0 * kDblEps * fmax(fabs(a), fabs(b)) + 2.0 * pivmin) break;
    }
    out[idx] = 0.5 * (a + b);
  }
  co.sync();
}

// eigenvalues (ascending) of the symmetric n x n matrix in A (column-major; destroyed)
RVT_HD void coop_sym_eigvals(const Coop& co, double* A, int n, double* d, double* e, double* v, double* w,
                             double* out) {
  if (n == 1) {
    if (co.tid == 0) out[0] = A[0];
    co.sync();
    return;
  }
  coop_tridiagonalize(co, A, n, d, e, v, w);
  coop_tridiag_eigvals(co, d, e, n, out);
}

}  // namespace rvt
