// rvtests_amd — workgroup-cooperative primitives for the per-gene statistics kernel.
//
// Code written against `Coop` runs as one 256-thread workgroup per gene on the GPU and as a single
// thread in the host test harness (tid = 0, nt = 1, sync() a no-op), so the CPU test-suite can walk
// the very same flip-algebra / eigenvalue code the kernel runs.
#pragma once
#include <string.h>
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

// ---- Sturm count, division-free -------------------------------------------------------------------------------------
// Number of eigenvalues of the tridiagonal (d, e) below x = number of sign changes in p_0 = 1, p_1 = d_0 - x,
// p_j = (d_{j-1} - x) p_{j-1} - e_{j-2}^2 p_{j-2}.  The quotient form q_j = p_j / p_{j-1} (LAPACK's dlaebz) pays a division
// per row — on this machine a quarter-rate reciprocal plus its Newton step, and the count loop is THE instruction stream
// of the eigenvalue stage (13 problems x ~55 bisection steps x n rows per gene).  The product form costs a subtraction, a
// multiplication and an FMA per row; its historical weakness, over- and underflow, is handled by rescaling the pair
// (p_{j-1}, p_j) by a power of two every two rows (exponent instructions, no rounding: signs and ratios are unchanged).
// The computed sequence is the exact one of a matrix whose entries are perturbed by a few ulp (Wilkinson), as for the
// quotient form.  `e2` holds max(e_j^2, floor) of the matrix scaled to a span below 1 (coop_tridiag_eigvals): the positive
// floor keeps an exactly split matrix from zeroing the whole tail of the sequence.
RVT_HD int rvt_hi_word(double v) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __double2hiint(v);
#else
  long long b;
  memcpy(&b, &v, 8);
  return (int)(b >> 32);
#endif
}
RVT_HD void sturm_rescale(double& p0, double& p1) {  // the larger of the pair into [1/2, 1)
  const double m = fmax(fabs(p0), fabs(p1));            // (> 0: the floor on e2 keeps the pair from vanishing together)
#if defined(__HIP_DEVICE_COMPILE__)
  const int k = __builtin_amdgcn_frexp_exp(m);
  p0 = __builtin_amdgcn_ldexp(p0, -k);
  p1 = __builtin_amdgcn_ldexp(p1, -k);
#else
  int k = 0;
  if (m != 0.0) (void)frexp(m, &k);
  p0 = ldexp(p0, -k);
  p1 = ldexp(p1, -k);
#endif
}
// An exactly zero member p_j (x hit an eigenvalue of a leading block) needs no special case: the fma returns +0, the next
// member is -e2 p_{j-1} != 0 (e2 is floored above zero), so the two comparisons around the zero count ONE sign change
// between p_{j-1} and p_{j+1}, whichever side the zero is booked on — the same total as dlaebz's "zero pivot counts as
// negative".  Only a zero LAST member differs (x is an eigenvalue of the whole matrix: whether it counts as "below x" is the
// open / closed end of the bisection interval, which converges to it either way).
// The same recurrence, also returning the LAST member p_n(x) = det(T - x I) up to sign, as mant * 2^expo (the rescalings
// are exact, their exponents are summed).  Signs and counts are exactly sturm_count's.
RVT_HD int sturm_eval(const double* d, const double* e2, int n, double x, double* mant, int* expo) {
  double p0 = 1.0, p1 = d[0] - x;
  int cnt = (int)((unsigned)rvt_hi_word(p1) >> 31), E = 0;
  int j = 1;
  // the rows of the NEXT pair are fetched while the current pair is multiplied (the reads are LDS broadcasts whose latency
  // would otherwise sit on the chain of dependent FMAs: round 4's loop waited for its own two reads every iteration)
  double dn0 = 0.0, dn1 = 0.0, en0 = 0.0, en1 = 0.0;
  if (n > 2) {
    dn0 = d[1];
    dn1 = d[2];
    en0 = e2[0];
    en1 = e2[1];
  }
  for (; j + 1 < n; j += 2) {
    const double c0 = dn0, c1 = dn1, f0 = en0, f1 = en1;
    {
      const int q = (j + 3 < n) ? j + 2 : j;  // (the last pair re-reads itself: nothing is fetched out of range)
      dn0 = d[q];
      dn1 = d[q + 1];
      en0 = e2[q - 1];
      en1 = e2[q];
    }
    double pn = fma(c0 - x, p1, -(f0 * p0));
    cnt += (int)((unsigned)(rvt_hi_word(pn) ^ rvt_hi_word(p1)) >> 31);
    p0 = p1;
    p1 = pn;
    pn = fma(c1 - x, p1, -(f1 * p0));
    cnt += (int)((unsigned)(rvt_hi_word(pn) ^ rvt_hi_word(p1)) >> 31);
    p0 = p1;
    p1 = pn;
    const double m = fmax(fabs(p0), fabs(p1));  // (> 0: the floor on e2 keeps the pair from vanishing together)
#if defined(__HIP_DEVICE_COMPILE__)
    const int k = __builtin_amdgcn_frexp_exp(m);
    p0 = __builtin_amdgcn_ldexp(p0, -k);
    p1 = __builtin_amdgcn_ldexp(p1, -k);
#else
    int k = 0;
    if (m != 0.0) (void)frexp(m, &k);
    p0 = ldexp(p0, -k);
    p1 = ldexp(p1, -k);
#endif
    E += k;
  }
  if (j < n) {
    const double pn = fma(d[j] - x, p1, -(e2[j - 1] * p0));
    cnt += (int)((unsigned)(rvt_hi_word(pn) ^ rvt_hi_word(p1)) >> 31);
    p1 = pn;
  }
  *mant = p1;
  *expo = E;
  return cnt;
}
RVT_HD int sturm_count(const double* d, const double* e2, int n, double x) {
  double m;
  int e;
  return sturm_eval(d, e2, n, x, &m, &e);
}

// Eigenvalue number idx (0 = smallest) of the scaled tridiagonal (d, e2) inside [lo, hi] (count(lo) = 0, count(hi) = n), to
// the width tol_abs + 2 eps max(|a|, |b|).  Round 5: bisection only until the values of det(T - x I) at the two ends of the
// bracket differ in sign, then interpolation on those values (regula falsi with the Illinois modification, bisection whenever
// two steps failed to halve the bracket) — the Sturm COUNT still decides on which side of x the eigenvalue lies, so the
// bracket invariant count(a) <= idx < count(b) is exactly the bisection's and a useless value can only cost steps.
// Pure bisection took ~55 evaluations of n rows per eigenvalue; this takes ~12 to isolate + ~8 (measured: hostcheck).
RVT_HD double sturm_eigenvalue(const double* d, const double* e2, int n, int idx, double lo, double hi, double span,
                               double pivmin, int* evals_out = nullptr) {
  double a = lo, b = hi, fa = 0.0, fb = 0.0;  // f = mant * 2^expo of det(T - x I); 0 = not known
  int ea = 0, eb = 0, side = 0, evals = 0;
  double w2 = b - a, w1 = b - a;  // bracket widths two / one step(s) ago
  for (int it = 0; it < 200; ++it) {
    const double tol = 2.0 * kDblEps * fmax(fabs(a), fabs(b)) + (span * 0x1p-62 + 2.0 * pivmin);
    const double width = b - a;
    if (width <= tol) break;
    double x = 0.5 * (a + b);
    const bool slow = width > 0.5 * w2;  // the last two steps did not halve the bracket: bisect now
    if (!slow && fa != 0.0 && fb != 0.0 && ((rvt_hi_word(fa) ^ rvt_hi_word(fb)) < 0)) {
      // x = b - (b - a) / (1 - fa / fb), fa / fb < 0
      const int de = ea - eb;
      if (de > -1000 && de < 1000) {
        const double r = ldexp(fa / fb, de);
        double t = width / (1.0 - r);          // distance from b, in (0, width)
        const double edge = 0.5 * tol;         // never closer to an end than half the final width: the bracket then closes
        if (!(t > edge)) t = edge;
        if (!(t < width - edge)) t = width - edge;
        x = b - t;
        if (!(x > a && x < b)) x = 0.5 * (a + b);
      }
    }
    if (x <= a || x >= b) break;
    double m;
    int e;
    const int cnt = sturm_eval(d, e2, n, x, &m, &e);
    ++evals;
    if (cnt <= idx) {
      a = x;
      fa = m;
      ea = e;
      if (side == 1) fb *= 0.5;  // Illinois: the end that stays twice in a row gives way
      side = 1;
    } else {
      b = x;
      fb = m;
      eb = e;
      if (side == 2) fa *= 0.5;
      side = 2;
    }
    w2 = w1;
    w1 = width;
  }
  if (evals_out) *evals_out = evals;
  return 0.5 * (a + b);
}

// All eigenvalues, ascending, into out[n].  d and e (n - 1 off-diagonal entries) are OVERWRITTEN: the matrix is scaled by
// the power of two that brings its Gershgorin span into [1/2, 1) (exact; the eigenvalues are scaled back the same way), so
// that a row multiplies the Sturm pair by at most 3, and e becomes max(e^2, 2^-200) — a floor of 2^-100 of the span on the
// coupling, 15 orders below rounding, which bounds how fast the pair can shrink (two rows between rescalings stay some 500
// binary orders clear of underflow).
// (the two halves separately: rvt_gene.h looks at the scaled form before it decides whether the eigenvalues are needed)
struct TridiagScale {
  double lo, hi, span;
  int sh;
};
RVT_HD TridiagScale coop_tridiag_scale(const Coop& co, double* d, double* e, int n) {
  // Gershgorin interval (every thread computes the same numbers)
  double lo = d[0], hi = d[0];
  for (int j = 0; j < n; ++j) {
    const double r = (j > 0 ? fabs(e[j - 1]) : 0.0) + (j < n - 1 ? fabs(e[j]) : 0.0);
    lo = fmin(lo, d[j] - r);
    hi = fmax(hi, d[j] + r);
  }
  const double span0 = fmax(fabs(lo), fabs(hi));
  int sh = 0;
  if (span0 > 0.0 && span0 < INFINITY) (void)frexp(span0, &sh);
  co.sync();  // (every thread has read d and e)
  for (int j = co.tid; j < n; j += co.nt) {
    d[j] = ldexp(d[j], -sh);
    if (j < n - 1) {
      const double es = ldexp(e[j], -sh);
      e[j] = fmax(es * es, 0x1p-200);
    }
  }
  lo = ldexp(lo, -sh);
  hi = ldexp(hi, -sh);
  const double span = fmax(fabs(lo), fabs(hi));  // in [1/2, 1), or 0 for the zero matrix
  const double pivmin = DBL_MIN * 1024.0;
  TridiagScale ts;
  ts.lo = lo - (2.0 * kDblEps * span * n + 2.0 * pivmin);
  ts.hi = hi + (2.0 * kDblEps * span * n + 2.0 * pivmin);
  ts.span = span;
  ts.sh = sh;
  co.sync();
  return ts;
}
RVT_HD void coop_tridiag_eigvals_scaled(const Coop& co, const double* d, const double* e2, int n, const TridiagScale& ts,
                                        double* out) {
  const double pivmin = DBL_MIN * 1024.0;
  for (int idx = co.tid; idx < n; idx += co.nt) {
    // eigenvalue number idx (0 = smallest): largest x with count(x) <= idx
    // (relative for eigenvalues of the matrix's own size; never finer than 2^-62 of the span — 2^-10 of the rounding
    //  the tridiagonal form itself carries: an eigenvalue that is zero to rounding stops after ~62 halvings, not 200)
    const double ev = sturm_eigenvalue(d, e2, n, idx, ts.lo, ts.hi, ts.span, pivmin);
    out[idx] = ldexp(ev, ts.sh);
  }
  co.sync();
}
RVT_HD void coop_tridiag_eigvals(const Coop& co, double* d, double* e, int n, double* out) {
  const TridiagScale ts = coop_tridiag_scale(co, d, e, n);
  coop_tridiag_eigvals_scaled(co, d, e, n, ts, out);
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
