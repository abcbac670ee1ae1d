// ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the shipped product path.
//
// Re-entrant CPU restatement of Davies' algorithm AS 155 as the reference runs it:
//   regression/qfc.c:297-436 (qf), :82-89 (counter), :95-113 (log1), :115-134 (order),
//   :137-155 (errbd), :157-178 (ctff), :180-215 (truncation), :217-238 (findu),
//   :241-270 (integrate), :272-304 (cfe)
// called from MixtureChiSquare::getPvalue (regression/MixtureChiSquare.cpp:7-29) with
// sigma = 0, lim = 10000, acc = 1e-6, df = 1, noncentrality = 0 (MixtureChiSquare.h:7,31-33).
// File-scope statics of the reference become members; the longjmp on count > lim becomes a C++
// exception caught in qf() (fault 4).  Operation order inside every loop is the reference's.
// Pinned against the compiled reference fragment (oracle/_ref) by tests/test_oracle_ref.py and
// against tests/golden/davies_liu.json.
#include <cmath>
#include <vector>
#include "orc_api.h"

namespace {

const double kPi = 3.14159265358979;  // qfc.c:22 (truncated on purpose)
const double kLog28 = .0866;          // qfc.c:23

struct CountOverflow {};

struct Davies {
  double sigsq, lmax, lmin, mean, c;
  double intl, ersm;
  int count, r, lim;
  bool ndtsrt, fail;
  const int* n;
  std::vector<int> th;
  const double* lb;
  const double* nc;

  static double exp1(double x) { return x < -50.0 ? 0.0 : std::exp(x); }
  static double square(double x) { return x * x; }
  static double cube(double x) { return x * x * x; }

  void counter() {
    count = count + 1;
    if (count > lim) throw CountOverflow();
  }

  static double log1(double x, bool first) {
    if (std::fabs(x) > 0.1) return first ? std::log(1.0 + x) : (std::log(1.0 + x) - x);
    double s, s1, term, y, k;
    y = x / (2.0 + x);
    term = 2.0 * cube(y);
    k = 3.0;
    s = (first ? 2.0 : -x) * y;
    y = square(y);
    for (s1 = s + term / k; s1 != s; s1 = s + term / k) {
      k = k + 2.0;
      term = term * y;
      s = s1;
    }
    return s;
  }

  void order() {
    for (int j = 0; j < r; j++) {
      const double lj = std::fabs(lb[j]);
      int k;
      bool placed = false;
      for (k = j - 1; k >= 0; k--) {
        if (lj > std::fabs(lb[th[k]]))
          th[k + 1] = th[k];
        else {
          placed = true;
          break;
        }
      }
      if (!placed) k = -1;
      th[k + 1] = j;
    }
    ndtsrt = false;
  }

  double errbd(double u, double* cx) {
    counter();
    double xconst = u * sigsq, sum1 = u * xconst;
    u = 2.0 * u;
    for (int j = r - 1; j >= 0; j--) {
      const int nj = n[j];
      const double lj = lb[j], ncj = nc[j];
      const double x = u * lj, y = 1.0 - x;
      xconst = xconst + lj * (ncj / y + nj) / y;
      sum1 = sum1 + ncj * square(x / y) + nj * (square(x) / y + log1(-x, false));
    }
    *cx = xconst;
    return exp1(-0.5 * sum1);
  }

  double ctff(double accx, double* upn) {
    double u1, u2, u, rb, xconst, c1, c2;
    u2 = *upn;
    u1 = 0.0;
    c1 = mean;
    rb = 2.0 * ((u2 > 0.0) ? lmax : lmin);
    for (u = u2 / (1.0 + u2 * rb); errbd(u, &c2) > accx; u = u2 / (1.0 + u2 * rb)) {
      u1 = u2;
      c1 = c2;
      u2 = 2.0 * u2;
    }
    for (u = (c1 - mean) / (c2 - mean); u < 0.9; u = (c1 - mean) / (c2 - mean)) {
      u = (u1 + u2) / 2.0;
      if (errbd(u / (1.0 + u * rb), &xconst) > accx) {
        u1 = u;
        c1 = xconst;
      } else {
        u2 = u;
        c2 = xconst;
      }
    }
    *upn = u2;
    return c2;
  }

  double truncation(double u, double tausq) {
    counter();
    double sum1 = 0.0, prod2 = 0.0, prod3 = 0.0;
    int s = 0;
    const double sum2 = (sigsq + tausq) * square(u);
    double prod1 = 2.0 * sum2;
    u = 2.0 * u;
    for (int j = 0; j < r; j++) {
      const double lj = lb[j], ncj = nc[j];
      const int nj = n[j];
      const double x = square(u * lj);
      sum1 = sum1 + ncj * x / (1.0 + x);
      if (x > 1.0) {
        prod2 = prod2 + nj * std::log(x);
        prod3 = prod3 + nj * log1(x, true);
        s = s + nj;
      } else
        prod1 = prod1 + nj * log1(x, true);
    }
    sum1 = 0.5 * sum1;
    prod2 = prod1 + prod2;
    prod3 = prod1 + prod3;
    double x = exp1(-sum1 - 0.25 * prod2) / kPi;
    const double y = exp1(-sum1 - 0.25 * prod3) / kPi;
    double err1 = (s == 0) ? 1.0 : x * 2.0 / s;
    double err2 = (prod3 > 1.0) ? 2.5 * y : 1.0;
    if (err2 < err1) err1 = err2;
    x = 0.5 * sum2;
    err2 = (x <= y) ? 1.0 : y / x;
    return (err1 < err2) ? err1 : err2;
  }

  void findu(double* utx, double accx) {
    static const double divis[] = {2.0, 1.4, 1.2, 1.1};
    double ut = *utx, u = ut / 4.0;
    if (truncation(u, 0.0) > accx) {
      for (u = ut; truncation(u, 0.0) > accx; u = ut) ut = ut * 4.0;
    } else {
      ut = u;
      for (u = u / 4.0; truncation(u, 0.0) <= accx; u = u / 4.0) ut = u;
    }
    for (int i = 0; i < 4; i++) {
      u = ut / divis[i];
      if (truncation(u, 0.0) <= accx) ut = u;
    }
    *utx = ut;
  }

  void integrate(int nterm, double interv, double tausq, bool mainx) {
    const double inpi = interv / kPi;
    for (int k = nterm; k >= 0; k--) {
      const double u = (k + 0.5) * interv;
      double sum1 = -2.0 * u * c, sum2 = std::fabs(sum1);
      double sum3 = -0.5 * sigsq * square(u);
      for (int j = r - 1; j >= 0; j--) {
        const int nj = n[j];
        double x = 2.0 * lb[j] * u, y = square(x);
        sum3 = sum3 - 0.25 * nj * log1(y, true);
        y = nc[j] * x / (1.0 + y);
        const double z = nj * std::atan(x) + y;
        sum1 = sum1 + z;
        sum2 = sum2 + std::fabs(z);
        sum3 = sum3 - 0.5 * x * y;
      }
      double x = inpi * exp1(sum3) / u;
      if (!mainx) x = x * (1.0 - exp1(-0.5 * tausq * square(u)));
      sum1 = std::sin(0.5 * sum1) * x;
      sum2 = 0.5 * sum2 * x;
      intl = intl + sum1;
      ersm = ersm + sum2;
    }
  }

  double cfe(double x) {
    counter();
    if (ndtsrt) order();
    double axl = std::fabs(x);
    const double sxl = (x > 0.0) ? 1.0 : -1.0;
    double sum1 = 0.0;
    for (int j = r - 1; j >= 0; j--) {
      const int t = th[j];
      if (lb[t] * sxl > 0.0) {
        const double lj = std::fabs(lb[t]);
        const double axl1 = axl - lj * (n[t] + nc[t]), axl2 = lj / kLog28;
        if (axl1 > axl2)
          axl = axl1;
        else {
          if (axl > axl2) axl = axl2;
          sum1 = (axl - axl1) / lj;
          for (int k = j - 1; k >= 0; k--) sum1 = sum1 + (n[th[k]] + nc[th[k]]);
          break;
        }
      }
    }
    if (sum1 > 100.0) {
      fail = true;
      return 1.0;
    }
    return std::pow(2.0, (sum1 / 4.0)) / (kPi * square(axl));
  }

  double qf(const double* lb1, const double* nc1, const int* n1, int r1, double sigma, double c1, int lim1,
            double acc, double* trace, int* ifault) {
    static const int rats[] = {1, 2, 4, 8};
    double qfval = -1.0;
    int nt, ntm;
    double acc1, almx, xlim, xnt, xntm;
    double utx, tausq, sd, intv, intv1, x, up, un, d1, d2;
    r = r1;
    lim = lim1;
    c = c1;
    n = n1;
    lb = lb1;
    nc = nc1;
    for (int j = 0; j < 7; j++) trace[j] = 0.0;
    *ifault = 0;
    count = 0;
    intl = 0.0;
    ersm = 0.0;
    acc1 = acc;
    ndtsrt = true;
    fail = false;
    xlim = (double)lim;
    th.assign(r > 0 ? r : 1, 0);
    try {
      sigsq = square(sigma);
      sd = sigsq;
      lmax = 0.0;
      lmin = 0.0;
      mean = 0.0;
      for (int j = 0; j < r; j++) {
        const int nj = n[j];
        const double lj = lb[j], ncj = nc[j];
        if (nj < 0 || ncj < 0.0) {
          *ifault = 3;
          goto endofproc;
        }
        sd = sd + square(lj) * (2 * nj + 4.0 * ncj);
        mean = mean + lj * (nj + ncj);
        if (lmax < lj)
          lmax = lj;
        else if (lmin > lj)
          lmin = lj;
      }
      if (sd == 0.0) {
        qfval = (c > 0.0) ? 1.0 : 0.0;
        goto endofproc;
      }
      if (lmin == 0.0 && lmax == 0.0 && sigma == 0.0) {
        *ifault = 3;
        goto endofproc;
      }
      sd = std::sqrt(sd);
      almx = (lmax < -lmin) ? -lmin : lmax;

      utx = 16.0 / sd;
      up = 4.5 / sd;
      un = -up;
      findu(&utx, .5 * acc1);
      if (c != 0.0 && (almx > 0.07 * sd)) {
        tausq = .25 * acc1 / cfe(c);
        if (fail)
          fail = false;
        else if (truncation(utx, tausq) < .2 * acc1) {
          sigsq = sigsq + tausq;
          findu(&utx, .25 * acc1);
          trace[5] = std::sqrt(tausq);
        }
      }
      trace[4] = utx;
      acc1 = 0.5 * acc1;

      for (;;) {  // l1:
        d1 = ctff(acc1, &up) - c;
        if (d1 < 0.0) {
          qfval = 1.0;
          goto endofproc;
        }
        d2 = c - ctff(acc1, &un);
        if (d2 < 0.0) {
          qfval = 0.0;
          goto endofproc;
        }
        intv = 2.0 * kPi / ((d1 > d2) ? d1 : d2);
        xnt = utx / intv;
        xntm = 3.0 / std::sqrt(acc1);
        if (xnt > xntm * 1.5) {
          if (xntm > xlim) {
            *ifault = 1;
            goto endofproc;
          }
          ntm = (int)std::floor(xntm + 0.5);
          intv1 = utx / ntm;
          x = 2.0 * kPi / intv1;
          if (x <= std::fabs(c)) break;  // goto l2
          tausq = .33 * acc1 / (1.1 * (cfe(c - x) + cfe(c + x)));
          if (fail) break;  // goto l2
          acc1 = .67 * acc1;
          integrate(ntm, intv1, tausq, false);
          xlim = xlim - xntm;
          sigsq = sigsq + tausq;
          trace[2] = trace[2] + 1;
          trace[1] = trace[1] + ntm + 1;
          findu(&utx, .25 * acc1);
          acc1 = 0.75 * acc1;
          continue;  // goto l1
        }
        break;
      }
      // l2: main integration
      trace[3] = intv;
      if (xnt > xlim) {
        *ifault = 1;
        goto endofproc;
      }
      nt = (int)std::floor(xnt + 0.5);
      integrate(nt, intv, 0.0, true);
      trace[2] = trace[2] + 1;
      trace[1] = trace[1] + nt + 1;
      qfval = 0.5 - intl;
      trace[0] = ersm;
      up = ersm;
      x = up + acc / 10.0;
      for (int j = 0; j < 4; j++) {
        if (rats[j] * x == rats[j] * up) *ifault = 2;
      }
    } catch (const CountOverflow&) {
      *ifault = 4;
    }
  endofproc:
    trace[6] = (double)count;
    return qfval;
  }
};

}  // namespace

extern "C" {

// qf() with the reference's signature (qfc.c:297-299)
double orc_qf(const double* lb, const double* nc, const int* n, int r, double sigma, double c, int lim,
              double acc, double* trace, int* ifault) {
  Davies d;
  return d.qf(lb, nc, n, r, sigma, c, lim, acc, trace, ifault);
}

}  // extern "C"
