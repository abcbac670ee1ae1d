// ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the shipped product path.
//
// CPU restatement of the adaptive integrator the reference reaches through
//   Integration::integrateLU  regression/GSLIntegration.cpp:37-49  (limit 1000, :7-15)
//   -> gsl_integration_qags (GSL 1.16, vendored as third/gsl-1.16.tar.gz):
//      integration/qags.c (driver), qk.c + qk21.c (21-point Gauss-Kronrod), qelg.c (Wynn epsilon),
//      qpsrt.c / qpsrt2.c / util.c (interval bookkeeping), err.c (error rescaling).
// GSL's routine is itself a transcription of QUADPACK DQAGSE (Piessens et al. 1983); this file
// restates that published algorithm with the same bookkeeping so that the sequence of
// integrand abscissae — and therefore the SKAT-O p-value digits — is the same.
// Pinned against GSL 1.16 through tests/golden/gsl_scalar.json (qags cases) and cross-checked
// against scipy.integrate.quad (QUADPACK) in tests/test_oracle_scalar.py.
#include <cfloat>
#include <cmath>
#include <vector>
#include "orc_api.h"

namespace {

const double xgk[11] = {0.995657163025808080735527280689003, 0.973906528517171720077964012084452,
                        0.930157491355708226001207180059508, 0.865063366688984510732096688423493,
                        0.780817726586416897063717578345042, 0.679409568299024406234327365114874,
                        0.562757134668604683339000099272694, 0.433395394129247190799265943165784,
                        0.294392862701460198131126603103866, 0.148874338981631210884826001129720,
                        0.000000000000000000000000000000000};
const double wg[5] = {0.066671344308688137593568809893332, 0.149451349150580593145776339657697,
                      0.219086362515982043995534934228163, 0.269266719309996355091226921569469,
                      0.295524224714752870173892994651338};
const double wgk[11] = {0.011694638867371874278064396062192, 0.032558162307964727478818972459390,
                        0.054755896574351996031381300244580, 0.075039674810919952767043140916190,
                        0.093125454583697605535065465083366, 0.109387158802297641899210590325805,
                        0.123491976262065851077958109831074, 0.134709217311473325928054001771707,
                        0.142775938577060080797094273138717, 0.147739104901338491374841515972068,
                        0.149445554002916905664936468389821};

double rescale_error(double err, double result_abs, double result_asc) {
  err = std::fabs(err);
  if (result_asc != 0 && err != 0) {
    const double scale = std::pow((200 * err / result_asc), 1.5);
    err = (scale < 1) ? result_asc * scale : result_asc;
  }
  if (result_abs > DBL_MIN / (50 * DBL_EPSILON)) {
    const double min_err = 50 * DBL_EPSILON * result_abs;
    if (min_err > err) err = min_err;
  }
  return err;
}

void qk21(orc_integrand f, void* p, double a, double b, double* result, double* abserr, double* resabs,
          double* resasc) {
  const int n = 11;
  double fv1[11], fv2[11];
  const double center = 0.5 * (a + b), half_length = 0.5 * (b - a), abs_half_length = std::fabs(half_length);
  const double f_center = f(center, p);
  double result_gauss = 0, result_kronrod = f_center * wgk[n - 1];
  double result_abs = std::fabs(result_kronrod), result_asc = 0;
  for (int j = 0; j < (n - 1) / 2; j++) {
    const int jtw = j * 2 + 1;
    const double abscissa = half_length * xgk[jtw];
    const double fval1 = f(center - abscissa, p), fval2 = f(center + abscissa, p);
    const double fsum = fval1 + fval2;
    fv1[jtw] = fval1;
    fv2[jtw] = fval2;
    result_gauss += wg[j] * fsum;
    result_kronrod += wgk[jtw] * fsum;
    result_abs += wgk[jtw] * (std::fabs(fval1) + std::fabs(fval2));
  }
  for (int j = 0; j < n / 2; j++) {
    const int jtwm1 = j * 2;
    const double abscissa = half_length * xgk[jtwm1];
    const double fval1 = f(center - abscissa, p), fval2 = f(center + abscissa, p);
    fv1[jtwm1] = fval1;
    fv2[jtwm1] = fval2;
    result_kronrod += wgk[jtwm1] * (fval1 + fval2);
    result_abs += wgk[jtwm1] * (std::fabs(fval1) + std::fabs(fval2));
  }
  const double mean = result_kronrod * 0.5;
  result_asc = wgk[n - 1] * std::fabs(f_center - mean);
  for (int j = 0; j < n - 1; j++) result_asc += wgk[j] * (std::fabs(fv1[j] - mean) + std::fabs(fv2[j] - mean));
  const double err = (result_kronrod - result_gauss) * half_length;
  result_kronrod *= half_length;
  result_abs *= abs_half_length;
  result_asc *= abs_half_length;
  *result = result_kronrod;
  *resabs = result_abs;
  *resasc = result_asc;
  *abserr = rescale_error(err, result_abs, result_asc);
}

struct Workspace {
  size_t limit, size, nrmax, i, maximum_level;
  std::vector<double> alist, blist, rlist, elist;
  std::vector<size_t> order, level;
  explicit Workspace(size_t lim)
      : limit(lim), size(0), nrmax(0), i(0), maximum_level(0), alist(lim), blist(lim), rlist(lim), elist(lim),
        order(lim), level(lim) {}
  void initialise(double a, double b) {
    size = 0;
    nrmax = 0;
    i = 0;
    alist[0] = a;
    blist[0] = b;
    rlist[0] = 0.0;
    elist[0] = 0.0;
    order[0] = 0;
    level[0] = 0;
    maximum_level = 0;
  }
  void set_initial_result(double result, double error) {
    size = 1;
    rlist[0] = result;
    elist[0] = error;
  }
  void qpsrt() {
    const size_t last = size - 1;
    double errmax, errmin;
    int ii, k, top;
    size_t i_nrmax = nrmax;
    size_t i_maxerr = order[i_nrmax];
    if (last < 2) {
      order[0] = 0;
      order[1] = 1;
      i = i_maxerr;
      return;
    }
    errmax = elist[i_maxerr];
    while (i_nrmax > 0 && errmax > elist[order[i_nrmax - 1]]) {
      order[i_nrmax] = order[i_nrmax - 1];
      i_nrmax--;
    }
    if (last < (limit / 2 + 2))
      top = (int)last;
    else
      top = (int)(limit - last + 1);
    ii = (int)i_nrmax + 1;
    while (ii < top && errmax < elist[order[ii]]) {
      order[ii - 1] = order[ii];
      ii++;
    }
    order[ii - 1] = i_maxerr;
    errmin = elist[last];
    k = top - 1;
    while (k > ii - 2 && errmin >= elist[order[k]]) {
      order[k + 1] = order[k];
      k--;
    }
    order[k + 1] = last;
    i_maxerr = order[i_nrmax];
    i = i_maxerr;
    nrmax = i_nrmax;
  }
  void update(double a1, double b1, double area1, double error1, double a2, double b2, double area2,
              double error2) {
    const size_t i_max = i, i_new = size;
    const size_t new_level = level[i_max] + 1;
    if (error2 > error1) {
      alist[i_max] = a2;
      rlist[i_max] = area2;
      elist[i_max] = error2;
      level[i_max] = new_level;
      alist[i_new] = a1;
      blist[i_new] = b1;
      rlist[i_new] = area1;
      elist[i_new] = error1;
      level[i_new] = new_level;
    } else {
      blist[i_max] = b1;
      rlist[i_max] = area1;
      elist[i_max] = error1;
      level[i_max] = new_level;
      alist[i_new] = a2;
      blist[i_new] = b2;
      rlist[i_new] = area2;
      elist[i_new] = error2;
      level[i_new] = new_level;
    }
    size++;
    if (new_level > maximum_level) maximum_level = new_level;
    qpsrt();
  }
  double sum_results() const {
    double s = 0;
    for (size_t k = 0; k < size; k++) s += rlist[k];
    return s;
  }
  bool large_interval() const { return level[i] < maximum_level; }
  void reset_nrmax() {
    nrmax = 0;
    i = order[0];
  }
  bool increase_nrmax() {
    const int id = (int)nrmax;
    int jupbnd;
    const size_t last = size - 1;
    if (last > (1 + limit / 2))
      jupbnd = (int)(limit + 1 - last);
    else
      jupbnd = (int)last;
    for (int k = id; k <= jupbnd; k++) {
      const size_t i_max = order[nrmax];
      i = i_max;
      if (level[i_max] < maximum_level) return true;
      nrmax++;
    }
    return false;
  }
};

struct ExtrapolationTable {
  size_t n;
  double rlist2[52];
  size_t nres;
  double res3la[3];
  ExtrapolationTable() : n(0), nres(0) {}
  void append(double y) { rlist2[n++] = y; }
  void qelg(double* result, double* abserr) {
    double* epstab = rlist2;
    const size_t nn = n - 1;
    const double current = epstab[nn];
    double absolute = DBL_MAX;
    double relative = 5 * DBL_EPSILON * std::fabs(current);
    const size_t newelm = nn / 2, n_orig = nn;
    size_t n_final = nn;
    const size_t nres_orig = nres;
    *result = current;
    *abserr = DBL_MAX;
    if (nn < 2) {
      *result = current;
      *abserr = std::fmax(absolute, relative);
      return;
    }
    epstab[nn + 2] = epstab[nn];
    epstab[nn] = DBL_MAX;
    for (size_t i = 0; i < newelm; i++) {
      double res = epstab[nn - 2 * i + 2];
      const double e0 = epstab[nn - 2 * i - 2], e1 = epstab[nn - 2 * i - 1], e2 = res;
      const double e1abs = std::fabs(e1), delta2 = e2 - e1, err2 = std::fabs(delta2);
      const double tol2 = std::fmax(std::fabs(e2), e1abs) * DBL_EPSILON;
      const double delta3 = e1 - e0, err3 = std::fabs(delta3);
      const double tol3 = std::fmax(e1abs, std::fabs(e0)) * DBL_EPSILON;
      if (err2 <= tol2 && err3 <= tol3) {
        *result = res;
        absolute = err2 + err3;
        relative = 5 * DBL_EPSILON * std::fabs(res);
        *abserr = std::fmax(absolute, relative);
        return;
      }
      const double e3 = epstab[nn - 2 * i];
      epstab[nn - 2 * i] = e1;
      const double delta1 = e1 - e3, err1 = std::fabs(delta1);
      const double tol1 = std::fmax(e1abs, std::fabs(e3)) * DBL_EPSILON;
      if (err1 <= tol1 || err2 <= tol2 || err3 <= tol3) {
        n_final = 2 * i;
        break;
      }
      const double ss = (1 / delta1 + 1 / delta2) - 1 / delta3;
      if (std::fabs(ss * e1) <= 0.0001) {
        n_final = 2 * i;
        break;
      }
      res = e1 + 1 / ss;
      epstab[nn - 2 * i] = res;
      const double error = err2 + std::fabs(res - e2) + err3;
      if (error <= *abserr) {
        *abserr = error;
        *result = res;
      }
    }
    const size_t limexp = 50 - 1;
    if (n_final == limexp) n_final = 2 * (limexp / 2);
    if (n_orig % 2 == 1) {
      for (size_t i = 0; i <= newelm; i++) epstab[1 + i * 2] = epstab[i * 2 + 3];
    } else {
      for (size_t i = 0; i <= newelm; i++) epstab[i * 2] = epstab[i * 2 + 2];
    }
    if (n_orig != n_final) {
      for (size_t i = 0; i <= n_final; i++) epstab[i] = epstab[n_orig - n_final + i];
    }
    n = n_final + 1;
    if (nres_orig < 3) {
      res3la[nres_orig] = *result;
      *abserr = DBL_MAX;
    } else {
      *abserr = (std::fabs(*result - res3la[2]) + std::fabs(*result - res3la[1]) + std::fabs(*result - res3la[0]));
      res3la[0] = res3la[1];
      res3la[1] = res3la[2];
      res3la[2] = *result;
    }
    nres = nres_orig + 1;
    *abserr = std::fmax(*abserr, 5 * DBL_EPSILON * std::fabs(*result));
  }
};

bool subinterval_too_small(double a1, double a2, double b2) {
  const double e = DBL_EPSILON, u = DBL_MIN;
  const double tmp = (1 + 100 * e) * (std::fabs(a2) + 1000 * u);
  return std::fabs(a1) <= tmp && std::fabs(b2) <= tmp;
}

}  // namespace

extern "C" {

// gsl_integration_qags. Returns 0 on success, otherwise a GSL-style error code:
// 11 EMAXITER, 18 EROUND, 21 ESING, 22 EDIVERGE, 5 EFAILED, 13 EBADTOL.  neval_out counts integrand calls.
int orc_qags(orc_integrand f, void* params, double a, double b, double epsabs, double epsrel, int limit_i,
             double* result, double* abserr, int* neval_out) {
  const size_t limit = (size_t)limit_i;
  Workspace ws(limit);
  double area, errsum, res_ext, err_ext;
  double result0, abserr0, resabs0, resasc0;
  double tolerance;
  double ertest = 0, error_over_large_intervals = 0, reseps = 0, abseps = 0, correc = 0;
  size_t ktmin = 0;
  int roundoff_type1 = 0, roundoff_type2 = 0, roundoff_type3 = 0;
  int error_type = 0, error_type2 = 0;
  size_t iteration = 0;
  int positive_integrand = 0, extrapolate = 0, disallow_extrapolation = 0;
  ExtrapolationTable table;
  int neval = 0;
  struct Counted {
    orc_integrand f;
    void* p;
    int* n;
  } cf{f, params, &neval};
  auto fc = [](double x, void* q) -> double {
    Counted* c = (Counted*)q;
    ++*c->n;
    return c->f(x, c->p);
  };

  ws.initialise(a, b);
  *result = 0;
  *abserr = 0;
  if (neval_out) *neval_out = 0;
  if (epsabs <= 0 && (epsrel < 50 * DBL_EPSILON || epsrel < 0.5e-28)) return 13;

  qk21(fc, &cf, a, b, &result0, &abserr0, &resabs0, &resasc0);
  ws.set_initial_result(result0, abserr0);
  tolerance = std::fmax(epsabs, epsrel * std::fabs(result0));

  auto finish = [&](int code) {
    if (neval_out) *neval_out = neval;
    return code;
  };

  if (abserr0 <= 100 * DBL_EPSILON * resabs0 && abserr0 > tolerance) {
    *result = result0;
    *abserr = abserr0;
    return finish(18);
  } else if ((abserr0 <= tolerance && abserr0 != resasc0) || abserr0 == 0.0) {
    *result = result0;
    *abserr = abserr0;
    return finish(0);
  } else if (limit == 1) {
    *result = result0;
    *abserr = abserr0;
    return finish(11);
  }

  table.append(result0);
  area = result0;
  errsum = abserr0;
  res_ext = result0;
  err_ext = DBL_MAX;
  positive_integrand = (std::fabs(result0) >= (1 - 50 * DBL_EPSILON) * resabs0);
  iteration = 1;

  bool go_compute = false;
  do {
    size_t current_level;
    double a1, b1, a2, b2, a_i, b_i, r_i, e_i;
    double area1 = 0, area2 = 0, area12 = 0, error1 = 0, error2 = 0, error12 = 0;
    double resasc1, resasc2, resabs1, resabs2, last_e_i;

    a_i = ws.alist[ws.i];
    b_i = ws.blist[ws.i];
    r_i = ws.rlist[ws.i];
    e_i = ws.elist[ws.i];
    current_level = ws.level[ws.i] + 1;
    a1 = a_i;
    b1 = 0.5 * (a_i + b_i);
    a2 = b1;
    b2 = b_i;
    iteration++;
    qk21(fc, &cf, a1, b1, &area1, &error1, &resabs1, &resasc1);
    qk21(fc, &cf, a2, b2, &area2, &error2, &resabs2, &resasc2);
    area12 = area1 + area2;
    error12 = error1 + error2;
    last_e_i = e_i;
    errsum = errsum + error12 - e_i;
    area = area + area12 - r_i;
    tolerance = std::fmax(epsabs, epsrel * std::fabs(area));
    if (resasc1 != error1 && resasc2 != error2) {
      const double delta = r_i - area12;
      if (std::fabs(delta) <= 1.0e-5 * std::fabs(area12) && error12 >= 0.99 * e_i) {
        if (!extrapolate)
          roundoff_type1++;
        else
          roundoff_type2++;
      }
      if (iteration > 10 && error12 > e_i) roundoff_type3++;
    }
    if (roundoff_type1 + roundoff_type2 >= 10 || roundoff_type3 >= 20) error_type = 2;
    if (roundoff_type2 >= 5) error_type2 = 1;
    if (subinterval_too_small(a1, a2, b2)) error_type = 4;
    ws.update(a1, b1, area1, error1, a2, b2, area2, error2);
    if (errsum <= tolerance) {
      go_compute = true;
      break;
    }
    if (error_type) break;
    if (iteration >= limit - 1) {
      error_type = 1;
      break;
    }
    if (iteration == 2) {
      error_over_large_intervals = errsum;
      ertest = tolerance;
      table.append(area);
      continue;
    }
    if (disallow_extrapolation) continue;
    error_over_large_intervals += -last_e_i;
    if (current_level < ws.maximum_level) error_over_large_intervals += error12;
    if (!extrapolate) {
      if (ws.large_interval()) continue;
      extrapolate = 1;
      ws.nrmax = 1;
    }
    if (!error_type2 && error_over_large_intervals > ertest) {
      if (ws.increase_nrmax()) continue;
    }
    table.append(area);
    table.qelg(&reseps, &abseps);
    ktmin++;
    if (ktmin > 5 && err_ext < 0.001 * errsum) error_type = 5;
    if (abseps < err_ext) {
      ktmin = 0;
      err_ext = abseps;
      res_ext = reseps;
      correc = error_over_large_intervals;
      ertest = std::fmax(epsabs, epsrel * std::fabs(reseps));
      if (err_ext <= ertest) break;
    }
    if (table.n == 1) disallow_extrapolation = 1;
    if (error_type == 5) break;
    ws.reset_nrmax();
    extrapolate = 0;
    error_over_large_intervals = errsum;
  } while (iteration < limit);

  bool return_error = false;
  if (!go_compute) {
    *result = res_ext;
    *abserr = err_ext;
    if (err_ext == DBL_MAX) {
      go_compute = true;
    } else {
      if (error_type || error_type2) {
        if (error_type2) err_ext += correc;
        if (error_type == 0) error_type = 3;
        if (res_ext != 0.0 && area != 0.0) {
          if (err_ext / std::fabs(res_ext) > errsum / std::fabs(area)) go_compute = true;
        } else if (err_ext > errsum) {
          go_compute = true;
        } else if (area == 0.0) {
          return_error = true;
        }
      }
      if (!go_compute && !return_error) {
        const double max_area = std::fmax(std::fabs(res_ext), std::fabs(area));
        if (!positive_integrand && max_area < 0.01 * resabs0) {
          return_error = true;
        } else {
          const double ratio = res_ext / area;
          if (ratio < 0.01 || ratio > 100.0 || errsum > std::fabs(area)) error_type = 6;
          return_error = true;
        }
      }
    }
  }
  if (go_compute) {
    *result = ws.sum_results();
    *abserr = errsum;
  }
  if (error_type > 2) error_type--;
  switch (error_type) {
    case 0: return finish(0);
    case 1: return finish(11);
    case 2: return finish(18);
    case 3: return finish(21);
    case 4: return finish(18);
    case 5: return finish(22);
    default: return finish(5);
  }
}

}  // extern "C"
