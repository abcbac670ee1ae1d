// rvtests_amd — adaptive quadrature for the SKAT-O p-value, restructured for one-wave-per-gene.
//
// The reference integrates its SKAT-O integrand with gsl_integration_qags (QUADPACK QAGS: bisection
// on the worst interval, 21-point Gauss–Kronrod panels, Wynn epsilon extrapolation) through
// Integration::integrateLU (regression/GSLIntegration.cpp:37-49; limit 1000, :7-15) at
// regression/SkatO.cpp:236-256.  Each panel needs 21 integrand values and each integrand value is a
// full Davies evaluation, so on the GPU the 21 (first panel) or 42 (two halves of a bisected interval)
// abscissae are evaluated by 21 / 42 lanes of one wave AT ONCE, and only the cheap, inherently
// sequential interval bookkeeping below is done by a single lane.  To make that split possible the
// integrator is an explicit state machine:
//
//     QagsMachine m;  m.begin(a, b, epsabs, epsrel, limit, workspace);
//     <evaluate 21 abscissae of [a,b]>            m.first_panel(fv);
//     while (m.running()) { m.bisect(&a1,&b1,&b2); <evaluate 2 x 21 abscissae>  m.advance(fvL, fvR); }
//     m.result / m.abserr / m.status (GSL error numbers) / m.iterations
//
// The panel arithmetic and the bookkeeping follow QAGS step for step (same abscissae, same
// accumulation order, same interval ordering and extrapolation table), because the p-value digits
// depend on which abscissae are visited.
#pragma once
#include "rvt_special.h"

namespace rvt {

// abscissa t (0..20) of the 21-point Kronrod rule on [a,b]: t = 0 centre, 1..10 centre - h*x[t-1],
// 11..20 centre + h*x[t-11]
RVT_HD double gk21_node(int t) {  // |node| on [-1,1]
  static const double x[11] = {0.995657163025808080735527280689003, 0.973906528517171720077964012084452,
                        0.930157491355708226001207180059508, 0.865063366688984510732096688423493,
                        0.780817726586416897063717578345042, 0.679409568299024406234327365114874,
                        0.562757134668604683339000099272694, 0.433395394129247190799265943165784,
                        0.294392862701460198131126603103866, 0.148874338981631210884826001129720, 0.0};
  if (t == 0) return 0.0;
  return (t <= 10) ? x[t - 1] : x[t - 11];
}
RVT_HD double gk21_abscissa(double a, double b, int t) {
  const double center = 0.5 * (a + b), half = 0.5 * (b - a);
  if (t == 0) return center;
  const double off = half * gk21_node(t);  // "half_length * xgk[j]"
  return (t <= 10) ? center - off : center + off;
}

struct GkPanel {
  double result, abserr, resabs, resasc;
};

// Combine the 21 values fv[t] (layout of gk21_abscissa) exactly as the QUADPACK panel routine does.
RVT_HD GkPanel gk21_combine(const double* fv, double a, double b) {
  static const double wg[5] = {0.066671344308688137593568809893332, 0.149451349150580593145776339657697,
                        0.219086362515982043995534934228163, 0.269266719309996355091226921569469,
                        0.295524224714752870173892994651338};
  static const double wgk[11] = {0.011694638867371874278064396062192, 0.032558162307964727478818972459390,
                          0.054755896574351996031381300244580, 0.075039674810919952767043140916190,
                          0.093125454583697605535065465083366, 0.109387158802297641899210590325805,
                          0.123491976262065851077958109831074, 0.134709217311473325928054001771707,
                          0.142775938577060080797094273138717, 0.147739104901338491374841515972068,
                          0.149445554002916905664936468389821};
  const double half_length = 0.5 * (b - a), abs_half_length = fabs(half_length);
  const double f_center = fv[0];
  const double* fv1 = fv + 1;   // centre - abscissa_j
  const double* fv2 = fv + 11;  // centre + abscissa_j
  double result_gauss = 0, result_kronrod = f_center * wgk[10];
  double result_abs = fabs(result_kronrod);
  for (int j = 0; j < 5; j++) {
    const int jtw = j * 2 + 1;
    const double fsum = fv1[jtw] + fv2[jtw];
    result_gauss += wg[j] * fsum;
    result_kronrod += wgk[jtw] * fsum;
    result_abs += wgk[jtw] * (fabs(fv1[jtw]) + fabs(fv2[jtw]));
  }
  for (int j = 0; j < 5; j++) {
    const int jtwm1 = j * 2;
    result_kronrod += wgk[jtwm1] * (fv1[jtwm1] + fv2[jtwm1]);
    result_abs += wgk[jtwm1] * (fabs(fv1[jtwm1]) + fabs(fv2[jtwm1]));
  }
  const double mean = result_kronrod * 0.5;
  double result_asc = wgk[10] * fabs(f_center - mean);
  for (int j = 0; j < 10; j++) result_asc += wgk[j] * (fabs(fv1[j] - mean) + fabs(fv2[j] - mean));
  double err = (result_kronrod - result_gauss) * half_length;
  result_kronrod *= half_length;
  result_abs *= abs_half_length;
  result_asc *= abs_half_length;
  // error rescaling
  err = fabs(err);
  if (result_asc != 0 && err != 0) {
    const double scale = pow((200 * err / result_asc), 1.5);
    err = (scale < 1) ? result_asc * scale : result_asc;
  }
  if (result_abs > DBL_MIN / (50 * kDblEps)) {
    const double min_err = 50 * kDblEps * result_abs;
    if (min_err > err) err = min_err;
  }
  GkPanel p;
  p.result = result_kronrod;
  p.abserr = err;
  p.resabs = result_abs;
  p.resasc = result_asc;
  return p;
}

// interval store: 4 doubles + 2 ints per interval, `limit` intervals
struct QagsWorkspace {
  double* alist;
  double* blist;
  double* rlist;
  double* elist;
  int* order;
  int* level;
};
RVT_HD size_t qags_workspace_bytes(int limit) { return (size_t)limit * (4 * sizeof(double) + 2 * sizeof(int)); }
RVT_HD QagsWorkspace qags_workspace_carve(void* mem, int limit) {
  QagsWorkspace w;
  double* d = (double*)mem;
  w.alist = d;
  w.blist = d + limit;
  w.rlist = d + 2 * (size_t)limit;
  w.elist = d + 3 * (size_t)limit;
  w.order = (int*)(d + 4 * (size_t)limit);
  w.level = w.order + limit;
  return w;
}

struct QagsMachine {
  // configuration
  double epsabs, epsrel;
  int limit;
  QagsWorkspace w;
  // interval list state
  int size, nrmax, cur, maximum_level;
  // integration state
  double area, errsum, res_ext, err_ext, tolerance, resabs0;
  double ertest, error_over_large_intervals, correc;
  int ktmin, roundoff_type1, roundoff_type2, roundoff_type3, error_type, error_type2;
  int iteration, positive_integrand, extrapolate, disallow_extrapolation;
  // epsilon table
  int tab_n, tab_nres;
  double rlist2[52], res3la[3];
  // pending bisection
  double a1, b1, a2, b2, r_i, e_i;
  int current_level;
  // outputs
  bool active;
  double result, abserr;
  int status;  // 0 ok, 11 max iterations, 18 round-off, 21 singularity, 22 divergent, 5 failed, 13 bad tolerance
  int iterations;

  RVT_HD bool running() const { return active; }

  RVT_HD void begin(double a, double b, double epsabs_, double epsrel_, int limit_, QagsWorkspace ws) {
    epsabs = epsabs_;
    epsrel = epsrel_;
    limit = limit_;
    w = ws;
    size = 0;
    nrmax = 0;
    cur = 0;
    maximum_level = 0;
    w.alist[0] = a;
    w.blist[0] = b;
    w.rlist[0] = 0.0;
    w.elist[0] = 0.0;
    w.order[0] = 0;
    w.level[0] = 0;
    ertest = 0;
    error_over_large_intervals = 0;
    correc = 0;
    ktmin = 0;
    roundoff_type1 = roundoff_type2 = roundoff_type3 = 0;
    error_type = error_type2 = 0;
    iteration = 0;
    positive_integrand = extrapolate = disallow_extrapolation = 0;
    tab_n = tab_nres = 0;
    result = abserr = 0;
    status = 0;
    iterations = 0;
    active = true;
    if (epsabs <= 0 && (epsrel < 50 * kDblEps || epsrel < 0.5e-28)) {
      status = 13;
      active = false;
    }
  }

  RVT_HD void table_append(double y) { rlist2[tab_n++] = y; }

  RVT_HD void first_panel(const double* fv) {
    const GkPanel p = gk21_combine(fv, w.alist[0], w.blist[0]);
    size = 1;
    w.rlist[0] = p.result;
    w.elist[0] = p.abserr;
    resabs0 = p.resabs;
    tolerance = fmax(epsabs, epsrel * fabs(p.result));
    iterations = 1;
    if (p.abserr <= 100 * kDblEps * p.resabs && p.abserr > tolerance) {
      result = p.result;
      abserr = p.abserr;
      status = 18;
      active = false;
      return;
    }
    if ((p.abserr <= tolerance && p.abserr != p.resasc) || p.abserr == 0.0) {
      result = p.result;
      abserr = p.abserr;
      status = 0;
      active = false;
      return;
    }
    if (limit == 1) {
      result = p.result;
      abserr = p.abserr;
      status = 11;
      active = false;
      return;
    }
    table_append(p.result);
    area = p.result;
    errsum = p.abserr;
    res_ext = p.result;
    err_ext = DBL_MAX;
    positive_integrand = (fabs(p.result) >= (1 - 50 * kDblEps) * p.resabs);
    iteration = 1;
  }

  // the interval to split next; [a1,b1] and [b1,b2] are the halves
  RVT_HD void bisect(double* a1_out, double* b1_out, double* b2_out) {
    const double a_i = w.alist[cur], b_i = w.blist[cur];
    r_i = w.rlist[cur];
    e_i = w.elist[cur];
    current_level = w.level[cur] + 1;
    a1 = a_i;
    b1 = 0.5 * (a_i + b_i);
    a2 = b1;
    b2 = b_i;
    *a1_out = a1;
    *b1_out = b1;
    *b2_out = b2;
  }

  RVT_HD void sort_after_insert() {
    const int last = size - 1;
    int i_nrmax = nrmax;
    int i_maxerr = w.order[i_nrmax];
    if (last < 2) {
      w.order[0] = 0;
      w.order[1] = 1;
      cur = i_maxerr;
      return;
    }
    const double errmax = w.elist[i_maxerr];
    while (i_nrmax > 0 && errmax > w.elist[w.order[i_nrmax - 1]]) {
      w.order[i_nrmax] = w.order[i_nrmax - 1];
      i_nrmax--;
    }
    const int top = (last < (limit / 2 + 2)) ? last : limit - last + 1;
    int i = i_nrmax + 1;
    while (i < top && errmax < w.elist[w.order[i]]) {
      w.order[i - 1] = w.order[i];
      i++;
    }
    w.order[i - 1] = i_maxerr;
    const double errmin = w.elist[last];
    int k = top - 1;
    while (k > i - 2 && errmin >= w.elist[w.order[k]]) {
      w.order[k + 1] = w.order[k];
      k--;
    }
    w.order[k + 1] = last;
    cur = w.order[i_nrmax];
    nrmax = i_nrmax;
  }

  RVT_HD void store_halves(double area1, double error1, double area2, double error2) {
    const int i_max = cur, i_new = size;
    const int new_level = w.level[i_max] + 1;
    if (error2 > error1) {
      w.alist[i_max] = a2;
      w.rlist[i_max] = area2;
      w.elist[i_max] = error2;
      w.level[i_max] = new_level;
      w.alist[i_new] = a1;
      w.blist[i_new] = b1;
      w.rlist[i_new] = area1;
      w.elist[i_new] = error1;
      w.level[i_new] = new_level;
    } else {
      w.blist[i_max] = b1;
      w.rlist[i_max] = area1;
      w.elist[i_max] = error1;
      w.level[i_max] = new_level;
      w.alist[i_new] = a2;
      w.blist[i_new] = b2;
      w.rlist[i_new] = area2;
      w.elist[i_new] = error2;
      w.level[i_new] = new_level;
    }
    size++;
    if (new_level > maximum_level) maximum_level = new_level;
    sort_after_insert();
  }

  RVT_HD bool raise_nrmax() {
    const int last = size - 1;
    const int jupbnd = (last > (1 + limit / 2)) ? limit + 1 - last : last;
    for (int k = nrmax; k <= jupbnd; k++) {
      const int i_max = w.order[nrmax];
      cur = i_max;
      if (w.level[i_max] < maximum_level) return true;
      nrmax++;
    }
    return false;
  }

  // Wynn epsilon algorithm on the table of partial areas
  RVT_HD void extrapolate_table(double* res_out, double* abserr_out) {
    double* epstab = rlist2;
    const int n = tab_n - 1;
    const double current = epstab[n];
    double absolute = DBL_MAX;
    double relative = 5 * kDblEps * fabs(current);
    const int newelm = n / 2, n_orig = n;
    int n_final = n;
    const int nres_orig = tab_nres;
    *res_out = current;
    *abserr_out = DBL_MAX;
    if (n < 2) {
      *abserr_out = fmax(absolute, relative);
      return;
    }
    epstab[n + 2] = epstab[n];
    epstab[n] = DBL_MAX;
    for (int i = 0; i < newelm; i++) {
      double res = epstab[n - 2 * i + 2];
      const double e0 = epstab[n - 2 * i - 2], e1 = epstab[n - 2 * i - 1], e2 = res;
      const double e1abs = fabs(e1), delta2 = e2 - e1, err2 = fabs(delta2);
      const double tol2 = fmax(fabs(e2), e1abs) * kDblEps;
      const double delta3 = e1 - e0, err3 = fabs(delta3);
      const double tol3 = fmax(e1abs, fabs(e0)) * kDblEps;
      if (err2 <= tol2 && err3 <= tol3) {
        *res_out = res;
        absolute = err2 + err3;
        relative = 5 * kDblEps * fabs(res);
        *abserr_out = fmax(absolute, relative);
        return;
      }
      const double e3 = epstab[n - 2 * i];
      epstab[n - 2 * i] = e1;
      const double delta1 = e1 - e3, err1 = fabs(delta1);
      const double tol1 = fmax(e1abs, fabs(e3)) * kDblEps;
      if (err1 <= tol1 || err2 <= tol2 || err3 <= tol3) {
        n_final = 2 * i;
        break;
      }
      const double ss = (1 / delta1 + 1 / delta2) - 1 / delta3;
      if (fabs(ss * e1) <= 0.0001) {
        n_final = 2 * i;
        break;
      }
      res = e1 + 1 / ss;
      epstab[n - 2 * i] = res;
      const double error = err2 + fabs(res - e2) + err3;
      if (error <= *abserr_out) {
        *abserr_out = error;
        *res_out = res;
      }
    }
    const int limexp = 50 - 1;
    if (n_final == limexp) n_final = 2 * (limexp / 2);
    if (n_orig % 2 == 1) {
      for (int i = 0; i <= newelm; i++) epstab[1 + i * 2] = epstab[i * 2 + 3];
    } else {
      for (int i = 0; i <= newelm; i++) epstab[i * 2] = epstab[i * 2 + 2];
    }
    if (n_orig != n_final) {
      for (int i = 0; i <= n_final; i++) epstab[i] = epstab[n_orig - n_final + i];
    }
    tab_n = n_final + 1;
    if (nres_orig < 3) {
      res3la[nres_orig] = *res_out;
      *abserr_out = DBL_MAX;
    } else {
      *abserr_out = (fabs(*res_out - res3la[2]) + fabs(*res_out - res3la[1]) + fabs(*res_out - res3la[0]));
      res3la[0] = res3la[1];
      res3la[1] = res3la[2];
      res3la[2] = *res_out;
    }
    tab_nres = nres_orig + 1;
    *abserr_out = fmax(*abserr_out, 5 * kDblEps * fabs(*res_out));
  }

  RVT_HD double sum_results() const {
    double s = 0;
    for (int k = 0; k < size; k++) s += w.rlist[k];
    return s;
  }

  RVT_HD void finish(bool from_sum) {
    bool go_sum = from_sum;
    bool return_error = false;
    if (!go_sum) {
      result = res_ext;
      abserr = err_ext;
      if (err_ext == DBL_MAX) {
        go_sum = true;
      } else {
        if (error_type || error_type2) {
          if (error_type2) err_ext += correc;
          if (error_type == 0) error_type = 3;
          if (res_ext != 0.0 && area != 0.0) {
            if (err_ext / fabs(res_ext) > errsum / fabs(area)) go_sum = true;
          } else if (err_ext > errsum) {
            go_sum = true;
          } else if (area == 0.0) {
            return_error = true;
          }
        }
        if (!go_sum && !return_error) {
          const double max_area = fmax(fabs(res_ext), fabs(area));
          if (!(!positive_integrand && max_area < 0.01 * resabs0)) {
            const double ratio = res_ext / area;
            if (ratio < 0.01 || ratio > 100.0 || errsum > fabs(area)) error_type = 6;
          }
        }
      }
    }
    if (go_sum) {
      result = sum_results();
      abserr = errsum;
    }
    if (error_type > 2) error_type--;
    switch (error_type) {
      case 0: status = 0; break;
      case 1: status = 11; break;
      case 2: status = 18; break;
      case 3: status = 21; break;
      case 4: status = 18; break;
      case 5: status = 22; break;
      default: status = 5; break;
    }
    active = false;
  }

  // absorb the two half-interval panels (21 values each, layout of gk21_abscissa)
  RVT_HD void advance(const double* fvL, const double* fvR) {
    const GkPanel p1 = gk21_combine(fvL, a1, b1);
    const GkPanel p2 = gk21_combine(fvR, a2, b2);
    iteration++;
    iterations = iteration;
    const double area1 = p1.result, area2 = p2.result, error1 = p1.abserr, error2 = p2.abserr;
    const double area12 = area1 + area2, error12 = error1 + error2;
    const double last_e_i = e_i;
    errsum = errsum + error12 - e_i;
    area = area + area12 - r_i;
    tolerance = fmax(epsabs, epsrel * fabs(area));
    if (p1.resasc != error1 && p2.resasc != error2) {
      const double delta = r_i - area12;
      if (fabs(delta) <= 1.0e-5 * fabs(area12) && error12 >= 0.99 * e_i) {
        if (!extrapolate)
          roundoff_type1++;
        else
          roundoff_type2++;
      }
      if (iteration > 10 && error12 > e_i) roundoff_type3++;
    }
    if (roundoff_type1 + roundoff_type2 >= 10 || roundoff_type3 >= 20) error_type = 2;
    if (roundoff_type2 >= 5) error_type2 = 1;
    {
      const double tmp = (1 + 100 * kDblEps) * (fabs(a2) + 1000 * DBL_MIN);
      if (fabs(a1) <= tmp && fabs(b2) <= tmp) error_type = 4;
    }
    store_halves(area1, error1, area2, error2);
    if (errsum <= tolerance) {
      finish(true);
      return;
    }
    if (error_type) {
      finish(false);
      return;
    }
    if (iteration >= limit - 1) {
      error_type = 1;
      finish(false);
      return;
    }
    if (iteration == 2) {
      error_over_large_intervals = errsum;
      ertest = tolerance;
      table_append(area);
      return;
    }
    if (disallow_extrapolation) {
      if (!(iteration < limit)) finish(false);
      return;
    }
    error_over_large_intervals += -last_e_i;
    if (current_level < maximum_level) error_over_large_intervals += error12;
    bool do_extrap = true;
    if (!extrapolate) {
      if (w.level[cur] < maximum_level) {
        do_extrap = false;
      } else {
        extrapolate = 1;
        nrmax = 1;
      }
    }
    if (do_extrap && !error_type2 && error_over_large_intervals > ertest) {
      if (raise_nrmax()) do_extrap = false;
    }
    if (do_extrap) {
      double reseps, abseps;
      table_append(area);
      extrapolate_table(&reseps, &abseps);
      ktmin++;
      if (ktmin > 5 && err_ext < 0.001 * errsum) error_type = 5;
      if (abseps < err_ext) {
        ktmin = 0;
        err_ext = abseps;
        res_ext = reseps;
        correc = error_over_large_intervals;
        ertest = fmax(epsabs, epsrel * fabs(reseps));
        if (err_ext <= ertest) {
          finish(false);
          return;
        }
      }
      if (tab_n == 1) disallow_extrapolation = 1;
      if (error_type == 5) {
        finish(false);
        return;
      }
      nrmax = 0;
      cur = w.order[0];
      extrapolate = 0;
      error_over_large_intervals = errsum;
    }
    if (!(iteration < limit)) finish(false);
  }
};

}  // namespace rvt
