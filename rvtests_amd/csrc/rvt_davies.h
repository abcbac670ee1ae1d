// rvtests_amd — mixture-of-chi-square tail probabilities, one evaluation per GPU lane.
//
// Device implementation (RVT_HD: also compiled on the host ONLY for the test harness) of what the
// reference computes in
//   MixtureChiSquare::getPvalue / getLiuPvalue   regression/MixtureChiSquare.cpp:7-29, 44-83
//   qf (Davies 1980, AS 155)                     regression/qfc.c:297-436 and helpers :82-304
//   cdfchn(which=1) -> cumchn                    regression/cdflib.cpp:2634-2800, 5172-5350
// as called on the hot path: sigma = 0, every term 1 df and non-centrality 0, lim = 10000,
// acc = 1e-6 (regression/MixtureChiSquare.h:7,31-33).
//
// GPU shape: the reference keeps qf's state in file-scope statics and leaves through longjmp; here
// the state is a small struct in registers, the coefficient array lives in LDS (read as a
// broadcast by every lane of the wave), and "count > lim" is a sticky flag tested by each search
// loop.  Each lane of a wave evaluates a DIFFERENT point c against the SAME coefficients, which
// is exactly the access pattern of SKAT-O's quadrature (21 or 42 abscissae per QAGS step).
// The order of floating-point operations inside every sum is the reference's, so results differ
// from it only through libm (ocml vs glibc) rounding.
#pragma once
#include "rvt_special.h"

// The search routines below are force-inlined into their callers and specialised on the form of the coefficient sums
// (template parameter FAST), so that DaviesState lives in registers and the p-value kernel that uses the product
// form carries none of the term-by-term code.
#if defined(__HIPCC__)
#define RVT_HDI __host__ __device__ __forceinline__
#else
#define RVT_HDI inline __attribute__((always_inline))
#endif

// On the GPU every array the searches and the integrand terms read in their inner loops — the coefficients, their order,
// the memo, the prelude — lives in LDS, and the pointer types SAY so (address space 3): through generic pointers the
// compiler emitted flat loads, one fully exposed round trip per coefficient (measured: the loops ran at the flat-load latency,
// ~150 cycles per coefficient).  On the host the qualifier is empty.
#if defined(__HIP_DEVICE_COMPILE__)
#define RVT_LDSQ __attribute__((address_space(3)))
#else
#define RVT_LDSQ
#endif

namespace rvt {

typedef const RVT_LDSQ double* dv_coefs;  // coefficient array
typedef const RVT_LDSQ int* dv_index;     // its order (davies_order)

// ---- memo of the coefficient sums of errbd() / truncation() ---------------------------------------------------------
// Both routines are  f(u, sigsq, tausq) = g(u * u * (sigsq + tausq), S(u))  where S(u) — the O(r) part: sums and products over
// the coefficients — depends on the evaluation point u ALONE.  And the points the searches of qf() ask for come from a tiny
// lattice that does not depend on the quantile c either: ctff() doubles and bisects from up = 4.5 / sd (qfc.c:157-178), findu()
// walks utx * 4^k and then divides by 2, 1.4, 1.2, 1.1 (qfc.c:217-238); c and the convergence factor only decide which of
// those points are visited.  The ~15 000 search evaluations of one SKAT-O quadrature (1 000 abscissae x 12) therefore ask for
// about 35 distinct points (measured: tools/davies_divergence.py), and every abscissa performs the same floating-point
// operations on the way to a point, so the keys agree bit for bit.  The memo is a small open-addressing table keyed by the
// bits of u; a hit returns exactly the doubles a miss computes, so results do not depend on what the table holds.
// On the GPU the table sits in LDS and is shared by the lanes of a wave (one lane per abscissa): a slot is claimed with a
// compare-and-swap, filled, and published by storing its key last.
constexpr int kMemoSlotsE = 32, kMemoSlotsT = 32;    // errbd / truncation points; a full table just stops memoising
constexpr unsigned long long kMemoEmpty = ~0ull;     // (NaN bit patterns: never the bits of a finite u)
constexpr unsigned long long kMemoBusy = ~0ull - 1;
struct DaviesMemo {
  unsigned long long ekey[kMemoSlotsE];  // errbd: bits of 2u
  unsigned long long tkey[kMemoSlotsT];  // truncation: bits of 2u
  double eval[kMemoSlotsE][2];           // sum_j [x^2/y + log y + x],  sum_j lb_j / y        (y = 1 - x, x = 2 u lb_j)
  double tval[kMemoSlotsT][4];           // log prod_{x<=1}(1+x), log prod_{x>1} x, log prod_{x>1}(1+x), #{x>1}   (x = (2 u lb_j)^2)
};
typedef RVT_LDSQ DaviesMemo* dv_memo_p;
RVT_HDI void dv_memo_clear(dv_memo_p m, int lane = 0, int nlanes = 1) {
  for (int i = lane; i < kMemoSlotsE; i += nlanes) m->ekey[i] = kMemoEmpty;
  for (int i = lane; i < kMemoSlotsT; i += nlanes) m->tkey[i] = kMemoEmpty;
}
// 1 = found (slot), 0 = absent (slot = where it may be inserted, or -1 when the table is full)
template <int SLOTS>
RVT_HDI int dv_memo_find(const RVT_LDSQ unsigned long long* keys, unsigned long long k, int* slot) {
  const volatile RVT_LDSQ unsigned long long* vk = keys;
  // the points differ in their exponent and leading mantissa bits (dyadic steps and a few fixed divisors)
  int s = (int)(((((unsigned)(k >> 32)) ^ (unsigned)(k >> 13)) * 2654435761u) >> 16) & (SLOTS - 1);
  int free_slot = -1;
  for (int probe = 0; probe < SLOTS; ++probe) {
    const unsigned long long cur = vk[s];
    if (cur == k) {
      *slot = s;
      return 1;
    }
    if (cur == kMemoEmpty) {
      free_slot = s;
      break;
    }
    s = (s + 1) & (SLOTS - 1);
  }
  *slot = free_slot;
  return 0;
}
// claim the slot (it may have been taken since it was seen empty: then the values are simply not stored)
RVT_HDI bool dv_memo_claim(RVT_LDSQ unsigned long long* keys, int slot) {
#if defined(__HIP_DEVICE_COMPILE__)
  return atomicCAS((unsigned long long*)&keys[slot], kMemoEmpty, kMemoBusy) == kMemoEmpty;  // (a known LDS address: ds_cmpst)
#else
  if (keys[slot] != kMemoEmpty) return false;
  keys[slot] = kMemoBusy;
  return true;
#endif
}
RVT_HDI void dv_memo_publish(RVT_LDSQ unsigned long long* keys, int slot, unsigned long long k) {
#if defined(__HIP_DEVICE_COMPILE__)
  __threadfence_block();
  *(volatile RVT_LDSQ unsigned long long*)&keys[slot] = k;
#else
  keys[slot] = k;
#endif
}
RVT_HDI unsigned long long dv_bits(double u) {
  unsigned long long b;
#if defined(__HIP_DEVICE_COMPILE__)
  b = (unsigned long long)__double_as_longlong(u);
#else
  __builtin_memcpy(&b, &u, 8);
#endif
  return b;
}

struct DaviesState {
  dv_memo_p memo;    // optional (product form only)
  dv_coefs lb;       // coefficients (all > 0 on the hot path), length r
  dv_index th;       // indices of lb ordered by decreasing |lb|
  dv_coefs ls;       // optional: the coefficients IN that order, ls[j] = lb[th[j]] (then th is not read)
  int r;
  int lim;
  int count;
  bool over;   // count exceeded lim  (reference: longjmp -> fault 4)
  bool fail;   // cfe() could not bound the error
  double sigsq, lmax, lmin, mean, c;
  double intl, ersm;
};

constexpr double kDaviesPi = 3.14159265358979;  // value used by the reference (qfc.c:22)
constexpr double kDaviesLog28 = .0866;          // qfc.c:23

RVT_HD double dv_exp1(double x) { return x < -50.0 ? 0.0 : exp(x); }

// ---- product form of the coefficient sums -------------------------------------------------------------------------
// Every O(r) loop of qf() is a sum of logarithms and arc tangents over the coefficients:
//   integrate():  sum_j atan(x_j),  sum_j log(1 + x_j^2)            x_j = 2 lb_j u           (qfc.c:250-262)
//   errbd():      sum_j [log(1 - x_j) + x_j],  sum_j lb_j / (1 - x_j), sum_j x_j^2 / (1 - x_j)   (qfc.c:143-152)
//   truncation(): sum_j log(1 + x_j), sum_j log(x_j)                 x_j = (u lb_j)^2         (qfc.c:192-205)
// i.e. the argument and the log-modulus of prod_j (1 + i x_j), and the logs of three real products.  The
// products are formed by two FMAs (complex) or one multiplication (real) per coefficient, with the binary exponent
// carried separately, and ONE atan2 / log is taken per sum instead of one atan + one log (83 + 98 fp64 instructions in
// the device library) per coefficient: the p-value stage costs ~8x fewer instructions.  The products carry a relative
// error of ~r ulp, the same size as the accumulated rounding of the term-by-term sums, so results move by ~1e-15
// absolute — the level at which ocml's and glibc's atan / log already differ.  The term-by-term form stays
// available (template parameter FAST = false; RVT_TEST_EXACT_DAVIES at the ABI).
constexpr double kLn2 = 0.693147180559945309417232121458;
constexpr double kTwoPiHi = 6.283185307179586232;        // 2 pi rounded to double
constexpr double kTwoPiLo = 2.4492935982947064e-16;      // 2 pi - kTwoPiHi

struct ScaledProd {  // value = m * 2^e
  double m;
  int e;
};
RVT_HD void sp_renorm(ScaledProd& p) {
  int k;
  p.m = frexp(p.m, &k);
  p.e += k;
}
RVT_HD double sp_log(ScaledProd p) {  // log(m 2^e), m > 0
  sp_renorm(p);
  return log(p.m) + (double)p.e * kLn2;
}

// sum_j atan(x_j), sum_j |atan(x_j)| and sum_j log(1 + x_j^2) for x_j = lb_j * u2 (u2 > 0; lb_j of either sign).
// The factors with x >= 0 and those with x < 0 go into two products: the argument of the first only grows, that of
// the second only falls (every factor turns by less than pi / 2), so the crossings of the negative real axis can be
// counted and the principal value of atan2 unwrapped.  (The sign of a coefficient is the same for every lane of a
// wave, so the branch does not diverge.)
// allpos: the caller knows that every coefficient is > 0 (the hot path: davies_all_positive) — the second product then
// stays 1 and its renormalisations, its atan2 and its log are skipped (they contribute exact zeros).
RVT_HDI void dv_arg_logmod(dv_coefs lb, int r, double u2, double* sum_atan, double* sum_abs_atan,
                           double* sum_log1p_sq, bool allpos = false) {
  double ap = 1.0, bp = 0.0, an = 1.0, bn = 0.0;  // prod (1 + i x_j) / 2^E over x >= 0 / over x < 0
  int Ep = 0, En = 0, wp = 0, wn = 0;
  if (allpos) {
    // One complex multiplication per coefficient, j = r-1 .. 0, renormalised after every j that is a multiple of 4 — written
    // as groups of four whose coefficients are fetched one group AHEAD of the multiplications (the chain of dependent FMAs
    // is the critical path; the LDS reads must not sit on it).
    auto step = [&](double lj) {
      const double x = lj * u2;
      const double na = fma(-bp, x, ap), nb = fma(ap, x, bp);
      wp += (bp >= 0.0 && nb < 0.0) ? 1 : 0;
      ap = na;
      bp = nb;
    };
    auto renorm = [&]() {
      int k;
      (void)frexp(fmax(fabs(ap), fabs(bp)), &k);
      ap = ldexp(ap, -k);
      bp = ldexp(bp, -k);
      Ep += k;
    };
    int j = r - 1;
    for (; j >= 0 && (j & 3) != 3; --j) {  // the incomplete top group
      step(lb[j]);
      if ((j & 3) == 0) renorm();
    }
    if (j >= 3) {
      double n3 = lb[j], n2 = lb[j - 1], n1 = lb[j - 2], n0 = lb[j - 3];
      for (; j >= 3; j -= 4) {
        const double c3 = n3, c2 = n2, c1 = n1, c0 = n0;
        if (j >= 7) {
          n3 = lb[j - 4];
          n2 = lb[j - 5];
          n1 = lb[j - 6];
          n0 = lb[j - 7];
        }
        step(c3);
        step(c2);
        step(c1);
        step(c0);
        renorm();
      }
    }
    const double fp = (double)wp;
    const double tp = (atan2(bp, ap) + fp * kTwoPiHi) + fp * kTwoPiLo;   // >= 0
    *sum_atan = tp;
    *sum_abs_atan = tp;
    *sum_log1p_sq = log(fma(ap, ap, bp * bp)) + (double)(2 * Ep) * kLn2;
    return;
  }
  for (int j = r - 1; j >= 0; --j) {
    const double x = lb[j] * u2;
    if (x >= 0.0) {
      const double na = fma(-bp, x, ap), nb = fma(ap, x, bp);
      wp += (bp >= 0.0 && nb < 0.0) ? 1 : 0;
      ap = na;
      bp = nb;
    } else {
      const double na = fma(-bn, x, an), nb = fma(an, x, bn);
      wn += (bn <= 0.0 && nb > 0.0) ? 1 : 0;
      an = na;
      bn = nb;
    }
    if ((j & 3) == 0) {
      int k;
      (void)frexp(fmax(fabs(ap), fabs(bp)), &k);
      ap = ldexp(ap, -k);
      bp = ldexp(bp, -k);
      Ep += k;
      (void)frexp(fmax(fabs(an), fabs(bn)), &k);
      an = ldexp(an, -k);
      bn = ldexp(bn, -k);
      En += k;
    }
  }
  const double fp = (double)wp, fn = (double)wn;
  const double tp = (atan2(bp, ap) + fp * kTwoPiHi) + fp * kTwoPiLo;   // >= 0
  const double tn = (atan2(bn, an) - fn * kTwoPiHi) - fn * kTwoPiLo;   // <= 0
  *sum_atan = tp + tn;
  *sum_abs_atan = tp - tn;
  *sum_log1p_sq = (log(fma(ap, ap, bp * bp)) + log(fma(an, an, bn * bn))) + (double)(2 * (Ep + En)) * kLn2;
}

// log(1+x) if first, else log(1+x) - x      (qfc.c:95-113)
RVT_HD double dv_log1(double x, bool first) {
  if (fabs(x) > 0.1) return first ? log(1.0 + x) : (log(1.0 + x) - x);
  double y = x / (2.0 + x);
  double term = 2.0 * (y * y * y);
  double k = 3.0;
  double s = (first ? 2.0 : -x) * y;
  y = y * y;
  double s1 = s + term / k;
  for (int guard = 0; s1 != s && guard < 64; ++guard) {  // converges in < 10 steps for |x| <= 0.1
    k = k + 2.0;
    term = term * y;
    s = s1;
    s1 = s + term / k;
  }
  return s;
}

// Four independent evaluations of dv_log1 at once.  Values are exactly dv_log1's; the point is instruction-level
// parallelism: one wave evaluates these sums essentially alone on its SIMD, so a chain of dependent fp64 ops runs
// at their latency.  The common case (|x| > 0.1: a plain log) is issued for all four elements back to back; the
// series branch of qfc.c:103-112 is taken only for the elements that need it.
RVT_HD void dv_log1_x4(const double (&x)[4], int cnt, bool first, double (&out)[4]) {
  bool any_small = false;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    out[k] = first ? log(1.0 + x[k]) : (log(1.0 + x[k]) - x[k]);
    if (k < cnt && !(fabs(x[k]) > 0.1)) any_small = true;
  }
  if (any_small) {
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (k < cnt && !(fabs(x[k]) > 0.1)) out[k] = dv_log1(x[k], first);
  }
}

RVT_HDI void dv_tick(DaviesState& st) {
  st.count = st.count + 1;
  if (st.count > st.lim) st.over = true;
}

// bound on the tail probability from the mgf; cutoff returned in *cx      (qfc.c:137-155)
#if defined(RVT_DV_PROFILE) && !defined(__HIP_DEVICE_COMPILE__)
#define RVT_DV_HIT(k) (++rvt_dv_profile[k])   // host-side call counters of a profiling build (tools/davies_calls.cpp)
extern long long rvt_dv_profile[8];
void rvt_dv_eval_mark();
void rvt_dv_key(int kind, double u);          // every (kind, u) an errbd / truncation evaluation is asked for
#define RVT_DV_KEY(kind, u) rvt_dv_key(kind, u)
#else
#define RVT_DV_HIT(k) ((void)0)
#define RVT_DV_KEY(kind, u) ((void)0)
#endif
// 1 / y for y in (0, 1] (errbd's denominators 1 - 2 u lb_j): on the device the hardware reciprocal with two Newton steps
// (relative error ~1e-16; 5 instructions where the IEEE division sequence takes 12 — errbd is three quarters of the
// instructions qf()'s searches issue, and they are half of the p-value stage); the host divides
RVT_HDI double dv_recip(double y) {
#if defined(__HIP_DEVICE_COMPILE__)
  double r = __builtin_amdgcn_rcp(y);
  r = fma(fma(-y, r, 1.0), r, r);
  r = fma(fma(-y, r, 1.0), r, r);
  return r;
#else
  return 1.0 / y;
#endif
}
template <bool FAST>
RVT_HDI double dv_errbd(DaviesState& st, double u, double* cx) {
  RVT_DV_HIT(0);
  dv_tick(st);
  if (st.over) {
    *cx = 0.0;
    return 0.0;
  }
  double xconst = u * st.sigsq, sum1 = u * xconst;
  u = 2.0 * u;
  if (FAST) {
    // sum_j [x^2 / y + log(y) + x] with y = 1 - x in (0, 1]: one reciprocal per coefficient, log of the product.
    // Both sums start from zero (not from the sigsq terms), so that they are functions of u alone: see DaviesMemo.
    double B, X;
    const unsigned long long key = dv_bits(u);
    int slot = -1;
    const bool use_memo = st.memo != nullptr && key < kMemoBusy;
    if (use_memo && dv_memo_find<kMemoSlotsE>(st.memo->ekey, key, &slot)) {
      const volatile RVT_LDSQ double* v = st.memo->eval[slot];
      B = v[0];
      X = v[1];
    } else {
      RVT_DV_KEY(0, u);
      ScaledProd py{1.0, 0};
      double sx = 0.0, sq = 0.0;
      X = 0.0;
#pragma unroll 4
      for (int j = st.r - 1; j >= 0; --j) {
        const double lj = st.lb[j], x = u * lj, y = 1.0 - x, ry = dv_recip(y);
        X = fma(lj, ry, X);
        sq = fma(x * x, ry, sq);
        sx += x;
        py.m *= y;
        if ((j & 7) == 0) sp_renorm(py);
      }
      B = sq + (sp_log(py) + sx);
      if (use_memo && slot >= 0 && dv_memo_claim(st.memo->ekey, slot)) {
        st.memo->eval[slot][0] = B;
        st.memo->eval[slot][1] = X;
        dv_memo_publish(st.memo->ekey, slot, key);
      }
    }
    sum1 = sum1 + B;
    *cx = xconst + X;
    return dv_exp1(-0.5 * sum1);
  }
  for (int j0 = st.r - 1; j0 >= 0; j0 -= 4) {  // elements evaluated 4 at a time, accumulated in the reference's order
    const int cnt = (j0 >= 3) ? 4 : j0 + 1;
    double lj[4], x[4], y[4], mx[4], lg[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      lj[k] = (k < cnt) ? st.lb[j0 - k] : 1.0;
      x[k] = u * lj[k];
      y[k] = 1.0 - x[k];
      mx[k] = (k < cnt) ? -x[k] : 1.0;
    }
    dv_log1_x4(mx, cnt, false, lg);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (k < cnt) {
        xconst = xconst + lj[k] / y[k];
        sum1 = sum1 + ((x[k] * x[k]) / y[k] + lg[k]);
      }
    }
  }
  *cx = xconst;
  return dv_exp1(-0.5 * sum1);
}

// find ctff so that p(qf > ctff) < accx (upn > 0) or p(qf < ctff) < accx      (qfc.c:157-178)
template <bool FAST>
RVT_HDI double dv_ctff(DaviesState& st, double accx, double* upn) {
  double u1, u2, u, rb, xconst = 0.0, c1, c2 = 0.0;
  u2 = *upn;
  u1 = 0.0;
  c1 = st.mean;
  rb = 2.0 * ((u2 > 0.0) ? st.lmax : st.lmin);
  for (u = u2 / (1.0 + u2 * rb); dv_errbd<FAST>(st, u, &c2) > accx && !st.over; u = u2 / (1.0 + u2 * rb)) {
    RVT_DV_HIT(2);  // a doubling
    u1 = u2;
    c1 = c2;
    u2 = 2.0 * u2;
  }
  for (u = (c1 - st.mean) / (c2 - st.mean); u < 0.9 && !st.over; u = (c1 - st.mean) / (c2 - st.mean)) {
    u = (u1 + u2) / 2.0;
    RVT_DV_HIT(3);  // a bisection
    if (dv_errbd<FAST>(st, u / (1.0 + u * rb), &xconst) > accx) {
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

// bound on the integration error due to truncation at u      (qfc.c:180-215)
template <bool FAST>
RVT_HDI double dv_truncation(DaviesState& st, double u, double tausq) {
  RVT_DV_HIT(1);
  dv_tick(st);
  if (st.over) return 0.0;
  double sum1 = 0.0, prod2 = 0.0, prod3 = 0.0;
  int s = 0;
  const double sum2 = (st.sigsq + tausq) * (u * u);
  double prod1 = 2.0 * sum2;
  u = 2.0 * u;
  if (FAST) {
    // prod1 += sum_{x <= 1} log(1 + x), prod2 = sum_{x > 1} log(x), prod3 = sum_{x > 1} log(1 + x): three products —
    // functions of u alone (DaviesMemo)
    double la;
    const unsigned long long key = dv_bits(u);
    int slot = -1;
    const bool use_memo = st.memo != nullptr && key < kMemoBusy;
    if (use_memo && dv_memo_find<kMemoSlotsT>(st.memo->tkey, key, &slot)) {
      const volatile RVT_LDSQ double* v = st.memo->tval[slot];
      la = v[0];
      prod2 = v[1];
      prod3 = v[2];
      s = (int)v[3];
    } else {
      RVT_DV_KEY(1, u);
      ScaledProd pa{1.0, 0}, pb{1.0, 0}, pc{1.0, 0};
#pragma unroll 4
      for (int j = 0; j < st.r; ++j) {
        const double t = u * st.lb[j], x = t * t;
        const bool big = x > 1.0;
        pa.m *= big ? 1.0 : 1.0 + x;
        pb.m *= big ? x : 1.0;
        pc.m *= big ? 1.0 + x : 1.0;
        s += big ? 1 : 0;
        if ((j & 3) == 3) {
          sp_renorm(pa);
          sp_renorm(pb);
          sp_renorm(pc);
        }
      }
      la = sp_log(pa);
      prod2 = sp_log(pb);
      prod3 = sp_log(pc);
      if (use_memo && slot >= 0 && dv_memo_claim(st.memo->tkey, slot)) {
        RVT_LDSQ double* v = st.memo->tval[slot];
        v[0] = la;
        v[1] = prod2;
        v[2] = prod3;
        v[3] = (double)s;
        dv_memo_publish(st.memo->tkey, slot, key);
      }
    }
    prod1 = prod1 + la;
  } else
  for (int j0 = 0; j0 < st.r; j0 += 4) {
    const int cnt = (st.r - j0 >= 4) ? 4 : st.r - j0;
    double x[4], l1[4], lx[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const double t = u * ((k < cnt) ? st.lb[j0 + k] : 1.0);
      x[k] = t * t;
    }
    dv_log1_x4(x, cnt, true, l1);
#pragma unroll
    for (int k = 0; k < 4; ++k) lx[k] = log(x[k]);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (k < cnt) {
        if (x[k] > 1.0) {
          prod2 = prod2 + lx[k];
          prod3 = prod3 + l1[k];
          s = s + 1;
        } else
          prod1 = prod1 + l1[k];
      }
    }
  }
  sum1 = 0.5 * sum1;
  prod2 = prod1 + prod2;
  prod3 = prod1 + prod3;
  double x = dv_exp1(-sum1 - 0.25 * prod2) / kDaviesPi;
  const double y = dv_exp1(-sum1 - 0.25 * prod3) / kDaviesPi;
  double err1 = (s == 0) ? 1.0 : x * 2.0 / s;
  double err2 = (prod3 > 1.0) ? 2.5 * y : 1.0;
  if (err2 < err1) err1 = err2;
  x = 0.5 * sum2;
  err2 = (x <= y) ? 1.0 : y / x;
  return (err1 < err2) ? err1 : err2;
}

// find u with truncation(u) < accx and truncation(u/1.2) > accx      (qfc.c:217-238)
template <bool FAST>
RVT_HDI void dv_findu(DaviesState& st, double* utx, double accx) {
  double ut = *utx, u = ut / 4.0;
  if (dv_truncation<FAST>(st, u, 0.0) > accx) {
    for (u = ut; dv_truncation<FAST>(st, u, 0.0) > accx && !st.over; u = ut) ut = ut * 4.0;
  } else {
    ut = u;
    for (u = u / 4.0; dv_truncation<FAST>(st, u, 0.0) <= accx && !st.over; u = u / 4.0) ut = u;
  }
  for (int i = 0; i < 4; i++) {
    const double dv = (i == 0) ? 2.0 : (i == 1 ? 1.4 : (i == 2 ? 1.2 : 1.1));
    u = ut / dv;
    if (dv_truncation<FAST>(st, u, 0.0) <= accx) ut = u;
  }
  *utx = ut;
}

// nterm+1 terms of the inversion integral at step interv      (qfc.c:241-270)
template <bool FAST>
RVT_HDI void dv_integrate(DaviesState& st, int nterm, double interv, double tausq, bool mainx) {
  const double inpi = interv / kDaviesPi;
  for (int k = nterm; k >= 0; k--) {
    const double u = (k + 0.5) * interv;
    double sum1 = -2.0 * u * st.c, sum2 = fabs(sum1);
    double sum3 = -0.5 * st.sigsq * (u * u);
    if (FAST) {
      double sa, sb, sl;
      dv_arg_logmod(st.lb, st.r, 2.0 * u, &sa, &sb, &sl, st.lmin == 0.0);  // (lmin stays 0 when no coefficient is negative)
      sum1 = sum1 + sa;
      sum2 = sum2 + sb;
      sum3 = sum3 - 0.25 * sl;
    } else
    for (int j = st.r - 1; j >= 0; j--) {
      const double x = 2.0 * st.lb[j] * u;
      const double y = x * x;
      sum3 = sum3 - 0.25 * dv_log1(y, true);
      const double z = atan(x);
      sum1 = sum1 + z;
      sum2 = sum2 + fabs(z);
    }
    double x = inpi * dv_exp1(sum3) / u;
    if (!mainx) x = x * (1.0 - dv_exp1(-0.5 * tausq * (u * u)));
    sum1 = sin(0.5 * sum1) * x;
    sum2 = 0.5 * sum2 * x;
    st.intl = st.intl + sum1;
    st.ersm = st.ersm + sum2;
  }
}

// coefficient of tausq in the error when the convergence factor is used      (qfc.c:272-304)
template <bool FAST>
RVT_HDI double dv_cfe(DaviesState& st, double x) {
  dv_tick(st);
  if (st.over) return 1.0;
  double axl = fabs(x);
  const double sxl = (x > 0.0) ? 1.0 : -1.0;
  double sum1 = 0.0;
  // (product form: lj / log28 as a multiplication by the rounded reciprocal — one ulp beside the quotient, on a quantity that
  //  only sets the size of the convergence factor — and the `+ 1.0` loop of qfc.c:291 as one addition of the exact count)
  constexpr double kInvLog28 = 1.0 / kDaviesLog28;
  for (int j = st.r - 1; j >= 0; j--) {
    const double lt = st.ls ? st.ls[j] : st.lb[st.th[j]];
    if (lt * sxl > 0.0) {
      const double lj = fabs(lt);
      const double axl1 = axl - lj, axl2 = FAST ? lj * kInvLog28 : lj / kDaviesLog28;
      if (axl1 > axl2)
        axl = axl1;
      else {
        if (axl > axl2) axl = axl2;
        sum1 = (axl - axl1) / lj;
        if (FAST)
          sum1 = sum1 + (double)j;
        else
          for (int k = j - 1; k >= 0; k--) sum1 = sum1 + 1.0;
        break;
      }
    }
  }
  if (sum1 > 100.0) {
    st.fail = true;
    return 1.0;
  }
  return (FAST ? exp2(sum1 * 0.25) : pow(2.0, (sum1 / 4.0))) / (kDaviesPi * (axl * axl));
}

// ---- the part of qf() that does not depend on the evaluation point c ------------------------------------
// SKAT-O evaluates qf() at ~10^3 points c against ONE coefficient set (every abscissa of every QAGS panel,
// regression/SkatO.cpp:303-325).  In qfc.c:343-377 the moments, the first truncation search
// findu(&utx, .5*acc) and — as long as no convergence factor has been added to sigsq — the two cutoff
// searches ctff(acc1, &up) / ctff(acc1, &un) are functions of the coefficients only.  They are computed
// once per coefficient set here and replayed by davies_qf(); the replay is value-for-value what qf()
// would recompute, so results are unchanged.
struct DaviesPrelude {
  bool valid;        // false: not usable (degenerate coefficients or the search overran lim) -> full path
  bool fast;         // the searches used (and davies_qf_front will use) the product form
  dv_memo_p memo;    // memo of the coefficient sums shared by every evaluation against these coefficients (may be null)
  double sd, mean, lmax, lmin;
  double utx;        // after findu(&utx, .5*acc)
  int cnt_findu;     // errbd/truncation/cfe calls spent so far (qf's `count`)
  double up, c_up;   // ctff(.5*acc, &up) with sigsq = 0
  int cnt_up;
  double un, c_un;   // ctff(.5*acc, &un) with sigsq = 0
  int cnt_un;
};

// all coefficients strictly positive: the product form applies
RVT_HD bool davies_all_positive(const double* lb, int r) {
  bool ok = true;
  for (int j = 0; j < r; ++j) ok = ok && (lb[j] > 0.0);
  return ok;
}

typedef RVT_LDSQ DaviesPrelude* dv_pre_p;
typedef const RVT_LDSQ DaviesPrelude* dv_pre_cp;
template <bool FAST>
// The wrappers below take ORDINARY pointers and hand them to the templated bodies as address-space-3 types — a no-op on the
// host (the harness, the oracle's checks), a truncation to an LDS offset in device code: they are host-only, so that a device
// caller cannot be compiled against them by accident (the kernels call the _t forms with real LDS pointers).
#if defined(__HIP_DEVICE_COMPILE__)
#define RVT_HOST_ONLY __host__ inline
#else
#define RVT_HOST_ONLY RVT_HD
#endif
RVT_HDI void davies_prelude_t(dv_coefs lb, dv_index th, int r, int lim, double acc, dv_pre_p P,
                              dv_memo_p memo = nullptr, dv_coefs ls = nullptr) {
  DaviesState st;
  P->fast = FAST;
  P->memo = FAST ? memo : nullptr;
  st.memo = P->memo;
  st.lb = lb;
  st.th = th;
  st.ls = ls;
  st.r = r;
  st.lim = lim;
  st.c = 0.0;
  st.count = 0;
  st.over = false;
  st.fail = false;
  st.intl = st.ersm = 0.0;
  st.sigsq = 0.0;
  double sd = 0.0;
  st.lmax = st.lmin = st.mean = 0.0;
  for (int j = 0; j < r; j++) {
    const double lj = lb[j];
    sd = sd + (lj * lj) * 2.0;
    st.mean = st.mean + lj;
    if (st.lmax < lj)
      st.lmax = lj;
    else if (st.lmin > lj)
      st.lmin = lj;
  }
  P->valid = false;
  if (sd == 0.0 || (st.lmin == 0.0 && st.lmax == 0.0)) return;
  sd = sqrt(sd);
  P->sd = sd;
  P->mean = st.mean;
  P->lmax = st.lmax;
  P->lmin = st.lmin;
  double utx = 16.0 / sd, up = 4.5 / sd, un = -up;
  dv_findu<FAST>(st, &utx, .5 * acc);
  P->utx = utx;
  P->cnt_findu = st.count;
  const double acc1 = 0.5 * acc;
  P->c_up = dv_ctff<FAST>(st, acc1, &up);
  P->up = up;
  P->cnt_up = st.count - P->cnt_findu;
  const int before = st.count;
  P->c_un = dv_ctff<FAST>(st, acc1, &un);
  P->un = un;
  P->cnt_un = st.count - before;
  P->valid = !st.over;
}
// fast: product form of the coefficient sums (else term by term)
// (the wrappers without template arguments take ordinary pointers: the host test harness and the serial form call them)
RVT_HOST_ONLY void davies_prelude(const double* lb, const int* th, int r, int lim, double acc, DaviesPrelude* P,
                           bool fast = true, DaviesMemo* memo = nullptr) {
  if (fast)
    davies_prelude_t<true>((dv_coefs)lb, (dv_index)th, r, lim, acc, (dv_pre_p)P, (dv_memo_p)memo);
  else
    davies_prelude_t<false>((dv_coefs)lb, (dv_index)th, r, lim, acc, (dv_pre_p)P);
}

// One term of the inversion integral (the body of qfc.c:250-268 for term k, main integration).
template <bool FAST>
RVT_HDI void davies_term_t(dv_coefs lb, int r, double c, double sigsq, double interv, int k, double* t1,
                           double* t2, bool allpos = false) {
  const double inpi = interv / kDaviesPi;
  const double u = (k + 0.5) * interv;
  double sum1 = -2.0 * u * c, sum2 = fabs(sum1);
  double sum3 = -0.5 * sigsq * (u * u);
  if (FAST) {
    double sa, sb, sl;
    dv_arg_logmod(lb, r, 2.0 * u, &sa, &sb, &sl, allpos);
    sum1 = sum1 + sa;
    sum2 = sum2 + sb;
    sum3 = sum3 - 0.25 * sl;
  } else
  for (int j0 = r - 1; j0 >= 0; j0 -= 4) {
    const int cnt = (j0 >= 3) ? 4 : j0 + 1;
    double x[4], y[4], l1[4], z[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      x[k] = 2.0 * ((k < cnt) ? lb[j0 - k] : 1.0) * u;
      y[k] = x[k] * x[k];
    }
    dv_log1_x4(y, cnt, true, l1);
#pragma unroll
    for (int k = 0; k < 4; ++k) z[k] = atan(x[k]);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (k < cnt) {
        sum3 = sum3 - 0.25 * l1[k];
        sum1 = sum1 + z[k];
        sum2 = sum2 + fabs(z[k]);
      }
    }
  }
  const double x = inpi * dv_exp1(sum3) / u;
  *t1 = sin(0.5 * sum1) * x;
  *t2 = 0.5 * sum2 * x;
}
RVT_HOST_ONLY void davies_term(const double* lb, int r, double c, double sigsq, double interv, int k, double* t1,
                        double* t2, bool fast = false, bool allpos = false) {
  if (fast)
    davies_term_t<true>((dv_coefs)lb, r, c, sigsq, interv, k, t1, t2, allpos);
  else
    davies_term_t<false>((dv_coefs)lb, r, c, sigsq, interv, k, t1, t2);
}

// qf() split at its main integration so that the GPU can spread the nt+1 independent terms of MANY
// evaluation points over the lanes of a wave (gene_pvalue_kernel) while keeping the reference's
// summation order:  front -> { for k = nt..0: intl += term1(k); ersm += term2(k) } -> back.
struct DaviesTask {
  bool need_main;   // the main integration has to run: nt, intv, sigsq, c are set
  int nt;
  double intv, sigsq, c, acc;
  double intl, ersm;  // running sums (non-zero when an auxiliary integration ran)
  double qfval;       // result when !need_main
  int fault;
  bool over;
  bool fast;          // evaluate the main integration's terms in the product form
  bool allpos;        // no coefficient is negative (dv_arg_logmod's short form applies)
  double nterms;      // terms already evaluated (auxiliary integrations)
};

// `pre` (optional) = davies_prelude() of the same coefficients: the c-independent searches are replayed.
// the caller guarantees that `pre` (if any) was computed in the same form
template <bool FAST>
RVT_HDI void davies_qf_front_t(dv_coefs lb, dv_index th, int r, double c, int lim, double acc,
                               dv_pre_cp pre, DaviesTask* task, dv_memo_p memo = nullptr, dv_coefs ls = nullptr) {
  DaviesState st;
  task->fast = FAST;
  // (the GPU kernel passes the memo directly, so that the compiler sees an LDS address; the host takes the prelude's)
  st.memo = !FAST ? nullptr : (memo ? memo : (pre ? pre->memo : nullptr));
  if (pre && !pre->valid) pre = nullptr;
  st.lb = lb;
  st.th = th;
  st.ls = ls;
  st.r = r;
  st.lim = lim;
  st.c = c;
  st.count = 0;
  st.over = false;
  st.fail = false;
  st.intl = 0.0;
  st.ersm = 0.0;
  task->need_main = false;
  task->fault = 0;
  task->qfval = -1.0;
  task->nterms = 0.0;
  task->c = c;
  task->acc = acc;
  double acc1 = acc;
  double xlim = (double)lim;
  st.sigsq = 0.0;
  double sd = 0.0;
  st.lmax = 0.0;
  st.lmin = 0.0;
  st.mean = 0.0;
  if (pre) {  // (a valid prelude holds the moments: the same loop over the same coefficients)
    sd = pre->sd;  // (non-zero, which is all that is tested below; the root itself is taken from the prelude there)
    st.mean = pre->mean;
    st.lmax = pre->lmax;
    st.lmin = pre->lmin;
  } else
  for (int j = 0; j < r; j++) {
    const double lj = lb[j];
    sd = sd + (lj * lj) * 2.0;
    st.mean = st.mean + lj;
    if (st.lmax < lj)
      st.lmax = lj;
    else if (st.lmin > lj)
      st.lmin = lj;
  }
  task->allpos = (st.lmin == 0.0);  // (lmin starts at 0 and only a negative coefficient lowers it)
  bool done = false;
  double utx = 0, up = 0, un = 0, intv = 0, xnt = 0;
  if (sd == 0.0) {
    task->qfval = (c > 0.0) ? 1.0 : 0.0;
    done = true;
  } else if (st.lmin == 0.0 && st.lmax == 0.0) {
    task->fault = 3;
    done = true;
  }
  if (!done) {
    sd = pre ? pre->sd : sqrt(sd);
    const double almx = (st.lmax < -st.lmin) ? -st.lmin : st.lmax;
    utx = 16.0 / sd;
    up = 4.5 / sd;
    un = -up;
    if (pre) {
      utx = pre->utx;
      st.count = pre->cnt_findu;
    } else {
      dv_findu<FAST>(st, &utx, .5 * acc1);
    }
    bool sig_changed = false;  // has a convergence factor been added to sigsq?
    if (c != 0.0 && (almx > 0.07 * sd)) {
      const double tausq = .25 * acc1 / dv_cfe<FAST>(st, c);
      if (st.fail)
        st.fail = false;
      else if (dv_truncation<FAST>(st, utx, tausq) < .2 * acc1) {
        st.sigsq = st.sigsq + tausq;
        sig_changed = true;
        dv_findu<FAST>(st, &utx, .25 * acc1);
      }
    }
    acc1 = 0.5 * acc1;
    bool to_main = false;
    while (!done && !to_main && !st.over) {
      double cut_up, cut_un;
      const bool replay = pre && !sig_changed;  // first pass, sigsq still 0, acc1 still .5*acc
      if (replay) {
        cut_up = pre->c_up;
        up = pre->up;
        st.count += pre->cnt_up;
        if (st.count > st.lim) st.over = true;
      } else {
        cut_up = dv_ctff<FAST>(st, acc1, &up);
      }
      const double d1 = cut_up - c;
      if (st.over) break;
      if (d1 < 0.0) {
        task->qfval = 1.0;
        done = true;
        break;
      }
      if (replay) {
        cut_un = pre->c_un;
        un = pre->un;
        st.count += pre->cnt_un;
        if (st.count > st.lim) st.over = true;
      } else {
        cut_un = dv_ctff<FAST>(st, acc1, &un);
      }
      const double d2 = c - cut_un;
      if (st.over) break;
      if (d2 < 0.0) {
        task->qfval = 0.0;
        done = true;
        break;
      }
      intv = 2.0 * kDaviesPi / ((d1 > d2) ? d1 : d2);
      xnt = utx / intv;
      const double xntm = 3.0 / sqrt(acc1);
      if (xnt > xntm * 1.5) {
        if (xntm > xlim) {
          task->fault = 1;
          done = true;
          break;
        }
        const int ntm = (int)floor(xntm + 0.5);
        const double intv1 = utx / ntm;
        const double x = 2.0 * kDaviesPi / intv1;
        if (x <= fabs(c)) {
          to_main = true;
          break;
        }
        const double cf = dv_cfe<FAST>(st, c - x) + dv_cfe<FAST>(st, c + x);
        if (st.over) break;
        const double tausq = .33 * acc1 / (1.1 * cf);
        if (st.fail) {
          to_main = true;
          break;
        }
        acc1 = .67 * acc1;
        RVT_DV_HIT(4);  // an auxiliary integration
        dv_integrate<FAST>(st, ntm, intv1, tausq, false);
        task->nterms += ntm + 1;
        xlim = xlim - xntm;
        st.sigsq = st.sigsq + tausq;
        sig_changed = true;
        dv_findu<FAST>(st, &utx, .25 * acc1);
        acc1 = 0.75 * acc1;
        continue;
      }
      to_main = true;
    }
    if (!done && !st.over) {
      if (xnt > xlim) {
        task->fault = 1;
      } else {
        task->need_main = true;
        task->nt = (int)floor(xnt + 0.5);
        task->intv = intv;
        task->sigsq = st.sigsq;
      }
    }
  }
  task->intl = st.intl;
  task->ersm = st.ersm;
  task->over = st.over;
  if (st.over) {
    task->fault = 4;
    task->need_main = false;
  }
}
RVT_HOST_ONLY void davies_qf_front(const double* lb, const int* th, int r, double c, int lim, double acc,
                            const DaviesPrelude* pre, DaviesTask* task, bool fast = true) {
  if (fast)
    davies_qf_front_t<true>((dv_coefs)lb, (dv_index)th, r, c, lim, acc, (dv_pre_cp)pre, task);
  else
    davies_qf_front_t<false>((dv_coefs)lb, (dv_index)th, r, c, lim, acc, (dv_pre_cp)pre, task);
}

// after the main integration: qfval and the round-off test (qfc.c:424-431)
RVT_HD double davies_qf_back(const DaviesTask& task, double intl, double ersm, int* ifault) {
  *ifault = task.fault;
  if (!task.need_main) return task.qfval;
  const double qfval = 0.5 - intl;
  const double up2 = ersm;
  const double x = up2 + task.acc / 10.0;
  if (1.0 * x == 1.0 * up2 || 2.0 * x == 2.0 * up2 || 4.0 * x == 4.0 * up2 || 8.0 * x == 8.0 * up2) *ifault = 2;
  return qfval;
}

// P[ sum_j lb_j chi²_1 < c ]; *ifault as in the reference (0 ok, 1 accuracy, 2 round-off, 3 invalid,
// 4 search overran lim).  nterms_out (optional) = number of integrand terms evaluated.
RVT_HD double davies_qf(const double* lb, const int* th, int r, double c, int lim, double acc, int* ifault,
                        double* nterms_out, const DaviesPrelude* pre = nullptr, bool fast = true) {
  DaviesTask task;
  if (pre) fast = pre->fast;
  davies_qf_front(lb, th, r, c, lim, acc, pre, &task, fast);
  double intl = task.intl, ersm = task.ersm;
  if (task.need_main) {
    for (int k = task.nt; k >= 0; k--) {
      double t1, t2;
      davies_term(lb, r, task.c, task.sigsq, task.intv, k, &t1, &t2, task.fast, task.allpos);
      intl = intl + t1;
      ersm = ersm + t2;
    }
    task.nterms += task.nt + 1;
  }
  if (nterms_out) *nterms_out = task.nterms;
  return davies_qf_back(task, intl, ersm, ifault);
}

// ---- Liu et al. moment matching through the non-central chi-square  -----------------------------
// cumulative central chi-square, both tails       (cdflib cumchi -> cumgam -> gamma_inc)
RVT_HD void chisq_both_tails(double x, double df, double* cum, double* ccum) {
  const double a = df * 0.5, xx = x * 0.5;
  if (xx <= 0.0) {
    *cum = 0.0;
    *ccum = 1.0;
    return;
  }
  *cum = igam_P(a, xx);
  *ccum = igam_Q(a, xx);
}

// non-central chi-square CDF: Poisson-weighted sum started at the central term, truncated with the
// reference's eps = 1e-5 / 1000-term rules (they set the digits, so they are kept), including its
// `sumadj = sum + adj` update in the forward sweep        (regression/cdflib.cpp:5172-5350)
// The pieces that do not depend on x (central index, its Poisson weight, the lgamma of the adjustment term)
// are split off so that many evaluations against one (df, pnonc) — every abscissa of a SKAT-O quadrature —
// pay for them once; values are exactly those the one-shot form computes.
struct NoncentralPre {
  double df, pnonc;
  bool central;      // pnonc <= 1e-10: plain chi-square
  double xnonc;
  int icent;
  double centwt;     // exp(-xnonc + icent log(xnonc) - lgamma(icent + 1))
  double dfd2c;      // (df + 2 icent) / 2
  double lgam_adj;   // lgamma(1 + dfd2c)
};

RVT_HD NoncentralPre noncentral_prepare(double df, double pnonc) {
  NoncentralPre q;
  q.df = df;
  q.pnonc = pnonc;
  q.central = (pnonc <= 1.0e-10);
  q.xnonc = pnonc / 2.0;
  q.icent = 1;
  q.centwt = 0.0;
  q.dfd2c = 0.0;
  q.lgam_adj = 0.0;
  if (q.central) return q;
  int icent = (int)q.xnonc;
  if (icent == 0) icent = 1;
  q.icent = icent;
  q.centwt = exp(-q.xnonc + (double)icent * log(q.xnonc) - lgamma((double)(icent + 1)));
  q.dfd2c = (df + 2.0 * (double)icent) / 2.0;
  q.lgam_adj = lgamma(1.0 + q.dfd2c);
  return q;
}

RVT_HD void noncentral_chisq_tails_pre(const NoncentralPre& q, double x, double* cum, double* ccum) {
  const double eps = 1.0e-5;
  const int ntired = 1000;
  if (x <= 0.0) {
    *cum = 0.0;
    *ccum = 1.0;
    return;
  }
  if (q.central) {
    chisq_both_tails(x, q.df, cum, ccum);
    return;
  }
  const double df = q.df, xnonc = q.xnonc;
  const int icent = q.icent;
  const double chid2 = x / 2.0;
  const double centwt = q.centwt;
  double pcent, tmp;
  chisq_both_tails(x, df + 2.0 * (double)icent, &pcent, &tmp);
  double dfd2 = q.dfd2c;
  const double centaj = exp(dfd2 * log(chid2) - chid2 - q.lgam_adj);
  double sum = centwt * pcent;
  // towards zero
  double sumadj = 0.0, adj = centaj, wt = centwt, term;
  int i = icent, iter = 0;
  do {
    dfd2 = (df + 2.0 * (double)i) / 2.0;
    adj = adj * dfd2 / chid2;
    sumadj = sumadj + adj;
    wt *= ((double)i / xnonc);
    term = wt * (pcent + sumadj);
    sum = sum + term;
    i -= 1;
    iter += 1;
  } while (!(iter > ntired || (sum < 1.0e-20 || term < eps * sum) || i == 0));
  // towards infinity
  sumadj = adj = centaj;
  wt = centwt;
  i = icent;
  iter = 0;
  do {
    wt *= (xnonc / (double)(i + 1));
    term = wt * (pcent - sumadj);
    sum = sum + term;
    i += 1;
    dfd2 = (df + 2.0 * (double)i) / 2.0;
    adj = adj * chid2 / dfd2;
    sumadj = sum + adj;
    iter += 1;
  } while (!(iter > ntired || (sum < 1.0e-20 || term < eps * sum)));
  *cum = sum;
  *ccum = 0.5 + (0.5 - sum);
}

RVT_HD void noncentral_chisq_tails(double x, double df, double pnonc, double* cum, double* ccum) {
  if (x <= 0.0) {
    *cum = 0.0;
    *ccum = 1.0;
    return;
  }
  const NoncentralPre q = noncentral_prepare(df, pnonc);
  noncentral_chisq_tails_pre(q, x, cum, ccum);
}

RVT_HD double dv_powsum(const double* d, int n, int power) {
  double r = 0.0;
  for (int i = 0; i < n; ++i) {
    double t = d[i];
    for (int j = 1; j < power; ++j) t *= d[i];
    r += t;
  }
  return r;
}

// MixtureChiSquare::getLiuPvalue      (regression/MixtureChiSquare.cpp:44-83), split into the part that only
// depends on the coefficients (moments -> a, delta, l) and the part that depends on Q.
struct LiuPre {
  double muQ, sigmaQ, muX, sigmaX, l, delta;
  bool bad_args;  // cdfchn would return status != 0 regardless of x (df <= 0 or ncp < 0)
  NoncentralPre nc;
};

RVT_HD LiuPre liu_prepare(const double* lambda, int n) {
  LiuPre p;
  const double c1 = dv_powsum(lambda, n, 1), c2 = dv_powsum(lambda, n, 2), c3 = dv_powsum(lambda, n, 3),
               c4 = dv_powsum(lambda, n, 4);
  const double s1 = c3 / c2 / sqrt(c2), s2 = c4 / c2 / c2;
  p.muQ = c1;
  p.sigmaQ = sqrt(2.0 * c2);
  double a, delta, l;
  if (s1 * s1 > s2) {
    a = 1 / (s1 - sqrt(s1 * s1 - s2));
    delta = (s1 * a - 1) * a * a;
    l = a * a - 2.0 * delta;
  } else {
    a = 1.0 / s1;
    delta = 0.0;
    l = c2 * c2 * c2 / c3 / c3;
  }
  p.l = l;
  p.delta = delta;
  p.muX = l + delta;
  p.sigmaX = sqrt(2.0) * a;
  p.bad_args = (l <= 0.0 || delta < 0.0);
  p.nc = noncentral_prepare(l, delta);
  return p;
}

RVT_HD double liu_pvalue_pre(const LiuPre& p, double Q) {
  const double tstar = (Q - p.muQ) / p.sigmaQ;
  const double x = tstar * p.sigmaX + p.muX;
  if (x < 0.0 || p.bad_args) return 1.0;  // cdfchn status != 0
  double cum, q;
  noncentral_chisq_tails_pre(p.nc, x, &cum, &q);
  return q;
}

RVT_HD double liu_pvalue(const double* lambda, int n, double Q) {
  const LiuPre p = liu_prepare(lambda, n);
  return liu_pvalue_pre(p, Q);
}

// MixtureChiSquare::getPvalue        (regression/MixtureChiSquare.cpp:7-29)
// All coefficients on the hot path are > 0 (Skat.cpp:92 keeps lambda > 1e-30, SkatO.cpp:365-374 keeps
// lambda >= mean/1e5), so for Q < 0 the reference's qf() stops at "d2 < 0 -> qfval = 0" (or faults) and
// getPvalue returns exactly 1 (or -1); every caller on the hot path then replaces that value by Liu's
// (Skat.cpp:100-103, SkatO.cpp:318-321).  The search is therefore skipped for Q < 0 and 1.0 returned —
// pinned against the compiled reference by tests/test_oracle_ref.py::test_negative_q_is_one_or_fault.
RVT_HD double davies_pvalue(const double* lambda, const int* th, int n, double Q, int* fault_out,
                            double* nterms_out, const DaviesPrelude* pre = nullptr, bool fast = true) {
  if (nterms_out) *nterms_out = 0.0;
  if (fault_out) *fault_out = 0;
  if (n == 1) return liu_pvalue(lambda, n, Q);
  if (Q < 0.0) return 1.0;
  int fault;
  double p = 1.0 - davies_qf(lambda, th, n, Q, 10000, 0.000001, &fault, nterms_out, pre, fast);
  if (p > 1.0) p = 1.0;
  if (fault) p = -1.0;
  if (fault_out) *fault_out = fault;
  return p;
}

// indices of lb by decreasing |lb| (stable), as qfc.c:115-134 builds them
RVT_HD void davies_order(const double* lb, int r, int* th) {
  for (int j = 0; j < r; j++) {
    const double lj = fabs(lb[j]);
    int k = j - 1;
    for (; k >= 0; k--) {
      if (lj > fabs(lb[th[k]]))
        th[k + 1] = th[k];
      else
        break;
    }
    th[k + 1] = j;
  }
}

}  // namespace rvt
