// rvtests_amd — eigendecomposition of a DENSE kinship (a genetic relationship matrix) by Householder tridiagonalisation,
// bisection and inverse iteration (round 5).
//
// Replaces, for dense matrices, KinshipHolder::decompose (base/KinshipHolder.cpp:270-290: Eigen::SelfAdjointEigenSolver on the
// N x N kinship; eigenvalues ascending in matS, eigenvectors in the columns of matU).  The one-sided block Jacobi iteration of
// jacobi_kernels.hip.h needs 16-30 sweeps on a gap-free spectrum, every sweep N^3 bytes of HBM traffic: 44 s at N = 12 000.
// Here:
//   1. K = Q T Q'.  Blocked Householder tridiagonalisation (LAPACK's dsytrd / dlatrd scheme, panels of kTdNb columns, the full
//      symmetric matrix kept so that the O(N^2) product of every column is a plain column-dot-product pass: 8 N^3 / 3 bytes of
//      traffic in all, the rank-2nb updates 16 N^3 / nb).  The reflector v_j overwrites column j of the matrix.
//   2. The eigenvalues of T by the Sturm bisection + interpolation of rvt_coop.h, one thread per eigenvalue.
//   3. The eigenvectors of T by inverse iteration, one thread per eigenvector (tridiagonal LU with partial pivoting, three
//      solves), WITHOUT reorthogonalisation: the vectors of eigenvalues a gap g apart are orthogonal to ~eps / g, and the
//      boundary stores U as FLOAT (EigenMatrix = Eigen::MatrixXf) — a gap of 1e-7 of the spectrum's width is enough.  A matrix
//      with a tighter cluster (repeated eigenvalues: pedigree kinships, rank-deficient matrices) is left to the Jacobi
//      iteration, which does not care; so is a result that fails the residual / orthogonality check that closes the procedure.
//   4. U = Q Z: the reflectors applied panel by panel in compact WY form, Z <- Z - V (T_p (V'Z)); V'Z is the fp64 matrix-core
//      product of gemm_f64.hip.h (both operands run along the rows).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "rvt_coop.h"

namespace rvt {

constexpr int kTdNb = 32;  // panel width

// float K (column-major n x n, leading dimension n) -> fp64 A (n columns, leading dimension ld >= n, pad rows zero)
static __global__ void td_init_kernel(const float* __restrict__ K, long long n, long long ld, double* __restrict__ A) {
  const long long total = n * ld;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const long long j = idx / ld, i = idx % ld;
    A[idx] = (i < n) ? (double)K[j * n + i] : 0.0;
  }
}

__device__ __forceinline__ double td_block_sum(double v, double* red) {  // sum over a workgroup of <= 1024 threads
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  double s = 0.0;
  for (int w = 0; w < nw; ++w) s += red[w];  // fixed order
  return s;
}

// Column j of the matrix, i = j - j0 columns of the panel before it (V = columns j0 .. j0 + i - 1 of A, W = the panel's W):
//   a[r] -= sum_p V[r, p] W[j, p] + W[r, p] V[j, p]   for r >= j          (dlatrd's update of the column)
//   d[j] = a[j];  reflector of x = a[j + 1 .. n): beta = -sign(x0) |x|, tau = (beta - x0) / beta, v = [1; x[1:] / (x0 - beta)]
//   e[j] = beta;  column j of A becomes v (zeros in rows 0 .. j)
// ONE workgroup of 1024 threads.
static __global__ __launch_bounds__(1024) void td_col_house_kernel(double* __restrict__ A, long long n, long long ld, int j, int j0,
                                                            const double* __restrict__ W, double* __restrict__ dvec,
                                                            double* __restrict__ evec, double* __restrict__ tau) {
  __shared__ double red[16];
  __shared__ double wj[kTdNb], vj[kTdNb];
  const int i = j - j0, tid = threadIdx.x;
  double* a = A + (long long)j * ld;
  if (tid < i) {
    wj[tid] = W[(long long)tid * ld + j];
    vj[tid] = A[(long long)(j0 + tid) * ld + j];
  }
  __syncthreads();
  double ss = 0.0;  // sum of squares of x[1:]
  for (long long r = j + tid; r < n; r += 1024) {
    double v = a[r];
    for (int p = 0; p < i; ++p) v -= A[(long long)(j0 + p) * ld + r] * wj[p] + W[(long long)p * ld + r] * vj[p];
    a[r] = v;
    if (r > j + 1) ss += v * v;
  }
  const double xss = td_block_sum(ss, red);
  __syncthreads();
  const double ajj = a[j], x0 = (j + 1 < n) ? a[j + 1] : 0.0;
  double beta = x0, t = 0.0, scale = 0.0;
  if (xss > 0.0) {
    const double nrm = sqrt(x0 * x0 + xss);
    beta = (x0 >= 0.0) ? -nrm : nrm;
    t = (beta - x0) / beta;
    scale = 1.0 / (x0 - beta);
  }
  __syncthreads();
  for (long long r = tid; r < n; r += 1024) {
    double v = 0.0;
    if (r == j + 1)
      v = 1.0;
    else if (r > j + 1)
      v = a[r] * scale;
    a[r] = v;
  }
  if (tid == 0) {
    dvec[j] = ajj;
    if (j + 1 < n) evec[j] = beta;
    tau[j] = t;
  }
}

// Column dot products against v = column j of A (rows j + 1 .. n):
//   group 0: y[c] = A[:, c]' v                 for c in (j, n)         (the trailing matrix: A22 v, A symmetric)
//   group 1: t1[p] = W[:, p]' v,  group 2: t2[p] = V[:, p]' v          for p < i
// grid = (n - j - 1) + 2 i workgroups of 256 threads, one per column.
static __global__ __launch_bounds__(256) void td_dots_kernel(const double* __restrict__ A, long long n, long long ld, int j, int j0,
                                                       const double* __restrict__ W, double* __restrict__ y,
                                                       double* __restrict__ t12) {
  __shared__ double red[4];
  const int i = j - j0, b = blockIdx.x;
  const long long ntrail = n - j - 1;
  const double* col;
  double* out;
  if (b < ntrail) {
    col = A + (long long)(j + 1 + b) * ld;
    out = y + (j + 1 + b);
  } else if (b < ntrail + i) {
    col = W + (long long)(b - ntrail) * ld;
    out = t12 + (b - ntrail);
  } else {
    col = A + (long long)(j0 + (b - ntrail - i)) * ld;
    out = t12 + kTdNb + (b - ntrail - i);
  }
  const double* v = A + (long long)j * ld;
  double s = 0.0;
  for (long long r = j + 1 + threadIdx.x; r < n; r += 256) s = fma(col[r], v[r], s);
  const double tot = td_block_sum(s, red);
  if (threadIdx.x == 0) *out = tot;
}

// w'[r] = tau (y[r] - sum_p V[r, p] t1[p] + W[r, p] t2[p]),  r > j; per-workgroup partial of w' . v into part[blockIdx.x]
static __global__ __launch_bounds__(256) void td_w_comb_kernel(const double* __restrict__ A, long long n, long long ld, int j, int j0,
                                                         double* __restrict__ W, const double* __restrict__ y,
                                                         const double* __restrict__ t12, const double* __restrict__ tau,
                                                         double* __restrict__ part) {
  __shared__ double red[4];
  __shared__ double t1[kTdNb], t2[kTdNb];
  const int i = j - j0;
  if ((int)threadIdx.x < i) {
    t1[threadIdx.x] = t12[threadIdx.x];
    t2[threadIdx.x] = t12[kTdNb + threadIdx.x];
  }
  __syncthreads();
  const double tj = tau[j];
  const double* v = A + (long long)j * ld;
  double* w = W + (long long)i * ld;
  double dot = 0.0;
  for (long long r = (long long)blockIdx.x * 256 + threadIdx.x; r < n; r += (long long)gridDim.x * 256) {
    double val = 0.0;
    if (r > j) {
      val = y[r];
      for (int p = 0; p < i; ++p) val -= A[(long long)(j0 + p) * ld + r] * t1[p] + W[(long long)p * ld + r] * t2[p];
      val *= tj;
      dot = fma(val, v[r], dot);
    }
    w[r] = val;
  }
  const double tot = td_block_sum(dot, red);
  if (threadIdx.x == 0) part[blockIdx.x] = tot;
}
// w = w' - (tau / 2) (w' . v) v        (one workgroup; nparts partial dots)
static __global__ __launch_bounds__(1024) void td_w_final_kernel(const double* __restrict__ A, long long n, long long ld, int j, int j0,
                                                           double* __restrict__ W, const double* __restrict__ tau,
                                                           const double* __restrict__ part, int nparts) {
  double dot = 0.0;
  for (int k = 0; k < nparts; ++k) dot += part[k];  // fixed order, every thread the same
  const double alpha = -0.5 * tau[j] * dot;
  const double* v = A + (long long)j * ld;
  double* w = W + (long long)(j - j0) * ld;
  for (long long r = j + 1 + threadIdx.x; r < n; r += 1024) w[r] = fma(alpha, v[r], w[r]);
}

// A[r, c] -= sum_p V[r, p] W[c, p] + W[r, p] V[c, p]   for r, c >= j1 (the columns behind the panel j0 .. j1 - 1)
// grid (tiles, tiles) of 64 x 64 outputs, 256 threads (4 x 4 outputs each); the panel rows of both tiles in LDS
static __global__ __launch_bounds__(256) void td_rank2k_kernel(double* __restrict__ A, long long n, long long ld, int j0, int j1,
                                                         const double* __restrict__ W) {
  __shared__ double vr[kTdNb][64 + 1], wr[kTdNb][64 + 1], vc[kTdNb][64 + 1], wc[kTdNb][64 + 1];
  const long long r0 = j1 + (long long)blockIdx.x * 64, c0 = j1 + (long long)blockIdx.y * 64;
  const int nb = j1 - j0;
  for (int idx = threadIdx.x; idx < nb * 64; idx += 256) {
    const int p = idx / 64, q = idx % 64;
    const long long r = r0 + q, c = c0 + q;
    vr[p][q] = r < n ? A[(long long)(j0 + p) * ld + r] : 0.0;
    wr[p][q] = r < n ? W[(long long)p * ld + r] : 0.0;
    vc[p][q] = c < n ? A[(long long)(j0 + p) * ld + c] : 0.0;
    wc[p][q] = c < n ? W[(long long)p * ld + c] : 0.0;
  }
  __syncthreads();
  const int tr = (threadIdx.x & 15) * 4, tc = (threadIdx.x >> 4) * 4;
  double acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = 0.0;
  for (int p = 0; p < nb; ++p) {
    double a_v[4], a_w[4], b_v[4], b_w[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      a_v[a] = vr[p][tr + a];
      a_w[a] = wr[p][tr + a];
      b_v[a] = vc[p][tc + a];
      b_w[a] = wc[p][tc + a];
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = fma(a_v[a], b_w[b], fma(a_w[a], b_v[b], acc[a][b]));
  }
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const long long r = r0 + tr + a, c = c0 + tc + b;
      if (r < n && c < n) A[c * ld + r] -= acc[a][b];
    }
}

// ---- eigenvalues and eigenvectors of the tridiagonal ---------------------------------------------------------------------
// ds / e2s: the diagonal and the floored squares of the couplings of the matrix scaled by 2^-sh (as coop_tridiag_eigvals
// scales); lam[idx] = eigenvalue idx (ascending) of the UNSCALED matrix.  One thread per eigenvalue.
static __global__ __launch_bounds__(64) void td_eigvals_kernel(const double* __restrict__ ds, const double* __restrict__ e2s, int n,
                                                        double lo, double hi, double span, int sh, double* __restrict__ lam) {
  const int idx = blockIdx.x * 64 + threadIdx.x;
  if (idx >= n) return;
  lam[idx] = ldexp(sturm_eigenvalue(ds, e2s, n, idx, lo, hi, span, DBL_MIN * 1024.0), sh);
}

// Inverse iteration, one thread per eigenvector k: (T - lam_k I) z = b by LU with partial pivoting, three times; the rows of
// the factor and the vector live in [row][k] arrays (nk = padded count: consecutive threads touch consecutive doubles).
// zt[i * nk + k] = component i of eigenvector k, 2-norm 1.
static __global__ __launch_bounds__(64) void td_invit_kernel(const double* __restrict__ d, const double* __restrict__ e, int n,
                                                      const double* __restrict__ lam, long long nk, double pert,
                                                      double* __restrict__ ud, double* __restrict__ uu, double* __restrict__ uw,
                                                      double* __restrict__ zt) {
  const long long k = (long long)blockIdx.x * 64 + threadIdx.x;
  if (k >= n) return;
  const double l = lam[k];
  // start vector: deterministic, no zero component, not aligned with anything in particular
  for (int i = 0; i < n; ++i) {
    unsigned h = (unsigned)i * 2654435761u ^ ((unsigned)k * 40503u + 0x9E3779B9u);
    h ^= h >> 15;
    h *= 2246822519u;
    h ^= h >> 13;
    zt[(long long)i * nk + k] = 0.5 + (double)(h & 0xffffu) * (1.0 / 131072.0);
  }
  for (int iter = 0; iter < 3; ++iter) {
    // forward elimination; row i of U = (ud, uu, uw) at columns i, i + 1, i + 2; the transformed right-hand side in zt
    double cd = d[0] - l, cu = (n > 1) ? e[0] : 0.0, cw = 0.0, cb = zt[k];
    for (int i = 0; i + 1 < n; ++i) {
      const double sub = e[i], nd = d[i + 1] - l, nu = (i + 2 < n) ? e[i + 1] : 0.0, nb = zt[(long long)(i + 1) * nk + k];
      const long long at = (long long)i * nk + k;
      if (fabs(cd) >= fabs(sub)) {
        if (cd == 0.0) cd = pert;
        const double f = sub / cd;
        ud[at] = cd;
        uu[at] = cu;
        uw[at] = cw;
        zt[at] = cb;
        cd = nd - f * cu;
        cu = nu - f * cw;
        cw = 0.0;
        cb = nb - f * cb;
      } else {
        const double f = cd / sub;
        ud[at] = sub;
        uu[at] = nd;
        uw[at] = nu;
        zt[at] = nb;
        cd = cu - f * nd;
        cu = cw - f * nu;
        cw = 0.0;
        cb = cb - f * nb;
      }
    }
    if (fabs(cd) < pert) cd = (cd < 0.0) ? -pert : pert;
    // back substitution
    double z2 = 0.0, z1 = cb / cd, mx = fabs(z1);
    zt[(long long)(n - 1) * nk + k] = z1;
    for (int i = n - 2; i >= 0; --i) {
      const long long at = (long long)i * nk + k;
      double piv = ud[at];
      if (fabs(piv) < pert) piv = (piv < 0.0) ? -pert : pert;
      const double z0 = (zt[at] - uu[at] * z1 - uw[at] * z2) / piv;
      zt[at] = z0;
      mx = fmax(mx, fabs(z0));
      z2 = z1;
      z1 = z0;
    }
    // scale (max norm, then the 2-norm after the last solve)
    const double sc = (mx > 0.0 && mx < INFINITY) ? 1.0 / mx : 1.0;
    double ss = 0.0;
    for (int i = 0; i < n; ++i) {
      const long long at = (long long)i * nk + k;
      const double v = zt[at] * sc;
      zt[at] = v;
      ss = fma(v, v, ss);
    }
    if (iter == 2) {
      const double nrm = 1.0 / sqrt(ss);
      for (int i = 0; i < n; ++i) zt[(long long)i * nk + k] *= nrm;
    }
  }
}

// zc[i + k * ld] = zt[i * nk + k]   (32 x 32 tiles through LDS)
static __global__ __launch_bounds__(256) void td_transpose_kernel(const double* __restrict__ zt, long long nk, int n, long long ld,
                                                            double* __restrict__ zc) {
  __shared__ double tile[32][33];
  const long long i0 = (long long)blockIdx.x * 32, k0 = (long long)blockIdx.y * 32;
  for (int t = threadIdx.x; t < 1024; t += 256) {
    const int a = t / 32, b = t % 32;  // a: row i, b: column k (k fastest in zt)
    const long long i = i0 + a, k = k0 + b;
    tile[a][b] = (i < n && k < n) ? zt[i * nk + k] : 0.0;
  }
  __syncthreads();
  for (int t = threadIdx.x; t < 1024; t += 256) {
    const int b = t / 32, a = t % 32;  // a fastest: row i contiguous in zc
    const long long i = i0 + a, k = k0 + b;
    if (i < n && k < n) zc[k * ld + i] = tile[a][b];
  }
}

// ---- back-transformation U = Q Z, panel by panel ----------------------------------------------------------------------------
// The triangular factor of the panel's block reflector H_j0 ... H_{j1-1} = I - V T V' (dlarft, forward, columnwise):
//   T[i][i] = tau_i,  T[0:i, i] = -tau_i T[0:i, 0:i] (V[:, 0:i]' v_i).
// td_gram_kernel: G[a][b] = v_a' v_b for a < b, one workgroup per pair (grid nb x nb);  td_larft_kernel: T from G and tau, one
// thread (nb^3 / 6 multiply-adds).  g, tp: kTdNb x kTdNb, row-major.
static __global__ __launch_bounds__(256) void td_gram_kernel(const double* __restrict__ A, long long n, long long ld, int j0,
                                                       double* __restrict__ g) {
  __shared__ double red[4];
  const int a = blockIdx.x, b = blockIdx.y;
  if (a >= b) return;
  const double* va = A + (long long)(j0 + a) * ld;
  const double* vb = A + (long long)(j0 + b) * ld;
  double s = 0.0;
  for (long long r = j0 + b + 1 + threadIdx.x; r < n; r += 256) s = fma(va[r], vb[r], s);
  const double tot = td_block_sum(s, red);
  if (threadIdx.x == 0) g[a * kTdNb + b] = tot;
}
static __global__ __launch_bounds__(64) void td_larft_kernel(const double* __restrict__ g, int j0, int nb,
                                                       const double* __restrict__ tau, double* __restrict__ tp) {
  __shared__ double T[kTdNb][kTdNb];
  if (threadIdx.x != 0) return;
  for (int i = 0; i < nb; ++i) {
    for (int a = 0; a < kTdNb; ++a) T[a][i] = 0.0;
    const double ti = tau[j0 + i];
    T[i][i] = ti;
    for (int a = 0; a < i; ++a) {
      double s = 0.0;
      for (int b = a; b < i; ++b) s += T[a][b] * g[b * kTdNb + i];
      T[a][i] = -ti * s;
    }
  }
  for (int a = 0; a < kTdNb; ++a)
    for (int b = 0; b < kTdNb; ++b) tp[a * kTdNb + b] = (a < nb && b < nb) ? T[a][b] : 0.0;
}
// y[k, p] = sum_q cz[k, q] T[p, q]   (cz = Z'V, n x nb column-major with leading dimension ldc;  y likewise)
static __global__ __launch_bounds__(256) void td_apply_t_kernel(const double* __restrict__ cz, long long ldc, int n, int nb,
                                                          const double* __restrict__ tp, double* __restrict__ y) {
  __shared__ double T[kTdNb][kTdNb];
  for (int t = threadIdx.x; t < kTdNb * kTdNb; t += 256) T[t / kTdNb][t % kTdNb] = tp[t];
  __syncthreads();
  const long long k = (long long)blockIdx.x * 256 + threadIdx.x;
  if (k >= n) return;
  double c[kTdNb];
  for (int q = 0; q < nb; ++q) c[q] = cz[(long long)q * ldc + k];
  for (int p = 0; p < nb; ++p) {
    double s = 0.0;
    for (int q = p; q < nb; ++q) s = fma(c[q], T[p][q], s);  // T upper triangular
    y[(long long)p * ldc + k] = s;
  }
}
// Z[r, k] -= sum_p V[r, p] y[k, p]    grid (row tiles, column tiles) of 64 x 64, 256 threads
static __global__ __launch_bounds__(256) void td_update_z_kernel(double* __restrict__ Z, long long n, long long ld,
                                                           const double* __restrict__ A, int j0, int nb,
                                                           const double* __restrict__ y, long long ldc) {
  __shared__ double vr[kTdNb][64 + 1], yk[kTdNb][64 + 1];
  const long long r0 = (long long)blockIdx.x * 64, k0 = (long long)blockIdx.y * 64;
  for (int idx = threadIdx.x; idx < nb * 64; idx += 256) {
    const int p = idx / 64, q = idx % 64;
    const long long r = r0 + q, k = k0 + q;
    vr[p][q] = r < n ? A[(long long)(j0 + p) * ld + r] : 0.0;
    yk[p][q] = k < n ? y[(long long)p * ldc + k] : 0.0;
  }
  __syncthreads();
  const int tr = (threadIdx.x & 15) * 4, tc = (threadIdx.x >> 4) * 4;
  double acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = 0.0;
  for (int p = 0; p < nb; ++p) {
    double av[4], bv[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      av[a] = vr[p][tr + a];
      bv[a] = yk[p][tc + a];
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = fma(av[a], bv[b], acc[a][b]);
  }
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const long long r = r0 + tr + a, k = k0 + tc + b;
      if (r < n && k < n) Z[k * ld + r] -= acc[a][b];
    }
}

// ---- the closing check and the hand-over --------------------------------------------------------------------------------
// kz = K Z (n x n, leading dimension ldk);  out[0] = max |kz[:, k] - lam[k] z[:, k]|  (bits of a non-negative double, atomicMax)
static __global__ __launch_bounds__(256) void td_residual_kernel(const double* __restrict__ kz, long long ldk,
                                                           const double* __restrict__ Z, long long ld, int n,
                                                           const double* __restrict__ lam, unsigned long long* __restrict__ out) {
  const long long k = blockIdx.x;
  double mx = 0.0;
  for (long long i = threadIdx.x; i < n; i += 256) mx = fmax(mx, fabs(kz[k * ldk + i] - lam[k] * Z[k * ld + i]));
  for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_down(mx, o));
  if ((threadIdx.x & 63) == 0) atomicMax(out, (unsigned long long)__double_as_longlong(mx));
}
// zz = Z'Z;  out[1] = max |zz - I|
// (zz holds the columns k0 .. k0 + gridDim.x - 1 of the product)
static __global__ __launch_bounds__(256) void td_orth_kernel(const double* __restrict__ zz, long long ldz, int n, int k0,
                                                       unsigned long long* __restrict__ out) {
  const long long k = blockIdx.x;
  double mx = 0.0;
  for (long long i = threadIdx.x; i < n; i += 256) mx = fmax(mx, fabs(zz[k * ldz + i] - (i == k + k0 ? 1.0 : 0.0)));
  for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_down(mx, o));
  if ((threadIdx.x & 63) == 0) atomicMax(out + 1, (unsigned long long)__double_as_longlong(mx));
}
// float U (n x n, leading dimension n) from Z
static __global__ void td_to_float_kernel(const double* __restrict__ Z, long long ld, long long n, float* __restrict__ U) {
  const long long total = n * n;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const long long k = idx / n, i = idx % n;
    U[idx] = (float)Z[k * ld + i];
  }
}

}  // namespace rvt
