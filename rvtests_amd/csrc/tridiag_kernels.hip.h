// rvtests_amd — eigendecomposition of a DENSE kinship (a genetic relationship matrix) by Householder tridiagonalisation,
// bisection and inverse iteration (round 5).
//
// Replaces, for dense matrices, KinshipHolder::decompose (base/KinshipHolder.cpp:270-290: Eigen::SelfAdjointEigenSolver on the
// N x N kinship; eigenvalues ascending in matS, eigenvectors in the columns of matU).  The one-sided block Jacobi iteration of
// jacobi_kernels.hip.h needs 16-30 sweeps on a gap-free spectrum, every sweep N^3 bytes of HBM traffic: 44 s at N = 12 000.
// Here:
//   1. K = Q T Q'.  Blocked Householder tridiagonalisation (LAPACK's dsytrd / dlatrd scheme, panels of kTdNb columns).  The
//      O(N^2) product of every column reads the LOWER triangle of the trailing matrix once, tile by tile (td_symv_kernel:
//      4 N^3 / 3 bytes of traffic in all — the HBM-bound part), the rank-2nb updates touch the lower triangle only.  The
//      reflector v_j overwrites column j of the matrix.
//   2. The eigenvalues of T by the Sturm bisection + interpolation of rvt_coop.h, one thread per eigenvalue.
//   3. The eigenvectors of T by inverse iteration, one thread per eigenvector (tridiagonal LU with partial pivoting, three
//      solves), WITHOUT reorthogonalisation: the vectors of eigenvalues a gap g apart are orthogonal to ~eps / g, and the
//      boundary stores U as FLOAT (EigenMatrix = Eigen::MatrixXf) — a gap of 4e-9 of the spectrum's width is enough.  A matrix
//      with a tighter cluster (repeated eigenvalues: pedigree kinships, rank-deficient matrices) is left to the Jacobi
//      iteration, which does not care; so is a result that fails the residual / orthogonality check that closes the procedure.
//   4. U = Q Z: the reflectors applied 256 at a time in compact WY form, Z <- Z - V (T (V'Z)); V'Z, V'V and the rank-256 update
//      are the fp64 matrix-core product of gemm_f64.hip.h, T is applied as a back substitution with T^-1 = triu(V'V, 1) + diag(1 / tau).
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
// td_col_update_kernel: 256 rows per workgroup; ss[blockIdx.x] = the workgroup's share of sum_{r > j + 1} a[r]^2.
static __global__ __launch_bounds__(256) void td_col_update_kernel(double* __restrict__ A, long long n, long long ld, int j, int j0,
                                                             const double* __restrict__ W, double* __restrict__ ss) {
  __shared__ double red[4];
  __shared__ double wj[kTdNb], vj[kTdNb];
  const int i = j - j0, tid = threadIdx.x;
  double* a = A + (long long)j * ld;
  if (tid < i) {
    wj[tid] = W[(long long)tid * ld + j];
    vj[tid] = A[(long long)(j0 + tid) * ld + j];
  }
  __syncthreads();
  const long long r = j + (long long)blockIdx.x * 256 + tid;
  double sq = 0.0;
  if (r < n) {
    double v = a[r];
    for (int p = 0; p < i; ++p) v -= A[(long long)(j0 + p) * ld + r] * wj[p] + W[(long long)p * ld + r] * vj[p];
    a[r] = v;
    if (r > j + 1) sq = v * v;
  }
  const double tot = td_block_sum(sq, red);
  if (tid == 0) ss[blockIdx.x] = tot;
}
// td_col_house_kernel (ONE workgroup of 1024 threads, behind td_col_update_kernel's n_ss workgroups):
//   d[j] = a[j];  reflector of x = a[j + 1 .. n): beta = -sign(x0) |x|, tau = (beta - x0) / beta, v = [1; x[1:] / (x0 - beta)]
//   e[j] = beta;  column j of A becomes v (zeros in rows 0 .. j)
static __global__ __launch_bounds__(1024) void td_col_house_kernel(double* __restrict__ A, long long n, long long ld, int j,
                                                            const double* __restrict__ ss, int n_ss, double* __restrict__ dvec,
                                                            double* __restrict__ evec, double* __restrict__ tau) {
  const int tid = threadIdx.x;
  double* a = A + (long long)j * ld;
  double xss = 0.0;
  for (int k = 0; k < n_ss; ++k) xss += ss[k];  // fixed order, every thread the same
  const double ajj = a[j], x0 = (j + 1 < n) ? a[j + 1] : 0.0;
  double beta = x0, t = 0.0, scale = 0.0;
  if (xss > 0.0) {
    const double nrm = sqrt(x0 * x0 + xss);
    beta = (x0 >= 0.0) ? -nrm : nrm;
    t = (beta - x0) / beta;
    scale = 1.0 / (x0 - beta);
  }
  __syncthreads();  // (every thread has read a[j], a[j + 1])
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

// y = A22 v (v = column j of A, zero in rows 0 .. j) from the LOWER triangle of the trailing matrix only — the O(N^2) half of
// the reduction is a pass over HBM, and reading one triangle halves it.  The matrix is cut into 64 x 64 tiles on absolute
// multiples of 64; a workgroup takes ONE tile (R, C), R >= C >= B0 = (j + 1) / 64, through LDS and produces
//   the tile's share of the rows    sum_c A[r, c] v[c]   (c <= r inside a diagonal tile)   -> P[C][r]
//   the tile's share of the columns sum_r A[r, c] v[r]   (r >  c inside a diagonal tile)   -> P[R + 1][c]
// (rows / columns <= j contribute nothing: v is zero there).  For an index i of block B the slots B0 .. B come from the row
// shares and B + 1 .. nblk from the column shares: every slot B0 .. nblk of P[.][i] is written in every step, and
// td_w_comb_kernel adds them in slot order.  P: (nblk + 1) x ld.
// The same launch computes the panel's dot products  t1[p] = W[:, p]' v,  t2[p] = V[:, p]' v  (p < i): workgroups behind the tiles.
constexpr int kSyT = 64, kSyRun = 4;
static __global__ __launch_bounds__(256) void td_symv_kernel(const double* __restrict__ A, long long n, long long ld, int j, int j0,
                                                       int nbt, const double* __restrict__ W,
                                                       double* __restrict__ P, double* __restrict__ t12) {
  __shared__ double tile[kSyT][kSyT + 1];
  __shared__ double vr[kSyT], vc[kSyT];
  __shared__ double red[4];
  const double* v = A + (long long)j * ld;
  const int tid = threadIdx.x;
  if ((int)blockIdx.y >= nbt) {  // a column of the panel
    const int i = j - j0, q = ((int)blockIdx.y - nbt) * (int)gridDim.x + (int)blockIdx.x;
    if (q >= 2 * i) return;
    const double* col = q < i ? W + (long long)q * ld : A + (long long)(j0 + q - i) * ld;
    double sacc = 0.0;
    for (long long r = j + 1 + tid; r < n; r += 256) sacc = fma(col[r], v[r], sacc);
    const double tot = td_block_sum(sacc, red);
    if (tid == 0) t12[q < i ? q : kTdNb + (q - i)] = tot;
    return;
  }
  // A workgroup walks kSyRun tiles DOWN one column strip (C fixed, R = C + kSyRun g ...): the loads of the next tile are in
  // flight while the shares of this one are computed.  grid (groups, strips): strip C holds nbt - C tiles.
  const int C = blockIdx.y, L = nbt - C, first = kSyRun * (int)blockIdx.x;
  if (first >= L) return;
  const int last = (first + kSyRun < L ? first + kSyRun : L) - 1;
  const int B0 = (j + 1) / kSyT, Cb = B0 + C;
  const long long c0 = (long long)Cb * kSyT;
  const int rp = 2 * (tid & 31), cq = tid >> 5;  // 32 row pairs x 8 columns per pass
  double2 buf[8];
  auto load = [&](int R) {  // (ld is a multiple of 64: the pair behind the last row reads zero pad rows)
    const long long r0 = (long long)(B0 + R) * kSyT;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int c = cq + 8 * k;
      buf[k] = double2{0.0, 0.0};
      if (c0 + c < n) buf[k] = *reinterpret_cast<const double2*>(A + (c0 + c) * ld + r0 + rp);
    }
  };
  load(C + first);
  if (tid < kSyT) vc[tid] = (c0 + tid < n) ? v[c0 + tid] : 0.0;
  for (int t = first; t <= last; ++t) {
    const int R = C + t, Rb = B0 + R;
    const long long r0 = (long long)Rb * kSyT;
    if (tid < kSyT) vr[tid] = (r0 + tid < n) ? v[r0 + tid] : 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      tile[rp][cq + 8 * k] = buf[k].x;
      tile[rp + 1][cq + 8 * k] = buf[k].y;
    }
    __syncthreads();
    if (t < last) load(R + 1);
    const bool diag = R == C;
    if (tid < kSyT) {
      const int r = tid, cend = diag ? r + 1 : kSyT;
      double sacc = 0.0;
      for (int c = 0; c < cend; ++c) sacc = fma(tile[r][c], vc[c], sacc);
      if (r0 + r < n) P[(long long)Cb * ld + r0 + r] = sacc;
    } else if (tid < 2 * kSyT) {
      const int c = tid - kSyT, rbeg = diag ? c + 1 : 0;
      double sacc = 0.0;
      for (int r = rbeg; r < kSyT; ++r) sacc = fma(tile[r][c], vr[r], sacc);
      if (c0 + c < n) P[(long long)(Rb + 1) * ld + c0 + c] = sacc;
    }
    __syncthreads();
  }
}

// w'[r] = tau (y[r] - sum_p V[r, p] t1[p] + W[r, p] t2[p]),  r > j; per-workgroup partial of w' . v into part[blockIdx.x].
// y[r] = the sum of the slots B0 .. nblk of P[.][r] (td_symv_kernel).  64 rows per workgroup: wave q adds the slots B0 + q,
// B0 + q + 4, ... of its 64 rows, wave 0 adds the four sums in wave order and finishes the row.
static __global__ __launch_bounds__(256) void td_w_comb_kernel(const double* __restrict__ A, long long n, long long ld, int j, int j0,
                                                         double* __restrict__ W, const double* __restrict__ P, int nblk,
                                                         const double* __restrict__ t12, const double* __restrict__ tau,
                                                         double* __restrict__ part) {
  __shared__ double quarter[4][64];
  __shared__ double t1[kTdNb], t2[kTdNb];
  const int i = j - j0, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if ((int)threadIdx.x < i) {
    t1[threadIdx.x] = t12[threadIdx.x];
    t2[threadIdx.x] = t12[kTdNb + threadIdx.x];
  }
  const long long r = (long long)blockIdx.x * 64 + lane;
  double sacc = 0.0;
  if (r > j && r < n)
    for (int bq = (j + 1) / kSyT + wave; bq <= nblk; bq += 4) sacc += P[(long long)bq * ld + r];
  quarter[wave][lane] = sacc;
  __syncthreads();
  if (wave != 0) return;
  const double tj = tau[j];
  const double* v = A + (long long)j * ld;
  double val = 0.0, dot = 0.0;
  if (r < n) {
    if (r > j) {
      val = ((quarter[0][lane] + quarter[1][lane]) + quarter[2][lane]) + quarter[3][lane];
      for (int p = 0; p < i; ++p) val -= A[(long long)(j0 + p) * ld + r] * t1[p] + W[(long long)p * ld + r] * t2[p];
      val *= tj;
      dot = val * v[r];
    }
    W[(long long)i * ld + r] = val;
  }
  for (int o = 32; o > 0; o >>= 1) dot += __shfl_down(dot, o);
  if (lane == 0) part[blockIdx.x] = dot;
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

// A[r, c] -= sum_p V[r, p] W[c, p] + W[r, p] V[c, p]   for r >= c >= j1 (the columns behind the panel j0 .. j1 - 1; tiles that
// meet the lower triangle)
// grid (tiles, tiles) of 64 x 64 outputs, 256 threads (4 x 4 outputs each); the panel rows of both tiles in LDS
static __global__ __launch_bounds__(256) void td_rank2k_kernel(double* __restrict__ A, long long n, long long ld, int j0, int j1,
                                                         const double* __restrict__ W) {
  __shared__ double vr[kTdNb][64 + 1], wr[kTdNb][64 + 1], vc[kTdNb][64 + 1], wc[kTdNb][64 + 1];
  if (blockIdx.x < blockIdx.y) return;  // (only the lower triangle of the trailing matrix is read afterwards: td_symv_kernel)
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
// (lam: the nv eigenvalues of this launch, a batch of the spectrum)
static __global__ __launch_bounds__(64) void td_invit_kernel(const double* __restrict__ d, const double* __restrict__ e, int n,
                                                      const double* __restrict__ lam, int nv, long long k_first, long long nk, double pert,
                                                      double* __restrict__ ud, double* __restrict__ uu, double* __restrict__ uw,
                                                      double* __restrict__ zt) {
  const long long k = (long long)blockIdx.x * 64 + threadIdx.x;
  if (k >= nv) return;
  const double l = lam[k];
  // start vector: deterministic, no zero component, not aligned with anything in particular
  for (int i = 0; i < n; ++i) {
    unsigned h = (unsigned)i * 2654435761u ^ ((unsigned)(k_first + k) * 40503u + 0x9E3779B9u);  // (the eigenvector's number in the spectrum: batch sizes do not change the result)
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

// zc[i + k * ld] = zt[i * nk + k]   for i < n, k < nv   (32 x 32 tiles through LDS; grid (n / 32, nv / 32))
static __global__ __launch_bounds__(256) void td_transpose_kernel(const double* __restrict__ zt, long long nk, int n, int nv,
                                                            long long ld, double* __restrict__ zc) {
  __shared__ double tile[32][33];
  const long long i0 = (long long)blockIdx.x * 32, k0 = (long long)blockIdx.y * 32;
  for (int t = threadIdx.x; t < 1024; t += 256) {
    const int a = t / 32, b = t % 32;  // a: row i, b: column k (k fastest in zt)
    const long long i = i0 + a, k = k0 + b;
    tile[a][b] = (i < n && k < nv) ? zt[i * nk + k] : 0.0;
  }
  __syncthreads();
  for (int t = threadIdx.x; t < 1024; t += 256) {
    const int b = t / 32, a = t % 32;  // a fastest: row i contiguous in zc
    const long long i = i0 + a, k = k0 + b;
    if (i < n && k < nv) zc[k * ld + i] = tile[a][b];
  }
}

// ---- back-transformation U = Q Z: kTdNbb reflectors at a time, every O(N^2 nb) product on the matrix cores -----------------
// The block reflector of a panel is I - V T V' with T^-1 = triu(V'V, 1) + diag(1 / tau) (from T^-1 + T^-T = V'V: the product is
// orthogonal), so T is never formed:  Z <- Z - V Y',  Y' = T (V'Z)  <=>  T^-1 Y' = V'Z, a back substitution per column of Z.
constexpr int kTdNbb = 256;

// One thread per column k of Z:  y_p = tau_p (c_p - sum_{p' > p} G[p, p'] y_p'),  c = cz[k, :] (cz: n x nb, leading dimension
// ldc; overwritten), G = V'V (column-major, leading dimension kTdNbb, entries above the diagonal), in blocks of 32 from the
// last: a block is solved in registers, stored, and subtracted from the c's in front of it.  yt[p + k kTdNbb] (p >= nb: 0).
static __global__ __launch_bounds__(64) void td_rsolve_kernel(double* __restrict__ cz, long long ldc, int n, int nb,
                                                        const double* __restrict__ G, const double* __restrict__ tau,
                                                        double* __restrict__ yt) {
  const long long k = (long long)blockIdx.x * 64 + threadIdx.x;
  if (k >= n) return;
  double* yk = yt + k * kTdNbb;
  for (int p = nb; p < kTdNbb; ++p) yk[p] = 0.0;
  for (int p0 = ((nb - 1) / 32) * 32; p0 >= 0; p0 -= 32) {
    double y[32];
#pragma unroll
    for (int q = 0; q < 32; ++q) y[q] = (p0 + q < nb) ? cz[k + (long long)(p0 + q) * ldc] : 0.0;
#pragma unroll
    for (int q = 31; q >= 0; --q) {
      const int p = p0 + q;
      if (p < nb) {  // (uniform)
        double sacc = y[q];
#pragma unroll
        for (int q2 = q + 1; q2 < 32; ++q2) sacc = fma(-G[p + (long long)(p0 + q2) * kTdNbb], y[q2], sacc);  // (y = 0 beyond nb)
        y[q] = sacc * tau[p];
      }
    }
#pragma unroll
    for (int q = 0; q < 32; ++q)
      if (p0 + q < nb) yk[p0 + q] = y[q];
    for (int i = 0; i < p0; ++i) {
      double sacc = cz[k + (long long)i * ldc];
#pragma unroll
      for (int q = 0; q < 32; ++q) sacc = fma(-G[i + (long long)(p0 + q) * kTdNbb], y[q], sacc);
      cz[k + (long long)i * ldc] = sacc;
    }
  }
}
// vt[p + r kTdNbb] = V[r, p] = A[r + (j0 + p) ld]  (p >= nb: 0);  32 x 32 tiles through LDS.  grid (rows / 32, kTdNbb / 32)
static __global__ __launch_bounds__(256) void td_panel_transpose_kernel(const double* __restrict__ A, long long n, long long ld,
                                                                  int j0, int nb, double* __restrict__ vt) {
  __shared__ double tile[32][33];
  const long long r0 = (long long)blockIdx.x * 32;
  const int p0 = blockIdx.y * 32;
  for (int t = threadIdx.x; t < 1024; t += 256) {
    const int pp = t / 32, rr = t % 32;  // rows fastest: contiguous in A
    const long long r = r0 + rr;
    const int p = p0 + pp;
    tile[pp][rr] = (r < n && p < nb) ? A[(long long)(j0 + p) * ld + r] : 0.0;
  }
  __syncthreads();
  for (int t = threadIdx.x; t < 1024; t += 256) {
    const int rr = t / 32, pp = t % 32;  // p fastest: contiguous in vt
    const long long r = r0 + rr;
    if (r < n) vt[r * kTdNbb + p0 + pp] = tile[pp][rr];
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
