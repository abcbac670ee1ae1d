// rvtests_amd — eigendecomposition of the kinship matrix on the device (SURVEY §8f "next" #3).
//
// Replaces KinshipHolder::decompose (base/KinshipHolder.cpp:270-290: Eigen::SelfAdjointEigenSolver<MatrixXf> on the N x N
// kinship; eigenvalues ascending in matS, eigenvectors in the columns of matU) with a ONE-SIDED BLOCK JACOBI iteration
// in fp64 — every O(N^3) step is a tall-skinny fp64 matrix-core product over contiguous column panels, and there is no
// sequential reduction to tridiagonal form:
//     W = K V,  V = I.   For a pair of column blocks (I, J), 32 columns each:
//        G = [W_I W_J]' [W_I W_J]                      (64 x 64 Gram matrix, jac_gram_kernel)
//        G = R D R'   by cyclic Jacobi in LDS           (jac_small_eig_kernel; eigen-columns ordered by decreasing D)
//        [W_I W_J] <- [W_I W_J] R,  [V_I V_J] <- [V_I V_J] R      (jac_apply_kernel)
//     nb / 2 disjoint pairs per round run side by side, nb - 1 rounds (round-robin tournament) make a sweep, sweeps
//     repeat until every cross-block cosine is below the tolerance.  Then the columns of W are orthogonal, K v_j = w_j is
//     parallel to v_j, and lambda_j = v_j' w_j.
// One-sided Jacobi orthogonalises the columns of K V, which pins the eigenvectors only when no two eigenvalues are
// (nearly) opposite; a positive semi-definite kinship has none.  The engine checks the residuals ||K v - lambda v|| at
// the end and, if one is large, repeats the iteration on K + shift I (shift above the spectral radius).
// The matrix order is padded to a multiple of 64 with diagonal entries -mu (|mu| above the spectral radius): those
// columns never mix with the others (their cross products are exactly zero and a zero off-diagonal is never rotated)
// and are dropped at the end.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rvt {

constexpr int kJacB = 32;        // columns per block
constexpr int kJacP = 2 * kJacB; // columns per pair
typedef double jd4_t __attribute__((ext_vector_type(4)));

// pair k (0 .. n/2-1) of round r (0 .. n-2) of the round-robin tournament of n players (n even)
__host__ __device__ inline void jac_rr_pair(int n, int r, int k, int* a, int* b) {
  if (k == 0) {
    *a = n - 1;
    *b = r;
  } else {
    *a = (r + k) % (n - 1);
    *b = (r - k + (n - 1)) % (n - 1);
  }
  if (*a > *b) {
    const int t = *a;
    *a = *b;
    *b = t;
  }
}

// float (column-major, n x n, leading dimension n) -> padded fp64 W (np x np, leading dimension np) with -mu on the
// pad diagonal; V = identity
#if !defined(RVT_K_SPLIT) || defined(RVT_K_FAM)
static __global__ void jac_init_kernel(const float* __restrict__ K, long long n, long long np, double mu, double shift,
                                double* __restrict__ W, double* __restrict__ V) {
  const long long total = np * np;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const long long j = idx / np, i = idx % np;
    double w = 0.0;
    if (i < n && j < n)
      w = (double)K[j * n + i] + (i == j ? shift : 0.0);
    else if (i == j)
      w = -mu;
    W[idx] = w;
    V[idx] = (i == j) ? 1.0 : 0.0;
  }
}
#endif  // RVT_K_FAM

// ---- Gram matrices of one round ---------------------------------------------------------------------------------------
// grid (pairs, splits), 256 threads: every wave accumulates the upper 16 x 16 tiles of G over its rows with
// v_mfma_f64_16x16x4_f64 (the register holding 16 columns x 4 rows is the A operand of a tile row and the B operand of a
// tile column) and writes its partial to part[pair][split * 4 + wave][64 x 64] (row-major, upper tiles only).
#if !defined(RVT_K_SPLIT) || defined(RVT_K_FAM)
static __global__ __launch_bounds__(256) void jac_gram_kernel(const double* __restrict__ W, long long np, int nb, int round,
                                                       int splits, double* __restrict__ part) {
  int bi, bj;
  jac_rr_pair(nb, round, blockIdx.x, &bi, &bj);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ci = lane & 15, kg = lane >> 4;
  const int nw = splits * 4, wid = blockIdx.y * 4 + wave;
  const long long steps = np / 16;  // 16 rows per step
  const long long s0 = steps * wid / nw, s1 = steps * (wid + 1) / nw;
  const double* col[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const long long c = (long long)(t < 2 ? bi : bj) * kJacB + 16 * (t & 1) + ci;
    col[t] = W + c * np + 4 * kg;
  }
  jd4_t acc[10];
#pragma unroll
  for (int t = 0; t < 10; ++t) acc[t] = jd4_t{0.0, 0.0, 0.0, 0.0};
  for (long long s = s0; s < s1; ++s) {
    double v[4][4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const double2 lo = *reinterpret_cast<const double2*>(col[t] + 16 * s);
      const double2 hi = *reinterpret_cast<const double2*>(col[t] + 16 * s + 2);
      v[t][0] = lo.x;
      v[t][1] = lo.y;
      v[t][2] = hi.x;
      v[t][3] = hi.y;
    }
    int q = 0;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = a; b < 4; ++b) {
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(v[a][m], v[b][m], acc[q], 0, 0, 0);
        ++q;
      }
  }
  double* out = part + ((long long)blockIdx.x * nw + wid) * (kJacP * kJacP);
  int q = 0;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = a; b < 4; ++b) {
#pragma unroll
      for (int r = 0; r < 4; ++r) out[(16 * a + kg + 4 * r) * kJacP + 16 * b + ci] = acc[q][r];
      ++q;
    }
}
#endif  // RVT_K_FAM

// ---- the 64 x 64 eigenproblem of every pair --------------------------------------------------------------------------
// grid (pairs), 256 threads.  Sums the partial Gram matrices, records the largest cross-block cosine (before rotating)
// into *maxcos (bits of a non-negative double, atomicMax), diagonalises G by cyclic Jacobi with the round-robin ordering
// (32 disjoint rotations per step) and writes R (row-major 64 x 64, columns ordered by decreasing eigenvalue).
#if !defined(RVT_K_SPLIT) || defined(RVT_K_FAM)
static __global__ __launch_bounds__(256) void jac_small_eig_kernel(const double* __restrict__ part, int nparts, double tol,
                                                            double* __restrict__ Rout,
                                                            unsigned long long* __restrict__ maxcos, int sort_mode,
                                                            double* __restrict__ lam_out = nullptr) {
  __shared__ double G[kJacP][kJacP + 1];
  __shared__ double R[kJacP][kJacP + 1];
  __shared__ double cs[kJacB][2];
  __shared__ int pq[kJacB][2];
  __shared__ int rotated;
  __shared__ double red[256];
  __shared__ int order[kJacP];
  const int tid = threadIdx.x;
  const double* src = part + (long long)blockIdx.x * nparts * (kJacP * kJacP);
  for (int e = tid; e < kJacP * kJacP; e += 256) {
    const int i = e / kJacP, j = e % kJacP;
    double s = 0.0;
    if ((j >> 4) >= (i >> 4)) {  // upper tiles were written
      for (int p = 0; p < nparts; ++p) s += src[(long long)p * (kJacP * kJacP) + e];
      G[i][j] = s;
      if ((j >> 4) > (i >> 4)) G[j][i] = s;
    }
    R[i][j] = (i == j) ? 1.0 : 0.0;
  }
  __syncthreads();
  // inside a diagonal tile both triangles were written by the matrix cores (equal up to rounding): symmetrise
  for (int e = tid; e < kJacP * kJacP; e += 256) {
    const int i = e / kJacP, j = e % kJacP;
    if (i < j && (i >> 4) == (j >> 4)) {
      const double m = 0.5 * (G[i][j] + G[j][i]);
      G[i][j] = m;
      G[j][i] = m;
    }
  }
  __syncthreads();
  {  // convergence measure: cosines between the two blocks
    double mx = 0.0;
    for (int e = tid; e < kJacB * kJacB; e += 256) {
      const int i = e / kJacB, j = kJacB + e % kJacB;
      const double d = G[i][i] * G[j][j];
      if (d > 0.0) mx = fmax(mx, fabs(G[i][j]) / sqrt(d));
    }
    red[tid] = mx;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
      if (tid < w) red[tid] = fmax(red[tid], red[tid + w]);
      __syncthreads();
    }
    if (tid == 0) atomicMax(maxcos, (unsigned long long)__double_as_longlong(red[0]));
  }
  for (int sweep = 0; sweep < 30; ++sweep) {
    if (tid == 0) rotated = 0;
    __syncthreads();
    for (int step = 0; step < kJacP - 1; ++step) {
      if (tid < kJacB) {
        int p, q;
        jac_rr_pair(kJacP, step, tid, &p, &q);
        pq[tid][0] = p;
        pq[tid][1] = q;
        const double gpq = G[p][q], gpp = G[p][p], gqq = G[q][q];
        double c = 1.0, s = 0.0;
        if (gpq != 0.0 && fabs(gpq) > tol * sqrt(fabs(gpp * gqq))) {
          const double theta = (gqq - gpp) / (2.0 * gpq);
          const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
          c = 1.0 / sqrt(t * t + 1.0);
          s = t * c;
          rotated = 1;
        }
        cs[tid][0] = c;
        cs[tid][1] = s;
      }
      __syncthreads();
      // columns p, q of G and of R
      for (int e = tid; e < kJacB * kJacP; e += 256) {
        const int k = e / kJacP, i = e % kJacP;
        const int p = pq[k][0], q = pq[k][1];
        const double c = cs[k][0], s = cs[k][1];
        if (s != 0.0) {
          const double gp = G[i][p], gq = G[i][q];
          G[i][p] = c * gp - s * gq;
          G[i][q] = s * gp + c * gq;
          const double rp = R[i][p], rq = R[i][q];
          R[i][p] = c * rp - s * rq;
          R[i][q] = s * rp + c * rq;
        }
      }
      __syncthreads();
      // rows p, q of G
      for (int e = tid; e < kJacB * kJacP; e += 256) {
        const int k = e / kJacP, j = e % kJacP;
        const int p = pq[k][0], q = pq[k][1];
        const double c = cs[k][0], s = cs[k][1];
        if (s != 0.0) {
          const double gp = G[p][j], gq = G[q][j];
          G[p][j] = c * gp - s * gq;
          G[q][j] = s * gp + c * gq;
        }
      }
      __syncthreads();
    }
    if (!rotated) break;
    __syncthreads();
  }
  // order the eigen-columns by decreasing eigenvalue (rank by counting; ties by index)
  if (tid < kJacP) {
    const double d = G[tid][tid];
    int rank = 0;
    for (int j = 0; j < kJacP; ++j) {
      const double e = G[j][j];
      if (e > d || (e == d && j < tid)) ++rank;
    }
    if (sort_mode == 0) rank = tid;
    order[rank] = tid;
    if (lam_out) lam_out[(long long)blockIdx.x * kJacP + rank] = d;  // eigenvalue of output column `rank`
  }
  __syncthreads();
  double* out = Rout + (long long)blockIdx.x * (kJacP * kJacP);
  for (int e = tid; e < kJacP * kJacP; e += 256) {
    const int i = e / kJacP, j = e % kJacP;
    out[e] = R[i][order[j]];
  }
}
#endif  // RVT_K_FAM

// ---- [X_I X_J] <- [X_I X_J] R for X = W and X = V ----------------------------------------------------------------------
// grid (pairs, slabs, 2), 256 threads; blockIdx.z selects W or V.  Transposed formulation C' = R' X': the A operand is R'
// (held in registers for the lifetime of the wave), the B operand a 16-row slab of X (16 consecutive rows of one
// column per 16 lanes: whole 128-byte lines), and the result tile stores 16 consecutive rows per column.
#if !defined(RVT_K_SPLIT) || defined(RVT_K_FAM)
static __global__ __launch_bounds__(256) void jac_apply_kernel(double* __restrict__ W, double* __restrict__ V, long long np,
                                                        int nb, int round, const double* __restrict__ Rall) {
  int bi, bj;
  jac_rr_pair(nb, round, blockIdx.x, &bi, &bj);
  double* X = blockIdx.z ? V : W;
  const double* R = Rall + (long long)blockIdx.x * (kJacP * kJacP);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, kg = lane >> 4;
  // A[i][k] of (output tile ti, k chunk kc) = R[4 kc + k][16 ti + i]
  double ra[4][16];
#pragma unroll
  for (int ti = 0; ti < 4; ++ti)
#pragma unroll
    for (int kc = 0; kc < 16; ++kc) ra[ti][kc] = R[(4 * kc + kg) * kJacP + 16 * ti + li];
  const long long groups = np / 16;
  const int nwaves = gridDim.y * 4;
  for (long long g = (long long)blockIdx.y * 4 + wave; g < groups; g += nwaves) {
    const long long r0 = 16 * g;
    double xb[16];  // B[k][j] of k chunk kc = X[r0 + j, column 4 kc + k of the pair]
#pragma unroll
    for (int kc = 0; kc < 16; ++kc) {
      const int cp = 4 * kc + kg;  // column inside the pair
      const long long c = (long long)(cp < kJacB ? bi : bj) * kJacB + (cp & (kJacB - 1));
      xb[kc] = X[c * np + r0 + li];
    }
#pragma unroll
    for (int ti = 0; ti < 4; ++ti) {
      jd4_t acc = jd4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int kc = 0; kc < 16; ++kc) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ra[ti][kc], xb[kc], acc, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int cp = 16 * ti + kg + 4 * r;  // output column inside the pair
        const long long c = (long long)(cp < kJacB ? bi : bj) * kJacB + (cp & (kJacB - 1));
        X[c * np + r0 + li] = acc[r];
      }
    }
  }
}
#endif  // RVT_K_FAM

// lambda_j = v_j' w_j and the residual ||w_j - lambda_j v_j||; one workgroup per column
#if !defined(RVT_K_SPLIT) || defined(RVT_K_FAM)
static __global__ __launch_bounds__(256) void jac_lambda_kernel(const double* __restrict__ W, const double* __restrict__ V,
                                                         long long np, double* __restrict__ lambda,
                                                         double* __restrict__ resid) {
  __shared__ double red[256];
  const double* w = W + (long long)blockIdx.x * np;
  const double* v = V + (long long)blockIdx.x * np;
  double s = 0.0;
  for (long long i = threadIdx.x; i < np; i += 256) s += w[i] * v[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if (threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
    __syncthreads();
  }
  const double lam = red[0];
  __syncthreads();
  s = 0.0;
  for (long long i = threadIdx.x; i < np; i += 256) {
    const double d = w[i] - lam * v[i];
    s += d * d;
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if (threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    lambda[blockIdx.x] = lam;
    resid[blockIdx.x] = sqrt(red[0]);
  }
}
#endif  // RVT_K_FAM

// U (float, n x n, column-major, leading dimension n): column j = column src[j] of V (first n rows), normalised
#if !defined(RVT_K_SPLIT) || defined(RVT_K_FAM)
static __global__ __launch_bounds__(256) void jac_gather_kernel(const double* __restrict__ V, long long np, long long n,
                                                         const int* __restrict__ src, float* __restrict__ U) {
  __shared__ double red[256];
  const double* v = V + (long long)src[blockIdx.x] * np;
  double s = 0.0;
  for (long long i = threadIdx.x; i < n; i += 256) s += v[i] * v[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if (threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
    __syncthreads();
  }
  const double inv = red[0] > 0.0 ? 1.0 / sqrt(red[0]) : 0.0;
  float* u = U + (long long)blockIdx.x * n;
  for (long long i = threadIdx.x; i < n; i += 256) u[i] = (float)(v[i] * inv);
}
#endif  // RVT_K_FAM

// Family-wise decomposition (rvt_kinship_decompose on a kinship of separate families): column k of U (float, n x n,
// column-major, pre-cleared) gets the `len[k]` entries of eigenvector column `col[k]` of tile `tile[k]` (R: [tile][64 x 64]
// row-major) at the sample rows of that tile (rows: [tile][64]); one thread per column.
#if !defined(RVT_K_SPLIT) || defined(RVT_K_FAM)
static __global__ void jac_scatter_blocks_kernel(const double* __restrict__ R, const int* __restrict__ tile,
                                          const int* __restrict__ col, const int* __restrict__ len,
                                          const int* __restrict__ rows, long long n, float* __restrict__ U) {
  const long long k = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (k >= n) return;
  const double* r = R + (long long)tile[k] * (kJacP * kJacP) + col[k];
  const int* rw = rows + (long long)tile[k] * kJacP;
  float* u = U + k * n;
  for (int i = 0; i < len[k]; ++i) u[rw[i]] = (float)r[(long long)i * kJacP];
}
#endif  // RVT_K_FAM

}  // namespace rvt
