// rvtests_amd — the rows of MetaCov's band behind the tile products of band_gemm.hip.h / gemm_f64.hip.h: the centring and
// covariate algebra of computeScaledXX (src/Model.h:3997-4005) in the layout the adapter prints from.
// Include after kernels.hip.h (CovConsts).
#pragma once
#include "band_gemm.hip.h"

namespace rvt {

// One workgroup per head h of the pass: S(h, h + t), t = 0 .. halo, from the partial tiles; then the algebra of
// cov_rect_rows_kernel (the same expressions in the same order: the rows are bit-identical to rvt_cov_rect's).
// cs / xz: column sums and covXZ rows of the pass's columns (index 0 = the pass's first head).
// band_f32 (optional): (float)value * scale, the number the adapter prints with %g (src/Model.cpp:975-984 casts to float and
// divides by N in float); band_f64 (optional): the value itself.  Entries beyond the window (h + t >= W) are NaN.
// mu != null (round 6): some columns are hard calls plus ONE other value (g = h + mu m, mu[j] = 0 for a column without one);
// `part` then holds FOUR sets of partial tiles, set_stride ints apart — h'h, h'm, m'h, m'm (head side first) — and
//   S(h, j) = h_h'h_j + mu_j h_h'm_j + mu_h m_h'h_j + mu_h mu_j m_h'm_j,
// four exact integers combined in fp64 (three roundings; the fp64 product of the doubles themselves rounds N times).
__global__ __launch_bounds__(256) void band_finish_i32_kernel(CovConsts cc, const int* __restrict__ part, int n_slices,
                                                              int n_tiles, const double* __restrict__ cs,
                                                              const double* __restrict__ xz, int H, int W, int halo,
                                                              float scale, float* __restrict__ band_f32,
                                                              double* __restrict__ band_f64,
                                                              const double* __restrict__ mu = nullptr,
                                                              long long set_stride = 0) {
  const int h = blockIdx.x, d = cc.d;
  __shared__ double a[RVT_MAX_COV];
  __shared__ int tile0;
  if (threadIdx.x < d) {
    double t = 0.0;
    for (int k = 0; k < d; ++k) t += xz[(long long)h * d + k] * cc.zzinv[k * d + threadIdx.x];
    a[threadIdx.x] = t;
  }
  if (threadIdx.x == 64) tile0 = band_tile_of(h, h, W, halo);  // (the panel's diagonal tile; marker j adds (j >> 8) - (h >> 8))
  __syncthreads();
  const double sh = cs[h];
  const long long row = (long long)h * (halo + 1);
  // a thread takes FOUR consecutive markers j0 .. j0 + 3, j0 a multiple of 4 (one 16-byte load per slice; 256 is a multiple of
  // 4, so the four lie in one tile), of which those inside [h, h + halo] are written
  const int jbeg = h & ~3, ngroups = (h + halo - jbeg) / 4 + 1;
  for (int g = threadIdx.x; g < ngroups; g += blockDim.x) {
    const int j0 = jbeg + 4 * g;
    long long s[4] = {0, 0, 0, 0};
    double sm[4] = {0.0, 0.0, 0.0, 0.0};  // (masked columns: the three mixed products, already weighted by the mu's)
    if (j0 < W) {
      const int tile = tile0 + (j0 >> 8) - (h >> 8);
      const i4v_t* p = reinterpret_cast<const i4v_t*>(part + ((long long)tile << 16) + (h & 255) * kBandBT + (j0 & 255));
      for (int sl = 0; sl < n_slices; ++sl) {
        const i4v_t v = p[((long long)sl * n_tiles) << 14];
        s[0] += v[0];
        s[1] += v[1];
        s[2] += v[2];
        s[3] += v[3];
      }
      if (mu) {
        long long hm[4] = {0, 0, 0, 0}, mh[4] = {0, 0, 0, 0}, mm[4] = {0, 0, 0, 0};
        const i4v_t* p1 = reinterpret_cast<const i4v_t*>(reinterpret_cast<const int*>(p) + set_stride);
        const i4v_t* p2 = reinterpret_cast<const i4v_t*>(reinterpret_cast<const int*>(p) + 2 * set_stride);
        const i4v_t* p3 = reinterpret_cast<const i4v_t*>(reinterpret_cast<const int*>(p) + 3 * set_stride);
        for (int sl = 0; sl < n_slices; ++sl) {
          const long long o = ((long long)sl * n_tiles) << 14;
          const i4v_t a1 = p1[o], a2 = p2[o], a3 = p3[o];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            hm[e] += a1[e];
            mh[e] += a2[e];
            mm[e] += a3[e];
          }
        }
        const double muh = mu[h];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const double muj = (j0 + e < W) ? mu[j0 + e] : 0.0;
          sm[e] = muj * (double)hm[e] + muh * (double)mh[e] + (muh * muj) * (double)mm[e];
        }
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int j = j0 + e, t = j - h;
      if (t < 0 || t > halo) continue;
      double v = NAN;
      if (j < W) {
        const double sxx = mu ? (double)s[e] + sm[e] : (double)s[e];
        const double xx = cc.binary ? sxx : (sxx - sh * cs[j] * cc.inv_n) * cc.inv_sigma2;
        double quad = 0.0;
        for (int k = 0; k < d; ++k) quad += a[k] * xz[(long long)j * d + k];
        v = xx - quad;
      }
      if (band_f32) band_f32[row + t] = (float)v * scale;
      if (band_f64) band_f64[row + t] = v;
    }
  }
}

// The same band from a rectangle of doubles S (H x Wd, column-major, leading dimension lds: the fp64 matrix cores' product
// for dosages / a binary trait, or the rotated product of the family model).  fam: the centring algebra of
// cov_rect_fam_rows_kernel (t1 = G~' D u1).  b2: MetaCovFamBinary's factor on covXX (1 otherwise).
__global__ __launch_bounds__(256) void band_rows_f64_kernel(CovConsts cc, const double* __restrict__ S, long long lds,
                                                            const double* __restrict__ cs, const double* __restrict__ xz,
                                                            const double* __restrict__ t1, int H, int W, int halo, double b2,
                                                            float scale, float* __restrict__ band_f32,
                                                            double* __restrict__ band_f64) {
  const int h = blockIdx.x, d = cc.d;
  __shared__ double a[RVT_MAX_COV];
  if (threadIdx.x < d) {
    double t = 0.0;
    for (int k = 0; k < d; ++k) t += xz[(long long)h * d + k] * cc.zzinv[k * d + threadIdx.x];
    a[threadIdx.x] = t;
  }
  __syncthreads();
  const double sh = cs[h];
  const double mh = sh * cc.inv_n, t1h = t1 ? t1[h] : 0.0;
  const long long row = (long long)h * (halo + 1);
  for (int t = threadIdx.x; t <= halo; t += blockDim.x) {
    const int j = h + t;
    double v = NAN;
    if (j < W) {
      const double sxx = S[h + (long long)j * lds];
      double xx;
      if (t1) {
        const double mj = cs[j] * cc.inv_n;
        xx = sxx - mh * t1[j] - mj * t1h + mh * mj * cc.c11;
      } else {
        xx = cc.binary ? sxx : (sxx - sh * cs[j] * cc.inv_n) * cc.inv_sigma2;
      }
      double quad = 0.0;
      for (int k = 0; k < d; ++k) quad += a[k] * xz[(long long)j * d + k];
      v = xx - quad;
      if (b2 != 1.0) v *= b2;
    }
    if (band_f32) band_f32[row + t] = (float)v * scale;
    if (band_f64) band_f64[row + t] = v;
  }
}

// the column statistics a ring keeps per PHYSICAL column -> the work arrays of a band call: logical column j = physical
// (col0 + j) mod ring
__global__ void band_cache_gather_kernel(const double* __restrict__ cs_c, const int* __restrict__ poly_c,
                                         const double* __restrict__ T_c, int ring, int col0, int W, int d, int t_stride,
                                         double* __restrict__ colsum, int* __restrict__ poly, double* __restrict__ T,
                                         const double* __restrict__ mu_c = nullptr, double* __restrict__ mu = nullptr) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= W) return;
  long long p = (long long)col0 + j;
  if (ring > 0 && p >= ring) p -= ring;
  colsum[j] = cs_c[p];
  poly[j] = poly_c[p];
  if (mu) mu[j] = mu_c[p];
  for (int k = 0; k < d; ++k) T[j + (long long)k * W] = T_c[p * t_stride + k];
}

// Columns that crossed PCIe as PLINK 2-bit codes (host_stage.h pack_column_f64: 00 -> 0, 10 -> 1, 11 -> 2, 01 -> the column's one
// other value mu[j], e.g. an imputed mean) back to the doubles of the block: rows [column j][pitch bytes], sample i in bits
// 2 (i & 3) of byte i >> 2.  grid (ceil(N / 1024), columns), 256 threads, four samples per thread.
__global__ __launch_bounds__(256) void bed_expand_columns_kernel(const unsigned char* __restrict__ rows, long long pitch,
                                                                 const double* __restrict__ mu, long long N, long long ld,
                                                                 double* __restrict__ G) {
  const long long b = (long long)blockIdx.x * 256 + threadIdx.x, i = 4 * b;
  if (i >= N) return;
  const int j = blockIdx.y;
  const unsigned code = rows[(long long)j * pitch + b];
  const double m = mu[j];
  double* g = G + (long long)j * ld + i;
  double v[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const unsigned q = (code >> (2 * e)) & 3u;
    v[e] = q == 0u ? 0.0 : (q == 2u ? 1.0 : (q == 3u ? 2.0 : m));
  }
  if (i + 4 <= N) {
    *reinterpret_cast<double2*>(g) = double2{v[0], v[1]};
    *reinterpret_cast<double2*>(g + 2) = double2{v[2], v[3]};
  } else {
    for (int e = 0; e < 4 && i + e < N; ++e) g[e] = v[e];
  }
}

}  // namespace rvt
