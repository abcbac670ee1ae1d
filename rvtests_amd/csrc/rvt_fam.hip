// rvtests_amd — related samples: the kinship (installation, family structure, KinshipHolder::decompose on the device), the
// exact integer rotation G~ = U'G (rot_gemm.hip.h), the FastLMM null model, FamSKAT / the family burden tests / the family
// forms of MetaCov and MetaScore.  Part of librvtests_amd.so; the per-gene pipeline behind it is rvt_engine.hip's.
// this unit compiles (and ships) the FAM kernel family only: see "kernel families" in rvt_engine_int.h
#define RVT_K_SPLIT
#define RVT_K_FAM
#include "rvt_engine_int.h"
#include "tridiag_kernels.hip.h"

extern "C" {

int ensure_fam_cols(rvt_ctx* c, size_t T, int64_t ld) {
  // (the capacity is columns OF THIS LEADING DIMENSION: a context whose null model was replaced by one with more samples
  //  otherwise kept buffers of the old column length — a memory fault in the permutation stage, found in round 6)
  if (T <= c->fam_cols_cap && ld <= c->fam_cols_ld) return RVT_OK;
  if (c->d_Gp) hipFree(c->d_Gp);
  if (c->d_Gt) hipFree(c->d_Gt);
  c->d_Gp = c->d_Gt = nullptr;
  const size_t want = std::max(T + T / 4, c->fam_cols_cap);
  c->fam_cols_cap = 0;
  c->fam_cols_ld = 0;
  HIP_TRY(c, hipMalloc((void**)&c->d_Gp, sizeof(double) * (size_t)ld * want));
  HIP_TRY(c, hipMalloc((void**)&c->d_Gt, sizeof(double) * (size_t)ld * want));
  c->fam_cols_cap = want;
  c->fam_cols_ld = ld;
  return RVT_OK;
}


// ---- related samples: kinship, FastLMM null model, FamSKAT ----------------------------------------------------
int rvt_set_kinship(rvt_ctx* c, int64_t N, const float* U, const float* S) {
  if (!c || !U || !S || N < 2) return fail(c, RVT_E_INVALID, "bad kinship");
  hipSetDevice(c->device);
  int rc = rvt_sync(c);
  if (rc) return rc;
  for (double** p : {&c->d_S, &c->d_u1}) {
    if (*p) hipFree(*p);
    *p = nullptr;
  }
  if (c->d_Uq) hipFree(c->d_Uq);
  c->d_Uq = nullptr;
  if (c->d_uq_range) hipFree(c->d_uq_range);
  c->d_uq_range = nullptr;
  for (void** q : {(void**)&c->d_csc_ptr, (void**)&c->d_csc_rows, (void**)&c->d_csc_vals}) {
    if (*q) hipFree(*q);
    *q = nullptr;
  }
  c->uq_visit = 1.0;
  c->have_kin = c->have_fam = false;
  HIP_TRY(c, hipMalloc((void**)&c->d_S, sizeof(double) * N));
  HIP_TRY(c, hipMalloc((void**)&c->d_u1, sizeof(double) * N));
  // U -> fixed-point digit planes (rot_gemm.hip.h).  Eigenvectors have |u| <= 1; entries up to 2 are representable.
  c->uq_ldk = (N + 127) / 128 * 128;
  c->uq_rows_pad = (N + kRotBM - 1) / kRotBM * kRotBM;
  c->uq_plane = (size_t)c->uq_rows_pad * (size_t)c->uq_ldk;
  c->uq_sexp = 7 * kRotPlanesU - 3;
  HIP_TRY(c, hipMalloc((void**)&c->d_Uq, c->uq_plane * kRotPlanesU));
  HIP_TRY(c, hipMemsetAsync(c->d_Uq, 0, c->uq_plane * kRotPlanesU, c->stream));
  const long long csc_cap = 64ll * N;  // non-zeros the sparse form may hold
  std::vector<long long> csc_ptr((size_t)N + 1, 0);
  bool csc_ok = !getenv("RVT_KINSHIP_DENSE");
  int* d_cnt = nullptr;
  int* d_span = nullptr;  // first / last non-zero row of every column of U
  struct SpanGuard {
    int** p;
    ~SpanGuard() {
      if (*p) hipFree(*p);
    }
  } span_guard{&d_span};
  {  // whole columns at a time through a bounded staging buffer (the caller's U can be tens of GB): digits + column sums
    const int64_t cols_per = std::max<int64_t>(1, std::min<int64_t>(N, ((int64_t)256 << 20) / N));
    float* d_tmp = nullptr;
    double* d_tmp64 = nullptr;
    int* d_flag = nullptr;
    struct TmpGuard {  // (every early return below leaves through HIP_TRY)
      void** p[4];
      ~TmpGuard() {
        for (void** q : p)
          if (*q) hipFree(*q);
      }
    } tmp_guard{{(void**)&d_tmp, (void**)&d_tmp64, (void**)&d_flag, (void**)&d_cnt}};
    HIP_TRY(c, hipMalloc((void**)&d_span, sizeof(int) * 2 * (size_t)N));
    HIP_TRY(c, hipMalloc((void**)&d_cnt, sizeof(int) * (size_t)cols_per));
    HIP_TRY(c, hipMalloc((void**)&c->d_csc_ptr, sizeof(long long) * (size_t)(N + 1)));
    HIP_TRY(c, hipMalloc((void**)&c->d_csc_rows, sizeof(int) * (size_t)csc_cap));
    HIP_TRY(c, hipMalloc((void**)&c->d_csc_vals, sizeof(double) * (size_t)csc_cap));
    HIP_TRY(c, hipMalloc((void**)&d_tmp, sizeof(float) * (size_t)cols_per * N));
    HIP_TRY(c, hipMalloc((void**)&d_tmp64, sizeof(double) * (size_t)cols_per * N));
    HIP_TRY(c, hipMalloc((void**)&d_flag, sizeof(int)));
    HIP_TRY(c, hipMemsetAsync(d_flag, 0, sizeof(int), c->stream));
    for (int64_t k0 = 0; k0 < N; k0 += cols_per) {
      const int64_t nc = std::min(cols_per, N - k0);
      const size_t n = (size_t)nc * N;
      // (hipMemcpyDefault: rvt_kinship_decompose hands over eigenvectors that are already on the device)
      HIP_TRY(c, hipMemcpyAsync(d_tmp, U + (size_t)k0 * N, sizeof(float) * n, hipMemcpyDefault, c->stream));
      hipLaunchKernelGGL(rot_quantize_f32_kernel, dim3(2048), dim3(256), 0, c->stream, d_tmp, (long long)N, (long long)nc,
                         (long long)N, c->uq_sexp, kRotPlanesU, c->d_Uq, (long long)c->uq_ldk, (long long)c->uq_plane,
                         (long long)k0, d_flag);
      hipLaunchKernelGGL(rot_span_kernel, dim3((unsigned)nc), dim3(256), 0, c->stream, d_tmp, (long long)N, (long long)N,
                         d_span + k0, d_span + N + k0);
      if (csc_ok) {  // column-compressed copy of the chunk while it fits the budget of 64 non-zeros per column
        hipLaunchKernelGGL(rot_nnz_count_kernel, dim3((unsigned)nc), dim3(256), 0, c->stream, d_tmp, (long long)N,
                           (long long)N, d_cnt);
        std::vector<int> cnt((size_t)nc);
        HIP_TRY(c, hipMemcpyAsync(cnt.data(), d_cnt, sizeof(int) * (size_t)nc, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, sync_stream(c->stream));
        for (int64_t j = 0; j < nc; ++j) csc_ptr[k0 + j + 1] = csc_ptr[k0 + j] + cnt[j];
        if (csc_ptr[k0 + nc] > csc_cap) {
          csc_ok = false;
        } else {
          HIP_TRY(c, hipMemcpyAsync(c->d_csc_ptr + k0, csc_ptr.data() + k0, sizeof(long long) * (size_t)(nc + 1),
                                    hipMemcpyHostToDevice, c->stream));
          hipLaunchKernelGGL(rot_nnz_fill_kernel, dim3((unsigned)nc), dim3(256), 0, c->stream, d_tmp, (long long)N,
                             (long long)N, c->d_csc_ptr + k0, c->d_csc_rows, c->d_csc_vals);
        }
      }
      hipLaunchKernelGGL(cvt_f32_f64_kernel, dim3(1024), dim3(256), 0, c->stream, d_tmp, d_tmp64, n);
      hipLaunchKernelGGL(column_sums_kernel, dim3((unsigned)nc), dim3(256), 0, c->stream, d_tmp64, (long long)N,
                         (long long)N, c->d_u1 + k0);
      HIP_TRY(c, sync_stream(c->stream));
    }
    int bad = 0;
    HIP_TRY(c, hipMemcpy(&bad, d_flag, sizeof(int), hipMemcpyDeviceToHost));
    if (!csc_ok)
      for (void** q : {(void**)&c->d_csc_ptr, (void**)&c->d_csc_rows, (void**)&c->d_csc_vals}) {
        if (*q) hipFree(*q);
        *q = nullptr;
      }
    if (bad) return fail(c, RVT_E_INVALID, "kinship eigenvectors have entries >= 2 in magnitude (not unit vectors)");
  }
  c->h_S.resize(N);
  for (int64_t i = 0; i < N; ++i) c->h_S[i] = (double)S[i];
  c->h_u1.resize(N);
  HIP_TRY(c, hipMemcpy(c->h_u1.data(), c->d_u1, sizeof(double) * N, hipMemcpyDeviceToHost));
  {
    // Structure of U.  A kinship matrix of unrelated families is block diagonal and so are its eigenvectors: column k
    // of U is non-zero on the rows [lo_k, hi_k] of one family only.  The statistics do not depend on the ORDER of the
    // eigenpairs, so they are re-ordered by lo_k (stable; a dense U keeps its order): a 256-row panel of the planes
    // then holds eigenvectors of neighbouring families, its non-zeros fall into a few K chunks, and the rotation GEMM
    // visits only those (rot_gemm.hip.h: a_krange).  Exact — the skipped chunks are exact zeros.
    std::vector<int> span(2 * (size_t)N);
    HIP_TRY(c, hipMemcpy(span.data(), d_span, sizeof(int) * 2 * (size_t)N, hipMemcpyDeviceToHost));
    hipFree(d_span);
    d_span = nullptr;
    const int* lo = span.data();
    const int* hi = span.data() + N;
    std::vector<int> order((size_t)N);
    for (int64_t k = 0; k < N; ++k) order[k] = (int)k;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return lo[a] < lo[b]; });
    const int64_t nrp = c->uq_rows_pad / kRotBM, nchunk = c->uq_ldk / kRotKC;
    std::vector<int2> range((size_t)nrp);
    double visited = 0.0;
    for (int64_t rp = 0; rp < nrp; ++rp) {
      int l = (int)N, h = -1;
      for (int64_t r = rp * kRotBM; r < std::min<int64_t>(N, (rp + 1) * kRotBM); ++r) {
        l = std::min(l, lo[order[r]]);
        h = std::max(h, hi[order[r]]);
      }
      range[rp] = (h < l) ? int2{0, 0} : int2{l / kRotKC, h / kRotKC + 1};
      visited += range[rp].y - range[rp].x;
    }
    const double frac = visited / ((double)nrp * (double)nchunk);
    if (frac < 0.5 && !getenv("RVT_KINSHIP_DENSE")) {
      bool identity = true;
      for (int64_t k = 0; k < N && identity; ++k) identity = order[k] == (int)k;
      if (!identity) {  // re-order the rows of every plane (one plane-sized scratch buffer) and S, U'1 with them
        signed char* d_scratch = nullptr;
        int* d_order = nullptr;
        HIP_TRY(c, hipMalloc((void**)&d_scratch, c->uq_plane));
        HIP_TRY(c, hipMalloc((void**)&d_order, sizeof(int) * (size_t)N));
        HIP_TRY(c, hipMemcpy(d_order, order.data(), sizeof(int) * (size_t)N, hipMemcpyHostToDevice));
        for (int p = 0; p < kRotPlanesU; ++p) {
          signed char* plane = c->d_Uq + (size_t)p * c->uq_plane;
          HIP_TRY(c, hipMemcpyAsync(d_scratch, plane, (size_t)N * c->uq_ldk, hipMemcpyDeviceToDevice, c->stream));
          hipLaunchKernelGGL(rot_gather_rows_kernel, dim3((unsigned)N), dim3(256), 0, c->stream, d_scratch, d_order,
                             (long long)c->uq_ldk, plane);
        }
        HIP_TRY(c, sync_stream(c->stream));
        hipFree(d_scratch);
        hipFree(d_order);
        std::vector<double> s2((size_t)N), u2((size_t)N);
        for (int64_t r = 0; r < N; ++r) {
          s2[r] = c->h_S[order[r]];
          u2[r] = c->h_u1[order[r]];
        }
        c->h_S.swap(s2);
        c->h_u1.swap(u2);
        HIP_TRY(c, hipMemcpy(c->d_u1, c->h_u1.data(), sizeof(double) * N, hipMemcpyHostToDevice));
      }
      HIP_TRY(c, hipMalloc((void**)&c->d_uq_range, sizeof(int2) * (size_t)nrp));
      HIP_TRY(c, hipMemcpy(c->d_uq_range, range.data(), sizeof(int2) * (size_t)nrp, hipMemcpyHostToDevice));
      c->uq_visit = frac;
      for (void** q : {(void**)&c->d_csc_ptr, (void**)&c->d_csc_rows, (void**)&c->d_csc_vals}) {  // not needed then
        if (*q) hipFree(*q);
        *q = nullptr;
      }
    } else if (c->d_csc_ptr) {
      // sparse eigenvectors whose supports are scattered over the samples (families interleaved in the sample order):
      // the rotation gathers (rot_sparse_kernel); eigenpairs stay in the caller's order.  The digit planes (6 N^2 bytes)
      // are never read in this mode: give them back.
      c->uq_visit = (double)csc_ptr[N] / ((double)N * (double)N);
      hipFree(c->d_Uq);
      c->d_Uq = nullptr;
    }
  }
  HIP_TRY(c, hipMemcpy(c->d_S, c->h_S.data(), sizeof(double) * N, hipMemcpyHostToDevice));
  c->kin_N = N;
  c->have_kin = true;
  return RVT_OK;
}

int rvt_kinship_structure(rvt_ctx* c, double* visited_fraction) {
  if (!c || !visited_fraction) return RVT_E_INVALID;
  if (!c->have_kin) return fail(c, RVT_E_STATE, "no kinship decomposition installed");
  *visited_fraction = c->uq_visit;
  return RVT_OK;
}

// Family-wise form of rvt_kinship_decompose.  When the sparsity pattern of K splits the samples into groups that do not
// interact (the connected components of its non-zeros: families, in whatever order the samples are listed) and none is
// larger than 64, the eigenproblem is that of its blocks: families are packed into 64 x 64 tiles (block diagonal inside a
// tile, distinct negative pads on the rest of the diagonal: zero off-diagonals are never rotated, so nothing mixes),
// every tile is diagonalised by the two-sided cyclic Jacobi kernel of the dense iteration (jac_small_eig_kernel) and the
// eigenpairs are merged in ascending order.  *done = false: K is not of that form, take the dense iteration.
static int decompose_by_family(rvt_ctx* c, int64_t N, const float* K,
                               const std::vector<std::vector<std::pair<int, int>>>& edges, double mu, int64_t np,
                               float* U_out, float* S_out, int install, rvt_decompose_info* info, bool* done) {
  *done = false;
  // families = connected components of the sparsity pattern (union-find over the off-diagonal non-zeros)
  std::vector<int> parent((size_t)N), csize((size_t)N, 1);
  for (int64_t i = 0; i < N; ++i) parent[i] = (int)i;
  auto find = [&](int x) {
    while (parent[x] != x) {
      parent[x] = parent[parent[x]];
      x = parent[x];
    }
    return x;
  };
  for (const auto& ed : edges)
    for (const auto& e : ed) {
      int a = find(e.first), b = find(e.second);
      if (a == b) continue;
      if (csize[a] < csize[b]) std::swap(a, b);
      parent[b] = a;
      csize[a] += csize[b];
      if (csize[a] > kJacP) return RVT_OK;  // a family larger than a tile: the dense iteration
    }
  // members of every family in ascending order; families in the order of their first member
  std::vector<int> first_of((size_t)N, -1);
  std::vector<std::vector<int>> fam;
  for (int64_t i = 0; i < N; ++i) {
    const int r = find((int)i);
    if (first_of[r] < 0) {
      first_of[r] = (int)fam.size();
      fam.emplace_back();
    }
    fam[first_of[r]].push_back((int)i);
  }
  if (fam.size() < 2) return RVT_OK;
  // tiles of consecutive families: trow[t] = the sample index behind every row of the tile
  std::vector<std::vector<int>> trow;
  for (const auto& f : fam) {
    if (!trow.empty() && trow.back().size() + f.size() <= (size_t)kJacP)
      trow.back().insert(trow.back().end(), f.begin(), f.end());
    else
      trow.push_back(f);
  }
  std::vector<int> tlen;
  for (const auto& r : trow) tlen.push_back((int)r.size());
  const size_t nt = trow.size();
  std::vector<double> A(nt * (size_t)kJacP * kJacP, 0.0);
  for (size_t t = 0; t < nt; ++t) {
    double* a = A.data() + t * (size_t)kJacP * kJacP;
    const std::vector<int>& rows = trow[t];
    for (int q = 0; q < tlen[t]; ++q)
      for (int r = 0; r < tlen[t]; ++r) a[(size_t)r * kJacP + q] = (double)K[(size_t)rows[r] + (size_t)rows[q] * N];
    for (int r = tlen[t]; r < kJacP; ++r) a[(size_t)r * kJacP + r] = -mu * (1.0 + (double)r / kJacP);  // pads: last, apart
  }
  hipStream_t st = c->stream;
  struct Bufs {
    double *A = nullptr, *R = nullptr, *lam = nullptr;
    unsigned long long* maxcos = nullptr;
    float* dU = nullptr;
    int* meta = nullptr;
    ~Bufs() {
      for (void* p : {(void*)A, (void*)R, (void*)lam, (void*)maxcos, (void*)dU, (void*)meta})
        if (p) hipFree(p);
    }
  } b;
  const size_t tile_bytes = sizeof(double) * (size_t)kJacP * kJacP;
  HIP_TRY(c, hipMalloc((void**)&b.A, tile_bytes * nt));
  HIP_TRY(c, hipMalloc((void**)&b.R, tile_bytes * nt));
  HIP_TRY(c, hipMalloc((void**)&b.lam, sizeof(double) * kJacP * nt));
  HIP_TRY(c, hipMalloc((void**)&b.maxcos, sizeof(unsigned long long)));
  HIP_TRY(c, hipMemsetAsync(b.maxcos, 0, sizeof(unsigned long long), st));
  HIP_TRY(c, hipMemcpyAsync(b.A, A.data(), tile_bytes * nt, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(jac_small_eig_kernel, dim3((unsigned)nt), dim3(256), 0, st, b.A, 1, 1e-14, b.R, b.maxcos, 1, b.lam);
  HIP_TRY(c, hipGetLastError());
  std::vector<double> R(nt * (size_t)kJacP * kJacP), lam(nt * (size_t)kJacP);
  HIP_TRY(c, hipMemcpyAsync(R.data(), b.R, tile_bytes * nt, hipMemcpyDeviceToHost, st));
  HIP_TRY(c, hipMemcpyAsync(lam.data(), b.lam, sizeof(double) * kJacP * nt, hipMemcpyDeviceToHost, st));
  HIP_TRY(c, sync_stream(st));
  // the first tlen[t] output columns of a tile are its own eigenpairs (decreasing; the pads are below all of them)
  struct Pair {
    double lam;
    int tile, col;
  };
  std::vector<Pair> pairs;
  pairs.reserve((size_t)N);
  double worst = 0.0;
  for (size_t t = 0; t < nt; ++t) {
    const double* a = A.data() + t * (size_t)kJacP * kJacP;
    const double* r = R.data() + t * (size_t)kJacP * kJacP;
    for (int q = 0; q < tlen[t]; ++q) {
      const double l = lam[t * kJacP + q];
      pairs.push_back({l, (int)t, q});
      double res2 = 0.0;  // || K u - lambda u || inside the tile (K is zero outside)
      for (int i = 0; i < tlen[t]; ++i) {
        double s = 0.0;
        for (int k = 0; k < tlen[t]; ++k) s += a[(size_t)i * kJacP + k] * r[(size_t)k * kJacP + q];
        s -= l * r[(size_t)i * kJacP + q];
        res2 += s * s;
      }
      worst = std::max(worst, std::sqrt(res2));
    }
  }
  if ((int64_t)pairs.size() != N) return fail(c, RVT_E_STATE, "family-wise decomposition lost eigenpairs");
  // the 30 cyclic sweeps of jac_small_eig_kernel converge for every family tile seen so far; should one not have, the
  // dense iteration (which checks its own convergence and residuals) takes the matrix instead
  if (worst > 1e-9 * mu) return RVT_OK;  // (*done stays false)
  std::stable_sort(pairs.begin(), pairs.end(), [](const Pair& x, const Pair& y) { return x.lam < y.lam; });  // ascending
  std::vector<float> S((size_t)N);
  std::vector<int> meta(3 * (size_t)N + nt * (size_t)kJacP, 0);  // tile | column | length per eigenpair, rows per tile
  for (int64_t k = 0; k < N; ++k) {
    S[k] = (float)pairs[k].lam;
    meta[k] = pairs[k].tile;
    meta[(size_t)N + k] = pairs[k].col;
    meta[2 * (size_t)N + k] = tlen[pairs[k].tile];
  }
  for (size_t t = 0; t < nt; ++t)
    for (int r = 0; r < tlen[t]; ++r) meta[3 * (size_t)N + t * kJacP + r] = trow[t][r];
  if (U_out || install) {
    HIP_TRY(c, hipMalloc((void**)&b.dU, sizeof(float) * (size_t)N * (size_t)N));
    HIP_TRY(c, hipMemsetAsync(b.dU, 0, sizeof(float) * (size_t)N * (size_t)N, st));
    HIP_TRY(c, hipMalloc((void**)&b.meta, sizeof(int) * meta.size()));
    HIP_TRY(c, hipMemcpyAsync(b.meta, meta.data(), sizeof(int) * meta.size(), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(jac_scatter_blocks_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, b.R, b.meta,
                       b.meta + N, b.meta + 2 * N, b.meta + 3 * N, (long long)N, b.dU);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, sync_stream(st));
    if (U_out) HIP_TRY(c, hipMemcpy(U_out, b.dU, sizeof(float) * (size_t)N * (size_t)N, hipMemcpyDeviceToHost));
  }
  if (S_out) std::memcpy(S_out, S.data(), sizeof(float) * (size_t)N);
  if (info) {
    info->sweeps = 0;  // no block sweeps: every family is an eigenproblem of its own
    info->max_cosine = 0.0;
    info->padded_order = np;
    info->shift = 0.0;
    info->max_residual = worst;
  }
  *done = true;
  if (install) return rvt_set_kinship(c, N, b.dU, S.data());
  return RVT_OK;
}

// ---- dense kinship: tridiagonalisation + bisection + inverse iteration (tridiag_kernels.hip.h) ------------------------------
// *done = true when U and S were produced AND passed the closing check (residual, orthogonality); false (and RVT_OK) when the
// matrix has eigenvalues closer than the inverse iteration separates, or the check failed: the caller then runs the Jacobi
// iteration.  mu: 4 x the largest absolute row sum (a bound on 4 x the spectral radius).
static int decompose_dense_tridiag(rvt_ctx* c, int64_t N, const float* K, double mu, float* U_out, float* S_out, int install,
                                   rvt_decompose_info* info, bool* done) {
  *done = false;
  // an allocation the device cannot serve (fragmentation, another context, the K-slice buffers of gemm_tn_f64) is not an error of
  // the call: the matrix goes to the Jacobi iteration (*done stays false), as the size check below promises
#define TD_ALLOC(call)                                                                                                \
  do {                                                                                                                \
    if ((call) != hipSuccess) {                                                                                       \
      (void)hipGetLastError();                                                                                        \
      if (c->d_rot_part) {                                                                                            \
        hipFree(c->d_rot_part);                                                                                       \
        c->d_rot_part = nullptr;                                                                                      \
        c->rot_part_cap = 0;                                                                                          \
      }                                                                                                               \
      if (trace) fprintf(stderr, "[rvt] tridiag: device allocation failed: left to the Jacobi iteration\n");          \
      return RVT_OK;                                                                                                  \
    }                                                                                                                 \
  } while (0)
  hipStream_t st = c->stream;
  const bool trace = getenv("RVT_TRIDIAG_TRACE") != nullptr;
  const double t_start = now_s();
  const int n = (int)N;
  const int64_t ld = (N + 63) / 64 * 64;
  struct Bufs {
    float* dK = nullptr;
    float* dU = nullptr;
    double *A = nullptr, *B1 = nullptr, *B2 = nullptr, *B3 = nullptr, *B4 = nullptr, *W = nullptr, *vec = nullptr, *cz = nullptr,
           *yy = nullptr, *P = nullptr;
    unsigned long long* worst = nullptr;
    ~Bufs() {
      for (void* p : {(void*)dK, (void*)dU, (void*)A, (void*)B1, (void*)B2, (void*)B3, (void*)B4, (void*)W, (void*)vec, (void*)cz,
                      (void*)yy, (void*)P, (void*)worst})
        if (p) hipFree(p);
    }
  } b;
  const size_t mat = sizeof(double) * (size_t)ld * (size_t)N;
  {
    // Peak: the matrix with its reflectors (A), the eigenvectors (Z) and the float matrix for the closing check — 20 N^2 bytes
    // (the inverse iteration's factors are kept for a batch of eigenvectors at a time and the check's fp64 copy of K re-uses A).
    // When the device cannot hold that (other contexts, a large block pool) the Jacobi iteration (16 N^2 bytes) gets the matrix
    // instead of an allocation failure.
    size_t free_b = 0, total_b = 0;
    const size_t need = 2 * mat + sizeof(float) * (size_t)N * (size_t)N + sizeof(double) * (size_t)ld * (4 * kTdNbb + kTdNb + 16 * 1024) +
                        ((size_t)2 << 30);
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < need) {
      (void)hipGetLastError();
      if (trace) fprintf(stderr, "[rvt] tridiag: %zu MB needed, %zu MB free: left to the Jacobi iteration\n", need >> 20, free_b >> 20);
      return RVT_OK;
    }
  }
  TD_ALLOC(hipMalloc((void**)&b.dK, sizeof(float) * (size_t)N * (size_t)N));
  TD_ALLOC(hipMalloc((void**)&b.A, mat));
  TD_ALLOC(hipMalloc((void**)&b.W, sizeof(double) * (size_t)ld * kTdNb));
  // vec: d | e | tau | y | t12 (2 kTdNb) | partial dots | scaled d | scaled e^2 | lambda
  const int64_t vs = std::max<int64_t>(ld, 1024);  // (y doubles as the panel's Gram matrix later)
  const int comb_blocks = (int)((N + 63) / 64);                   // workgroups of td_w_comb_kernel = partial dots per column
  const int64_t n_part = std::max<int64_t>(1024, comb_blocks);
  const size_t nv = 7 * (size_t)vs + 2 * kTdNb + (size_t)n_part;
  TD_ALLOC(hipMalloc((void**)&b.vec, sizeof(double) * nv));
  HIP_TRY(c, hipMemsetAsync(b.vec, 0, sizeof(double) * nv, st));
  HIP_TRY(c, hipMemsetAsync(b.W, 0, sizeof(double) * (size_t)ld * kTdNb, st));
  double *d_d = b.vec, *d_e = d_d + vs, *d_tau = d_e + vs, *d_y = d_tau + vs, *d_t12 = d_y + vs, *d_part = d_t12 + 2 * kTdNb, *d_ss = d_y,
         *d_ds = d_part + n_part, *d_e2s = d_ds + vs, *d_lam = d_e2s + vs;
  HIP_TRY(c, hipMemcpyAsync(b.dK, K, sizeof(float) * (size_t)N * (size_t)N, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(td_init_kernel, dim3(4096), dim3(256), 0, st, b.dK, (long long)N, (long long)ld, b.A);
  // 1. K = Q T Q'
  const int nblk = (int)((N + kSyT - 1) / kSyT);  // 64-row blocks of the matrix; P: the (nblk + 1) x ld partial products of a column step
  TD_ALLOC(hipMalloc((void**)&b.P, sizeof(double) * (size_t)(nblk + 1) * (size_t)ld));
  for (int j0 = 0; j0 < n; j0 += kTdNb) {
    const int j1 = std::min(n, j0 + kTdNb);
    for (int j = j0; j < j1; ++j) {
      const int n_ss = (n - j + 255) / 256;
      hipLaunchKernelGGL(td_col_update_kernel, dim3((unsigned)n_ss), dim3(256), 0, st, b.A, (long long)N, (long long)ld, j, j0, b.W,
                         d_ss);
      hipLaunchKernelGGL(td_col_house_kernel, dim3(1), dim3(1024), 0, st, b.A, (long long)N, (long long)ld, j, d_ss, n_ss, d_d, d_e,
                         d_tau);
      if (j + 1 >= n) continue;
      const int i = j - j0;
      const int nbt = nblk - (j + 1) / kSyT, groups = (nbt + kSyRun - 1) / kSyRun;
      hipLaunchKernelGGL(td_symv_kernel, dim3((unsigned)groups, (unsigned)(nbt + (2 * i + groups - 1) / groups)), dim3(256), 0, st, b.A,
                         (long long)N, (long long)ld, j, j0, nbt, b.W, b.P, d_t12);
      hipLaunchKernelGGL(td_w_comb_kernel, dim3((unsigned)comb_blocks), dim3(256), 0, st, b.A, (long long)N, (long long)ld, j, j0,
                         b.W, b.P, nblk, d_t12, d_tau, d_part);
      hipLaunchKernelGGL(td_w_final_kernel, dim3(1), dim3(1024), 0, st, b.A, (long long)N, (long long)ld, j, j0, b.W, d_tau, d_part,
                         comb_blocks);
    }
    if (j1 < n) {
      const unsigned tiles = (unsigned)((n - j1 + 63) / 64);
      hipLaunchKernelGGL(td_rank2k_kernel, dim3(tiles, tiles), dim3(256), 0, st, b.A, (long long)N, (long long)ld, j0, j1, b.W);
    }
  }
  HIP_TRY(c, hipGetLastError());
  std::vector<double> hd((size_t)N), he((size_t)N, 0.0);
  HIP_TRY(c, hipMemcpyAsync(hd.data(), d_d, sizeof(double) * (size_t)N, hipMemcpyDeviceToHost, st));
  HIP_TRY(c, hipMemcpyAsync(he.data(), d_e, sizeof(double) * (size_t)(N - 1), hipMemcpyDeviceToHost, st));
  HIP_TRY(c, sync_stream(st));
  hipFree(b.dK);
  b.dK = nullptr;
  const double t_tri = now_s();
  // 2. eigenvalues of T (scaled as coop_tridiag_eigvals scales: Gershgorin span into [1/2, 1), squares floored)
  double lo = hd[0], hi = hd[0];
  for (int64_t j = 0; j < N; ++j) {
    if (!std::isfinite(hd[j]) || !std::isfinite(he[j])) return RVT_OK;  // (the Jacobi iteration reports what is wrong)
    const double r = (j > 0 ? std::fabs(he[j - 1]) : 0.0) + (j < N - 1 ? std::fabs(he[j]) : 0.0);
    lo = std::min(lo, hd[j] - r);
    hi = std::max(hi, hd[j] + r);
  }
  const double span0 = std::max(std::fabs(lo), std::fabs(hi));
  if (!(span0 > 0.0)) return RVT_OK;
  int sh = 0;
  (void)std::frexp(span0, &sh);
  {
    std::vector<double> ds((size_t)N), e2s((size_t)N, 0.0);
    for (int64_t j = 0; j < N; ++j) {
      ds[j] = std::ldexp(hd[j], -sh);
      if (j < N - 1) {
        const double es = std::ldexp(he[j], -sh);
        e2s[j] = std::max(es * es, 0x1p-200);
      }
    }
    HIP_TRY(c, hipMemcpyAsync(d_ds, ds.data(), sizeof(double) * (size_t)N, hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipMemcpyAsync(d_e2s, e2s.data(), sizeof(double) * (size_t)N, hipMemcpyHostToDevice, st));
    HIP_TRY(c, sync_stream(st));
  }
  const double slo = std::ldexp(lo, -sh), shi = std::ldexp(hi, -sh), span = std::max(std::fabs(slo), std::fabs(shi));
  const double pivmin = DBL_MIN * 1024.0, slack = 2.0 * kDblEps * span * (double)N + 2.0 * pivmin;
  hipLaunchKernelGGL(td_eigvals_kernel, dim3((unsigned)((N + 63) / 64)), dim3(64), 0, st, d_ds, d_e2s, n, slo - slack, shi + slack,
                     span, sh, d_lam);
  std::vector<double> lam((size_t)N);
  HIP_TRY(c, hipMemcpyAsync(lam.data(), d_lam, sizeof(double) * (size_t)N, hipMemcpyDeviceToHost, st));
  HIP_TRY(c, sync_stream(st));
  const double t_eig = now_s();
  // gaps: the inverse iteration is not reorthogonalised
  double min_gap = span0;
  for (int64_t j = 1; j < N; ++j) min_gap = std::min(min_gap, lam[j] - lam[j - 1]);
  if (trace) fprintf(stderr, "[rvt] tridiag: N %lld, reduction %.3f s, eigenvalues %.3f s, smallest gap %.3g of %.3g\n", (long long)N,
                     t_tri - t_start, t_eig - t_tri, min_gap, span0);
  // neighbours a gap g apart come out orthogonal to ~ eps |T| / g: 4e-9 of the norm promises 6e-8, float rounding of the U the
  // boundary stores (the closing check below holds the result to 1e-7 whatever this predicts)
  if (!(min_gap > 4e-9 * span0)) return RVT_OK;
  // 3. eigenvectors of T, a batch of columns at a time (the factors of a batch: three [row][eigenvector] arrays + the vectors)
  TD_ALLOC(hipMalloc((void**)&b.B2, mat));
  HIP_TRY(c, hipMemsetAsync(b.B2, 0, mat, st));
  {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = (size_t)8 << 30;
    int64_t nkb = (int64_t)((free_b / 2) / (4 * sizeof(double) * (size_t)ld)) / 64 * 64;  // half of what is free
    if (const char* e = getenv("RVT_TRIDIAG_BATCH")) nkb = std::max(64, atoi(e) / 64 * 64);  // (tests: several batches on a small matrix)
    nkb = std::max<int64_t>(64, std::min<int64_t>(nkb, ld));
    const size_t arr = sizeof(double) * (size_t)ld * (size_t)nkb;
    TD_ALLOC(hipMalloc((void**)&b.B1, 4 * arr));
    double *zt = b.B1, *ud = zt + (size_t)ld * nkb, *uu = ud + (size_t)ld * nkb, *uw = uu + (size_t)ld * nkb;
    for (int64_t k0 = 0; k0 < N; k0 += nkb) {
      const int nk = (int)std::min<int64_t>(nkb, N - k0);
      hipLaunchKernelGGL(td_invit_kernel, dim3((unsigned)((nk + 63) / 64)), dim3(64), 0, st, d_d, d_e, n, d_lam + k0, nk, (long long)k0, (long long)nkb,
                         kDblEps * span0, ud, uu, uw, zt);
      hipLaunchKernelGGL(td_transpose_kernel, dim3((unsigned)((N + 31) / 32), (unsigned)((nk + 31) / 32)), dim3(256), 0, st, zt,
                         (long long)nkb, n, nk, (long long)ld, b.B2 + (size_t)k0 * ld);
    }
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, sync_stream(st));
    hipFree(b.B1);
    b.B1 = nullptr;
  }
  double* Z = b.B2;
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, sync_stream(st));
  const double t_vec = now_s();
  // 4. U = Q Z, the reflectors kTdNbb at a time in reverse order: G = V'V and V'Z (K = N), the back substitution with
  //    T^-1 = triu(G, 1) + diag(1 / tau), and Z -= V Y' (K = kTdNbb) — all three products on the matrix cores
  TD_ALLOC(hipMalloc((void**)&b.cz, sizeof(double) * (size_t)ld * kTdNbb));
  TD_ALLOC(hipMalloc((void**)&b.yy, sizeof(double) * (size_t)ld * kTdNbb * 2 + sizeof(double) * kTdNbb * kTdNbb));
  double *d_yt = b.yy, *d_vt = b.yy + (size_t)ld * kTdNbb, *d_gram = d_vt + (size_t)ld * kTdNbb;
  HIP_TRY(c, hipMemsetAsync(d_gram, 0, sizeof(double) * kTdNbb * kTdNbb, st));
  for (int j0 = ((n - 1) / kTdNbb) * kTdNbb; j0 >= 0; j0 -= kTdNbb) {
    const int nbp = std::min(n, j0 + kTdNbb) - j0;
    const double* Vp = b.A + (size_t)j0 * ld;
    int rc = gemm_tn_f64(c, Vp, ld, nbp, Vp, ld, nbp, nullptr, 0, 0, nullptr, ld, d_gram, kTdNbb, true, st);
    if (rc) return (rc == RVT_E_HIP && strstr(rvt_last_error(c), "out of memory")) ? (int)((void)hipGetLastError(), RVT_OK) : rc;  // (its K-slice buffer: Jacobi takes over)
    rc = gemm_tn_f64(c, Z, ld, n, Vp, ld, nbp, nullptr, 0, 0, nullptr, ld, b.cz, ld, false, st);
    if (rc) return (rc == RVT_E_HIP && strstr(rvt_last_error(c), "out of memory")) ? (int)((void)hipGetLastError(), RVT_OK) : rc;  // (its K-slice buffer: Jacobi takes over)
    hipLaunchKernelGGL(td_rsolve_kernel, dim3((unsigned)((N + 63) / 64)), dim3(64), 0, st, b.cz, (long long)ld, n, nbp, d_gram,
                       d_tau + j0, d_yt);
    hipLaunchKernelGGL(td_panel_transpose_kernel, dim3((unsigned)((N + 31) / 32), kTdNbb / 32), dim3(256), 0, st, b.A, (long long)N,
                       (long long)ld, j0, nbp, d_vt);
    rc = gemm_tn_f64(c, d_vt, kTdNbb, n, d_yt, kTdNbb, n, nullptr, 0, 0, nullptr, kTdNbb, Z, ld, false, st, true);
    if (rc) return (rc == RVT_E_HIP && strstr(rvt_last_error(c), "out of memory")) ? (int)((void)hipGetLastError(), RVT_OK) : rc;  // (its K-slice buffer: Jacobi takes over)
  }
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, sync_stream(st));
  const double t_back = now_s();
  // 5. the closing check: max |K u - lambda u|, max |U'U - I|, in column batches (the product's K slices need room)
  TD_ALLOC(hipMalloc((void**)&b.worst, 2 * sizeof(unsigned long long)));
  HIP_TRY(c, hipMemsetAsync(b.worst, 0, 2 * sizeof(unsigned long long), st));
  // (the reflectors are not needed any more: A takes the fp64 copy of K; the float upload is freed behind the conversion)
  for (double** q : {&b.cz, &b.yy, &b.W, &b.P}) {
    if (*q) hipFree(*q);
    *q = nullptr;
  }
  TD_ALLOC(hipMalloc((void**)&b.dK, sizeof(float) * (size_t)N * (size_t)N));
  HIP_TRY(c, hipMemcpyAsync(b.dK, K, sizeof(float) * (size_t)N * (size_t)N, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(td_init_kernel, dim3(4096), dim3(256), 0, st, b.dK, (long long)N, (long long)ld, b.A);
  HIP_TRY(c, sync_stream(st));
  hipFree(b.dK);
  b.dK = nullptr;
  constexpr int kBatch = 1024;
  TD_ALLOC(hipMalloc((void**)&b.B3, sizeof(double) * (size_t)ld * kBatch));
  for (int k0 = 0; k0 < n; k0 += kBatch) {
    const int nk = std::min(kBatch, n - k0);
    int rc = gemm_tn_f64(c, b.A, ld, n, Z + (size_t)k0 * ld, ld, nk, nullptr, 0, 0, nullptr, ld, b.B3, ld, false, st);
    if (rc) return (rc == RVT_E_HIP && strstr(rvt_last_error(c), "out of memory")) ? (int)((void)hipGetLastError(), RVT_OK) : rc;  // (its K-slice buffer: Jacobi takes over)
    hipLaunchKernelGGL(td_residual_kernel, dim3((unsigned)nk), dim3(256), 0, st, b.B3, (long long)ld, Z + (size_t)k0 * ld,
                       (long long)ld, n, d_lam + k0, b.worst);
    rc = gemm_tn_f64(c, Z, ld, n, Z + (size_t)k0 * ld, ld, nk, nullptr, 0, 0, nullptr, ld, b.B3, ld, false, st);
    if (rc) return (rc == RVT_E_HIP && strstr(rvt_last_error(c), "out of memory")) ? (int)((void)hipGetLastError(), RVT_OK) : rc;  // (its K-slice buffer: Jacobi takes over)
    hipLaunchKernelGGL(td_orth_kernel, dim3((unsigned)nk), dim3(256), 0, st, b.B3, (long long)ld, n, k0, b.worst);
  }
  unsigned long long bits[2] = {0, 0};
  HIP_TRY(c, hipMemcpyAsync(bits, b.worst, sizeof(bits), hipMemcpyDeviceToHost, st));
  HIP_TRY(c, sync_stream(st));
  double resid, orth;
  std::memcpy(&resid, &bits[0], 8);
  std::memcpy(&orth, &bits[1], 8);
  const double t_chk = now_s();
  if (trace)
    fprintf(stderr, "[rvt] tridiag: vectors %.3f s, back-transformation %.3f s, check %.3f s: residual %.3g (scale %.3g), |U'U - I| %.3g\n",
            t_vec - t_eig, t_back - t_vec, t_chk - t_back, resid, span0, orth);
  if (!(resid <= 1e-9 * mu) || !(orth <= 1e-7)) {  // (mu = 4 x a bound on the spectral radius; Jacobi's own bar)
    if (c->d_rot_part) {  // (the products' K slices: the Jacobi iteration that takes over needs the room)
      hipFree(c->d_rot_part);
      c->d_rot_part = nullptr;
      c->rot_part_cap = 0;
    }
    return RVT_OK;
  }
  std::vector<float> S((size_t)N);
  for (int64_t j = 0; j < N; ++j) S[j] = (float)lam[j];
  hipFree(b.A);
  b.A = nullptr;
  TD_ALLOC(hipMalloc((void**)&b.dU, sizeof(float) * (size_t)N * (size_t)N));
  hipLaunchKernelGGL(td_to_float_kernel, dim3(4096), dim3(256), 0, st, Z, (long long)ld, (long long)N, b.dU);
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, sync_stream(st));
  hipFree(b.B2);  // (Z: the installation below needs room for the digit planes of U)
  b.B2 = nullptr;
  if (c->d_rot_part) {  // (the K slices of the check's products: GBs at this size)
    hipFree(c->d_rot_part);
    c->d_rot_part = nullptr;
    c->rot_part_cap = 0;
  }
  if (U_out) HIP_TRY(c, hipMemcpy(U_out, b.dU, sizeof(float) * (size_t)N * (size_t)N, hipMemcpyDeviceToHost));
  if (S_out) std::memcpy(S_out, S.data(), sizeof(float) * (size_t)N);
  if (info) {
    info->sweeps = 0;  // no Jacobi sweeps
    info->max_cosine = 0.0;
    info->padded_order = ld;
    info->shift = 0.0;
    info->max_residual = resid;
  }
  *done = true;
  if (install) return rvt_set_kinship(c, N, b.dU, S.data());
  return RVT_OK;
}
#undef TD_ALLOC

// ---- KinshipHolder::decompose on the device (jacobi_kernels.hip.h) ---------------------------------------------------------
int rvt_kinship_decompose(rvt_ctx* c, int64_t N, const float* K, float* U_out, float* S_out, int install,
                          rvt_decompose_info* info) {
  if (!c || !K || N < 2) return fail(c, RVT_E_INVALID, "bad kinship matrix");
  if (N > (int64_t)1 << 20) return fail(c, RVT_E_TOO_LARGE, "kinship of %lld samples", (long long)N);
  hipSetDevice(c->device);
  int rc = rvt_sync(c);
  if (rc) return rc;
  hipStream_t st = c->stream;
  const int64_t np = (N + kJacP - 1) / kJacP * kJacP;  // an even number of 32-column blocks
  const int nb = (int)(np / kJacB), pairs = nb / 2;
  // |lambda| <= max row sum of |K| (Gershgorin); the pad entries sit well outside
  double mu = 0.0;
  std::vector<std::vector<std::pair<int, int>>> edges;  // off-diagonal non-zeros (j, i), i > j, per scanning thread
  bool sparse_pattern = true;                            // false: more non-zeros than families of <= 64 could have
  {
    // one pass over the N^2 floats on the host (40 GB at N = 100 000): column ranges dealt to a few threads
    const int nthr = (int)std::max<int64_t>(1, std::min<int64_t>({(int64_t)16, (int64_t)std::thread::hardware_concurrency(),
                                                                 N / 512}));
    std::vector<std::vector<double>> part((size_t)nthr, std::vector<double>((size_t)N, 0.0));
    std::vector<int> bad((size_t)nthr, 0);
    edges.assign((size_t)nthr, {});
    std::vector<std::thread> pool;
    for (int t = 0; t < nthr; ++t)
      pool.emplace_back([&, t]() {
        std::vector<double>& rows = part[t];
        std::vector<std::pair<int, int>>& ed = edges[t];
        const int64_t j0 = N * t / nthr, j1 = N * (t + 1) / nthr;
        const size_t cap = (size_t)(j1 - j0) * kJacP;  // more non-zeros than families of 64 could have: a dense matrix
        bool dense_here = false;
        for (int64_t j = j0; j < j1; ++j) {
          const float* col = K + (size_t)j * N;
          for (int64_t i = 0; i < N; ++i) {
            if (!std::isfinite(col[i])) bad[t] = 1;
            rows[i] += std::fabs((double)col[i]);
            if (col[i] != 0.0f && i != j && !dense_here) {
              // edges come from the lower triangle; every off-diagonal non-zero of either triangle must have its mirror
              // image (an asymmetric matrix would silently lose cross-family entries otherwise)
              if (col[i] != K[(size_t)i * N + j]) bad[t] |= 4;
              if (i > j) {
                if (ed.size() >= cap)
                  dense_here = true;
                else
                  ed.emplace_back((int)j, (int)i);
              }
            }
          }
        }
        if (dense_here) bad[t] |= 2;
      });
    for (auto& th : pool) th.join();
    for (int t = 0; t < nthr; ++t) {
      if (bad[t] & 1) return fail(c, RVT_E_INVALID, "kinship matrix holds a non-finite entry");
      if (bad[t] & 4) return fail(c, RVT_E_INVALID, "kinship matrix is not symmetric");
      if (bad[t] & 2) sparse_pattern = false;
    }
    for (int64_t i = 0; i < N; ++i) {
      double r = 0.0;
      for (int t = 0; t < nthr; ++t) r += part[t][i];
      mu = std::max(mu, r);
    }
    mu = 4.0 * std::max(mu, 1e-300);
  }
  if (!getenv("RVT_KINSHIP_DENSE")) {  // a block-diagonal (pedigree) kinship is decomposed family by family
    bool done = false;
    if (sparse_pattern) rc = decompose_by_family(c, N, K, edges, mu, np, U_out, S_out, install, info, &done);
    if (rc || done) return rc;
  }
  // a dense matrix whose work areas fit (20 N^2 bytes at the peak): tridiagonalisation + bisection + inverse iteration;
  // repeated eigenvalues (and anything that fails its closing check) fall through to the Jacobi iteration
  if (!getenv("RVT_KINSHIP_JACOBI") && N >= 128) {
    bool done = false;
    rc = decompose_dense_tridiag(c, N, K, mu, U_out, S_out, install, info, &done);
    if (rc || done) return rc;
  }
  struct Bufs {
    float* dK = nullptr;
    double *W = nullptr, *V = nullptr, *part = nullptr, *R = nullptr, *lam = nullptr;
    unsigned long long* maxcos = nullptr;
    float* dU = nullptr;
    int* dsrc = nullptr;
    ~Bufs() {
      for (void* p : {(void*)dK, (void*)W, (void*)V, (void*)part, (void*)R, (void*)lam, (void*)maxcos, (void*)dU, (void*)dsrc})
        if (p) hipFree(p);
    }
  } b;
  const size_t nn = (size_t)np * (size_t)np;
  HIP_TRY(c, hipMalloc((void**)&b.V, sizeof(double) * nn));
  // row splits of the Gram pass / row slabs of the update: enough waves to fill the chip when there are few pairs
  const int splits = (int)std::max<int64_t>(1, std::min<int64_t>(np / 64, (1024 + pairs - 1) / pairs / 4));
  const int nparts = splits * 4, slabs = splits;
  HIP_TRY(c, hipMalloc((void**)&b.part, sizeof(double) * (size_t)pairs * nparts * kJacP * kJacP));
  HIP_TRY(c, hipMalloc((void**)&b.R, sizeof(double) * (size_t)pairs * kJacP * kJacP));
  HIP_TRY(c, hipMalloc((void**)&b.lam, sizeof(double) * 2 * (size_t)np));
  HIP_TRY(c, hipMalloc((void**)&b.maxcos, sizeof(unsigned long long)));
  const double tol = 1e-10;
  const int max_sweeps = 40;
  const int sort_mode = getenv("RVT_JACOBI_NOSORT") ? 0 : 1;
  int sweeps = 0, total_sweeps = 0;
  double last = 0.0, shift = 0.0, worst_resid = 0.0;
  std::vector<double> lam((size_t)np), resid((size_t)np);
  for (int attempt = 0; attempt < 2; ++attempt) {
    if (!b.W) HIP_TRY(c, hipMalloc((void**)&b.W, sizeof(double) * nn));
    HIP_TRY(c, hipMalloc((void**)&b.dK, sizeof(float) * (size_t)N * (size_t)N));
    HIP_TRY(c, hipMemcpyAsync(b.dK, K, sizeof(float) * (size_t)N * (size_t)N, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(jac_init_kernel, dim3(4096), dim3(256), 0, st, b.dK, (long long)N, (long long)np, mu + shift, shift,
                       b.W, b.V);
    HIP_TRY(c, sync_stream(st));
    hipFree(b.dK);
    b.dK = nullptr;
    for (sweeps = 0; sweeps < max_sweeps;) {
      HIP_TRY(c, hipMemsetAsync(b.maxcos, 0, sizeof(unsigned long long), st));
      for (int r = 0; r < nb - 1; ++r) {
        hipLaunchKernelGGL(jac_gram_kernel, dim3((unsigned)pairs, (unsigned)splits), dim3(256), 0, st, b.W, (long long)np, nb,
                           r, splits, b.part);
        hipLaunchKernelGGL(jac_small_eig_kernel, dim3((unsigned)pairs), dim3(256), 0, st, b.part, nparts, 1e-15, b.R,
                           b.maxcos, sort_mode);
        hipLaunchKernelGGL(jac_apply_kernel, dim3((unsigned)pairs, (unsigned)slabs, 2), dim3(256), 0, st, b.W, b.V,
                           (long long)np, nb, r, b.R);
      }
      HIP_TRY(c, hipGetLastError());
      unsigned long long bits = 0;
      HIP_TRY(c, hipMemcpyAsync(&bits, b.maxcos, sizeof(bits), hipMemcpyDeviceToHost, st));
      HIP_TRY(c, sync_stream(st));
      std::memcpy(&last, &bits, sizeof(last));
      ++sweeps;
      if (getenv("RVT_JACOBI_TRACE")) fprintf(stderr, "[rvt] jacobi sweep %d: max cosine %.3e\n", sweeps, last);
      if (last < tol) break;
    }
    total_sweeps += sweeps;
    if (!(last < tol))
      return fail(c, RVT_E_INVALID, "kinship decomposition did not converge (cosine %.3g after %d sweeps)", last, sweeps);
    hipLaunchKernelGGL(jac_lambda_kernel, dim3((unsigned)np), dim3(256), 0, st, b.W, b.V, (long long)np, b.lam, b.lam + np);
    HIP_TRY(c, hipMemcpyAsync(lam.data(), b.lam, sizeof(double) * (size_t)np, hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipMemcpyAsync(resid.data(), b.lam + np, sizeof(double) * (size_t)np, hipMemcpyDeviceToHost, st));
    HIP_TRY(c, sync_stream(st));
    worst_resid = 0.0;
    for (int64_t j = 0; j < np; ++j) worst_resid = std::max(worst_resid, resid[j]);
    if (worst_resid <= 1e-9 * mu || attempt == 1) break;
    shift = 0.26 * mu;  // mu = 4 x (bound on the spectral radius): K + shift I is positive definite
  }
  for (int64_t j = 0; j < np; ++j) lam[j] -= shift;
  hipFree(b.W);  // (80 GB at N = 100 000: not needed any more)
  b.W = nullptr;
  std::vector<int> src;
  src.reserve((size_t)N);
  for (int64_t j = 0; j < np; ++j)
    if (lam[j] > -0.5 * mu - shift) src.push_back((int)j);
  if ((int64_t)src.size() != N) return fail(c, RVT_E_INVALID, "kinship decomposition: %zu of %lld eigenpairs separated", src.size(), (long long)N);
  std::stable_sort(src.begin(), src.end(), [&](int x, int y) { return lam[x] < lam[y]; });  // ascending, as Eigen returns them
  std::vector<float> S((size_t)N);
  for (int64_t j = 0; j < N; ++j) S[j] = (float)lam[src[j]];
  HIP_TRY(c, hipMalloc((void**)&b.dsrc, sizeof(int) * (size_t)N));
  HIP_TRY(c, hipMemcpyAsync(b.dsrc, src.data(), sizeof(int) * (size_t)N, hipMemcpyHostToDevice, st));
  HIP_TRY(c, hipMalloc((void**)&b.dU, sizeof(float) * (size_t)N * (size_t)N));
  hipLaunchKernelGGL(jac_gather_kernel, dim3((unsigned)N), dim3(256), 0, st, b.V, (long long)np, (long long)N, b.dsrc, b.dU);
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, sync_stream(st));
  hipFree(b.V);
  b.V = nullptr;
  if (U_out) HIP_TRY(c, hipMemcpy(U_out, b.dU, sizeof(float) * (size_t)N * (size_t)N, hipMemcpyDeviceToHost));
  if (S_out) std::memcpy(S_out, S.data(), sizeof(float) * (size_t)N);
  if (info) {
    info->sweeps = total_sweeps;
    info->max_cosine = last;
    info->padded_order = np;
    info->shift = shift;
    info->max_residual = worst_resid;
  }
  if (install) return rvt_set_kinship(c, N, b.dU, S.data());
  return RVT_OK;
}

// ---- integer-plane GEMMs (rot_gemm.hip.h) ---------------------------------------------------------------------------
namespace {
constexpr int kRotMaxCols = 1 << 15;

struct QuantCols {       // digit planes of a set of columns, in a context-owned buffer
  signed char* d = nullptr;
  size_t plane_stride = 0;
  int planes = 0;
  std::vector<int> sexp;  // per column: entries were scaled by 2^sexp
};

int ensure_rot_scratch(rvt_ctx* c) {
  if (!c->d_rot_scale) {
    HIP_TRY(c, hipMalloc((void**)&c->d_rot_scale, sizeof(double) * 4 * kRotMaxCols));  // col | max | row | spare
    HIP_TRY(c, hipMalloc((void**)&c->d_rot_sexp, sizeof(int) * kRotMaxCols));
  }
  return RVT_OK;
}

// Quantise ncols <= kRotMaxCols columns of n_rows doubles (column-major, leading dimension ld_src) into planes laid out
// [plane][column (padded to `pad`)][ldk].  One plane when every column holds integers in [-127, 127], else kRotPlanesG.
int quantize_columns(rvt_ctx* c, const double* d_src, int64_t n_rows, int64_t ld_src, int ncols, int pad, int64_t ldk,
                     signed char** buf, size_t* cap, hipStream_t st, QuantCols* out, bool known_hard_calls = false) {
  int rc = ensure_rot_scratch(c);
  if (rc) return rc;
  double* d_max = c->d_rot_scale + kRotMaxCols;
  std::vector<double> cmax(ncols, 2.0);  // (hard calls: 0 / 1 / 2, no scan needed)
  if (!known_hard_calls) {
    hipLaunchKernelGGL(rot_colmax_kernel, dim3((unsigned)ncols), dim3(256), 0, st, d_src, (long long)n_rows,
                       (long long)ld_src, d_max);
    HIP_TRY(c, hipMemcpyAsync(cmax.data(), d_max, sizeof(double) * ncols, hipMemcpyDeviceToHost, st));
    HIP_TRY(c, sync_stream(st));
  }
  bool small = true;
  for (int j = 0; j < ncols; ++j) small = small && cmax[j] >= 0.0 && cmax[j] <= 127.0;
  const int PG = small ? 1 : kRotPlanesG;
  out->planes = PG;
  out->sexp.assign(ncols, 0);
  for (int j = 0; j < ncols; ++j) {
    const double mx = cmax[j] < 0.0 ? -cmax[j] - 1.0 : cmax[j];
    if (!std::isfinite(mx)) return fail(c, RVT_E_INVALID, "non-finite value in a column of an integer-plane product");
    out->sexp[j] = (PG == 1 || mx == 0.0) ? 0 : 7 * PG - 3 - std::ilogb(mx);
  }
  const int64_t cols_pad = ((int64_t)ncols + pad - 1) / pad * pad;
  out->plane_stride = (size_t)cols_pad * (size_t)ldk;
  const size_t need = out->plane_stride * PG;
  if (*cap < need) {
    if (*buf) hipFree(*buf);
    *buf = nullptr;
    *cap = 0;
    HIP_TRY(c, hipMalloc((void**)buf, need + need / 4));
    *cap = need + need / 4;
  }
  out->d = *buf;
  HIP_TRY(c, hipMemsetAsync(out->d, 0, need, st));
  HIP_TRY(c, hipMemcpyAsync(c->d_rot_sexp, out->sexp.data(), sizeof(int) * ncols, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(rot_quantize_f64_kernel, dim3(2048), dim3(256), 0, st, d_src, (long long)n_rows, (long long)ncols,
                     (long long)ld_src, c->d_rot_sexp, PG, out->d, (long long)ldk, (long long)out->plane_stride);
  HIP_TRY(c, sync_stream(st));  // d_rot_sexp / the host vectors are reused by the next call
  return RVT_OK;
}

// C[a + b * ldc] = sum_i A[i, a] B[i, b] from digit planes: nA rows (A columns), nB columns, K = n_rows samples.
// row_exp (host, may be null: uniform a_exp) / col_exp: binary scale exponents of the two sides.
int planes_gemm(rvt_ctx* c, const signed char* A, size_t a_stride, int PA, int nA, const int* row_exp, int a_exp,
                const signed char* B, size_t b_stride, int PB, int nB, const int* col_exp, int64_t n_rows, int64_t ldk,
                double* C, int64_t ldc, hipStream_t st, const int2* a_krange = nullptr) {
  int rc = ensure_rot_scratch(c);
  if (rc) return rc;
  std::vector<double> cs(nB), rs;
  for (int j = 0; j < nB; ++j) cs[j] = std::ldexp(1.0, -((row_exp ? 0 : a_exp) + (col_exp ? col_exp[j] : 0)));
  HIP_TRY(c, hipMemcpyAsync(c->d_rot_scale, cs.data(), sizeof(double) * nB, hipMemcpyHostToDevice, st));
  const double* d_rs = nullptr;
  if (row_exp) {
    rs.resize(nA);
    for (int a = 0; a < nA; ++a) rs[a] = std::ldexp(1.0, -row_exp[a]);
    HIP_TRY(c, hipMemcpyAsync(c->d_rot_scale + 2 * kRotMaxCols, rs.data(), sizeof(double) * nA, hipMemcpyHostToDevice, st));
    d_rs = c->d_rot_scale + 2 * kRotMaxCols;
  }
  const int nrp = (nA + kRotBM - 1) / kRotBM, nct = (nB + kRotBN - 1) / kRotBN;
  const long long sets = (long long)((nrp + 31) / 32) * ((nct + 7) / 8);
  const long long kbytes = (n_rows + kRotKC - 1) / kRotKC * kRotKC;
  // the kernel accumulates a whole K range in int32: |digit| <= 64 (several planes) or <= 127 (one plane of small
  // integers), so ranges longer than 2^31 / (bound_A bound_B) samples are cut and added in fp64
  const long long bound = (long long)(PA == 1 ? 127 : 64) * (PB == 1 ? 127 : 64);
  long long kmax = std::max<long long>(kRotKC, ((1LL << 31) - 1) / bound / kRotKC * kRotKC);
  if (const char* e = getenv("RVT_ROT_KMAX"))  // tests: force the cut on small problems
    kmax = std::max<long long>(kRotKC, std::min<long long>(kmax, atoll(e) / kRotKC * kRotKC));
  // Few output tiles (a short, wide product such as G'G of one MetaCov block: 16 tiles for 1024 x 1024) cannot fill
  // 256 CUs: K is then also split across workgroups (grid.y), every slice writes its own partial result and a
  // fixed-order reduction adds them.  Also used for the int32 range cut above.
  const long long tiles = (long long)nrp * nct;
  long long slices = 1;
  if (tiles < 256) slices = std::min<long long>((512 + tiles - 1) / tiles, std::max<long long>(1, kbytes / (16 * kRotKC)));
  slices = std::max(slices, (kbytes + kmax - 1) / kmax);
  if (const char* e = getenv("RVT_ROT_SLICES")) slices = std::max<long long>(1, atoll(e));
  long long kslice = ((kbytes + slices - 1) / slices + kRotKC - 1) / kRotKC * kRotKC;
  kslice = std::min(kslice, kmax);
  slices = (kbytes + kslice - 1) / kslice;
  double* d_part = nullptr;
  long long c_slice = 0;
  if (slices > 1) {
    c_slice = (long long)ldc * nB;
    const size_t need = sizeof(double) * (size_t)c_slice * (size_t)slices;
    if (c->rot_part_cap < need) {
      if (c->d_rot_part) hipFree(c->d_rot_part);
      c->d_rot_part = nullptr;
      c->rot_part_cap = 0;
      HIP_TRY(c, hipMalloc((void**)&c->d_rot_part, need));
      c->rot_part_cap = need;
    }
    d_part = c->d_rot_part;
  }
  if (a_krange && !row_exp) {
    // structured A: every plane of A inside one launch per plane of B, C written once (rot_gemm_i8_short_kernel)
    const int nct_s = (nB + kRotShortBN - 1) / kRotShortBN;
    const long long sets_s = (long long)((nrp + 31) / 32) * ((nct_s + 7) / 8);
    for (int q = 0; q < PB; ++q)
      hipLaunchKernelGGL(rot_gemm_i8_short, dim3((unsigned)(sets_s * 256)), dim3(512), 0, st, (const int8_t*)A,
                         (long long)a_stride, PA, (const int8_t*)(B + (size_t)q * b_stride), (long long)ldk, kbytes, C,
                         (long long)ldc, nA, nB, nrp, nct_s, c->d_rot_scale, std::ldexp(1.0, 7 * q), q > 0 ? 1 : 0, a_krange);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, sync_stream(st));
    return RVT_OK;
  }
  int first = 1;
  for (int sdeg = 0; sdeg <= (PA - 1) + (PB - 1); ++sdeg)  // least significant digit pairs first
    for (int p = 0; p < PA; ++p) {
      const int q = sdeg - p;
      if (q < 0 || q >= PB) continue;
      const dim3 grid((unsigned)(sets * 256), (unsigned)slices);
      if (slices == 1) {
        hipLaunchKernelGGL(rot_gemm_i8_kernel, grid, dim3(kRotThreads), 0, st, (const int8_t*)(A + (size_t)p * a_stride),
                           (const int8_t*)(B + (size_t)q * b_stride), (long long)ldk, kbytes, C, (long long)ldc, nA, nB, nrp,
                           nct, c->d_rot_scale, d_rs, std::ldexp(1.0, 7 * (p + q)), first ? 0 : 1, kbytes, 0LL);
      } else {
        hipLaunchKernelGGL(rot_gemm_i8_kernel, grid, dim3(kRotThreads), 0, st, (const int8_t*)(A + (size_t)p * a_stride),
                           (const int8_t*)(B + (size_t)q * b_stride), (long long)ldk, kbytes, d_part, (long long)ldc, nA, nB,
                           nrp, nct, c->d_rot_scale, d_rs, std::ldexp(1.0, 7 * (p + q)), 0, kslice, c_slice);
        hipLaunchKernelGGL(rot_reduce_slices_kernel, dim3(1024), dim3(256), 0, st, d_part, (long long)ldc, (long long)nA,
                           (long long)nB, c_slice, (int)slices, C, first ? 0 : 1);
      }
      first = 0;
    }
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, sync_stream(st));  // (the scale arrays are reused by the next call)
  return RVT_OK;
}
}  // namespace

// dst (N x ncols doubles, leading dimension ld_dst) = U' src (src: N x ncols doubles, leading dimension ld_src), exactly
// as the integer products of the digit planes.  Columns of small integers (hard calls after the flip, collapsed burden
// columns) are one digit plane; any other batch is quantised to kRotPlanesG digits per column.
// (planes_gemm for the other translation units: MetaCov's hard-call band)
int rvt_planes_gemm(rvt_ctx* c, const signed char* A, size_t a_stride, int PA, int nA, const int* row_exp, int a_exp,
                    const signed char* B, size_t b_stride, int PB, int nB, const int* col_exp, int64_t n_rows, int64_t ldk,
                    double* C, int64_t ldc, hipStream_t st, const int2* a_krange) {
  return planes_gemm(c, A, a_stride, PA, nA, row_exp, a_exp, B, b_stride, PB, nB, col_exp, n_rows, ldk, C, ldc, st, a_krange);
}
int rotate_columns(rvt_ctx* c, const double* d_src, int64_t ld_src, int ncols, double* d_dst, int64_t ld_dst,
                          hipStream_t st) {
  const int64_t N = c->kin_N;
  if (c->d_csc_ptr && !c->d_uq_range) {  // sparse U, scattered supports: a gather per output (fp64 products and sums)
    for (int c0 = 0; c0 < ncols; c0 += 65535) {  // (gridDim.y holds at most 65535 columns)
      const int nc = std::min(65535, ncols - c0);
      hipLaunchKernelGGL(rot_sparse_kernel, dim3((unsigned)((N + 255) / 256), (unsigned)nc), dim3(256), 0, st,
                         c->d_csc_ptr, c->d_csc_rows, c->d_csc_vals, (long long)N, d_src + (size_t)c0 * ld_src,
                         (long long)ld_src, d_dst + (size_t)c0 * ld_dst, (long long)ld_dst);
      HIP_TRY(c, hipGetLastError());
    }
    return RVT_OK;
  }
  for (int c0 = 0; c0 < ncols; c0 += kRotMaxCols) {  // long lists in pieces
    const int nc = std::min(kRotMaxCols, ncols - c0);
    QuantCols qb;
    int rc = quantize_columns(c, d_src + (size_t)c0 * ld_src, N, ld_src, nc, kRotBN, c->uq_ldk, &c->d_rotB, &c->rotB_cap, st,
                              &qb);
    if (rc) return rc;
    rc = planes_gemm(c, c->d_Uq, c->uq_plane, kRotPlanesU, (int)N, nullptr, c->uq_sexp, qb.d, qb.plane_stride, qb.planes,
                     nc, qb.sexp.data(), N, c->uq_ldk, d_dst + (size_t)c0 * ld_dst, ld_dst, st, c->d_uq_range);
    if (rc) return rc;
  }
  return RVT_OK;
}

// C (nA x nB, column-major, leading dimension ldc) = A' B for two double matrices with n_rows rows (column-major) through
// the integer planes: exact when both hold small integers, else to ~2^-40 relative of each column's largest entry.
int gemm_tn_planes(rvt_ctx* c, const double* dA, int64_t ldA, int nA, const double* dB, int64_t ldB, int nB,
                          int64_t n_rows, double* C, int64_t ldc, hipStream_t st) {
  if (nA > kRotMaxCols || nB > kRotMaxCols) return fail(c, RVT_E_TOO_LARGE, "integer-plane product: too many columns");
  const int64_t ldk = (n_rows + 127) / 128 * 128;
  QuantCols qa, qb;
  int rc = quantize_columns(c, dA, n_rows, ldA, nA, kRotBM, ldk, &c->d_rotA, &c->rotA_cap, st, &qa);
  if (rc) return rc;
  rc = quantize_columns(c, dB, n_rows, ldB, nB, kRotBN, ldk, &c->d_rotB, &c->rotB_cap, st, &qb);
  if (rc) return rc;
  return planes_gemm(c, qa.d, qa.plane_stride, qa.planes, nA, qa.sexp.data(), 0, qb.d, qb.plane_stride, qb.planes, nB,
                     qb.sexp.data(), n_rows, ldk, C, ldc, st);
}

namespace {
// GSL 1.16 Brent minimiser exactly as Minimizer::minimize drives it (regression/GSLMinimizer.cpp:18-66: set, then
// iterate until the bracket is narrower than epsabs = 1e-3 or 100 iterations).  The evaluation SEQUENCE matters:
// the reference's beta / sigma2 are side effects of the last evaluation (regression/FastLMM.cpp:812-817).
int brent_like_gsl(const std::function<double(double)>& f, double start, double lb, double ub, double* xmin) {
  const double golden = 0.3819660, sqrt_eps = 1.4901161193847656e-08;
  double xl = lb, xu = ub, xm = start;
  const double fl = f(xl);
  if (!std::isfinite(fl)) return -1;
  const double fu = f(xu);
  if (!std::isfinite(fu)) return -1;
  double fm = f(xm);
  if (!std::isfinite(fm)) return -1;
  if (xl > xu || xm >= xu || xm <= xl || fm >= fl || fm >= fu) return -1;
  double v = xl + golden * (xu - xl), w = v, st_d = 0, st_e = 0;
  double fv = f(v);
  if (!std::isfinite(fv)) return -1;
  double fw = fv;
  for (int iter = 1;; ++iter) {
    const double z = xm, fz = fm;
    double d = st_e, e = st_d;  // the roles of the two saved steps are exchanged on entry, as in GSL
    const double w_lower = z - xl, w_upper = xu - z, tol = sqrt_eps * std::fabs(z), mid = 0.5 * (xl + xu);
    double p = 0, q = 0, r = 0;
    if (std::fabs(e) > tol) {  // parabola through (v, w, z)
      r = (z - w) * (fz - fv);
      q = (z - v) * (fz - fw);
      p = (z - v) * q - (z - w) * r;
      q = 2 * (q - r);
      if (q > 0)
        p = -p;
      else
        q = -q;
      r = e;
      e = d;
    }
    double u;
    if (std::fabs(p) < std::fabs(0.5 * q * r) && p < q * w_lower && p < q * w_upper) {
      d = p / q;
      u = z + d;
      if ((u - xl) < 2 * tol || (xu - u) < 2 * tol) d = (z < mid) ? tol : -tol;
    } else {  // golden section into the larger part
      e = (z < mid) ? xu - z : -(z - xl);
      d = golden * e;
    }
    u = (std::fabs(d) >= tol) ? z + d : z + ((d > 0) ? tol : -tol);
    st_e = e;
    st_d = d;
    const double fuu = f(u);
    if (!std::isfinite(fuu)) return -1;
    if (fuu <= fz) {
      if (u < z)
        xu = z;
      else
        xl = z;
      v = w;
      fv = fw;
      w = z;
      fw = fz;
      xm = u;
      fm = fuu;
    } else {
      if (u < z)
        xl = u;
      else
        xu = u;
      if (fuu <= fw || w == z) {
        v = w;
        fv = fw;
        w = u;
        fw = fuu;
      } else if (fuu <= fv || v == z || v == w) {
        v = u;
        fv = fuu;
      }
    }
    *xmin = xm;
    if (std::fabs(xu - xl) < 0.001 || iter >= 100) return 0;
  }
}
}  // namespace

int rvt_fit_fam_null(rvt_ctx* c, int64_t N, int d, const double* X, const double* y, rvt_fam_null* out) {
  if (!c || !X || !y || !out || d < 1 || d + 2 > RVT_MAX_COV) return fail(c, RVT_E_INVALID, "bad arguments");
  if (!c->have_kin || c->kin_N != N) return fail(c, RVT_E_STATE, "rvt_set_kinship with the same N first");
  if (c->have_null && c->nc.N != N) return fail(c, RVT_E_STATE, "sample count differs from the installed null model");
  hipSetDevice(c->device);
  int rc = rvt_sync(c);
  if (rc) return rc;
  hipStream_t st = c->stream;
  const int64_t ld = rvt_padded_ld(N);
  for (double** p : {&c->d_uxy, &c->d_lmm_part, &c->d_fX, &c->d_frr, &c->d_fv, &c->d_fzeros, &c->d_fbeta}) {
    if (*p) hipFree(*p);
    *p = nullptr;
  }
  c->have_fam = false;
  c->famcov_b2 = 1.0;
  const int dx = d + 1;
  double* d_xy = nullptr;  // N x (d+1): X | y
  HIP_TRY(c, hipMalloc((void**)&d_xy, sizeof(double) * (size_t)N * dx));
  HIP_TRY(c, hipMalloc((void**)&c->d_uxy, sizeof(double) * (size_t)N * dx));
  HIP_TRY(c, hipMemcpy(d_xy, X, sizeof(double) * (size_t)N * d, hipMemcpyHostToDevice));
  HIP_TRY(c, hipMemcpy(d_xy + (size_t)N * d, y, sizeof(double) * (size_t)N, hipMemcpyHostToDevice));
  {  // ux = U'X, uy = U'y  (FastLMM.cpp:55-57)
    int rcr = rotate_columns(c, d_xy, N, dx, c->d_uxy, N, st);
    if (rcr) {
      hipFree(d_xy);
      return rcr;
    }
    HIP_TRY(c, sync_stream(st));
  }
  hipFree(d_xy);
  // |lambda| for the likelihood (FastLMM.cpp:50); the raw S stays in d_S for FamSkat's Sigma
  std::vector<double> absS(N);
  for (int64_t i = 0; i < N; ++i) absS[i] = std::fabs(c->h_S[i]);
  double* d_abs = nullptr;
  HIP_TRY(c, hipMalloc((void**)&d_abs, sizeof(double) * N));
  HIP_TRY(c, hipMemcpy(d_abs, absS.data(), sizeof(double) * N, hipMemcpyHostToDevice));
  const int rec = lmm_rec_len(d);
  HIP_TRY(c, hipMalloc((void**)&c->d_lmm_part, sizeof(double) * (size_t)kLmmBlocks * rec));
  std::vector<double> part((size_t)kLmmBlocks * rec), sums(rec);
  std::vector<double> beta(d, 0.0);
  double sigma2 = 0.0;
  bool hip_failed = false;
  // device sums for one delta: A, b, yy, sum log|lambda + delta|
  auto device_sums = [&](const double* lam, double delta, int take_abs) {
    hipLaunchKernelGGL(lmm_sums_kernel, dim3(kLmmBlocks), dim3(256), sizeof(double) * 256, st, c->d_uxy, lam,
                       (long long)N, d, delta, take_abs, c->d_lmm_part);
    if (hipMemcpyAsync(part.data(), c->d_lmm_part, sizeof(double) * part.size(), hipMemcpyDeviceToHost, st) !=
            hipSuccess ||
        sync_stream(st) != hipSuccess)
      hip_failed = true;
    for (int q = 0; q < rec; ++q) {
      double s = 0.0;
      for (int b = 0; b < kLmmBlocks; ++b) s += part[(size_t)b * rec + q];  // fixed order
      sums[q] = s;
    }
  };
  // getBetaSigma2 + getLogLikelihood for one delta (FastLMM.cpp:297-346, model MLE); returns the log-likelihood
  auto evaluate = [&](double delta) {
    device_sums(d_abs, delta, 1);
    const double* A = sums.data();
    const double* b = A + d * d;
    const double yy = sums[d * d + d], slog = sums[d * d + d + 1];
    double Ai[RVT_MAX_COV * RVT_MAX_COV];
    if (!invert_spd(A, d, Ai)) return (double)NAN;
    for (int a = 0; a < d; ++a) {
      double s = 0.0;
      for (int k = 0; k < d; ++k) s += Ai[a * d + k] * b[k];
      beta[a] = s;
    }
    // sum (uy - ux beta)^2 / (lambda + delta) = yy - 2 beta'b + beta'A beta
    double bb = 0.0, bAb = 0.0;
    for (int a = 0; a < d; ++a) {
      bb += beta[a] * b[a];
      for (int k = 0; k < d; ++k) bAb += beta[a] * A[a * d + k] * beta[k];
    }
    sigma2 = (yy - 2.0 * bb + bAb) / (double)N;
    const double n = (double)N;
    return -0.5 * (n * std::log(2.0 * 3.14159265358979323846) + slog + n + n * std::log(sigma2));
  };
  int maxIndex = -1;
  double maxLL = 0.0, delta = 0.0;
  for (int i = 0; i <= 100; ++i) {
    delta = std::exp(-10. + i * 0.2);
    const double ll = evaluate(delta);
    if (std::isnan(ll)) continue;
    if (maxIndex < 0 || ll > maxLL) {
      maxIndex = i;
      maxLL = ll;
    }
  }
  int evals = 0;
  if (maxIndex > 0 && maxIndex < 100) {
    const double lb = std::exp(-10. + (maxIndex - 1) * 0.2), ub = std::exp(-10. + (maxIndex + 1) * 0.2);
    const double start = std::exp(-10. + maxIndex * 0.2);
    double xmin = start;
    auto goal = [&](double x) {
      ++evals;
      return -evaluate(x);
    };
    delta = brent_like_gsl(goal, start, lb, ub, &xmin) ? start : xmin;
  }  // else: on the boundary delta (and beta, sigma2) stay at the LAST grid point, as in the reference
  if (hip_failed) {
    hipFree(d_abs);
    return fail(c, RVT_E_HIP, "device evaluation of the FastLMM likelihood failed");
  }
  out->delta = delta;
  out->sigma2_g = sigma2;
  c->fam_delta = delta;
  std::memset(out->beta, 0, sizeof(out->beta));
  for (int a = 0; a < d; ++a) out->beta[a] = beta[a];
  out->max_index = maxIndex;
  out->brent_evals = evals;
  // ---- what FamSkat::FitNullModel prepares (FamSkat.cpp:34-64), in rotated / folded form --------------------
  NullConsts& fn = c->fam_nc;
  std::memset(&fn, 0, sizeof(fn));
  fn.N = N;
  fn.ld = ld;
  fn.d = dx;
  fn.binary = 1;  // the sufficient statistics are weighted by V = sigma2 (S + delta)
  fn.sigma2 = 1.0;
  {  // C = X' Sigma^-1 X = sum ux ux' / (sigma2 (S + delta)) with the RAW S (FamSkat.cpp:48-56)
    device_sums(c->d_S, delta, 0);
    double C[RVT_MAX_COV * RVT_MAX_COV], Ci[RVT_MAX_COV * RVT_MAX_COV];
    for (int a = 0; a < d * d; ++a) C[a] = sums[a] / sigma2;
    if (hip_failed || !invert_spd(C, d, Ci)) {
      hipFree(d_abs);
      return fail(c, RVT_E_INVALID, "X' Sigma^-1 X is singular");
    }
    for (int a = 0; a < dx; ++a)
      for (int b = 0; b < dx; ++b) {
        const bool in = a < d && b < d;
        fn.C[a * dx + b] = in ? C[a * d + b] : (a == b ? 1.0 : 0.0);
        fn.Cinv[a * dx + b] = in ? Ci[a * d + b] : (a == b ? 1.0 : 0.0);
      }
  }
  hipFree(d_abs);
  {  // denom of FastGetAF: u1' |S|^-1 u1 (FastLMM.cpp:414-420), kept in the otherwise unused rss slot
    double den = 0.0;
    for (int64_t i = 0; i < N; ++i) den += c->h_u1[i] / std::fabs(c->h_S[i]) * c->h_u1[i];
    fn.rss = den;
  }
  const size_t vb = sizeof(double) * (size_t)ld;
  HIP_TRY(c, hipMalloc((void**)&c->d_fX, vb * dx));
  HIP_TRY(c, hipMalloc((void**)&c->d_frr, vb));
  HIP_TRY(c, hipMalloc((void**)&c->d_fv, vb));
  HIP_TRY(c, hipMalloc((void**)&c->d_fzeros, vb));
  HIP_TRY(c, hipMalloc((void**)&c->d_fbeta, sizeof(double) * RVT_MAX_COV));
  HIP_TRY(c, hipMemsetAsync(c->d_fX, 0, vb * dx, st));
  HIP_TRY(c, hipMemsetAsync(c->d_frr, 0, vb, st));
  HIP_TRY(c, hipMemsetAsync(c->d_fv, 0, vb, st));
  HIP_TRY(c, hipMemsetAsync(c->d_fzeros, 0, vb, st));
  HIP_TRY(c, hipMemcpyAsync(c->d_fbeta, out->beta, sizeof(double) * RVT_MAX_COV, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(fam_build_null_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, c->d_uxy, c->d_S,
                     c->d_u1, (long long)N, (long long)ld, d, sigma2, delta, c->d_fbeta, c->d_fX, c->d_frr, c->d_fv);
  if (!c->d_fam_nc) HIP_TRY(c, hipMalloc((void**)&c->d_fam_nc, sizeof(NullConsts)));
  HIP_TRY(c, hipMemcpyAsync(c->d_fam_nc, &fn, sizeof(NullConsts), hipMemcpyHostToDevice, st));
  // ---- the family MetaCov's constants and null set (MetaCovFamQtl over FastLMM::GetCov*, FastLMM.cpp:510-625) ----
  {
    // [U'X | u1] with weights 1/|lambda + delta|: A = ux'W ux, b = ux'W u1, yy = u1'W u1
    double* d_xu = nullptr;
    HIP_TRY(c, hipMalloc((void**)&d_xu, sizeof(double) * (size_t)N * dx));
    HIP_TRY(c, hipMemcpyAsync(d_xu, c->d_uxy, sizeof(double) * (size_t)N * d, hipMemcpyDeviceToDevice, st));
    HIP_TRY(c, hipMemcpyAsync(d_xu + (size_t)N * d, c->d_u1, sizeof(double) * (size_t)N, hipMemcpyDeviceToDevice, st));
    double* d_abs2 = nullptr;
    HIP_TRY(c, hipMalloc((void**)&d_abs2, sizeof(double) * N));
    HIP_TRY(c, hipMemcpyAsync(d_abs2, absS.data(), sizeof(double) * N, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(lmm_sums_kernel, dim3(kLmmBlocks), dim3(256), sizeof(double) * 256, st, d_xu, d_abs2,
                       (long long)N, d, delta, 1, c->d_lmm_part);
    HIP_TRY(c, hipMemcpyAsync(part.data(), c->d_lmm_part, sizeof(double) * part.size(), hipMemcpyDeviceToHost, st));
    HIP_TRY(c, sync_stream(st));
    hipFree(d_xu);
    hipFree(d_abs2);
    for (int q = 0; q < rec; ++q) {
      double s2 = 0.0;
      for (int b = 0; b < kLmmBlocks; ++b) s2 += part[(size_t)b * rec + q];
      sums[q] = s2;
    }
    for (int a = 0; a < d * d; ++a) c->famcov_zz[a] = sums[a] / sigma2;
    for (int a = 0; a < d; ++a) c->famcov_c1x[a] = sums[d * d + a] / sigma2;
    c->famcov_c11 = sums[d * d + d] / sigma2;
    if (!invert_spd(c->famcov_zz, d, c->famcov_zzinv)) return fail(c, RVT_E_INVALID, "covZZ is singular");
    {  // k1r = u1' D uResid = (u1'W uy - (ux'W u1)' beta) / sigma2 : one more reduction over [u1 | uy]
      double* d_uy2 = nullptr;
      HIP_TRY(c, hipMalloc((void**)&d_uy2, sizeof(double) * (size_t)N * 2));
      HIP_TRY(c, hipMemcpyAsync(d_uy2, c->d_u1, sizeof(double) * (size_t)N, hipMemcpyDeviceToDevice, st));
      HIP_TRY(c, hipMemcpyAsync(d_uy2 + (size_t)N, c->d_uxy + (size_t)N * d, sizeof(double) * (size_t)N,
                                hipMemcpyDeviceToDevice, st));
      double* d_abs3 = nullptr;
      HIP_TRY(c, hipMalloc((void**)&d_abs3, sizeof(double) * N));
      HIP_TRY(c, hipMemcpyAsync(d_abs3, absS.data(), sizeof(double) * N, hipMemcpyHostToDevice, st));
      hipLaunchKernelGGL(lmm_sums_kernel, dim3(kLmmBlocks), dim3(256), sizeof(double) * 256, st, d_uy2, d_abs3,
                         (long long)N, 1, delta, 1, c->d_lmm_part);
      const int rec1 = lmm_rec_len(1);
      std::vector<double> p1((size_t)kLmmBlocks * rec1);
      HIP_TRY(c, hipMemcpyAsync(p1.data(), c->d_lmm_part, sizeof(double) * p1.size(), hipMemcpyDeviceToHost, st));
      HIP_TRY(c, sync_stream(st));
      hipFree(d_uy2);
      hipFree(d_abs3);
      double u1Wy = 0.0;
      for (int b = 0; b < kLmmBlocks; ++b) u1Wy += p1[(size_t)b * rec1 + 1];  // b[0] = u1' W uy
      double k = u1Wy;
      for (int a = 0; a < d; ++a) k -= sums[d * d + a] * beta[a];
      c->famcov_k1r = k / sigma2;
    }
    for (double** pp : {&c->d_cX, &c->d_cv, &c->d_cr}) {
      if (*pp) hipFree(*pp);
      *pp = nullptr;
    }
    const int dc = d + 2;  // U'X | u1 | allele-frequency column
    HIP_TRY(c, hipMalloc((void**)&c->d_cX, vb * dc));
    HIP_TRY(c, hipMalloc((void**)&c->d_cv, vb));
    HIP_TRY(c, hipMalloc((void**)&c->d_cr, vb));
    HIP_TRY(c, hipMemsetAsync(c->d_cX, 0, vb * dc, st));
    HIP_TRY(c, hipMemsetAsync(c->d_cv, 0, vb, st));
    HIP_TRY(c, hipMemsetAsync(c->d_cr, 0, vb, st));
    hipLaunchKernelGGL(famcov_build_null_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, c->d_uxy,
                       c->d_S, c->d_u1, (long long)N, (long long)ld, d, sigma2, delta, c->d_fbeta, c->d_cX, c->d_cr,
                       c->d_cv);
    NullConsts& cn = c->famcov_nc;
    std::memset(&cn, 0, sizeof(cn));
    cn.N = N;
    cn.ld = ld;
    cn.d = dc;
    cn.binary = 1;
    cn.sigma2 = 1.0;
    for (int a = 0; a < dc; ++a) cn.C[a * dc + a] = cn.Cinv[a * dc + a] = 1.0;  // unused by the covariance kernels
    if (!c->d_famcov_nc) HIP_TRY(c, hipMalloc((void**)&c->d_famcov_nc, sizeof(NullConsts)));
    HIP_TRY(c, hipMemcpyAsync(c->d_famcov_nc, &cn, sizeof(NullConsts), hipMemcpyHostToDevice, st));
  }
  HIP_TRY(c, sync_stream(st));
  c->have_fam = true;
  return RVT_OK;
}

// Run V already ROTATED columns (d_rot, leading dimension ld) through the sufficient statistics with the family
// covariance null set and finish with the covariance kernels in family mode.  d_cs / d_poly: raw (unrotated) column
// sums and polymorphic flags on the device.
namespace {
int famcov_run(rvt_ctx* c, const double* d_rot, int V, const double* d_cs, const int* d_poly, CovOut* co) {
  const int64_t ld = c->fam_nc.ld;
  std::vector<double> af(V, 0.01);
  rvt_gene_result r;
  co->fam = true;
  co->d_raw_colsum = d_cs;
  co->d_raw_poly = d_poly;
  const double* p = d_rot;
  // the batch code reads the null set from the context: install the family-covariance set for this call
  NullConsts keep_nc = c->nc;
  NullConsts* keep_dnc = c->d_nc;
  double *kX = c->d_X, *kres = c->d_res, *krr = c->d_rr, *kv = c->d_v, *kz = c->d_zeros;
  const bool khave = c->have_null;
  const int64_t kld = c->null_ld;
  c->nc = c->famcov_nc;
  c->d_nc = c->d_famcov_nc;
  c->d_X = c->d_cX;
  c->d_res = c->d_fzeros;
  c->d_rr = c->d_cr;
  c->d_v = c->d_cv;
  c->d_zeros = c->d_fzeros;
  c->have_null = true;
  c->null_ld = ld;
  int rc = run_batch(c, 1, &p, &V, af.data(), nullptr, 0u, nullptr, &r, nullptr, co);
  c->nc = keep_nc;
  c->d_nc = keep_dnc;
  c->d_X = kX;
  c->d_res = kres;
  c->d_rr = krr;
  c->d_v = kv;
  c->d_zeros = kz;
  c->have_null = khave;
  c->null_ld = kld;
  for (auto& sl : c->slots)
    if (sl.pending_out == &r) {
      sl.pending_out = nullptr;
      sl.pending_n = 0;
    }
  return rc;
}
}  // namespace

// Raw column statistics + rotation by U' of one block of <= RVT_MAX_VARIANTS raw columns, then famcov_run.
static int fam_block_run(rvt_ctx* c, const double* dG, int V, CovOut* cop) {
  if (V > RVT_MAX_VARIANTS) return fail(c, RVT_E_TOO_LARGE, "block of %d variants exceeds RVT_MAX_VARIANTS", V);
  if (!c->have_fam) return fail(c, RVT_E_STATE, "rvt_set_kinship + rvt_fit_fam_null first");
  hipSetDevice(c->device);
  int rc = rvt_sync(c);
  if (rc) return rc;
  hipStream_t st = c->stream;
  const int64_t N = c->fam_nc.N, ld = c->fam_nc.ld;
  rc = ensure_fam_cols(c, (size_t)V, ld);
  if (rc) return rc;
  double* d_cs = nullptr;
  int* d_poly = nullptr;
  HIP_TRY(c, hipMalloc((void**)&d_cs, sizeof(double) * (size_t)V));
  HIP_TRY(c, hipMalloc((void**)&d_poly, sizeof(int) * (size_t)V));
  struct Guard {
    void *a, *b;
    ~Guard() {
      hipFree(a);
      hipFree(b);
    }
  } guard{(void*)d_cs, (void*)d_poly};
  hipLaunchKernelGGL(raw_colstat_kernel, dim3((unsigned)V), dim3(256), 0, st, dG, (long long)N, (long long)ld, d_cs,
                     d_poly);
  HIP_TRY(c, hipMemsetAsync(c->d_Gt, 0, sizeof(double) * (size_t)ld * V, st));
  {
    int rcr = rotate_columns(c, dG, ld, V, c->d_Gt, ld, st);
    if (rcr) return rcr;
  }
  HIP_TRY(c, sync_stream(st));
  return famcov_run(c, c->d_Gt, V, d_cs, d_poly, cop);
}

// MetaCov with kinship (quantitative): rotate the block, then famcov_run.
int rvt_cov_block_fam(rvt_ctx* c, const double* dG, int V, double* cov, double* xz, double* zz, int* polymorphic) {
  if (!c || !dG || V < 1 || !cov || !xz || !polymorphic) return fail(c, RVT_E_INVALID, "bad arguments");
  CovOut co;
  co.cov = cov;
  co.xz = xz;
  co.zz = zz;
  co.poly = polymorphic;
  int rc = fam_block_run(c, dG, V, &co);
  if (!rc && c->famcov_b2 != 1.0) {  // MetaCovFamBinary: covXX, covXZ, covZZ each carry b^2 (Model.cpp:651-668)
    const double b2 = c->famcov_b2;
    const int du = c->famcov_nc.d - 2;
    for (int h = 0; h < V; ++h)
      for (int j = h; j < V; ++j) cov[(size_t)h + (size_t)j * V] *= b2;
    for (size_t i = 0; i < (size_t)V * du; ++i) xz[i] *= b2;
    if (zz)
      for (int i = 0; i < du * du; ++i) zz[i] *= b2;
  }
  return rc;
}

// FamAnalyticVT (AnalyticVT(RELATED), src/Model.h:2189-2214): af_i = FastLMM::FastGetAF of column i of the flipped,
// polymorphic genotype, (u, v) = FastLMM::CalculateUandV (regression/FastLMM.cpp:259-291: u = (U'g_c)' (lambda + delta)^-1
// uResid / sigma2, v = (U'g_c)' scaledK (U'g_c) / sigma2), then MultivariateVT::compute.  u, v and af of the RAW columns
// come from the family-covariance machinery (ustat / band / GLS frequency of fam_block_run); flipping a column to its
// minor allele (g -> 2 - g) changes the sign of its centred genotype and maps af to 1 - af, monomorphic columns drop out.
int rvt_fam_analytic_vt(rvt_ctx* c, int n, const double* const* dG, const int* Ms, rvt_gene_result* out) {
  if (!c || n < 0 || (n > 0 && (!dG || !Ms || !out))) return fail(c, RVT_E_INVALID, "bad batch arguments");
  if (!c->have_fam) return fail(c, RVT_E_STATE, "rvt_set_kinship + rvt_fit_fam_null first");
  hipSetDevice(c->device);
  const int64_t N = c->fam_nc.N, ld = c->fam_nc.ld;
  for (int g = 0; g < n; ++g) {
    const int M = Ms[g];
    rvt_gene_result& r = out[g];
    std::memset(&r, 0, sizeof(r));
    r.gene_id = g;
    r.n_variants = M;
    if (M < 1 || M > RVT_MAX_VARIANTS) return fail(c, RVT_E_INVALID, "gene %d has M=%d", g, M);
    int rc = rvt_sync(c);
    if (rc) return rc;
    hipStream_t st = c->stream;
    // flip / polymorphic decisions on the raw columns (DataConsolidator.cpp:46-69,94-116)
    std::vector<const double*> cols(M);
    for (int j = 0; j < M; ++j) cols[j] = dG[g] + (size_t)j * ld;
    const double** d_cols = nullptr;
    int* d_flags = nullptr;
    double* d_buf = nullptr;
    rvt_gene_result* d_res = nullptr;
    struct Guard {
      std::vector<void**> p;
      ~Guard() {
        for (void** q : p)
          if (*q) hipFree(*q);
      }
    } guard{{(void**)&d_cols, (void**)&d_flags, (void**)&d_buf, (void**)&d_res}};
    HIP_TRY(c, hipMalloc((void**)&d_cols, sizeof(double*) * (size_t)M));
    HIP_TRY(c, hipMalloc((void**)&d_flags, sizeof(int) * (size_t)M));
    HIP_TRY(c, hipMemcpyAsync(d_cols, cols.data(), sizeof(double*) * M, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(fam_colstat_kernel, dim3((unsigned)M), dim3(256), 0, st, d_cols, (long long)N, d_flags);
    std::vector<int> flags(M);
    HIP_TRY(c, hipMemcpyAsync(flags.data(), d_flags, sizeof(int) * M, hipMemcpyDeviceToHost, st));
    HIP_TRY(c, sync_stream(st));
    std::vector<double> cov((size_t)M * M), xz((size_t)M * RVT_MAX_COV), ustat(M), vstat(M), af(M), pval(M);
    std::vector<int> poly(M);
    CovOut co;
    co.cov = cov.data();
    co.xz = xz.data();
    co.poly = poly.data();
    co.ustat = ustat.data();
    co.vstat = vstat.data();
    co.af = af.data();
    co.pval = pval.data();
    rc = fam_block_run(c, dG[g], M, &co);
    if (rc) return rc;
    std::vector<int> kept;
    for (int j = 0; j < M; ++j)
      if (flags[j] & 2) kept.push_back(j);
    const int m = (int)kept.size();
    r.n_poly = m;
    if (m == 0) continue;  // genotype.cols == 0 -> NA row
    const int Mp = (m + 15) / 16 * 16;
    const size_t nbuf = 2 * (size_t)m + (size_t)m * m + gene_vt_doubles(Mp);
    std::vector<double> host(2 * (size_t)m + (size_t)m * m);
    for (int a = 0; a < m; ++a) {
      const int j = kept[a];
      const double sa = (flags[j] & 1) ? -1.0 : 1.0;
      host[a] = (flags[j] & 1) ? 1.0 - af[j] : af[j];
      host[m + a] = sa * ustat[j];
      for (int b = 0; b < m; ++b) {
        const int k = kept[b];
        const double sb = (flags[k] & 1) ? -1.0 : 1.0;
        const double v = j <= k ? cov[(size_t)j + (size_t)k * M] : cov[(size_t)k + (size_t)j * M];
        host[2 * (size_t)m + (size_t)b * m + a] = sa * sb * v;
      }
    }
    HIP_TRY(c, hipMalloc((void**)&d_buf, sizeof(double) * nbuf));
    HIP_TRY(c, hipMalloc((void**)&d_res, sizeof(rvt_gene_result)));
    HIP_TRY(c, hipMemcpyAsync(d_buf, host.data(), sizeof(double) * host.size(), hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipMemcpyAsync(d_res, &r, sizeof(r), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(vt_direct_kernel, dim3(1), dim3(256), 0, st, m, Mp, d_buf, d_res);
    {
      GeneDesc gdv;
      std::memset(&gdv, 0, sizeof(gdv));
      gdv.Mp = Mp;
      gdv.vt_mem = d_buf + 2 * (size_t)m + (size_t)m * m;
      gdv.result = d_res;
      GeneDesc* d_gdv = nullptr;
      HIP_TRY(c, hipMalloc((void**)&d_gdv, sizeof(GeneDesc)));
      hipError_t e = hipMemcpyAsync(d_gdv, &gdv, sizeof(gdv), hipMemcpyHostToDevice, st);
      for (int stage = 0; stage < 2 && e == hipSuccess; ++stage) {
        k_vt_integrate(dim3(1, kMvnShifts), st, d_gdv, stage);
        k_vt_finish(dim3(1), st, d_gdv, 1, stage);
      }
      if (e == hipSuccess) e = sync_stream(st);
      hipFree(d_gdv);
      HIP_TRY(c, e);
    }
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(&r, d_res, sizeof(r), hipMemcpyDeviceToHost, st));
    HIP_TRY(c, sync_stream(st));
  }
  return RVT_OK;
}

// MetaScore with kinship: FastLMM score test of every raw column, the block in pieces of RVT_MAX_VARIANTS.
// binary = 0: MetaFamQtl; binary = 1: MetaFamBinary (genotype not centred; U b, V b^2 with the b of rvt_fam_binary_scale).
int rvt_score_block_fam(rvt_ctx* c, const double* dG, int V, int binary, int* ok, double* ustat, double* vstat,
                        double* af, double* pvalue) {
  if (!c || !dG || V < 1 || !ok || !ustat || !vstat || !af || !pvalue) return fail(c, RVT_E_INVALID, "bad arguments");
  if (!c->have_fam) return fail(c, RVT_E_STATE, "rvt_set_kinship + rvt_fit_fam_null first");
  const int64_t ld = c->fam_nc.ld;
  for (int c0 = 0; c0 < V; c0 += RVT_MAX_VARIANTS) {
    const int n = std::min(RVT_MAX_VARIANTS, V - c0);
    CovOut co;
    co.poly = ok + c0;
    co.ustat = ustat + c0;
    co.vstat = vstat + c0;
    co.af = af + c0;
    co.pval = pvalue + c0;
    co.uncentred = binary != 0;
    int rc = fam_block_run(c, dG + (size_t)c0 * ld, n, &co);
    if (rc) return rc;
  }
  if (binary) {  // MetaFamBinary::GetU / GetV (src/Model.h:3647-3648); the p-value is the unscaled statistic's
    const double b2 = c->famcov_b2, b = std::sqrt(b2);
    for (int h = 0; h < V; ++h) {
      ustat[h] *= b;
      vstat[h] *= b2;
    }
  }
  return RVT_OK;
}

// FastLMM::GetNullCovB (regression/FastLMM.cpp:473-483) as MetaFamQtl::PrintNullModel prints it: the diagonal of
// (ux' diag(lambda + delta) ux)^-1 — literally the reference's expression (it multiplies by lambda + delta where its
// own comment derives the inverse weights).
int rvt_fam_null_summary(rvt_ctx* c, double* covb_diag) {
  if (!c || !covb_diag) return fail(c, RVT_E_INVALID, "bad arguments");
  if (!c->have_fam) return fail(c, RVT_E_STATE, "rvt_set_kinship + rvt_fit_fam_null first");
  hipSetDevice(c->device);
  int rc = rvt_sync(c);
  if (rc) return rc;
  const int64_t N = c->fam_nc.N;
  const int d = c->fam_nc.d - 1;
  std::vector<double> ux((size_t)N * d);
  HIP_TRY(c, hipMemcpy(ux.data(), c->d_uxy, sizeof(double) * (size_t)N * d, hipMemcpyDeviceToHost));
  double A[RVT_MAX_COV * RVT_MAX_COV] = {}, Ai[RVT_MAX_COV * RVT_MAX_COV];
  for (int a = 0; a < d; ++a)
    for (int b = a; b < d; ++b) {
      double t = 0.0;
      for (int64_t i = 0; i < N; ++i)
        t += ux[(size_t)a * N + i] * (std::fabs(c->h_S[i]) + c->fam_delta) * ux[(size_t)b * N + i];
      A[a * d + b] = A[b * d + a] = t;
    }
  if (!invert_spd(A, d, Ai)) return fail(c, RVT_E_INVALID, "ux' (lambda + delta) ux is singular");
  for (int a = 0; a < d; ++a) covb_diag[a] = Ai[a * d + a];
  return RVT_OK;
}

int rvt_fam_binary_scale(rvt_ctx* c, int64_t n_case, int64_t n_ctrl, double* alpha_out, double* b_out) {
  if (!c || n_case < 0 || n_ctrl < 0) return fail(c, RVT_E_INVALID, "bad arguments");
  if (!c->have_fam) return fail(c, RVT_E_STATE, "rvt_fit_fam_null first");
  const float alpha = (n_ctrl > 0) ? (float)std::log(1.0 * (double)n_case / (double)n_ctrl) : 500.f;
  // b = int logistic'(alpha + x) phi(x) dx.  The integrand is analytic and decays like exp(-x^2/2), for which the
  // trapezoidal rule converges geometrically: h = 1/64 over [-40, 40] is exact to rounding (the reference's QAGI
  // stops at 1e-7 relative).  A scalar constant of the null model, evaluated once.
  const double a = (double)alpha, h = 1.0 / 64.0;
  double sum = 0.0;
  for (int i = -40 * 64; i <= 40 * 64; ++i) {
    const double x = i * h;
    const double t = std::exp(a + x);
    const double k = 1.0 / std::sqrt(2.0 * 3.1415926535897);  // the reference's constant (src/Model.cpp:343)
    const double f = std::isfinite(t) ? t / (1. + t) / (1. + t) * k * std::exp(-x * x * 0.5) : 0.0;
    sum += f;
  }
  const float b = (float)(sum * h);  // `float b` member (src/Model.cpp:680)
  c->famcov_b2 = (double)b * (double)b;
  if (alpha_out) *alpha_out = alpha;
  if (b_out) *b_out = b;
  return RVT_OK;
}

int rvt_run_fam_blocks(rvt_ctx* c, int n, const double* const* dG, const int* Ms, const int64_t* ids,
                       rvt_gene_result* out) {
  return rvt_run_fam_tests(c, n, dG, Ms, ids, RVT_TEST_FAMSKAT, out);
}

int rvt_run_fam_tests(rvt_ctx* c, int n, const double* const* dG, const int* Ms, const int64_t* ids, uint32_t tests,
                      rvt_gene_result* out) {
  if (!c || n < 0 || (n > 0 && (!dG || !Ms || !out))) return fail(c, RVT_E_INVALID, "bad batch arguments");
  if (!(tests & (RVT_TEST_FAMSKAT | RVT_TEST_FAMCMC | RVT_TEST_FAMZEGGINI)) ||
      (tests & ~(RVT_TEST_FAMSKAT | RVT_TEST_FAMCMC | RVT_TEST_FAMZEGGINI)))
    return fail(c, RVT_E_INVALID, "rvt_run_fam_tests takes the FAMSKAT / FAMCMC / FAMZEGGINI bits");
  if (!c->have_fam) return fail(c, RVT_E_STATE, "rvt_set_kinship + rvt_fit_fam_null first");
  if (n == 0) return RVT_OK;
  hipSetDevice(c->device);
  int rc = rvt_sync(c);
  if (rc) return rc;
  hipStream_t st = c->stream;
  const int64_t N = c->fam_nc.N, ld = c->fam_nc.ld;
  // ---- 1. flip / monomorphic flags of every column (DataConsolidator.cpp:46-69,94-142) ------------------------
  size_t tot = 0;
  for (int g = 0; g < n; ++g) {
    if (Ms[g] < 1 || Ms[g] > RVT_MAX_VARIANTS) return fail(c, RVT_E_TOO_LARGE, "gene %d: M=%d", g, Ms[g]);
    tot += (size_t)Ms[g];
  }
  std::vector<const double*> cols(tot);
  {
    size_t k = 0;
    for (int g = 0; g < n; ++g)
      for (int j = 0; j < Ms[g]; ++j) cols[k++] = dG[g] + (size_t)j * ld;
  }
  // column pointer / flag lists of the batch: one grow-only allocation of the context (no hipMalloc / hipFree per batch)
  {
    const size_t need = (sizeof(double*) + sizeof(int)) * tot * 2 + 64;
    if (c->fam_list_cap < need) {
      if (c->d_fam_list) hipFree(c->d_fam_list);
      c->d_fam_list = nullptr;
      c->fam_list_cap = 0;
      HIP_TRY(c, hipMalloc((void**)&c->d_fam_list, need + need / 2));
      c->fam_list_cap = need + need / 2;
    }
  }
  const double** d_cols = reinterpret_cast<const double**>(c->d_fam_list);
  int* d_flags = reinterpret_cast<int*>(c->d_fam_list + sizeof(double*) * tot * 2);
  HIP_TRY(c, hipMemcpyAsync(d_cols, cols.data(), sizeof(double*) * tot, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(fam_colstat_kernel, dim3((unsigned)tot), dim3(256), 0, st, d_cols, (long long)N, d_flags);
  std::vector<int> flags(tot);
  HIP_TRY(c, hipMemcpyAsync(flags.data(), d_flags, sizeof(int) * tot, hipMemcpyDeviceToHost, st));
  HIP_TRY(c, sync_stream(st));
  // ---- 2. compact the kept columns (flipped where needed) and rotate them by U' --------------------------------
  std::vector<const double*> kept_cols;
  std::vector<int> kept_flip, Mk(n), off(n);
  bool all_hard = true;  // every kept column holds hard calls only
  {
    size_t k = 0;
    for (int g = 0; g < n; ++g) {
      off[g] = (int)kept_cols.size();
      for (int j = 0; j < Ms[g]; ++j, ++k)
        if (flags[k] & 2) {
          kept_cols.push_back(cols[k]);
          kept_flip.push_back(flags[k] & 1);
          all_hard = all_hard && (flags[k] & 4);
        }
      Mk[g] = (int)kept_cols.size() - off[g];
    }
  }
  const size_t T = kept_cols.size();
  for (int g = 0; g < n; ++g) {
    rvt_gene_result& r = out[g];
    std::memset(&r, 0, sizeof(r));
    r.gene_id = ids ? ids[g] : g;
    r.n_variants = Ms[g];
    r.n_poly = Mk[g];
  }
  if (T == 0) return RVT_OK;  // genotype.cols == 0 everywhere: all NA (src/Model.h:3066-3069)
  // genes with a polymorphic column, and — for the burden tests — two collapsed columns each after the T genotype ones
  const bool burden = (tests & (RVT_TEST_FAMCMC | RVT_TEST_FAMZEGGINI)) != 0;
  std::vector<int> kgene, koff, km;
  for (int g = 0; g < n; ++g)
    if (Mk[g] > 0) {
      kgene.push_back(g);
      koff.push_back(off[g]);
      km.push_back(Mk[g]);
    }
  const size_t nk = kgene.size(), TB = burden ? 2 * nk : 0;
  rc = ensure_fam_cols(c, T + TB, ld);
  if (rc) return rc;
  HIP_TRY(c, hipMemcpyAsync(d_cols + tot, kept_cols.data(), sizeof(double*) * T, hipMemcpyHostToDevice, st));
  HIP_TRY(c, hipMemcpyAsync(d_flags + tot, kept_flip.data(), sizeof(int) * T, hipMemcpyHostToDevice, st));
  if (ld != N)  // pad rows must be zero (the rotation writes rows 0 .. N-1 of every column)
    HIP_TRY(c, hipMemset2DAsync(c->d_Gt + N, sizeof(double) * (size_t)ld, 0, sizeof(double) * (size_t)(ld - N), T + TB, st));
  // FamSKAT alone on hard calls: the flipped columns go straight to the int8 plane of the rotation (no fp64 copy, no
  // column scan, no separate quantiser pass)
  const bool direct = !burden && all_hard && c->hc_enabled && !(c->d_csc_ptr && !c->d_uq_range) &&
                      !getenv("RVT_FAM_NO_DIRECT");
  if (!direct) {
    for (size_t t0 = 0; t0 < T; t0 += 65535) {  // (gridDim.y holds at most 65535 columns)
      const unsigned nt = (unsigned)std::min<size_t>(65535, T - t0);
      hipLaunchKernelGGL(fam_flip_compact_kernel, dim3(64, nt), dim3(256), 0, st, d_cols + tot + t0, d_flags + tot + t0,
                         (long long)N, (long long)ld, c->d_Gp + t0 * (size_t)ld);
    }
    HIP_TRY(c, hipGetLastError());
  }
  int* d_koff = nullptr;
  double* d_bcs = nullptr;
  int* d_bpoly = nullptr;
  struct Guard2 {
    void **a, **b, **c2;
    ~Guard2() {
      for (void** p : {a, b, c2})
        if (*p) hipFree(*p);
    }
  } guard2{(void**)&d_koff, (void**)&d_bcs, (void**)&d_bpoly};
  if (burden) {
    // cmcCollapse / zegginiCollapse of the flipped, filtered blocks into columns T .. T + 2 nk - 1, then their raw
    // sums (the score test centres the collapsed genotype, FastLMM.cpp:218-220)
    HIP_TRY(c, hipMalloc((void**)&d_koff, sizeof(int) * 2 * nk));
    HIP_TRY(c, hipMalloc((void**)&d_bcs, sizeof(double) * TB));
    HIP_TRY(c, hipMalloc((void**)&d_bpoly, sizeof(int) * TB));
    HIP_TRY(c, hipMemcpyAsync(d_koff, koff.data(), sizeof(int) * nk, hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipMemcpyAsync(d_koff + nk, km.data(), sizeof(int) * nk, hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipMemsetAsync(c->d_Gp + (size_t)T * ld, 0, sizeof(double) * (size_t)ld * TB, st));
    hipLaunchKernelGGL(fam_collapse_kernel, dim3(64, (unsigned)nk), dim3(256), 0, st, c->d_Gp, d_koff, d_koff + nk,
                       (long long)N, (long long)ld, c->d_Gp + (size_t)T * ld);
    hipLaunchKernelGGL(raw_colstat_kernel, dim3((unsigned)TB), dim3(256), 0, st, c->d_Gp + (size_t)T * ld,
                       (long long)N, (long long)ld, d_bcs, d_bpoly);
  }
  if (direct) {
    const int64_t ldk = c->uq_ldk;
    for (size_t c0 = 0; c0 < T; c0 += kRotMaxCols) {  // long lists in pieces, as rotate_columns
      const int ncp = (int)std::min<size_t>(kRotMaxCols, T - c0);
      const int64_t cols_pad = ((int64_t)ncp + kRotBN - 1) / kRotBN * kRotBN;
      const size_t need = (size_t)cols_pad * (size_t)ldk;
      if (c->rotB_cap < need) {
        if (c->d_rotB) hipFree(c->d_rotB);
        c->d_rotB = nullptr;
        c->rotB_cap = 0;
        HIP_TRY(c, hipMalloc((void**)&c->d_rotB, need + need / 4));
        c->rotB_cap = need + need / 4;
      }
      HIP_TRY(c, hipMemsetAsync(c->d_rotB, 0, need, st));
      hipLaunchKernelGGL(fam_flip_quant_kernel, dim3(64, (unsigned)ncp), dim3(256), 0, st, d_cols + tot + c0,
                         d_flags + tot + c0, (long long)N, (long long)ldk, c->d_rotB);
      std::vector<int> zero_exp((size_t)ncp, 0);
      int rcr = planes_gemm(c, c->d_Uq, c->uq_plane, kRotPlanesU, (int)N, nullptr, c->uq_sexp, c->d_rotB, need, 1, ncp,
                            zero_exp.data(), N, ldk, c->d_Gt + c0 * (size_t)ld, ld, st, c->d_uq_range);
      if (rcr) return rcr;
    }
  } else {  // the rotation of the whole batch: genotype columns + collapsed burden columns (exact int8 products, rot_gemm.hip.h)
    int rcr = rotate_columns(c, c->d_Gp, ld, (int)(T + TB), c->d_Gt, ld, st);
    if (rcr) return rcr;
  }
  HIP_TRY(c, sync_stream(st));
  if (burden) {
    // FamCMC / FamZeggini: the 2 nk rotated collapsed columns as blocks of the family covariance machinery
    // (V = cov(h,h), U from the uResid column, AF from the allele-frequency column)
    std::vector<double> us(TB), vs(TB), afs(TB), ps(TB);
    for (size_t b0 = 0; b0 < TB; b0 += RVT_MAX_VARIANTS) {
      const int V = (int)std::min<size_t>(RVT_MAX_VARIANTS, TB - b0);
      CovOut co;
      co.ustat = us.data() + b0;
      co.vstat = vs.data() + b0;
      co.af = afs.data() + b0;
      co.pval = ps.data() + b0;
      rc = famcov_run(c, c->d_Gt + (T + b0) * (size_t)ld, V, d_bcs + b0, d_bpoly + b0, &co);
      if (rc) return rc;
    }
    for (size_t k = 0; k < nk; ++k) {
      rvt_gene_result& r = out[kgene[k]];
      if (tests & RVT_TEST_FAMCMC) {
        r.famcmc_ok = 1;
        r.famcmc_af = afs[2 * k];
        r.famcmc_U = us[2 * k];
        r.famcmc_V = vs[2 * k];
        r.famcmc_p = ps[2 * k];
      }
      if (tests & RVT_TEST_FAMZEGGINI) {
        r.famzeg_ok = 1;
        r.famzeg_af = afs[2 * k + 1];
        r.famzeg_U = us[2 * k + 1];
        r.famzeg_V = vs[2 * k + 1];
        r.famzeg_p = ps[2 * k + 1];
      }
    }
  }
  if (!(tests & RVT_TEST_FAMSKAT)) return RVT_OK;
  // ---- 3. the rotated blocks go through the ordinary batch machinery with the FamSKAT null set ---------------
  std::vector<const double*> ptr;
  std::vector<int> mm, which;
  std::vector<int64_t> gid;
  size_t af_total = 0;
  for (int g = 0; g < n; ++g)
    if (Mk[g] > 0) {
      ptr.push_back(c->d_Gt + (size_t)off[g] * ld);
      mm.push_back(Mk[g]);
      which.push_back(g);
      gid.push_back(out[g].gene_id);
      af_total += (size_t)Mk[g];
    }
  std::vector<double> af(af_total, 0.0);  // unused: FamSKAT derives its allele frequencies on the device
  std::vector<rvt_gene_result> rec(ptr.size());
  struct Swap {  // the batch code reads the null set from the context
    rvt_ctx* c;
    NullConsts nc;
    NullConsts* d_nc;
    double *X, *res, *rr, *v, *zeros;
    bool have;
    int64_t nld;
    explicit Swap(rvt_ctx* c_) : c(c_), nc(c_->nc), d_nc(c_->d_nc), X(c_->d_X), res(c_->d_res), rr(c_->d_rr),
                                 v(c_->d_v), zeros(c_->d_zeros), have(c_->have_null), nld(c_->null_ld) {
      c->nc = c->fam_nc;
      c->d_nc = c->d_fam_nc;
      c->d_X = c->d_fX;
      c->d_res = c->d_fzeros;
      c->d_rr = c->d_frr;
      c->d_v = c->d_fv;
      c->d_zeros = c->d_fzeros;
      c->have_null = true;
      c->null_ld = c->fam_nc.ld;
    }
    ~Swap() {
      c->nc = nc;
      c->d_nc = d_nc;
      c->d_X = X;
      c->d_res = res;
      c->d_rr = rr;
      c->d_v = v;
      c->d_zeros = zeros;
      c->have_null = have;
      c->null_ld = nld;
    }
  };
  {
    Swap sw(c);
    // batches of <= 256 genes keep the per-batch arena bounded
    for (size_t b0 = 0; b0 < ptr.size(); b0 += 256) {
      const int nb = (int)std::min<size_t>(256, ptr.size() - b0);
      size_t afo = 0;
      for (size_t k = 0; k < b0; ++k) afo += (size_t)mm[k];
      rc = run_batch(c, nb, ptr.data() + b0, mm.data() + b0, af.data() + afo, gid.data() + b0, RVT_TEST_FAMSKAT,
                     nullptr, rec.data() + b0, nullptr);
      if (!rc) rc = rvt_sync(c);
      if (rc) return rc;
    }
  }
  for (size_t k = 0; k < which.size(); ++k) {
    rvt_gene_result& r = out[which[k]];
    r.status = rec[k].status;
    r.famskat_ok = rec[k].famskat_ok;
    r.famskat_Q = rec[k].famskat_Q;
    r.famskat_p = rec[k].famskat_p;
    r.skat_nlambda = rec[k].skat_nlambda;
    r.davies_terms = rec[k].davies_terms;
  }
  return RVT_OK;
}

}  // extern "C"

// ---- launchers of this unit's kernels for the other units (rvt_engine_int.h, "kernel families") -----------------------------
void k_lmm_sums(dim3 grid, hipStream_t st, const double* uxy, const double* lam, long long N, int d, double delta, int take_abs,
                double* partial) {
  hipLaunchKernelGGL(lmm_sums_kernel, grid, dim3(256), sizeof(double) * 256, st, uxy, lam, N, d, delta, take_abs, partial);
}
void k_fam_colstat(dim3 grid, hipStream_t st, const double* const* cols, long long N, int* flags) {
  hipLaunchKernelGGL(fam_colstat_kernel, grid, dim3(256), 0, st, cols, N, flags);
}
void k_fam_flip_compact(dim3 grid, hipStream_t st, const double* const* src_cols, const int* src_flip, long long N, long long ld,
                        double* dst) {
  hipLaunchKernelGGL(fam_flip_compact_kernel, grid, dim3(256), 0, st, src_cols, src_flip, N, ld, dst);
}
void k_raw_colstat(dim3 grid, hipStream_t st, const double* G, long long N, long long ld, double* colsum, int* poly) {
  hipLaunchKernelGGL(raw_colstat_kernel, grid, dim3(256), 0, st, G, N, ld, colsum, poly);
}
void k_rot_reduce_slices(hipStream_t st, const double* part, long long ldc, long long M, long long N, long long stride, int slices,
                         double* C, int accumulate) {
  hipLaunchKernelGGL(rot_reduce_slices_kernel, dim3(1024), dim3(256), 0, st, part, ldc, M, N, stride, slices, C, accumulate);
}
