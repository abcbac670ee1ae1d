// rvtests_amd — the tests that are not gene tests of the main pipeline: KBAC, MetaScore, MetaCov (bands and rectangles, the
// exact int8 band for hard calls) and the column operations of the adapters' device ring.  Part of librvtests_amd.so.
// this unit compiles (and ships) the META kernel family only: see "kernel families" in rvt_engine_int.h
#define RVT_K_SPLIT
#define RVT_K_META
#include "rvt_engine_int.h"
#include "gemm_f64.hip.h"
#include "band_rows.hip.h"

// row slices of MetaCov's column pass (cov_hc_prep_kernel): a function of N alone, so that a column's sums are the same numbers
// whether it is treated inside a block or alone behind its upload (rvt_block_upload_columns)
static constexpr int kCovSlices = 16;

extern "C" {
// C (M x (Nb + Nb2), column-major, leading dimension ldc) = A' D [B | B2] in fp64 on the matrix cores (gemm_f64.hip.h).
// A: M columns (lda apart), B: Nb columns, B2: Nb2 further columns (the null-model columns), all N samples long and
// zero-padded to a multiple of 16; w: optional weights along the samples (D = diag(w)), else D = I.
// symmetric: A and B are the same columns, only the tiles that meet the upper triangle are computed (the rest of C is
// left unspecified).  K is split over the chip; the partial results are added in a fixed order.
// subtract: C -= A' D B instead (every entry is read and written by one thread); the product then runs as ONE K slice, the
// grid being the tile list — meant for short K (a rank-k update of a large C).
// halo >= 0 (symmetric only): a band — row m needs the columns m .. m + halo, only those tiles are computed.
// ring > 0: A = B = the base of a block used as a ring of `ring` columns, column m is physical column (col0 + m) mod ring.
int gemm_tn_f64(rvt_ctx* c, const double* A, int64_t lda, int M, const double* B, int64_t ldb, int Nb, const double* B2,
                int64_t ldb2, int Nb2, const double* w, int64_t N, double* C, int64_t ldc, bool symmetric, hipStream_t st,
                bool subtract, int halo, int ring, int col0) {
  const int Ntot = Nb + Nb2;
  if (M < 1 || Ntot < 1) return RVT_OK;
  if (!symmetric) halo = -1;
  int nct = 0;
  const int n_tiles = gemm_f64_tiles(M, Ntot, symmetric, &nct, halo);
  const int64_t chunks = (N + kGemmKC - 1) / kGemmKC;
  int64_t slices = gemm_f64_slices(n_tiles, chunks);
  if (const char* e = getenv("RVT_GEMM64_SLICES")) slices = std::max<int64_t>(1, atoll(e));
  if (subtract) slices = 1;
  const int64_t kslice = ((chunks + slices - 1) / slices) * kGemmKC;
  slices = (N + kslice - 1) / kslice;
  double* d_out = C;
  int64_t c_slice = 0;
  if (slices > 1) {
    c_slice = ldc * Ntot;
    const size_t need = sizeof(double) * (size_t)c_slice * (size_t)slices;
    if (c->rot_part_cap < need) {
      if (c->d_rot_part) hipFree(c->d_rot_part);
      c->d_rot_part = nullptr;
      c->rot_part_cap = 0;
      HIP_TRY(c, hipMalloc((void**)&c->d_rot_part, need));
      c->rot_part_cap = need;
    }
    d_out = c->d_rot_part;
  }
  const int64_t groups = (slices + 7) / 8;
  const dim3 grid(subtract ? (unsigned)n_tiles : (unsigned)(8 * (int64_t)n_tiles * groups));
  if (subtract)
    hipLaunchKernelGGL((gemm_tn_f64_kernel<3, true>), grid, dim3(kGemmThreads), 0, st, A, (long long)lda, M, B, (long long)ldb, Nb,
                       B2 ? B2 : B, (long long)(B2 ? ldb2 : ldb), Nb2, w, (long long)N, (long long)kslice, 1, d_out, (long long)ldc,
                       0LL, n_tiles, nct, symmetric ? 1 : 0, halo, ring, col0);
  else
    hipLaunchKernelGGL((gemm_tn_f64_kernel<3, false>), grid, dim3(kGemmThreads), 0, st, A, (long long)lda, M, B, (long long)ldb, Nb,
                       B2 ? B2 : B, (long long)(B2 ? ldb2 : ldb), Nb2, w, (long long)N, (long long)kslice, (int)slices, d_out,
                       (long long)ldc, (long long)c_slice, n_tiles, nct, symmetric ? 1 : 0, halo, ring, col0);
  if (slices > 1)
    k_rot_reduce_slices(st, d_out, (long long)ldc, (long long)M, (long long)Ntot, (long long)c_slice, (int)slices, C, 0);
  HIP_TRY(c, hipGetLastError());
  return RVT_OK;
}
}

extern "C" {

// What the engine wrote into the first V columns of a block it filled column by column (rvt_block_upload_columns records a
// flag per column behind the copy): 1 = hard calls only, 0 = something else, -1 = nothing known (a caller's own
// allocation, or a block filled another way).  A HINT for choosing the kernel to start on: the integer paths test every
// value they read and fall back.  colflag (optional) receives the per-column flags when they exist (empty otherwise).
static int block_hard_calls(rvt_ctx* c, const double* dG, int V, std::vector<int>* colflag, bool* any) {
  if (colflag) colflag->clear();
  if (any) *any = false;
  if (!c->hc_enabled) return 0;
  auto it = c->col_kind.find(dG);
  if (it == c->col_kind.end() || !it->second.d_flags || V > it->second.cols) {
    if (c->content_hint == 0) return 0;  // the caller said dosages (rvt_set_content_hint): do not try the integer path first
    if (any) *any = true;
    return -1;
  }
  std::vector<int> f((size_t)V);
  if (hipMemcpyAsync(f.data(), it->second.d_flags, sizeof(int) * (size_t)V, hipMemcpyDeviceToHost, c->io_stream) != hipSuccess ||
      sync_stream(c->io_stream) != hipSuccess)
    return 0;
  bool all = true, some = false;
  for (int j = 0; j < V; ++j) {
    all = all && f[j] != 0;
    some = some || f[j] != 0;
  }
  if (any) *any = some;
  if (colflag) *colflag = std::move(f);
  return all ? 1 : 0;
}

// ---- KBAC (--kernel kbac): genotype-pattern permutation test, binary traits without covariates ---------------------------------
int rvt_kbac_blocks(rvt_ctx* c, int n, const double* const* dG, const int* M, const double* af, const double* y,
                    int nperm, double alpha, rvt_kbac_result* out) {
  if (!c || n < 0 || (n > 0 && (!dG || !M || !af || !y || !out)) || nperm < 0) return fail(c, RVT_E_INVALID, "bad arguments");
  if (!c->have_null) return fail(c, RVT_E_STATE, "no null model set (it defines the sample count)");
  hipSetDevice(c->device);
  int rc = rvt_sync(c);
  if (rc) return rc;
  const int64_t N = c->nc.N;
  std::vector<unsigned char> yb((size_t)N);
  for (int64_t i = 0; i < N; ++i) {
    if (y[i] != 0.0 && y[i] != 1.0) return fail(c, RVT_E_INVALID, "KBAC needs a 0 / 1 phenotype");
    yb[i] = y[i] == 1.0;
  }
  size_t afo = 0;
  for (int g = 0; g < n; ++g) {  // one gene at a time: the random stream is consumed in gene order
    if (M[g] < 1 || M[g] > RVT_MAX_VARIANTS) return fail(c, RVT_E_INVALID, "gene %d has M=%d", g, M[g]);
    rc = rvt_kbac_stage(c, dG[g], M[g], af + afo, yb, nperm, alpha, out + g);
    if (rc) return rc;
    afo += (size_t)M[g];
  }
  return RVT_OK;
}

// ---- MetaScore: single-variant score statistics of a block of variants (unrelated samples) -----------------
int rvt_score_block(rvt_ctx* c, const double* dG, int V, int* ok, double* ustat, double* vstat, double* effect,
                    double* effect_se, double* pvalue) {
  if (!c || !dG || V < 1 || !ok || !ustat || !vstat || !effect || !effect_se || !pvalue)
    return fail(c, RVT_E_INVALID, "bad arguments");
  if (!c->have_null) return fail(c, RVT_E_STATE, "no null model set");
  int rc = rvt_sync(c);  // processed synchronously
  if (rc) return rc;
  // Which slices START on the hard-call kernel: all of them unless the per-column flags of rvt_block_upload_columns say a
  // slice holds something else.  The kernel tests what it reads; a slice it hands back is computed by the fp64 kernel in
  // the same batch (run_batch).
  std::vector<int> colflag;
  bool any_hc = false;
  bool all_hc = block_hard_calls(c, dG, V, &colflag, &any_hc) != 0;
  if (!c->hc_enabled) all_hc = any_hc = false;
  if (c->nc.binary && !(c->d_nulltile_w && c->d_vq)) all_hc = any_hc = false;  // (no digit planes: fp64 kernel)
  if (!all_hc && colflag.empty()) any_hc = false;
  // columns per slice.  General kernel: with M = 32 - (d + 1) the slice and its [X | rr] columns fill exactly two column
  // tiles, tile class (2,2).  Hard-call kernel: the null-model columns have a tile of their own, so a slice is two
  // full genotype tiles (32 columns, class MT = 2) when the whole block qualifies.
  int kSlice = all_hc ? 32 : 32 - (c->nc.d + 1);
  if (const char* e = getenv("RVT_SCORE_SLICE")) kSlice = std::max(1, std::min(64, atoi(e)));
  constexpr int kChunk = 256;  // slices per launch
  const int64_t ld = c->null_ld;
  std::vector<double> af((size_t)kSlice * kChunk, 0.01);
  std::vector<rvt_gene_result> rs(kChunk);
  std::vector<unsigned char> shc(kChunk);
  for (int c0 = 0; c0 < V; c0 += kSlice * kChunk) {
    const int cols = std::min(V - c0, kSlice * kChunk), n = (cols + kSlice - 1) / kSlice;
    std::vector<const double*> ptr(n);
    std::vector<int> Ms(n);
    std::vector<int64_t> ids(n);
    for (int g = 0; g < n; ++g) {
      ptr[g] = dG + (size_t)(c0 + g * kSlice) * ld;
      Ms[g] = std::min(kSlice, cols - g * kSlice);
      ids[g] = (int64_t)g * kSlice;
      bool hc = all_hc;
      if (!all_hc && any_hc) {
        hc = true;
        for (int j = 0; j < Ms[g]; ++j) hc = hc && colflag[(size_t)c0 + (size_t)g * kSlice + j] != 0;
      }
      shc[g] = hc ? 1 : 0;
    }
    CovOut co;
    co.score = true;
    co.slice_hc = any_hc ? shc.data() : nullptr;
    co.ok = ok + c0;
    co.ustat = ustat + c0;
    co.vstat = vstat + c0;
    co.effect = effect + c0;
    co.se = effect_se + c0;
    co.pval = pvalue + c0;
    rc = run_batch(c, n, ptr.data(), Ms.data(), af.data(), ids.data(), 0u, nullptr, rs.data(), nullptr, &co);
    if (rc) return rc;
  }
  return RVT_OK;
}

int rvt_null_dims(rvt_ctx* c, int64_t* N, int* d) {
  if (!c) return RVT_E_INVALID;
  if (!c->have_null) return fail(c, RVT_E_STATE, "no null model set");
  if (N) *N = c->nc.N;
  if (d) *d = c->nc.d;
  return RVT_OK;
}

int rvt_null_summary(rvt_ctx* c, double* beta, double* covb_diag, double* sigma2) {
  if (!c || !covb_diag) return fail(c, RVT_E_INVALID, "bad arguments");
  if (!c->have_null) return fail(c, RVT_E_STATE, "no null model set");
  const NullConsts& nc = c->nc;
  if (beta)
    for (int k = 0; k < nc.d; ++k) beta[k] = c->have_null_beta ? c->null_beta[k] : NAN;
  for (int k = 0; k < nc.d; ++k) covb_diag[k] = nc.Cinv[k * nc.d + k] * (nc.binary ? 1.0 : nc.sigma2);
  if (sigma2) *sigma2 = nc.sigma2;
  return RVT_OK;
}

// ---- MetaCov: covariance band of one block of consecutive variants ---------------------------------------
static int cov_rect_impl(rvt_ctx* c, const double* dG, int col0, int H, int W, double* cov, double* xz, double* zz,
                         int* polymorphic, bool allow_fast);
int rvt_cov_block(rvt_ctx* c, const double* dG, int V, double* cov, double* xz, double* zz, int* polymorphic) {
  if (!c || !dG || V < 1 || !cov || !xz || !polymorphic) return fail(c, RVT_E_INVALID, "bad arguments");
  if (V > RVT_MAX_VARIANTS) return fail(c, RVT_E_TOO_LARGE, "block of %d variants exceeds RVT_MAX_VARIANTS", V);
  int rc = rvt_sync(c);  // the block is processed alone and synchronously
  if (rc) return rc;
  // Hard calls and an unweighted model: G'G is an integer matrix — the band comes from the exact int8 product
  // (rvt_cov_rect with heads = window) instead of the fp64 matrix cores, ~6x faster at V = 1024.
  if (c->have_null && !c->nc.binary && V >= 64 && !getenv("RVT_METACOV_FP64") && block_hard_calls(c, dG, V, nullptr, nullptr))
    return rvt_cov_rect(c, dG, 0, V, V, cov, xz, zz, polymorphic);  // (tests what it reads; falls back by itself)
  // Anything else — dosages, or a binary trait's weights: the LDS-tiled fp64 product (gemm_f64.hip.h) from V = 64 on;
  // RVT_METACOV_PANEL=1 keeps round 4's path through the one-wave sufficient-statistics kernel (22 TFLOP/s at V = 1024)
  if (c->have_null && V >= 64 && !getenv("RVT_METACOV_PANEL")) return cov_rect_impl(c, dG, 0, V, V, cov, xz, zz, polymorphic, false);
  std::vector<double> af(V, 0.01);
  rvt_gene_result r;
  CovOut co;
  co.cov = cov;
  co.xz = xz;
  co.zz = zz;
  co.poly = polymorphic;
  const double* p = dG;
  rc = run_batch(c, 1, &p, &V, af.data(), nullptr, 0u, nullptr, &r, nullptr, &co);
  if (rc) return rc;
  // run_batch marked the slot busy without enqueuing a record copy: clear it
  for (auto& sl : c->slots) {
    if (sl.pending_out == &r) {
      sl.pending_out = nullptr;
      sl.pending_n = 0;
    }
  }
  return RVT_OK;
}

static int cov_rect_impl(rvt_ctx* c, const double* dG, int col0, int H, int W, double* cov, double* xz, double* zz,
                         int* polymorphic, bool allow_fast);
int rvt_cov_rect(rvt_ctx* c, const double* dG, int col0, int H, int W, double* cov, double* xz, double* zz,
                 int* polymorphic) {
  return cov_rect_impl(c, dG, col0, H, W, cov, xz, zz, polymorphic, true);
}
static int cov_rect_impl(rvt_ctx* c, const double* dG, int col0, int H, int W, double* cov, double* xz, double* zz,
                         int* polymorphic, bool allow_fast) {
  if (!c || !dG || col0 < 0 || H < 1 || W < H || !cov || !xz || !polymorphic)
    return fail(c, RVT_E_INVALID, "bad arguments");
  if (!c->have_null) return fail(c, RVT_E_STATE, "no null model set");
  hipSetDevice(c->device);
  int rc = rvt_sync(c);
  if (rc) return rc;
  hipStream_t st = c->stream;
  const NullConsts& nc = c->nc;
  const int64_t N = nc.N, ld = nc.ld;
  const int d = nc.d;
  CovConsts cc;
  std::vector<double> zzv;
  rc = cov_constants(c, false, &cc, &zzv);
  if (rc) return rc;
  const double* GW = dG + (size_t)col0 * ld;
  // work space: one grow-only allocation of the context (a window walk calls this per eviction: six hipMalloc / hipFree
  // pairs per call cost more than the kernels of a small window)
  double *d_S = nullptr, *d_T = nullptr, *d_cs = nullptr, *d_xz = nullptr, *d_cov = nullptr, *d_tmp = nullptr;
  int* d_poly = nullptr;
  {
    auto up = [](size_t b) { return (b + 255) / 256 * 256; };
    const size_t bS = up(sizeof(double) * (size_t)H * (W + d)), bC = up(sizeof(double) * (size_t)H * W),
                 bT = up(sizeof(double) * (size_t)W * d), bV = up(sizeof(double) * (size_t)W), bP = up(sizeof(int) * (size_t)W);
    const size_t bM = up(sizeof(double) * (size_t)64 * W * (RVT_MAX_COV + 3));   // slice partials of the column pass (<= 64 slices)
    const size_t need = bS + bC + 2 * bT + bV + bP + bM;
    if (c->cov_work_cap < need) {
      if (c->d_cov_work) hipFree(c->d_cov_work);
      c->d_cov_work = nullptr;
      c->cov_work_cap = 0;
      HIP_TRY(c, hipMalloc((void**)&c->d_cov_work, need + need / 4));
      c->cov_work_cap = need + need / 4;
    }
    char* q = c->d_cov_work;
    d_S = reinterpret_cast<double*>(q);      // H x (W + d): T sits behind S when heads = window
    q += bS;
    d_cov = reinterpret_cast<double*>(q);
    q += bC;
    d_T = reinterpret_cast<double*>(q);
    q += bT;
    d_xz = reinterpret_cast<double*>(q);
    q += bT;
    d_cs = reinterpret_cast<double*>(q);
    q += bV;
    d_poly = reinterpret_cast<int*>(q);
    q += bP;
    d_tmp = reinterpret_cast<double*>(q);
  }
  const double* T_used = d_T;
  // T = G_W' D X (W x d) and S = G_H' D G_W (H x W): for a hard-call block under an unweighted model the exact integer
  // product of rot_gemm.hip.h, else the fp64 matrix cores (round 4 quantised such operands to six digit planes, 36
  // int8 products, ~2^-40 relative)
  // (hard calls are a prediction — the engine's own per-column flags when it filled the block, optimism otherwise —
  // that cov_hc_prep_kernel verifies on every value it converts; a block that fails is computed again the general way)
  // (round 5: rectangles too — heads against a wider window, as the adapter's ring hands them over from 1 025 columns on)
  const bool fast = allow_fast && !nc.binary && W >= 64 && block_hard_calls(c, dG, col0 + W, nullptr, nullptr) != 0;
  int* d_bad = nullptr;
  int h_bad = 0;
  // one pass over the window's columns for the column statistics and T = G_W' D X (and, on the hard-call path, the int8
  // copy): rows sliced across workgroups, partial results added in a fixed order
  const int dmax = d <= 4 ? 4 : (d <= 8 ? 8 : RVT_MAX_COV);
  const int wgs = (W + kCovHcCols - 1) / kCovHcCols;
  // (the slice count depends on N alone: a column's sums then come out the same whether the column is treated with 1 023
  //  others here or alone behind its upload, rvt_block_upload_columns)
  const int slices = (int)std::max<int64_t>(1, std::min<int64_t>(kCovSlices, N / 4096 + 1));
  if (!fast) {
    const dim3 grid((unsigned)wgs, (unsigned)slices);
    const double* wts = nc.binary ? c->d_v : nullptr;
    if (dmax == 4)
      hipLaunchKernelGGL((cov_hc_prep_kernel<4, false>), grid, dim3(256), 0, st, GW, (long long)N, (long long)ld, W, c->d_X,
                         (long long)ld, d, (signed char*)nullptr, 0LL, d_tmp, (int*)nullptr, wts);
    else if (dmax == 8)
      hipLaunchKernelGGL((cov_hc_prep_kernel<8, false>), grid, dim3(256), 0, st, GW, (long long)N, (long long)ld, W, c->d_X,
                         (long long)ld, d, (signed char*)nullptr, 0LL, d_tmp, (int*)nullptr, wts);
    else
      hipLaunchKernelGGL((cov_hc_prep_kernel<RVT_MAX_COV, false>), grid, dim3(256), 0, st, GW, (long long)N, (long long)ld, W,
                         c->d_X, (long long)ld, d, (signed char*)nullptr, 0LL, d_tmp, (int*)nullptr, wts);
    hipLaunchKernelGGL(cov_hc_finish_kernel, dim3((unsigned)((W * (dmax + 3) + 255) / 256)), dim3(256), 0, st, d_tmp, slices,
                       W, d, dmax, d_cs, d_poly, d_T);
  }
  // the block's own column cache (rvt_block_upload_columns made it behind the PCIe copies): every column of the window has its
  // int8 copy, sum, flag and row of T under THIS null model -> the call starts at the integer product
  const rvt_ctx::ColKind* ckc = nullptr;
  if (fast) {
    auto itc = c->col_kind.find(dG);
    if (itc != c->col_kind.end() && itc->second.d_i8 && itc->second.gen == c->null_gen &&
        itc->second.ldk == (N + 127) / 128 * 128 && col0 + W <= itc->second.cols && !getenv("RVT_METACOV_NO_CACHE")) {
      bool all = true;
      for (int j = col0; j < col0 + W && all; ++j) all = itc->second.valid[(size_t)j] == 1;  // (2 = hard calls + an other value: MXFP4 band only)
      if (all) ckc = &itc->second;
    }
  }
  if (fast && ckc) {
    const int64_t ldk = ckc->ldk;
    hipLaunchKernelGGL(cov_cache_gather_kernel, dim3((unsigned)((W + 255) / 256)), dim3(256), 0, st, ckc->d_cs + col0,
                       ckc->d_poly + col0, ckc->d_T + (size_t)col0 * RVT_MAX_COV, W, d, RVT_MAX_COV, d_cs, d_poly, d_T);
    const signed char* A8 = ckc->d_i8 + (size_t)col0 * (size_t)ldk;
    std::vector<int> zero_exp((size_t)W, 0);
    rc = rvt_planes_gemm(c, A8, 0, 1, H, zero_exp.data(), 0, A8, 0, 1, W, zero_exp.data(), N, ldk, d_S, H, st);
    if (rc) return rc;
  } else if (fast) {
    // a hard-call block (rvt_cov_block's fast path; heads = the first H of the W columns): ONE pass over G gives the column
    // statistics, T = G'X and the int8 copy (cov_hc_prep_kernel); S = G'G is then one exact integer product
    const int64_t ldk = (N + 127) / 128 * 128;
    const int64_t cols_pad = ((int64_t)W + kRotBM - 1) / kRotBM * kRotBM;
    const size_t need = (size_t)cols_pad * (size_t)ldk;
    if (c->rotB_cap < need) {
      if (c->d_rotB) hipFree(c->d_rotB);
      c->d_rotB = nullptr;
      c->rotB_cap = 0;
      HIP_TRY(c, hipMalloc((void**)&c->d_rotB, need + need / 4));
      c->rotB_cap = need + need / 4;
    }
    // the product reads cols_pad columns of ldk bytes: the column pass writes the W columns' rows up to N rounded to 4 — only
    // the pad rows behind them and the pad columns have to be zeroed (the whole 0.5 GB copy cost 0.1 ms of a 2.3 ms block)
    {
      const int64_t n4 = (N + 3) / 4 * 4;
      if (ldk > n4)
        HIP_TRY(c, hipMemset2DAsync(c->d_rotB + n4, (size_t)ldk, 0, (size_t)(ldk - n4), (size_t)W, st));
      if (cols_pad > W) HIP_TRY(c, hipMemsetAsync(c->d_rotB + (size_t)W * ldk, 0, (size_t)(cols_pad - W) * (size_t)ldk, st));
    }
    if (!c->d_kind) HIP_TRY(c, hipMalloc((void**)&c->d_kind, sizeof(int)));
    d_bad = c->d_kind;
    HIP_TRY(c, hipMemsetAsync(d_bad, 0, sizeof(int), st));
    {
      const dim3 grid((unsigned)wgs, (unsigned)slices);
      if (dmax == 4)
        hipLaunchKernelGGL((cov_hc_prep_kernel<4>), grid, dim3(256), 0, st, GW, (long long)N, (long long)ld, W, c->d_X,
                           (long long)ld, d, c->d_rotB, (long long)ldk, d_tmp, d_bad);
      else if (dmax == 8)
        hipLaunchKernelGGL((cov_hc_prep_kernel<8>), grid, dim3(256), 0, st, GW, (long long)N, (long long)ld, W, c->d_X,
                           (long long)ld, d, c->d_rotB, (long long)ldk, d_tmp, d_bad);
      else
        hipLaunchKernelGGL((cov_hc_prep_kernel<RVT_MAX_COV>), grid, dim3(256), 0, st, GW, (long long)N, (long long)ld, W,
                           c->d_X, (long long)ld, d, c->d_rotB, (long long)ldk, d_tmp, d_bad);
      hipLaunchKernelGGL(cov_hc_finish_kernel, dim3((unsigned)((W * (dmax + 3) + 255) / 256)), dim3(256), 0, st, d_tmp, slices,
                         W, d, dmax, d_cs, d_poly, d_T);
    }
    std::vector<int> zero_exp((size_t)W, 0);
    rc = rvt_planes_gemm(c, c->d_rotB, need, 1, H, zero_exp.data(), 0, c->d_rotB, need, 1, W, zero_exp.data(), N, ldk, d_S, H, st);
    if (rc) return rc;
  } else {
    // anything else — dosages, a binary trait's weights — on the fp64 matrix cores (gemm_f64.hip.h): the upper triangle of
    // S = G_H' D G_W
    // (T came out of the pass above: a 128-column tile for d columns of X would cost as much as a whole tile of S)
    const double* wts = nc.binary ? c->d_v : nullptr;
    rc = gemm_tn_f64(c, GW, ld, H, GW, ld, W, nullptr, 0, 0, wts, N, d_S, H, true, st);
    if (rc) return rc;
  }
  hipLaunchKernelGGL(cov_rect_xz_kernel, dim3((unsigned)((W + 255) / 256)), dim3(256), 0, st, cc, T_used, d_cs, W, d_xz);
  hipLaunchKernelGGL(cov_rect_rows_kernel, dim3((unsigned)H), dim3(256), 0, st, cc, d_S, d_cs, d_xz, H, W, d_cov);
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipMemcpyAsync(cov, d_cov, sizeof(double) * (size_t)H * W, hipMemcpyDeviceToHost, st));
  HIP_TRY(c, hipMemcpyAsync(xz, d_xz, sizeof(double) * (size_t)W * d, hipMemcpyDeviceToHost, st));
  HIP_TRY(c, hipMemcpyAsync(polymorphic, d_poly, sizeof(int) * (size_t)W, hipMemcpyDeviceToHost, st));
  if (d_bad) HIP_TRY(c, hipMemcpyAsync(&h_bad, d_bad, sizeof(int), hipMemcpyDeviceToHost, st));
  HIP_TRY(c, sync_stream(st));
  if (h_bad) return cov_rect_impl(c, dG, col0, H, W, cov, xz, zz, polymorphic, false);  // not hard calls after all
  if (zz) std::memcpy(zz, zzv.data(), sizeof(double) * (size_t)d * d);
  return RVT_OK;
}



// one launch of MetaCov's column pass (cov_hc_prep_kernel) for d covariates; pack: the hard-call variant (value test, int8 / E2M1
// copies), else the general one (optional weights)
static void launch_cov_prep(hipStream_t st, int d, bool pack, dim3 grid, const double* G, int64_t N, int64_t ld, int W, const double* X,
                            signed char* out8, int64_t ldk, double* part, int* bad, const double* wts, int* hard_flag, int ring,
                            int col0, unsigned char* out4, int64_t ldk4, const double* mu_known = nullptr,
                            unsigned char* out4m = nullptr) {
  const int dmax = d <= 4 ? 4 : (d <= 8 ? 8 : RVT_MAX_COV);
#define RVT_PREP(DM, PK)                                                                                                          \
  hipLaunchKernelGGL((cov_hc_prep_kernel<DM, PK>), grid, dim3(256), 0, st, G, (long long)N, (long long)ld, W, X, (long long)ld, d, \
                     out8, (long long)ldk, part, bad, wts, hard_flag, ring, col0, out4, (long long)ldk4, mu_known, out4m)
  if (pack) {
    if (dmax == 4) RVT_PREP(4, true);
    else if (dmax == 8) RVT_PREP(8, true);
    else RVT_PREP(RVT_MAX_COV, true);
  } else {
    if (dmax == 4) RVT_PREP(4, false);
    else if (dmax == 8) RVT_PREP(8, false);
    else RVT_PREP(RVT_MAX_COV, false);
  }
#undef RVT_PREP
}

// ---- MetaCov on a circular ring: the band of a sliding window ----------------------------------------------------------------
// flags of the W logical columns of a ring (physical (col0 + j) mod ring): 1 = every column holds hard calls only, 0 = some
// column holds something else, -1 = nothing known about the block
// (masked_ok: the block's per-column cache states — a column in state 2, hard calls plus ONE other value known from its packed
//  upload, counts as usable although its content flag says "not hard calls only")
static int ring_hard_calls(rvt_ctx* c, const double* dG, int ring, int col0, int W, const unsigned char* masked_ok = nullptr) {
  if (!c->hc_enabled) return 0;
  auto it = c->col_kind.find(dG);
  const int span = ring > 0 ? ring : col0 + W;
  if (it == c->col_kind.end() || !it->second.d_flags || span > it->second.cols) return c->content_hint == 0 ? 0 : -1;
  std::vector<int> f((size_t)span);
  if (hipMemcpyAsync(f.data(), it->second.d_flags, sizeof(int) * (size_t)span, hipMemcpyDeviceToHost, c->io_stream) != hipSuccess ||
      sync_stream(c->io_stream) != hipSuccess)
    return 0;
  for (int j = 0; j < W; ++j) {
    int p = col0 + j;
    if (ring > 0 && p >= ring) p -= ring;
    if (!f[(size_t)p] && !(masked_ok && masked_ok[(size_t)p] == 2)) return 0;
  }
  return 1;
}

static int cov_band_impl(rvt_ctx* c, const double* dG, int ring, int col0, int H, int W, int halo, float scale, float* band,
                         double* xz, double* zz, int* polymorphic, bool allow_fast) {
  if (!c || !dG || ring < 0 || col0 < 0 || H < 1 || W < H || halo < 0 || !band || !xz || !polymorphic)
    return fail(c, RVT_E_INVALID, "bad arguments");
  if (!c->have_null) return fail(c, RVT_E_STATE, "no null model set");
  if (ring > 0 && (col0 >= ring || W > ring)) return fail(c, RVT_E_INVALID, "window of %d columns from %d in a ring of %d", W, col0, ring);
  if ((long long)W > (long long)H + halo) W = H + halo;  // (markers behind the last head's window are never read)
  {
    auto it = c->col_kind.find(dG);
    if (it != c->col_kind.end() && (ring > 0 ? ring : col0 + W) > it->second.cols)
      return fail(c, RVT_E_INVALID, "the window leaves the block (%d columns)", it->second.cols);
  }
  hipSetDevice(c->device);
  int rc = rvt_sync(c);
  if (rc) return rc;
  hipStream_t st = c->stream;
  const NullConsts& nc = c->nc;
  const int64_t N = nc.N, ld = nc.ld;
  const int d = nc.d;
  CovConsts cc;
  std::vector<double> zzv;
  rc = cov_constants(c, false, &cc, &zzv);
  if (rc) return rc;
  const int ringk = (ring > 0 && col0 + W > ring) ? ring : 0;   // (a window that does not wrap is a linear range)
  // what the product reads: the columns' hard calls as E2M1 codes, two per byte, on the MXFP4 matrix instruction (exact, see
  // band_gemm.hip.h; RVT_BAND_INT8=1: one byte per genotype on the int8 one)
  const bool fp4 = !getenv("RVT_BAND_INT8");
  // the ring's own column cache (rvt_block_upload_columns made it behind the uploads): usable when every column of the window
  // has an entry made under THIS null model — state 1 hard calls only, state 2 hard calls plus one other value (MXFP4 only)
  const rvt_ctx::ColKind* ckc = nullptr;
  bool masked = false;
  {
    auto itc = c->col_kind.find(dG);
    if (itc != c->col_kind.end() && itc->second.d_i8 && itc->second.gen == c->null_gen &&
        itc->second.ldk == (N + 127) / 128 * 128 && !getenv("RVT_METACOV_NO_CACHE") && (!fp4 || itc->second.d_i4)) {
      bool all = true;
      for (int j = 0; j < W && all; ++j) {
        int p = col0 + j;
        if (ring > 0 && p >= ring) p -= ring;
        const unsigned char v = itc->second.valid[(size_t)p];
        all = v != 0 && (v == 1 || (fp4 && itc->second.d_m4));
        masked = masked || v == 2;
      }
      if (all) ckc = &itc->second;
      else masked = false;
    }
  }
  const bool fast = allow_fast && !nc.binary && !getenv("RVT_METACOV_FP64") &&
                    ring_hard_calls(c, dG, ring, col0, W, (ckc && masked) ? ckc->valid.data() : nullptr) != 0;
  if (!fast) {
    ckc = nullptr;
    masked = false;
  }
  c->band_last_path = !fast ? 0 : (masked ? 4 : (ckc ? (fp4 ? 1 : 11) : (fp4 ? 2 : 12)));
  // heads per pass: the rectangle of doubles (fp64) of a pass stays within a few hundred MB; the integer band in passes of 1 024
  // so that the rows of one pass cross PCIe while the next pass multiplies (two band buffers, the copies on copy_stream)
  int Hc = 1024;
  if (const char* e = getenv("RVT_BAND_PASS")) Hc = std::max(256, atoi(e) / 256 * 256);
  const int Hp = std::min(H, Hc), Wp = (int)std::min<long long>(W, (long long)Hp + halo);
  double *d_T = nullptr, *d_cs = nullptr, *d_xz = nullptr, *d_tmp = nullptr, *d_S = nullptr, *d_mu_l = nullptr;
  float* d_band = nullptr;
  int* d_poly = nullptr;
  {
    auto up = [](size_t b) { return (b + 255) / 256 * 256; };
    const size_t bT = up(sizeof(double) * (size_t)W * d), bV = up(sizeof(double) * (size_t)W), bP = up(sizeof(int) * (size_t)W);
    const size_t bM = up(sizeof(double) * (size_t)kCovSlices * W * (RVT_MAX_COV + 3));
    const size_t bB = 2 * up(sizeof(float) * (size_t)Hp * ((size_t)halo + 1));
    const size_t bS = fast ? 0 : up(sizeof(double) * (size_t)Hp * Wp);
    const size_t need = 2 * bT + 2 * bV + bP + bM + bB + bS;
    if (c->cov_work_cap < need) {
      if (c->d_cov_work) hipFree(c->d_cov_work);
      c->d_cov_work = nullptr;
      c->cov_work_cap = 0;
      HIP_TRY(c, hipMalloc((void**)&c->d_cov_work, need + need / 4));
      c->cov_work_cap = need + need / 4;
    }
    char* q = c->d_cov_work;
    d_T = reinterpret_cast<double*>(q);
    q += bT;
    d_xz = reinterpret_cast<double*>(q);
    q += bT;
    d_cs = reinterpret_cast<double*>(q);
    q += bV;
    d_mu_l = reinterpret_cast<double*>(q);
    q += bV;
    d_poly = reinterpret_cast<int*>(q);
    q += bP;
    d_tmp = reinterpret_cast<double*>(q);
    q += bM;
    d_band = reinterpret_cast<float*>(q);
    q += bB;
    d_S = reinterpret_cast<double*>(q);
  }
  const int dmax = d <= 4 ? 4 : (d <= 8 ? 8 : RVT_MAX_COV);
  const int wgs = (W + kCovHcCols - 1) / kCovHcCols;
  const int slices = (int)std::max<int64_t>(1, std::min<int64_t>(kCovSlices, N / 4096 + 1));
  const double* Gbase = ringk ? dG : dG + (size_t)col0 * ld;   // what the column pass indexes from
  const int pcol0 = ringk ? col0 : 0;
  int* d_bad = nullptr;
  int h_bad = 0;
  // (the store the product reads, its ring and first column)
  const int8_t* R8 = nullptr;
  int r8_ring = 0, r8_col0 = 0;
  const int64_t ldk8 = (N + 127) / 128 * 128, ldk4 = ((N + 1) / 2 + 127) / 128 * 128;
  int64_t ldk = fp4 ? ldk4 : ldk8;
  if (fast && ckc) {
    hipLaunchKernelGGL(band_cache_gather_kernel, dim3((unsigned)((W + 255) / 256)), dim3(256), 0, st, ckc->d_cs, ckc->d_poly,
                       ckc->d_T, ringk, col0, W, d, RVT_MAX_COV, d_cs, d_poly, d_T, masked ? ckc->d_mu : (const double*)nullptr,
                       masked ? d_mu_l : (double*)nullptr);
    R8 = fp4 ? reinterpret_cast<const int8_t*>(ckc->d_i4) : reinterpret_cast<const int8_t*>(ckc->d_i8);
    r8_ring = ringk;
    r8_col0 = col0;
  } else if (fast) {
    // no cache: ONE pass over the window's columns gives the statistics, T = G'X and a linear copy of the window's hard calls
    const size_t need = ((size_t)W + kBandBT) * (size_t)ldk;
    if (c->rotB_cap < need) {
      if (c->d_rotB) hipFree(c->d_rotB);
      c->d_rotB = nullptr;
      c->rotB_cap = 0;
      HIP_TRY(c, hipMalloc((void**)&c->d_rotB, need + need / 4));
      c->rotB_cap = need + need / 4;
    }
    const int64_t n4 = (N + 3) / 4 * 4, nw = fp4 ? n4 / 2 : n4;   // bytes of a column the pass writes; the pad behind them: zero
    if (ldk > nw) HIP_TRY(c, hipMemset2DAsync(c->d_rotB + nw, (size_t)ldk, 0, (size_t)(ldk - nw), (size_t)W, st));
    if (!c->d_kind) HIP_TRY(c, hipMalloc((void**)&c->d_kind, sizeof(int)));
    d_bad = c->d_kind;
    HIP_TRY(c, hipMemsetAsync(d_bad, 0, sizeof(int), st));
    launch_cov_prep(st, d, true, dim3((unsigned)wgs, (unsigned)slices), Gbase, N, ld, W, c->d_X, fp4 ? nullptr : c->d_rotB, ldk8, d_tmp,
                    d_bad, nullptr, nullptr, ringk, pcol0, fp4 ? reinterpret_cast<unsigned char*>(c->d_rotB) : nullptr, ldk4);
    hipLaunchKernelGGL(cov_hc_finish_kernel, dim3((unsigned)((W * (dmax + 3) + 255) / 256)), dim3(256), 0, st, d_tmp, slices,
                       W, d, dmax, d_cs, d_poly, d_T);
    R8 = reinterpret_cast<const int8_t*>(c->d_rotB);
  } else {
    launch_cov_prep(st, d, false, dim3((unsigned)wgs, (unsigned)slices), Gbase, N, ld, W, c->d_X, nullptr, 0, d_tmp, nullptr,
                    nc.binary ? c->d_v : nullptr, nullptr, ringk, pcol0, nullptr, 0);
    hipLaunchKernelGGL(cov_hc_finish_kernel, dim3((unsigned)((W * (dmax + 3) + 255) / 256)), dim3(256), 0, st, d_tmp, slices,
                       W, d, dmax, d_cs, d_poly, d_T);
  }
  hipLaunchKernelGGL(cov_rect_xz_kernel, dim3((unsigned)((W + 255) / 256)), dim3(256), 0, st, cc, d_T, d_cs, W, d_xz);
  HIP_TRY(c, hipGetLastError());
  // the heads in passes of up to Hc: pass (h0, nh) covers the logical columns [h0, h0 + wsub)
  for (int i = 0; i < 2; ++i) {
    if (!c->ev_band_fin[i]) HIP_TRY(c, hipEventCreateWithFlags(&c->ev_band_fin[i], hipEventDisableTiming));
    if (!c->ev_band_copied[i]) HIP_TRY(c, hipEventCreateWithFlags(&c->ev_band_copied[i], hipEventDisableTiming));
  }
  float* const d_band2[2] = {d_band, d_band + (size_t)Hp * ((size_t)halo + 1)};
  int pass = 0;
  for (int h0 = 0; h0 < H; h0 += Hc, ++pass) {
    float* const d_band = d_band2[pass & 1];
    if (pass >= 2) HIP_TRY(c, hipStreamWaitEvent(st, c->ev_band_copied[pass & 1], 0));  // (the buffer's previous rows have left)
    const int nh = std::min(Hc, H - h0);
    const int wsub = (int)std::min<long long>((long long)W - h0, (long long)nh + halo);
    if (fast) {
      const int n_tiles = band_tiles(nh, wsub, halo);
      const int64_t kbytes = ldk, chunks = kbytes / kRotKC;
      const int n_sets = masked ? 4 : 1;  // (mean-imputed columns: h'h, h'm, m'h, m'm)
      int64_t nsl = band_slices(n_tiles, chunks, ((size_t)3 << 30) / (size_t)n_sets);
      if (const char* e = getenv("RVT_BAND_SLICES"))
        if (atoll(e) > 0) nsl = atoll(e);
      int64_t kslice = ((chunks + nsl - 1) / nsl) * kRotKC;
      // a slice's sums are exact in the accumulator: int32 holds 4 N for any N the engine takes; fp32 holds integers below 2^24,
      // i.e. at most 2^22 samples = 2^21 bytes of E2M1 codes per slice
      if (fp4) kslice = std::min<int64_t>(kslice, (int64_t)1 << 21);
      nsl = (kbytes + kslice - 1) / kslice;
      const long long set_stride = (long long)n_tiles * (long long)nsl * kBandBT * kBandBT;
      const size_t need = sizeof(int) * (size_t)set_stride * (size_t)n_sets;
      if (c->rot_part_cap < need) {
        if (c->d_rot_part) hipFree(c->d_rot_part);
        c->d_rot_part = nullptr;
        c->rot_part_cap = 0;
        HIP_TRY(c, hipMalloc((void**)&c->d_rot_part, need));
        c->rot_part_cap = need;
      }
      int* d_part = reinterpret_cast<int*>(c->d_rot_part);
      int pc = r8_col0 + h0;
      if (r8_ring > 0 && pc >= r8_ring) pc -= r8_ring;
      const int pr = (r8_ring > 0 && pc + wsub > r8_ring) ? r8_ring : 0;
      const unsigned grid = (unsigned)(8 * (int64_t)n_tiles * ((nsl + 7) / 8));
      if (masked) {
        const int8_t* Hs = reinterpret_cast<const int8_t*>(ckc->d_i4);
        const int8_t* Ms = reinterpret_cast<const int8_t*>(ckc->d_m4);
        const int8_t* sideA[4] = {Hs, Hs, Ms, Ms};
        const int8_t* sideB[4] = {Hs, Ms, Hs, Ms};
        for (int k = 0; k < 4; ++k)
          hipLaunchKernelGGL(band_gemm_fp4, dim3(grid), dim3(kBandThreads), 0, st, sideA[k], sideB[k], (long long)ldk, pr, pc, nh, wsub,
                             halo, (long long)kbytes, (long long)kslice, (int)nsl, n_tiles, d_part + (size_t)k * (size_t)set_stride);
      } else if (fp4) {
        hipLaunchKernelGGL(band_gemm_fp4, dim3(grid), dim3(kBandThreads), 0, st, R8, R8, (long long)ldk, pr, pc, nh, wsub, halo,
                           (long long)kbytes, (long long)kslice, (int)nsl, n_tiles, d_part);
      } else {
        hipLaunchKernelGGL(band_gemm_i8, dim3(grid), dim3(kBandThreads), 0, st, R8, R8, (long long)ldk, pr, pc, nh, wsub, halo,
                           (long long)kbytes, (long long)kslice, (int)nsl, n_tiles, d_part);
      }
      hipLaunchKernelGGL(band_finish_i32_kernel, dim3((unsigned)nh), dim3(256), 0, st, cc, d_part, (int)nsl, n_tiles, d_cs + h0,
                         d_xz + (size_t)h0 * d, nh, wsub, halo, scale, d_band, (double*)nullptr,
                         masked ? d_mu_l + h0 : (const double*)nullptr, set_stride);
    } else {
      // dosages, a binary trait's weights: the band tiles of S = G_H' D G_W on the fp64 matrix cores (gemm_f64.hip.h)
      const double* wts = nc.binary ? c->d_v : nullptr;
      int pc = col0 + h0;
      if (ring > 0 && pc >= ring) pc -= ring;
      const int pr = (ring > 0 && pc + wsub > ring) ? ring : 0;
      const double* base = pr ? dG : dG + (size_t)pc * ld;
      rc = gemm_tn_f64(c, base, ld, nh, base, ld, wsub, nullptr, 0, 0, wts, N, d_S, nh, true, st, false, halo, pr, pr ? pc : 0);
      if (rc) return rc;
      hipLaunchKernelGGL(band_rows_f64_kernel, dim3((unsigned)nh), dim3(256), 0, st, cc, d_S, (long long)nh, d_cs + h0,
                         d_xz + (size_t)h0 * d, (const double*)nullptr, nh, wsub, halo, 1.0, scale, d_band, (double*)nullptr);
    }
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipEventRecord(c->ev_band_fin[pass & 1], st));
    HIP_TRY(c, hipStreamWaitEvent(c->copy_stream, c->ev_band_fin[pass & 1], 0));
    HIP_TRY(c, hipMemcpyAsync(band + (size_t)h0 * ((size_t)halo + 1), d_band, sizeof(float) * (size_t)nh * ((size_t)halo + 1),
                              hipMemcpyDeviceToHost, c->copy_stream));
    HIP_TRY(c, hipEventRecord(c->ev_band_copied[pass & 1], c->copy_stream));
  }
  HIP_TRY(c, hipMemcpyAsync(xz, d_xz, sizeof(double) * (size_t)W * d, hipMemcpyDeviceToHost, st));
  HIP_TRY(c, hipMemcpyAsync(polymorphic, d_poly, sizeof(int) * (size_t)W, hipMemcpyDeviceToHost, st));
  if (d_bad) HIP_TRY(c, hipMemcpyAsync(&h_bad, d_bad, sizeof(int), hipMemcpyDeviceToHost, st));
  HIP_TRY(c, sync_stream(st));
  HIP_TRY(c, sync_stream(c->copy_stream));
  if (h_bad) return cov_band_impl(c, dG, ring, col0, H, W, halo, scale, band, xz, zz, polymorphic, false);  // not hard calls after all
  if (zz) std::memcpy(zz, zzv.data(), sizeof(double) * (size_t)d * d);
  return RVT_OK;
}
int rvt_cov_band_last_path(rvt_ctx* c) { return c ? c->band_last_path : -1; }
int rvt_cov_band(rvt_ctx* c, const double* dG, int ring_cols, int col0, int H, int W, int halo, float scale, float* band,
                 double* xz, double* zz, int* polymorphic) {
  return cov_band_impl(c, dG, ring_cols, col0, H, W, halo, scale, band, xz, zz, polymorphic, true);
}

// MetaCov with kinship for windows wider than one block (MetaCovFamQtl / MetaCovFamBinary, src/Model.cpp:437-504,595-692):
// heads [col0, col0 + H) against markers [col0, col0 + W) of the RAW block dG.  The W columns are rotated by U'
// (integer planes), then S = (D G~_H)' G~_W and T = G~_W' D [U'X | u1] are two more integer-plane products and the
// centring algebra of the block kernel finishes the rows.  Same outputs as rvt_cov_rect.
// (ring > 0: dG is a block used as a ring of `ring` columns, logical column j = physical (col0 + j) mod ring.
//  halo < 0: the rectangle cov[h + j H]; halo >= 0: the band, band[h (halo + 1) + t] = (float)value(h, h + t) * scale.)
static int cov_rect_fam_impl(rvt_ctx* c, const double* dG, int ring, int col0, int H, int W, int halo, float scale, double* cov,
                             float* band, double* xz, double* zz, int* polymorphic) {
  if (!c || !dG || col0 < 0 || H < 1 || W < H || (!cov && !band) || !xz || !polymorphic)
    return fail(c, RVT_E_INVALID, "bad arguments");
  if (!c->have_fam) return fail(c, RVT_E_STATE, "rvt_set_kinship + rvt_fit_fam_null first");
  if (ring > 0 && (col0 >= ring || W > ring)) return fail(c, RVT_E_INVALID, "window of %d columns from %d in a ring of %d", W, col0, ring);
  hipSetDevice(c->device);
  int rc = rvt_sync(c);
  if (rc) return rc;
  hipStream_t st = c->stream;
  const int64_t N = c->fam_nc.N, ld = c->fam_nc.ld;
  const int du = c->famcov_nc.d - 2;  // columns of U'X
  CovConsts cc;
  std::vector<double> zzv;
  {
    const NullConsts keep = c->nc;
    c->nc = c->famcov_nc;
    rc = cov_constants(c, true, &cc, &zzv);
    c->nc = keep;
    if (rc) return rc;
  }
  rc = ensure_fam_cols(c, (size_t)W, ld);
  if (rc) return rc;
  // the window's columns as at most two contiguous runs of the ring
  struct Seg {
    int phys, n, at;
  };
  Seg segs[2];
  int nseg = 1;
  segs[0] = Seg{col0, W, 0};
  if (ring > 0 && col0 + W > ring) {
    segs[0].n = ring - col0;
    segs[1] = Seg{0, W - segs[0].n, segs[0].n};
    nseg = 2;
  }
  double *d_S = nullptr, *d_T = nullptr, *d_cs = nullptr, *d_xz = nullptr, *d_cov = nullptr, *d_w = nullptr, *d_t1 = nullptr;
  float* d_band = nullptr;
  int* d_poly = nullptr;
  struct Guard {
    std::vector<void**> p;
    ~Guard() {
      for (void** q : p)
        if (*q) hipFree(*q);
    }
  } guard{{(void**)&d_S, (void**)&d_T, (void**)&d_cs, (void**)&d_xz, (void**)&d_cov, (void**)&d_w, (void**)&d_t1,
           (void**)&d_poly, (void**)&d_band}};
  HIP_TRY(c, hipMalloc((void**)&d_S, sizeof(double) * (size_t)H * W));
  if (halo < 0) HIP_TRY(c, hipMalloc((void**)&d_cov, sizeof(double) * (size_t)H * W));
  else HIP_TRY(c, hipMalloc((void**)&d_band, sizeof(float) * (size_t)H * ((size_t)halo + 1)));
  HIP_TRY(c, hipMalloc((void**)&d_T, sizeof(double) * (size_t)W * (du + 1)));
  HIP_TRY(c, hipMalloc((void**)&d_xz, sizeof(double) * (size_t)W * du));
  HIP_TRY(c, hipMalloc((void**)&d_t1, sizeof(double) * (size_t)W));
  HIP_TRY(c, hipMalloc((void**)&d_cs, sizeof(double) * (size_t)W));
  HIP_TRY(c, hipMalloc((void**)&d_poly, sizeof(int) * (size_t)W));
  HIP_TRY(c, hipMalloc((void**)&d_w, sizeof(double) * (size_t)ld * (size_t)(du + 1 + H)));
  HIP_TRY(c, hipMemsetAsync(c->d_Gt, 0, sizeof(double) * (size_t)ld * W, st));
  for (int k = 0; k < nseg; ++k) {
    const double* GW = dG + (size_t)segs[k].phys * ld;
    k_raw_colstat(dim3((unsigned)segs[k].n), st, GW, (long long)N, (long long)ld, d_cs + segs[k].at, d_poly + segs[k].at);
    rc = rotate_columns(c, GW, ld, segs[k].n, c->d_Gt + (size_t)segs[k].at * ld, ld, st);
    if (rc) return rc;
  }
  // the weights D = 1 / ((|lambda| + delta) sigma2) ride on the small operands: D [U'X | u1] and D G~_H
  HIP_TRY(c, hipMemsetAsync(d_w, 0, sizeof(double) * (size_t)ld * (size_t)(du + 1 + H), st));
  hipLaunchKernelGGL(scale_rows_kernel, dim3(64, (unsigned)(du + 1)), dim3(256), 0, st, c->d_cX, c->d_cv, (long long)N,
                     (long long)ld, d_w);
  hipLaunchKernelGGL(scale_rows_kernel, dim3(64, (unsigned)H), dim3(256), 0, st, c->d_Gt, c->d_cv, (long long)N,
                     (long long)ld, d_w + (size_t)ld * (du + 1));
  rc = gemm_tn_planes(c, c->d_Gt, ld, W, d_w, ld, du + 1, N, d_T, W, st);
  if (rc) return rc;
  rc = gemm_tn_planes(c, d_w + (size_t)ld * (du + 1), ld, H, c->d_Gt, ld, W, N, d_S, H, st);
  if (rc) return rc;
  hipLaunchKernelGGL(cov_rect_fam_xz_kernel, dim3((unsigned)((W + 255) / 256)), dim3(256), 0, st, cc, d_T, d_cs, W, d_xz,
                     d_t1);
  const double b2 = c->famcov_b2;  // MetaCovFamBinary: covXX, covXZ, covZZ each carry b^2 (Model.cpp:651-668)
  if (halo < 0) {
    hipLaunchKernelGGL(cov_rect_fam_rows_kernel, dim3((unsigned)H), dim3(256), 0, st, cc, d_S, d_cs, d_xz, d_t1, H, W,
                       d_cov);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(cov, d_cov, sizeof(double) * (size_t)H * W, hipMemcpyDeviceToHost, st));
  } else {
    hipLaunchKernelGGL(band_rows_f64_kernel, dim3((unsigned)H), dim3(256), 0, st, cc, d_S, (long long)H, d_cs, d_xz,
                       (const double*)d_t1, H, W, halo, b2, scale, d_band, (double*)nullptr);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(band, d_band, sizeof(float) * (size_t)H * ((size_t)halo + 1), hipMemcpyDeviceToHost, st));
  }
  HIP_TRY(c, hipMemcpyAsync(xz, d_xz, sizeof(double) * (size_t)W * du, hipMemcpyDeviceToHost, st));
  HIP_TRY(c, hipMemcpyAsync(polymorphic, d_poly, sizeof(int) * (size_t)W, hipMemcpyDeviceToHost, st));
  HIP_TRY(c, sync_stream(st));
  if (zz) std::memcpy(zz, zzv.data(), sizeof(double) * (size_t)du * du);
  if (b2 != 1.0) {
    if (halo < 0)
      for (int h = 0; h < H; ++h)
        for (int j = h; j < W; ++j) cov[(size_t)h + (size_t)j * H] *= b2;
    for (size_t i = 0; i < (size_t)W * du; ++i) xz[i] *= b2;
    if (zz)
      for (int i = 0; i < du * du; ++i) zz[i] *= b2;
  }
  return RVT_OK;
}
int rvt_cov_rect_fam(rvt_ctx* c, const double* dG, int col0, int H, int W, double* cov, double* xz, double* zz,
                     int* polymorphic) {
  if (!cov) return fail(c, RVT_E_INVALID, "bad arguments");
  return cov_rect_fam_impl(c, dG, 0, col0, H, W, -1, 1.0f, cov, nullptr, xz, zz, polymorphic);
}
// The family counterpart of rvt_cov_band: the heads in passes of up to 1 024 (every pass rotates the columns of its heads and
// of the window behind them).  xz / polymorphic: the W logical columns.
int rvt_cov_band_fam(rvt_ctx* c, const double* dG, int ring_cols, int col0, int H, int W, int halo, float scale, float* band,
                     double* xz, double* zz, int* polymorphic) {
  if (!c || !dG || ring_cols < 0 || col0 < 0 || H < 1 || W < H || halo < 0 || !band || !xz || !polymorphic)
    return fail(c, RVT_E_INVALID, "bad arguments");
  if (!c->have_fam) return fail(c, RVT_E_STATE, "rvt_set_kinship + rvt_fit_fam_null first");
  if ((long long)W > (long long)H + halo) W = H + halo;
  const int du = c->famcov_nc.d - 2, Hc = 1024;
  std::vector<double> xzp;
  std::vector<int> pp;
  for (int h0 = 0; h0 < H; h0 += Hc) {
    const int nh = std::min(Hc, H - h0);
    const int wsub = (int)std::min<long long>((long long)W - h0, (long long)nh + halo);
    int pc = col0 + h0;
    if (ring_cols > 0 && pc >= ring_cols) pc -= ring_cols;
    xzp.assign((size_t)wsub * du, 0.0);
    pp.assign((size_t)wsub, 0);
    int rc = cov_rect_fam_impl(c, dG, ring_cols, pc, nh, wsub, halo, scale, nullptr, band + (size_t)h0 * ((size_t)halo + 1),
                               xzp.data(), zz, pp.data());
    if (rc) return rc;
    std::memcpy(xz + (size_t)h0 * du, xzp.data(), sizeof(double) * xzp.size());
    std::memcpy(polymorphic + h0, pp.data(), sizeof(int) * pp.size());
  }
  return RVT_OK;
}

// the column cache of a block (int8 copy, E2M1 copy, sums, flags, rows of T): an optimisation — a failed allocation leaves the
// block without one (every column invalid; the covariance calls then run their own column pass), it does not fail the call
static void free_col_cache(rvt_ctx::ColKind& ck) {
  for (void* q : {(void*)ck.d_i8, (void*)ck.d_i4, (void*)ck.d_m4, (void*)ck.d_mu, (void*)ck.d_cs, (void*)ck.d_poly, (void*)ck.d_T})
    if (q) hipFree(q);
  ck.d_i8 = nullptr;
  ck.d_i4 = nullptr;
  ck.d_m4 = nullptr;
  ck.d_mu = nullptr;
  ck.d_cs = nullptr;
  ck.d_poly = nullptr;
  ck.d_T = nullptr;
  std::fill(ck.valid.begin(), ck.valid.end(), 0);
}
static bool alloc_col_cache(rvt_ctx* c, rvt_ctx::ColKind& ck, int64_t ldk, int64_t ldk4, uint64_t gen, hipStream_t st) {
  const size_t cap = ((size_t)ck.cols + 255) / 256 * 256 + 256;  // (the products read whole tiles of columns)
  bool ok = hipMalloc((void**)&ck.d_i8, cap * (size_t)ldk) == hipSuccess &&
            hipMalloc((void**)&ck.d_i4, cap * (size_t)ldk4) == hipSuccess &&
            hipMalloc((void**)&ck.d_m4, cap * (size_t)ldk4) == hipSuccess &&
            hipMalloc((void**)&ck.d_mu, sizeof(double) * (size_t)ck.cols) == hipSuccess &&
            hipMalloc((void**)&ck.d_cs, sizeof(double) * (size_t)ck.cols) == hipSuccess &&
            hipMalloc((void**)&ck.d_poly, sizeof(int) * (size_t)ck.cols) == hipSuccess &&
            hipMalloc((void**)&ck.d_T, sizeof(double) * (size_t)ck.cols * RVT_MAX_COV) == hipSuccess;
  ok = ok && hipMemsetAsync(ck.d_i8, 0, cap * (size_t)ldk, st) == hipSuccess &&
       hipMemsetAsync(ck.d_i4, 0, cap * (size_t)ldk4, st) == hipSuccess &&
       hipMemsetAsync(ck.d_m4, 0, cap * (size_t)ldk4, st) == hipSuccess &&
       hipMemsetAsync(ck.d_mu, 0, sizeof(double) * (size_t)ck.cols, st) == hipSuccess;
  ck.valid.assign((size_t)ck.cols, 0);
  if (!ok) {
    (void)hipGetLastError();
    free_col_cache(ck);
    return false;
  }
  ck.ldk = ldk;
  ck.ldk4 = ldk4;
  ck.gen = gen;
  (void)c;
  return true;
}

// the column cache of ncols columns from (block s, column sc) to (block t, column tc), t == s allowed (a forward move: tc < sc);
// columns whose source has no valid entry become invalid in the target
static int move_col_cache(rvt_ctx* c, rvt_ctx::ColKind* t, int tc, const rvt_ctx::ColKind* s, int sc, int ncols, hipStream_t st) {
  if (!t || tc + ncols > t->cols) return RVT_OK;
  const bool src_ok = s && s->d_i8 && sc + ncols <= s->cols;
  if (!src_ok || (t->d_i8 && (t->ldk != s->ldk || t->gen != s->gen))) {
    if (!t->valid.empty())
      for (int k = 0; k < ncols; ++k) t->valid[(size_t)(tc + k)] = 0;
    return RVT_OK;
  }
  if (!t->d_i8) {  // the target block has no cache yet: same shape as the source's
    if (!s->d_i4 || !alloc_col_cache(c, *t, s->ldk, s->ldk4, s->gen, st)) {
      if (!t->valid.empty())
        for (int k = 0; k < ncols; ++k) t->valid[(size_t)(tc + k)] = 0;
      return RVT_OK;
    }
  }
  // (forward, in pieces no longer than the shift: a piece never overwrites what it has not read)
  const int shift = (t == s) ? sc - tc : ncols;
  for (int k0 = 0; k0 < ncols; k0 += std::max(shift, 1)) {
    const int nk = std::min(std::max(shift, 1), ncols - k0);
    HIP_TRY(c, hipMemcpyAsync(t->d_i8 + (size_t)(tc + k0) * (size_t)s->ldk, s->d_i8 + (size_t)(sc + k0) * (size_t)s->ldk,
                              (size_t)nk * (size_t)s->ldk, hipMemcpyDeviceToDevice, st));
    if (t->d_i4 && s->d_i4)
      HIP_TRY(c, hipMemcpyAsync(t->d_i4 + (size_t)(tc + k0) * (size_t)s->ldk4, s->d_i4 + (size_t)(sc + k0) * (size_t)s->ldk4,
                                (size_t)nk * (size_t)s->ldk4, hipMemcpyDeviceToDevice, st));
    if (t->d_m4 && s->d_m4) {
      HIP_TRY(c, hipMemcpyAsync(t->d_m4 + (size_t)(tc + k0) * (size_t)s->ldk4, s->d_m4 + (size_t)(sc + k0) * (size_t)s->ldk4,
                                (size_t)nk * (size_t)s->ldk4, hipMemcpyDeviceToDevice, st));
      HIP_TRY(c, hipMemcpyAsync(t->d_mu + tc + k0, s->d_mu + sc + k0, sizeof(double) * (size_t)nk, hipMemcpyDeviceToDevice, st));
    }
    HIP_TRY(c, hipMemcpyAsync(t->d_cs + tc + k0, s->d_cs + sc + k0, sizeof(double) * (size_t)nk, hipMemcpyDeviceToDevice, st));
    HIP_TRY(c, hipMemcpyAsync(t->d_poly + tc + k0, s->d_poly + sc + k0, sizeof(int) * (size_t)nk, hipMemcpyDeviceToDevice, st));
    HIP_TRY(c, hipMemcpyAsync(t->d_T + (size_t)(tc + k0) * RVT_MAX_COV, s->d_T + (size_t)(sc + k0) * RVT_MAX_COV,
                              sizeof(double) * (size_t)nk * RVT_MAX_COV, hipMemcpyDeviceToDevice, st));
  }
  for (int k = 0; k < ncols; ++k) t->valid[(size_t)(tc + k)] = s->valid[(size_t)(sc + k)];
  return RVT_OK;
}

int rvt_block_copy_columns(rvt_ctx* c, double* dst, int dst_col, const double* src, int src_col, int ncols) {
  if (!c || !dst || !src || dst_col < 0 || src_col < 0 || ncols < 0) return fail(c, RVT_E_INVALID, "bad copy");
  if (ncols == 0) return RVT_OK;
  hipSetDevice(c->device);
  if (c->colq.n > 0) {
    int rc = flush_col_queue(c);
    if (rc) return rc;
  }
  HIP_TRY(c, sync_stream(c->io_stream));  // (queued uploads and the passes behind them write what is copied here)
  const size_t ld = (size_t)(c->have_null ? c->null_ld : c->fam_nc.ld);
  HIP_TRY(c, hipMemcpy(dst + (size_t)dst_col * ld, src + (size_t)src_col * ld, sizeof(double) * ld * ncols,
                       hipMemcpyDeviceToDevice));
  // what the engine knows about the columns travels with them: content flags and the column cache
  auto itd = c->col_kind.find(dst);
  auto its = c->col_kind.find(src);
  if (itd != c->col_kind.end() && dst == src) {
    // a copy inside one block (not the forward move of rvt_block_move_columns): the target columns' flags and cache entries
    // describe what was there before — unknown from here on
    rvt_ctx::ColKind& t = itd->second;
    for (int k = 0; k < ncols && !t.valid.empty(); ++k)
      if ((size_t)(dst_col + k) < t.valid.size()) t.valid[(size_t)(dst_col + k)] = 0;
    if (t.d_flags && dst_col < t.cols) {
      HIP_TRY(c, hipMemsetAsync(t.d_flags + dst_col, 0, sizeof(int) * (size_t)std::min(ncols, t.cols - dst_col), c->io_stream));
      HIP_TRY(c, sync_stream(c->io_stream));
    }
  }
  if (itd != c->col_kind.end() && dst != src) {
    rvt_ctx::ColKind& t = itd->second;
    const rvt_ctx::ColKind* s = its != c->col_kind.end() ? &its->second : nullptr;
    if (dst_col + ncols <= t.cols) {
      if (s && s->d_flags && src_col + ncols <= s->cols) {
        if (!t.d_flags) {
          HIP_TRY(c, hipMalloc((void**)&t.d_flags, sizeof(int) * (size_t)t.cols));
          HIP_TRY(c, hipMemsetAsync(t.d_flags, 0x01, sizeof(int) * (size_t)t.cols, c->io_stream));
        }
        HIP_TRY(c, hipMemcpyAsync(t.d_flags + dst_col, s->d_flags + src_col, sizeof(int) * (size_t)ncols, hipMemcpyDeviceToDevice,
                                  c->io_stream));
      } else if (t.d_flags) {
        HIP_TRY(c, hipMemsetAsync(t.d_flags + dst_col, 0, sizeof(int) * (size_t)ncols, c->io_stream));  // (unknown: not hard calls)
      }
      int rc = move_col_cache(c, &t, dst_col, s, src_col, ncols, c->io_stream);
      if (rc) return rc;
      HIP_TRY(c, sync_stream(c->io_stream));
    }
  }
  return RVT_OK;
}

// What the engine keeps per column of a block filled by rvt_block_upload_columns, for n <= kColQueue columns that crossed PCIe
// PACKED (their content is known from the packing: hard[k] = hard calls only, otherwise hard calls plus the one other value
// mu[k]) and have been expanded into dG on io_stream: the content flags, and under an unweighted null model the column pass
// of MetaCov's band on these columns (int8 copy, E2M1 codes of the hard calls and of the other value's mask, sum, flag, row of
// T) — cache state 1 / 2.
static int packed_columns_pass(rvt_ctx* c, double* dG, int col0, int n, const double* mu, const int* hard) {
  hipStream_t st = c->io_stream;
  const size_t N = (size_t)(c->have_null ? c->nc.N : c->fam_nc.N);
  const size_t ld = (size_t)(c->have_null ? c->null_ld : c->fam_nc.ld);
  auto it = c->col_kind.find(dG);
  if (it == c->col_kind.end() || !c->hc_enabled || col0 + n > it->second.cols) return RVT_OK;
  rvt_ctx::ColKind& ck = it->second;
  if (!ck.d_flags) {
    HIP_TRY(c, hipMalloc((void**)&ck.d_flags, sizeof(int) * (size_t)ck.cols));
    HIP_TRY(c, hipMemsetAsync(ck.d_flags, 0x01, sizeof(int) * (size_t)ck.cols, st));
  }
  // the content of a packed column is KNOWN: hard calls only unless it has an other value
  int rc = small_h2d(c, ck.d_flags + col0, hard, sizeof(int) * (size_t)n);
  if (rc) return rc;
  bool cache = c->have_null && !c->nc.binary && !getenv("RVT_METACOV_NO_CACHE");
  if (cache) {
    const int d = c->nc.d, dmax = d <= 4 ? 4 : (d <= 8 ? 8 : RVT_MAX_COV);
    const int64_t ldk = ((int64_t)N + 127) / 128 * 128, ldk4 = (((int64_t)N + 1) / 2 + 127) / 128 * 128;
    if (ck.d_i8 && (ck.ldk != ldk || ck.gen != c->null_gen || !ck.d_i4)) free_col_cache(ck);
    if (!ck.d_i8 && !ck.cache_failed && !alloc_col_cache(c, ck, ldk, ldk4, c->null_gen, st)) ck.cache_failed = true;
    cache = ck.d_i8 != nullptr;
    if (cache) {
      const size_t part_doubles = (size_t)kCovSlices * rvt_ctx::kColQueue * (RVT_MAX_COV + 3);
      if (!c->d_cc_part) HIP_TRY(c, hipMalloc((void**)&c->d_cc_part, sizeof(double) * part_doubles));
      const int slices = (int)std::max<int64_t>(1, std::min<int64_t>(kCovSlices, (int64_t)N / 4096 + 1));
      // (the columns' other values, NaN where a column has none: the pass splits g = h + mu m by them)
      double mu_nan[rvt_ctx::kColQueue];
      for (int k = 0; k < n; ++k) mu_nan[k] = hard[k] ? (double)NAN : mu[k];
      if (!c->d_mu_nan) HIP_TRY(c, hipMalloc((void**)&c->d_mu_nan, sizeof(double) * (size_t)rvt_ctx::kColQueue));
      double* d_mu_nan = c->d_mu_nan;
      rc = small_h2d(c, d_mu_nan, mu_nan, sizeof(double) * (size_t)n);
      if (rc) return rc;
      rc = small_h2d(c, ck.d_mu + col0, mu, sizeof(double) * (size_t)n);
      if (rc) return rc;
      launch_cov_prep(st, d, true, dim3((unsigned)((n + kCovHcCols - 1) / kCovHcCols), (unsigned)slices), dG + (size_t)col0 * ld,
                      (int64_t)N, (int64_t)ld, n, c->d_X, ck.d_i8 + (size_t)col0 * (size_t)ldk, ldk, c->d_cc_part, nullptr, nullptr,
                      nullptr, 0, 0, ck.d_i4 + (size_t)col0 * (size_t)ldk4, ldk4, d_mu_nan, ck.d_m4 + (size_t)col0 * (size_t)ldk4);
      hipLaunchKernelGGL(cov_hc_finish_kernel, dim3((unsigned)((n * (dmax + 3) + 255) / 256)), dim3(256), 0, st, c->d_cc_part, slices,
                         n, d, dmax, ck.d_cs + col0, ck.d_poly + col0, ck.d_T + (size_t)col0 * RVT_MAX_COV, RVT_MAX_COV);
      HIP_TRY(c, hipGetLastError());
      for (int k = 0; k < n; ++k) ck.valid[(size_t)(col0 + k)] = (unsigned char)(hard[k] ? 1 : 2);
    }
  }
  return RVT_OK;
}

static void free_col_cache(rvt_ctx::ColKind& ck);
static bool alloc_col_cache(rvt_ctx* c, rvt_ctx::ColKind& ck, int64_t ldk, int64_t ldk4, uint64_t gen, hipStream_t st);
// What rvt_block_upload_columns queued (rvt_ctx::ColQueue: up to 32 consecutive columns of one block, packed to 2-bit rows in
// pinned memory, their other values and hard-call flags known on the host) goes to the device: ONE DMA of the rows, the other
// values, the flags; the expansion to doubles; and — when the block keeps a column cache — the column pass over all of them.
int flush_col_queue(rvt_ctx* c) {
  rvt_ctx::ColQueue& q = c->colq;
  const int n = q.n;
  if (n == 0) return RVT_OK;
  q.n = 0;
  hipSetDevice(c->device);
  const size_t N = (size_t)(c->have_null ? c->nc.N : c->fam_nc.N);
  const size_t ld = (size_t)(c->have_null ? c->null_ld : c->fam_nc.ld);
  const size_t pitch = q.pitch;
  const size_t need = pitch * (size_t)rvt_ctx::kColQueue + 2 * sizeof(double) * (size_t)rvt_ctx::kColQueue;
  if (c->colpack_cap < need) {
    HIP_TRY(c, sync_stream(c->io_stream));
    if (c->d_colpack) hipFree(c->d_colpack);
    c->d_colpack = nullptr;
    c->colpack_cap = 0;
    HIP_TRY(c, hipMalloc((void**)&c->d_colpack, need + need / 2));
    c->colpack_cap = need + need / 2;
  }
  hipStream_t st = c->io_stream;
  double* dG = q.dG;
  const int col0 = q.col0;
  double* d_mu = reinterpret_cast<double*>(c->d_colpack + pitch * (size_t)rvt_ctx::kColQueue);
  HIP_TRY(c, hipMemcpyAsync(c->d_colpack, q.h[q.cur], pitch * (size_t)n, hipMemcpyHostToDevice, st));
  if (!q.ev[q.cur]) HIP_TRY(c, hipEventCreateWithFlags(&q.ev[q.cur], hipEventDisableTiming));
  HIP_TRY(c, hipEventRecord(q.ev[q.cur], st));
  q.used[q.cur] = true;
  q.cur ^= 1;
  int rc = small_h2d(c, d_mu, q.mu, sizeof(double) * (size_t)n);
  if (rc) return rc;
  hipLaunchKernelGGL(bed_expand_columns_kernel, dim3((unsigned)((N + 1023) / 1024), (unsigned)n), dim3(256), 0, st,
                     reinterpret_cast<const unsigned char*>(c->d_colpack), (long long)pitch, d_mu, (long long)N, (long long)ld,
                     dG + (size_t)col0 * ld);
  HIP_TRY(c, hipGetLastError());
  return packed_columns_pass(c, dG, col0, n, q.mu, q.hard);
}

int rvt_block_upload_columns(rvt_ctx* c, double* dG, int col0, int ncols, const double* G) {
  if (!c || !dG || !G || col0 < 0 || ncols < 1) return fail(c, RVT_E_INVALID, "bad upload");
  if (!c->have_null && !c->have_fam) return fail(c, RVT_E_STATE, "set the null model first");
  hipSetDevice(c->device);
  const size_t N = (size_t)(c->have_null ? c->nc.N : c->fam_nc.N);
  const size_t ld = (size_t)(c->have_null ? c->null_ld : c->fam_nc.ld);
  // ONE column (a site of MetaCovTest / MetaScoreTest::fit): packed by the staging threads into pinned memory and queued; the
  // device work runs once per 32 consecutive columns (flush_col_queue)
  if (ncols == 1 && c->hc_enabled && !getenv("RVT_UPLOAD_FP64") && !getenv("RVT_UPLOAD_NO_QUEUE") && N >= 4096) {
    rvt_ctx::ColQueue& q = c->colq;
    const size_t pitch = ((N + 3) / 4 + 15) / 16 * 16;
    if (q.n > 0 && (q.dG != dG || col0 != q.col0 + q.n || q.n == rvt_ctx::kColQueue || q.pitch != pitch)) {
      int rc = flush_col_queue(c);
      if (rc) return rc;
    }
    if (q.pitch != pitch) {  // (another N: new pinned rows)
      HIP_TRY(c, sync_stream(c->io_stream));
      for (int i = 0; i < 2; ++i) {
        if (q.h[i]) hipHostFree(q.h[i]);
        q.h[i] = nullptr;
        q.used[i] = false;
      }
      q.pitch = pitch;
    }
    for (int i = 0; i < 2; ++i)
      if (!q.h[i]) HIP_TRY(c, hipHostMalloc((void**)&q.h[i], pitch * (size_t)rvt_ctx::kColQueue, hipHostMallocDefault));
    if (q.n == 0) {
      if (q.used[q.cur]) {  // the DMA that last read these rows has finished?
        while (hipEventQuery(q.ev[q.cur]) == hipErrorNotReady) {
          struct timespec ts = {0, 20000};
          nanosleep(&ts, nullptr);
        }
        (void)hipGetLastError();
      }
      q.dG = dG;
      q.col0 = col0;
    }
    PackedColumn pc;
    if (StageRing::pack_columns_to(reinterpret_cast<char*>(q.h[q.cur]) + pitch * (size_t)q.n, pitch, G, N, N, 1,
                                   CopyPool::column_instance(), &pc)) {
      // whatever the engine knew about the overwritten column is void from here on
      auto itq = c->col_kind.find(dG);
      if (itq != c->col_kind.end() && (size_t)col0 < itq->second.valid.size()) itq->second.valid[(size_t)col0] = 0;
      q.mu[q.n] = pc.has_mu ? pc.mu : 0.0;
      q.hard[q.n] = pc.has_mu ? 0 : 1;
      ++q.n;
      return RVT_OK;
    }
    // not representable (dosages): this column crosses as doubles — behind whatever was queued before it
    int rc = flush_col_queue(c);
    if (rc) return rc;
  } else if (c->colq.n > 0) {
    int rc = flush_col_queue(c);
    if (rc) return rc;
  }
  // The columns cross PCIe as 2-bit codes when they are what consolidate() almost always leaves — hard calls plus at most one
  // other value per column (the imputed mean): the staging threads pack them (host_stage.h pack_column_f64, as for the genes
  // of rvt_submit_gene), 1/32 of the bytes go over the link, a small kernel writes the doubles of the block back.  Until round 6
  // every site's column crossed as 4 MB of doubles out of pageable memory (~120 us per site at N = 500 000: the adapter's
  // `--meta cov` ran at 8 k sites/s whatever the window).  Dosages (a second other value) cross as doubles, as before.
  bool packed = false;
  std::vector<double> pmu;   // (packed: the columns' other values and whether they have none)
  std::vector<int> phard;
  if (c->hc_enabled && !getenv("RVT_UPLOAD_FP64") && N >= 4096) {
    int rc = stage_ready(c);
    if (rc) return rc;
    const size_t pitch = ((N + 3) / 4 + 15) / 16 * 16;
    const size_t need = pitch * (size_t)ncols + sizeof(double) * (size_t)ncols;
    if (c->colpack_cap < need) {
      if (c->d_colpack) hipFree(c->d_colpack);
      c->d_colpack = nullptr;
      c->colpack_cap = 0;
      HIP_TRY(c, hipMalloc((void**)&c->d_colpack, need + need / 2));
      c->colpack_cap = need + need / 2;
    }
    if (pitch <= c->stage.chunk_bytes) {
      std::vector<PackedColumn> pc((size_t)ncols);
      const int prc = c->stage.pack_f64(c->d_colpack, pitch, G, N, N, (size_t)ncols, CopyPool::pack_instance(), pc.data());
      if (prc == 1) return fail(c, RVT_E_HIP, "packing the uploaded columns failed");
      if (prc == 0) {
        double* d_mu = reinterpret_cast<double*>(c->d_colpack + pitch * (size_t)ncols);
        std::vector<double> mu((size_t)ncols);
        for (int j = 0; j < ncols; ++j) mu[(size_t)j] = pc[(size_t)j].has_mu ? pc[(size_t)j].mu : 0.0;
        rc = small_h2d(c, d_mu, mu.data(), sizeof(double) * (size_t)ncols);
        if (rc) return rc;
        hipLaunchKernelGGL(bed_expand_columns_kernel, dim3((unsigned)((N + 1023) / 1024), (unsigned)ncols), dim3(256), 0, c->io_stream,
                           reinterpret_cast<const unsigned char*>(c->d_colpack), (long long)pitch, d_mu, (long long)N, (long long)ld,
                           dG + (size_t)col0 * ld);
        HIP_TRY(c, hipGetLastError());
        packed = true;
        pmu = mu;
        phard.resize((size_t)ncols);
        for (int j = 0; j < ncols; ++j) phard[(size_t)j] = pc[(size_t)j].has_mu ? 0 : 1;
      }
    }
  }
  if (!packed)
    HIP_TRY(c, hipMemcpy2D(dG + (size_t)col0 * ld, sizeof(double) * ld, G, sizeof(double) * N, sizeof(double) * N, ncols,
                           hipMemcpyHostToDevice));
  // content of the new columns (hard calls or not), recorded per column: rvt_score_block picks its kernel by it.  One
  // read of data that has just crossed PCIe at a hundredth of the rate.  Under an unweighted null model that read is the
  // column pass of MetaCov's hard-call band itself (cov_hc_prep_kernel on the one column): int8 copy, sum, min / max and the
  // row of T = G'X stay with the block (ColKind), the flag falls out of its value test.
  auto it = c->col_kind.find(dG);
  if (it != c->col_kind.end()) {
    // whatever the engine knew about the overwritten columns is void from here on — also when the passes below do not run
    // (hard calls switched off, RVT_METACOV_NO_CACHE): a stale cache entry would hand the OLD column to the integer product,
    // which no longer tests what it reads once the entry is marked valid
    rvt_ctx::ColKind& ck = it->second;
    for (int k = 0; k < ncols && !ck.valid.empty(); ++k)
      if ((size_t)(col0 + k) < ck.valid.size()) ck.valid[(size_t)(col0 + k)] = 0;
    if (ck.d_flags && !c->hc_enabled && col0 < ck.cols)  // (flags are not recomputed below: unknown = not hard calls)
      HIP_TRY(c, hipMemsetAsync(ck.d_flags + col0, 0, sizeof(int) * (size_t)std::min(ncols, ck.cols - col0), c->io_stream));
  }
  if (packed) {  // content known from the packing: the same pass as behind the queued single columns, 32 columns at a time
    for (int k0 = 0; k0 < ncols; k0 += rvt_ctx::kColQueue) {
      const int nk = std::min(rvt_ctx::kColQueue, ncols - k0);
      int rc = packed_columns_pass(c, dG, col0 + k0, nk, pmu.data() + k0, phard.data() + k0);
      if (rc) return rc;
    }
    return RVT_OK;
  }
  if (it != c->col_kind.end() && c->hc_enabled && col0 + ncols <= it->second.cols) {
    rvt_ctx::ColKind& ck = it->second;
    if (!ck.d_flags) {
      HIP_TRY(c, hipMalloc((void**)&ck.d_flags, sizeof(int) * (size_t)ck.cols));
      HIP_TRY(c, hipMemsetAsync(ck.d_flags, 0x01, sizeof(int) * (size_t)ck.cols, c->io_stream));
    }
    bool cache = c->have_null && !c->nc.binary && !getenv("RVT_METACOV_NO_CACHE");
    const int d = c->nc.d, dmax = d <= 4 ? 4 : (d <= 8 ? 8 : RVT_MAX_COV);
    const int64_t ldk = ((int64_t)N + 127) / 128 * 128, ldk4 = (((int64_t)N + 1) / 2 + 127) / 128 * 128;
    if (cache) {
      if (ck.d_i8 && (ck.ldk != ldk || ck.gen != c->null_gen || !ck.d_i4)) free_col_cache(ck);  // another model: nothing of the old cache is used
      if (!ck.d_i8 && !ck.cache_failed && !alloc_col_cache(c, ck, ldk, ldk4, c->null_gen, c->io_stream)) ck.cache_failed = true;
      cache = ck.d_i8 != nullptr;
      if (cache && !c->d_cc_part)
        HIP_TRY(c, hipMalloc((void**)&c->d_cc_part, sizeof(double) * (size_t)kCovSlices * rvt_ctx::kColQueue * (RVT_MAX_COV + 3)));
    }
    for (int k = 0; k < ncols; ++k) {
      const int col = col0 + k;
      if (!cache) {
        int rc = enqueue_classify(c, dG + (size_t)col * ld, 1, (int64_t)N, (int64_t)ld, c->io_stream, ck.d_flags + col);
        if (rc) return rc;
        continue;
      }
      static const int one = 1;
      HIP_TRY(c, hipMemcpyAsync(ck.d_flags + col, &one, sizeof(int), hipMemcpyHostToDevice, c->io_stream));
      // (this pass knows no other value: the slot's mask of an earlier occupant must not survive it — a window that mixes
      //  this column with mean-imputed ones reads the masks of all its columns)
      if (ck.d_m4) HIP_TRY(c, hipMemsetAsync(ck.d_m4 + (size_t)col * (size_t)ldk4, 0, (size_t)ldk4, c->io_stream));
      const int slices = (int)std::max<int64_t>(1, std::min<int64_t>(kCovSlices, (int64_t)N / 4096 + 1));
      launch_cov_prep(c->io_stream, d, true, dim3(1, (unsigned)slices), dG + (size_t)col * ld, (int64_t)N, (int64_t)ld, 1, c->d_X,
                      ck.d_i8 + (size_t)col * (size_t)ldk, ldk, c->d_cc_part, nullptr, nullptr, ck.d_flags + col, 0, 0,
                      ck.d_i4 + (size_t)col * (size_t)ldk4, ldk4);
      hipLaunchKernelGGL(cov_hc_finish_kernel, dim3(1), dim3(256), 0, c->io_stream, c->d_cc_part, slices, 1, d, dmax,
                         ck.d_cs + col, ck.d_poly + col, ck.d_T + (size_t)col * RVT_MAX_COV);
      HIP_TRY(c, hipGetLastError());
      ck.valid[(size_t)col] = 1;
    }
  }
  return RVT_OK;
}

int rvt_block_move_columns(rvt_ctx* c, double* dG, int dst_col, int src_col, int ncols) {
  if (!c || !dG || dst_col < 0 || src_col < dst_col || ncols < 0) return fail(c, RVT_E_INVALID, "bad move");
  if (ncols == 0 || dst_col == src_col) return RVT_OK;
  hipSetDevice(c->device);
  if (c->colq.n > 0) {
    int rc = flush_col_queue(c);
    if (rc) return rc;
  }
  const size_t ld = (size_t)(c->have_null ? c->null_ld : c->fam_nc.ld);
  HIP_TRY(c, sync_stream(c->io_stream));  // (the passes behind the last uploads write the cache this call moves)
  // forward move of a possibly overlapping range: in pieces no longer than the shift, in increasing order — a piece never
  // overwrites data that has not been read (a ring that drops more than it keeps moves in ONE copy)
  {
    const int shift = src_col - dst_col;
    for (int k0 = 0; k0 < ncols; k0 += shift) {
      const int nk = std::min(shift, ncols - k0);
      HIP_TRY(c, hipMemcpyAsync(dG + (size_t)(dst_col + k0) * ld, dG + (size_t)(src_col + k0) * ld, sizeof(double) * ld * (size_t)nk,
                                hipMemcpyDeviceToDevice, c->stream));
    }
  }
  {
    auto itc = c->col_kind.find(dG);
    if (itc != c->col_kind.end()) {
      int rc = move_col_cache(c, &itc->second, dst_col, &itc->second, src_col, ncols, c->stream);
      if (rc) return rc;
    }
  }
  HIP_TRY(c, sync_stream(c->stream));
  auto it = c->col_kind.find(dG);
  if (it != c->col_kind.end() && it->second.d_flags && src_col + ncols <= it->second.cols) {
    std::vector<int> f((size_t)it->second.cols);
    HIP_TRY(c, hipMemcpyAsync(f.data(), it->second.d_flags, sizeof(int) * f.size(), hipMemcpyDeviceToHost, c->io_stream));
    HIP_TRY(c, sync_stream(c->io_stream));
    std::memmove(f.data() + dst_col, f.data() + src_col, sizeof(int) * (size_t)ncols);
    HIP_TRY(c, hipMemcpyAsync(it->second.d_flags, f.data(), sizeof(int) * f.size(), hipMemcpyHostToDevice, c->io_stream));
    HIP_TRY(c, sync_stream(c->io_stream));
  }
  return RVT_OK;
}

}  // extern "C"
