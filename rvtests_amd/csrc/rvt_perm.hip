// rvtests_amd — the SKAT permutation test (--kernel skat[nPerm=..]): the exact replay of the reference's rand() stream and the
// counter-based shuffles (perm_kernels.hip.h, perm_counter.h).  Part of librvtests_amd.so.
// this unit compiles (and ships) the PERM kernel family only: see "kernel families" in rvt_engine_int.h
#define RVT_K_SPLIT
#define RVT_K_PERM
#include "rvt_engine_int.h"

extern "C" {

// ---- SKAT permutations (exact replay of the reference's rand() stream) ------------------------------------------
namespace {
void mat31_mul(const uint32_t* A, const uint32_t* B, uint32_t* C) {  // C = A B over Z/2^32
  uint32_t T[31 * 31];
  for (int i = 0; i < 31; ++i)
    for (int j = 0; j < 31; ++j) {
      uint32_t s = 0;
      for (int k = 0; k < 31; ++k) s += A[i * 31 + k] * B[k * 31 + j];
      T[i * 31 + j] = s;
    }
  std::memcpy(C, T, sizeof(T));
}
// J = A^e, A the one-draw transition x' = (x[1..30], x[0] + x[28])
void jump_matrix(uint64_t e, uint32_t* J) {
  uint32_t P[31 * 31] = {0}, R[31 * 31] = {0};
  for (int t = 0; t < 30; ++t) P[t * 31 + t + 1] = 1;
  P[30 * 31 + 0] = 1;
  P[30 * 31 + 28] = 1;
  for (int t = 0; t < 31; ++t) R[t * 31 + t] = 1;
  while (e) {
    if (e & 1) mat31_mul(R, P, R);
    mat31_mul(P, P, P);
    e >>= 1;
  }
  std::memcpy(J, R, sizeof(R));
}
void mat31_apply(const uint32_t* J, const uint32_t* x, uint32_t* y) {
  uint32_t t[31];
  for (int i = 0; i < 31; ++i) {
    uint32_t s = 0;
    for (int k = 0; k < 31; ++k) s += J[i * 31 + k] * x[k];
    t[i] = s;
  }
  std::memcpy(y, t, sizeof(t));
}

// The permutation test of one gene whose analytic SKAT result (obs = skat_Q) and weights are already on the device.
//   dG: the gene's block (unflipped), g0: its descriptor of the batch that just finished (weights in its scratch)
int perm_stage(rvt_ctx* c, const double* dG, int M, const GeneDesc& g0, const rvt_params& prm, rvt_gene_result* r) {
  const int64_t N = c->nc.N, ld = c->nc.ld;
  const int nPerm = prm.skat_nperm;
  hipStream_t st = c->stream;
  // flipped, polymorphic genotype block (K_sqrt = diag(w^1/2) G', Skat.cpp:42-47)
  std::vector<const double*> cols(M);
  for (int j = 0; j < M; ++j) cols[j] = dG + (size_t)j * ld;
  const double** d_cols = nullptr;
  int* d_flags = nullptr;
  HIP_TRY(c, hipMalloc((void**)&d_cols, sizeof(double*) * (size_t)M * 2));
  HIP_TRY(c, hipMalloc((void**)&d_flags, sizeof(int) * (size_t)M * 2));
  struct Guard {
    void *a, *b;
    ~Guard() {
      hipFree(a);
      hipFree(b);
    }
  } guard{(void*)d_cols, (void*)d_flags};
  HIP_TRY(c, hipMemcpyAsync(d_cols, cols.data(), sizeof(double*) * M, hipMemcpyHostToDevice, st));
  k_fam_colstat(dim3((unsigned)M), st, d_cols, (long long)N, d_flags);
  std::vector<int> flags(M);
  HIP_TRY(c, hipMemcpyAsync(flags.data(), d_flags, sizeof(int) * M, hipMemcpyDeviceToHost, st));
  HIP_TRY(c, sync_stream(st));
  std::vector<const double*> kc;
  std::vector<int> kf;
  for (int j = 0; j < M; ++j)
    if (flags[j] & 2) {
      kc.push_back(cols[j]);
      kf.push_back(flags[j] & 1);
    }
  const int m = (int)kc.size();
  if (m != r->n_poly) return fail(c, RVT_E_STATE, "permutation stage: %d polymorphic columns, batch reported %d", m, r->n_poly);
  int rc = ensure_fam_cols(c, (size_t)m, ld);
  if (rc) return rc;
  HIP_TRY(c, hipMemcpyAsync(d_cols + M, kc.data(), sizeof(double*) * m, hipMemcpyHostToDevice, st));
  HIP_TRY(c, hipMemcpyAsync(d_flags + M, kf.data(), sizeof(int) * m, hipMemcpyHostToDevice, st));
  k_fam_flip_compact(dim3(64, (unsigned)m), st, d_cols + M, d_flags + M, (long long)N, (long long)ld, c->d_Gp);
  const double* d_bw = gene_scratch_carve(g0.scratch, g0.Mp, g0.Cp).bw;  // sqrt of the SKAT weights, filtered order
  if (!c->perm_exact) {
    // ---- counter-based permutations (perm_counter.h): no stream shared between genes, nothing stored per shuffle ----------
    constexpr int kChunk = 2048;
    const int Mp = (m + 15) / 16 * 16;
    const long long ngroups = (N + 15) / 16;
    if (!c->d_pc_Q) HIP_TRY(c, hipMalloc((void**)&c->d_pc_Q, sizeof(double) * kChunk));
    const double obs = r->skat_Q;
    // Permutation::init — `threshold` is an INT member of the reference's class (src/Permutation.h:153): the product is truncated
    // (nPerm = 100, alpha = 0.001 -> 0: the test stops before its first shuffle and reports p = 1; found by running the
    // reference's compiled class beside this rule, tests/test_oracle_ref.py)
    const double threshold = (double)(int)(1.0 * nPerm * prm.skat_alpha * 2);
    int actual = 0, numX = 0, numEq = 0;
    std::vector<double> Q(kChunk);
    bool more = true;
    while (more) {
      if (actual >= nPerm || numX + numEq >= threshold) break;  // Permutation::next() before every shuffle
      // the first chunk is short: a gene far from significance stops after ~2 threshold shuffles
      const int want = actual == 0 ? std::min<int>(kChunk, (int)std::max(64.0, 2.5 * threshold)) : kChunk;
      const int nb = std::min(want, nPerm - actual);
      const int n_bt = (nb + 63) / 64;
      // ~4096 waves per launch; a slice holds at least 64 groups of 16 samples
      int slices = (int)std::max<long long>(1, std::min<long long>(4096 / n_bt, (ngroups + 63) / 64));
      const int gps = (int)((ngroups + slices - 1) / slices);
      slices = (int)((ngroups + gps - 1) / gps);
      const size_t need = (size_t)slices * nb * Mp;
      if (c->pc_part_cap < need) {
        if (c->d_pc_part) hipFree(c->d_pc_part);
        c->d_pc_part = nullptr;
        c->pc_part_cap = 0;
        HIP_TRY(c, hipMalloc((void**)&c->d_pc_part, sizeof(double) * (need + need / 4)));
        c->pc_part_cap = need + need / 4;
      }
      hipLaunchKernelGGL(perm_counter_partial_kernel, dim3((unsigned)slices, (unsigned)n_bt), dim3(64), 0, st, c->d_Gp,
                         (long long)ld, (long long)N, m, c->d_res, (unsigned long long)c->perm_seed,
                         (unsigned long long)r->gene_id, (unsigned)actual, nb, gps, Mp, c->d_pc_part);
      hipLaunchKernelGGL(perm_counter_q_kernel, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, st, c->d_pc_part, slices,
                         nb, Mp, m, d_bw, c->d_pc_Q);
      HIP_TRY(c, hipGetLastError());
      HIP_TRY(c, hipMemcpyAsync(Q.data(), c->d_pc_Q, sizeof(double) * (size_t)nb, hipMemcpyDeviceToHost, st));
      HIP_TRY(c, sync_stream(st));
      for (int u = 0; u < nb; ++u) {
        if (actual >= nPerm || numX + numEq >= threshold) {
          more = false;
          break;
        }
        ++actual;  // Permutation::add
        if (Q[u] > obs) ++numX;
        if (Q[u] == obs) ++numEq;
      }
    }
    r->perm_ok = 1;
    r->perm_num_perm = nPerm;
    r->perm_actual_perm = actual;
    r->perm_num_greater = numX;
    r->perm_num_equal = numEq;
    r->perm_pvalue = actual == 0 ? 1.0 : 1.0 * (numX + 0.5 * numEq) / actual;
    return RVT_OK;
  }
  // chunk buffers
  const int B = std::max(1, std::min(nPerm, (int)std::min<int64_t>(2048, ((int64_t)6 << 30) / (8 * N))));
  if ((size_t)N * B > c->perm_cap_NB || B > c->perm_cap_B || (size_t)B * m > c->perm_cap_BM || (size_t)N > c->perm_cap_N) {
    for (void** p : {(void**)&c->d_perm_idx, (void**)&c->d_perm_states, (void**)&c->d_perm_R, (void**)&c->d_perm_C,
                     (void**)&c->d_perm_Q, (void**)&c->d_perm_cur}) {
      if (*p) hipFree(*p);
      *p = nullptr;
    }
    c->perm_cap_NB = c->perm_cap_BM = 0;
    c->perm_cap_B = 0;
    c->perm_cap_N = 0;
    const size_t bm = (size_t)B * std::max(m, RVT_MAX_VARIANTS / 4);
    HIP_TRY(c, hipMalloc((void**)&c->d_perm_idx, sizeof(uint32_t) * (size_t)N * B));
    HIP_TRY(c, hipMalloc((void**)&c->d_perm_states, sizeof(uint32_t) * 31 * (size_t)B));
    HIP_TRY(c, hipMalloc((void**)&c->d_perm_R, sizeof(double) * (size_t)N * B));
    HIP_TRY(c, hipMalloc((void**)&c->d_perm_C, sizeof(double) * bm));
    HIP_TRY(c, hipMalloc((void**)&c->d_perm_Q, sizeof(double) * (size_t)B));
    HIP_TRY(c, hipMalloc((void**)&c->d_perm_cur, sizeof(double) * (size_t)N * 2));
    c->perm_cap_NB = (size_t)N * B;
    c->perm_cap_B = B;
    c->perm_cap_BM = bm;
    c->perm_cap_N = (size_t)N;  // (d_perm_cur holds 2 N doubles whatever B is: fewer shuffles of more samples must not keep it)
  }
  if (c->jump_N != N) {
    c->jump.resize(31 * 31);
    jump_matrix((uint64_t)(N - 1), c->jump.data());  // one shuffle draws N-1 numbers (LinearAlgebra.h:12-14)
    c->jump_N = N;
  }
  // permutedRes = res (src/Model.h:2708)
  double* cur = c->d_perm_cur;
  double* nxt = c->d_perm_cur + N;
  HIP_TRY(c, hipMemcpyAsync(cur, c->d_res, sizeof(double) * (size_t)N, hipMemcpyDeviceToDevice, st));
  const double obs = r->skat_Q;
  // Permutation::init — `threshold` is an INT member of the reference's class (src/Permutation.h:153): the product is truncated
    // (nPerm = 100, alpha = 0.001 -> 0: the test stops before its first shuffle and reports p = 1; found by running the
    // reference's compiled class beside this rule, tests/test_oracle_ref.py)
    const double threshold = (double)(int)(1.0 * nPerm * prm.skat_alpha * 2);
  int actual = 0, numX = 0, numEq = 0;
  uint32_t s0[31];
  std::memcpy(s0, c->rand_state, sizeof(s0));
  std::vector<uint32_t> states((size_t)31 * (B + 1));
  std::vector<double> Q(B);
  bool more = true;
  while (more) {
    // Permutation::next() before every shuffle
    if (actual >= nPerm || numX + numEq >= threshold) break;
    const int nb = std::min(B, nPerm - actual);
    std::memcpy(states.data(), s0, sizeof(s0));
    for (int p = 0; p < nb; ++p) mat31_apply(c->jump.data(), &states[(size_t)31 * p], &states[(size_t)31 * (p + 1)]);
    HIP_TRY(c, hipMemcpyAsync(c->d_perm_states, states.data(), sizeof(uint32_t) * 31 * (size_t)nb,
                              hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(perm_init_kernel, dim3(2048), dim3(256), 0, st, c->d_perm_idx, (long long)N, B);
    hipLaunchKernelGGL(perm_fisher_yates_kernel, dim3((unsigned)((nb + 63) / 64)), dim3(64), 0, st,
                       c->d_perm_states, c->d_perm_idx, (long long)N, B);
    for (int p = 0; p < nb; ++p) {  // the shuffles are cumulative: apply them in order
      hipLaunchKernelGGL(perm_apply_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, c->d_perm_idx, cur,
                         nxt, c->d_perm_R, (long long)N, B, p);
      std::swap(cur, nxt);
    }
    if (N <= 2048) {  // few samples: sums in sample order, so that exact ties with the observed Q resolve as in the reference
      hipLaunchKernelGGL(perm_dot_sequential_kernel, dim3((unsigned)(((long long)nb * m + 255) / 256)), dim3(256), 0, st,
                         c->d_perm_R, c->d_Gp, (long long)N, (long long)ld, nb, m, B, c->d_perm_C);
    } else {  // C (nb x m) = Rp' G with Rp = the chunk's permuted residuals as columns (N x nb): integer-plane product
      int rcg = gemm_tn_planes(c, c->d_perm_R, N, nb, c->d_Gp, ld, m, N, c->d_perm_C, B, st);
      if (rcg) return rcg;
    }
    hipLaunchKernelGGL(perm_q_kernel, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, st, c->d_perm_C, d_bw, B, m,
                       c->d_perm_Q);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(Q.data(), c->d_perm_Q, sizeof(double) * (size_t)nb, hipMemcpyDeviceToHost, st));
    HIP_TRY(c, sync_stream(st));
    int used = 0;
    for (; used < nb; ++used) {
      if (actual >= nPerm || numX + numEq >= threshold) {
        more = false;
        break;
      }
      ++actual;  // Permutation::add
      if (Q[used] > obs) ++numX;
      if (Q[used] == obs) ++numEq;
    }
    std::memcpy(s0, &states[(size_t)31 * used], sizeof(s0));  // the stream continues after the shuffles performed
    if (used < nb) {
      // the residual vector of the next gene restarts from res anyway; nothing else carries over
      more = false;
    }
  }
  std::memcpy(c->rand_state, s0, sizeof(s0));
  r->perm_ok = 1;
  r->perm_num_perm = nPerm;
  r->perm_actual_perm = actual;
  r->perm_num_greater = numX;
  r->perm_num_equal = numEq;
  r->perm_pvalue = actual == 0 ? 1.0 : 1.0 * (numX + 0.5 * numEq) / actual;
  return RVT_OK;
}

// KBAC (KBACTest::fit, src/Model.h:2925-2998 over regression/kbac.cpp) of one gene.  y: the 0 / 1 phenotype (host).
int kbac_stage(rvt_ctx* c, const double* dG, int M, const double* af, const std::vector<unsigned char>& y, int nPerm,
               double alpha, rvt_kbac_result* r) {
  std::memset(r, 0, sizeof(*r));
  r->pvalue = 9.0;
  const int64_t N = c->nc.N, ld = c->nc.ld;
  hipStream_t st = c->stream;
  // ---- flipped, polymorphic block (dc->getFlippedToMinorPolymorphicGenotype()) ---------------------------------------
  std::vector<const double*> cols(M);
  for (int j = 0; j < M; ++j) cols[j] = dG + (size_t)j * ld;
  const double** d_cols = nullptr;
  int* d_flags = nullptr;
  double* d_id = nullptr;
  int* d_carrier = nullptr;
  unsigned char *d_y = nullptr, *d_sub = nullptr;
  struct Guard {
    std::vector<void**> p;
    ~Guard() {
      for (void** q : p)
        if (*q) hipFree(*q);
    }
  } guard{{(void**)&d_cols, (void**)&d_flags, (void**)&d_id, (void**)&d_carrier, (void**)&d_y, (void**)&d_sub}};
  HIP_TRY(c, hipMalloc((void**)&d_cols, sizeof(double*) * (size_t)M * 2));
  HIP_TRY(c, hipMalloc((void**)&d_flags, sizeof(int) * (size_t)M * 3));
  HIP_TRY(c, hipMemcpyAsync(d_cols, cols.data(), sizeof(double*) * M, hipMemcpyHostToDevice, st));
  k_fam_colstat(dim3((unsigned)M), st, d_cols, (long long)N, d_flags);
  std::vector<int> flags(M);
  HIP_TRY(c, hipMemcpyAsync(flags.data(), d_flags, sizeof(int) * M, hipMemcpyDeviceToHost, st));
  HIP_TRY(c, sync_stream(st));
  std::vector<const double*> kc;
  std::vector<int> kf;
  for (int j = 0; j < M; ++j)
    if (flags[j] & 2) {
      kc.push_back(cols[j]);
      kf.push_back(flags[j] & 1);
    }
  const int m = (int)kc.size();
  r->n_poly = m;
  if (m == 0) {  // genotype.cols == 0: xdat is empty, KbacTest's constructor would reject it; rvtests prints what it got
    r->fit_ok = 0;
    return RVT_OK;
  }
  int rc = ensure_fam_cols(c, (size_t)m, ld);
  if (rc) return rc;
  HIP_TRY(c, hipMemcpyAsync(d_cols + M, kc.data(), sizeof(double*) * m, hipMemcpyHostToDevice, st));
  HIP_TRY(c, hipMemcpyAsync(d_flags + M, kf.data(), sizeof(int) * m, hipMemcpyHostToDevice, st));
  k_fam_flip_compact(dim3(64, (unsigned)m), st, d_cols + M, d_flags + M, (long long)N, (long long)ld, c->d_Gp);
  // ---- m_trimXdat: columns with 0 < maf <= 1 (maf of filtered position j = counter of unfiltered column j) --------------
  std::vector<int> use;
  for (int j = 0; j < m; ++j)
    if (!(af[j] <= 0.0 || af[j] > 1.0)) use.push_back(j);
  const int n_used = (int)use.size();
  std::vector<double> id((size_t)N, 0.0);
  if (n_used > 0) {
    std::vector<double> p3((size_t)n_used + 1);
    for (int k = 0; k <= n_used; ++k) p3[k] = std::pow(3.0, 1.0 * k);  // the host's pow, as the reference evaluates it
    double* d_p3 = nullptr;
    HIP_TRY(c, hipMalloc((void**)&d_id, sizeof(double) * ((size_t)N + p3.size())));
    d_p3 = d_id + N;
    HIP_TRY(c, hipMemcpyAsync(d_p3, p3.data(), sizeof(double) * p3.size(), hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipMemcpyAsync(d_flags + 2 * M, use.data(), sizeof(int) * n_used, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(kbac_pattern_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, c->d_Gp, (long long)N,
                       (long long)ld, d_flags + 2 * M, n_used, d_p3, d_id);
    HIP_TRY(c, hipMemcpyAsync(id.data(), d_id, sizeof(double) * (size_t)N, hipMemcpyDeviceToHost, st));
    HIP_TRY(c, sync_stream(st));
  }
  // ---- unique patterns (ascending), their counts, the carriers --------------------------------------------------------
  std::vector<double> pat;
  for (int64_t i = 0; i < N; ++i)
    if (id[i] != 0.0) pat.push_back(id[i]);
  if (pat.empty()) {  // "non-wildtype genotype data is empty ... Return p-value 1.0"
    r->fit_ok = 1;
    r->pvalue = 1.0;
    return RVT_OK;
  }
  std::sort(pat.begin(), pat.end());
  pat.erase(std::unique(pat.begin(), pat.end()), pat.end());
  const int P = (int)pat.size();
  std::vector<int> carrier, cpat;
  std::vector<unsigned> cnt(P, 0);
  for (int64_t i = 0; i < N; ++i)
    if (id[i] != 0.0) {
      const int u = (int)(std::lower_bound(pat.begin(), pat.end(), id[i]) - pat.begin());
      carrier.push_back((int)i);
      cpat.push_back(u);
      ++cnt[u];
    }
  const int nc_ = (int)carrier.size();
  r->n_pattern = P;
  r->n_carrier = nc_;
  unsigned nCases = 0;
  for (int64_t i = 0; i < N; ++i) nCases += y[i] == 1;
  const unsigned nCtrls = (unsigned)N - nCases;
  // kernel weights for every possible case count of every pattern
  static const Hypergeometric hyper;
  std::vector<std::vector<double>> W(P);
  for (int u = 0; u < P; ++u) {
    W[u].resize(cnt[u] + 1);
    for (unsigned k = 0; k <= cnt[u]; ++k) W[u][k] = hyper.cdf(k, cnt[u], (unsigned)N - cnt[u], nCases);
  }
  std::vector<unsigned> sub(P);
  auto statistic = [&](const unsigned char* yc) {  // yc: phenotype of the carriers, in carrier order
    std::fill(sub.begin(), sub.end(), 0u);
    for (int q = 0; q < nc_; ++q) sub[cpat[q]] += yc[q] == 1;
    double kbac = 0.0;
    for (int u = 0; u < P; ++u)
      kbac = kbac + ((1.0 * sub[u]) / (1.0 * nCases) - (1.0 * (cnt[u] - sub[u])) / (1.0 * nCtrls)) * W[u][sub[u]];
    return kbac;
  };
  std::vector<unsigned char> yc(nc_);
  for (int q = 0; q < nc_; ++q) yc[q] = y[carrier[q]];
  const double observed = statistic(yc.data());
  r->stat = observed;
  // ---- permutations: cumulative std::random_shuffle of the phenotype, chunks of B shuffles -------------------------------
  const unsigned adaptive = alpha >= 1.0 ? 0u : 5000u;
  const int total = nPerm + 1;  // the loop shuffles once more after the last statistic (kbac.cpp:185,323-324)
  const int B = std::max(1, std::min(total, (int)std::min<int64_t>(2048, ((int64_t)6 << 30) / (4 * N))));
  if ((size_t)N * B > c->perm_cap_NB || B > c->perm_cap_B) {
    for (void** p : {(void**)&c->d_perm_idx, (void**)&c->d_perm_states, (void**)&c->d_perm_R, (void**)&c->d_perm_C,
                     (void**)&c->d_perm_Q, (void**)&c->d_perm_cur}) {
      if (*p) hipFree(*p);
      *p = nullptr;
    }
    c->perm_cap_NB = c->perm_cap_BM = 0;
    c->perm_cap_B = 0;
    c->perm_cap_N = 0;
    HIP_TRY(c, hipMalloc((void**)&c->d_perm_idx, sizeof(uint32_t) * (size_t)N * B));
    HIP_TRY(c, hipMalloc((void**)&c->d_perm_states, sizeof(uint32_t) * 31 * (size_t)B));
    c->perm_cap_NB = (size_t)N * B;
    c->perm_cap_B = B;
  }
  if (c->jump_N != N) {
    c->jump.resize(31 * 31);
    jump_matrix((uint64_t)(N - 1), c->jump.data());  // one shuffle draws N-1 numbers
    c->jump_N = N;
  }
  HIP_TRY(c, hipMalloc((void**)&d_y, (size_t)N * 2));
  HIP_TRY(c, hipMalloc((void**)&d_sub, (size_t)B * nc_));
  HIP_TRY(c, hipMalloc((void**)&d_carrier, sizeof(int) * (size_t)nc_));
  HIP_TRY(c, hipMemcpyAsync(d_carrier, carrier.data(), sizeof(int) * (size_t)nc_, hipMemcpyHostToDevice, st));
  unsigned char *cur = d_y, *nxt = d_y + N;
  HIP_TRY(c, hipMemcpyAsync(cur, y.data(), (size_t)N, hipMemcpyHostToDevice, st));
  uint32_t s0[31];
  std::memcpy(s0, c->rand_state, sizeof(s0));
  std::vector<uint32_t> states((size_t)31 * (B + 1));
  std::vector<unsigned char> ysub((size_t)B * nc_);
  unsigned pc1 = 0, pc2 = 0;
  int done = 0;  // shuffles performed = statistics evaluated after the observed one
  bool stop = false;
  while (!stop && done < nPerm) {
    const int nb = std::min(B, nPerm - done);
    std::memcpy(states.data(), s0, sizeof(s0));
    for (int p = 0; p < nb; ++p) mat31_apply(c->jump.data(), &states[(size_t)31 * p], &states[(size_t)31 * (p + 1)]);
    HIP_TRY(c, hipMemcpyAsync(c->d_perm_states, states.data(), sizeof(uint32_t) * 31 * (size_t)nb, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(perm_init_kernel, dim3(2048), dim3(256), 0, st, c->d_perm_idx, (long long)N, B);
    hipLaunchKernelGGL(perm_random_shuffle_kernel, dim3((unsigned)((nb + 63) / 64)), dim3(64), 0, st, c->d_perm_states,
                       c->d_perm_idx, (long long)N, B);
    for (int p = 0; p < nb; ++p) {  // the shuffles are cumulative: apply them in order, keep the carriers' phenotype
      hipLaunchKernelGGL(perm_apply_u8_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, c->d_perm_idx, cur, nxt,
                         (long long)N, B, p);
      hipLaunchKernelGGL(perm_gather_u8_kernel, dim3((unsigned)((nc_ + 255) / 256)), dim3(256), 0, st, nxt, d_carrier, nc_,
                         d_sub + (size_t)p * nc_);
      std::swap(cur, nxt);
    }
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(ysub.data(), d_sub, (size_t)nb * nc_, hipMemcpyDeviceToHost, st));
    HIP_TRY(c, sync_stream(st));
    int used = 0;
    for (; used < nb; ++used) {
      const double s = statistic(&ysub[(size_t)used * nc_]);
      ++done;
      if (s >= observed) ++pc1;
      if (s <= observed) ++pc2;
      if (adaptive != 0 && (unsigned)done % adaptive == 0) {  // m_checkAdaptivePvalue, alternative = 0 (kbac.cpp:338-372)
        const double ap = (1.0 * pc1 + 1.0) / (1.0 * done + 1.0);
        const double sd = std::sqrt(ap * (1.0 - ap) / (1.0 * done));
        if (ap - 6.0 * sd > alpha) {
          r->pvalue = ap;
          stop = true;
          ++used;
          break;
        }
      }
    }
    std::memcpy(s0, &states[(size_t)31 * used], sizeof(s0));
  }
  if (!stop) {  // every statistic evaluated: the reference's loop shuffles once more before it ends
    uint32_t t[31];
    mat31_apply(c->jump.data(), s0, t);
    std::memcpy(s0, t, sizeof(s0));
    r->pvalue = (1.0 * pc1 + 1.0) / (1.0 * nPerm + 1.0);
  }
  std::memcpy(c->rand_state, s0, sizeof(s0));
  r->fit_ok = 1;
  r->actual_perm = done;
  r->num_ge = (int)pc1;
  r->num_le = (int)pc2;
  return RVT_OK;
}

}  // namespace

// (kbac_stage for rvt_meta.hip's rvt_kbac_blocks)
int rvt_kbac_stage(rvt_ctx* c, const double* dG, int M, const double* af, const std::vector<unsigned char>& y, int nPerm,
                   double alpha, rvt_kbac_result* r) {
  return kbac_stage(c, dG, M, af, y, nPerm, alpha, r);
}

// analytic tests + permutation test, one gene at a time (the random stream is consumed in gene order)
int run_blocks_with_perm(rvt_ctx* c, int n, const double* const* dG, const int* M, const double* af,
                         const int64_t* ids, uint32_t tests, const rvt_params* prm, rvt_gene_result* out) {
  size_t afo = 0;
  for (int g = 0; g < n; ++g) {
    DebugOut dbg;
    GeneDesc g0;
    dbg.desc0 = &g0;
    int64_t id = ids ? ids[g] : g;
    int rc = run_batch(c, 1, dG + g, M + g, af + afo, &id, tests, prm, out + g, &dbg);
    if (!rc) rc = rvt_sync(c);
    if (rc) return rc;
    afo += (size_t)M[g];
    if (out[g].skat_ok) {  // genotype.cols == 0 returns before the permutations (src/Model.h:2665-2668)
      rc = perm_stage(c, dG[g], M[g], g0, *prm, out + g);
      if (rc) return rc;
    }
  }
  return RVT_OK;
}

int rvt_rand_seed(rvt_ctx* c, unsigned seed) {
  if (!c) return RVT_E_INVALID;
  seed_rand_state(c->rand_state, seed);
  c->perm_seed = seed;
  return RVT_OK;
}

int rvt_set_perm_exact(rvt_ctx* c, int on) {
  if (!c) return RVT_E_INVALID;
  c->perm_exact = on != 0;
  return RVT_OK;
}

}  // extern "C"
