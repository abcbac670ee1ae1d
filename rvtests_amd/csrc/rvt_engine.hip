// rvtests_amd — the engine behind include/rvtests_amd.h: device memory, the per-batch kernel pipeline
// and the C ABI.  Built by hipcc for gfx950 into librvtests_amd.so.  There is no CPU fallback: without a
// HIP device rvt_init() fails with RVT_E_NO_DEVICE.
// this unit compiles (and ships) the ENGINE kernel family only: see "kernel families" in rvt_engine_int.h
#define RVT_K_SPLIT
#define RVT_K_ENGINE
#include "rvt_engine_int.h"
#include <chrono>
#include <functional>
#include <thread>
#include <sched.h>
#include <sys/syscall.h>
#include <unistd.h>
#include <cctype>

extern "C" {

const char* rvt_version(void) { return "rvtests_amd 0.1 (gfx950)"; }
int64_t rvt_padded_ld(int64_t N) { return (N + 15) / 16 * 16; }
const char* rvt_last_error(const rvt_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }
void* rvt_stream(rvt_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int rvt_init(rvt_ctx** out, int device_id) {
  if (!out) return RVT_E_INVALID;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return RVT_E_NO_DEVICE;
  if (device_id < 0 || device_id >= ndev) return RVT_E_INVALID;
  if (hipSetDevice(device_id) != hipSuccess) return RVT_E_NO_DEVICE;
  rvt_ctx* c = new rvt_ctx();
  c->device = device_id;
  std::memset(&c->timing, 0, sizeof(c->timing));
  // Streams.  Every stream is created through hipExtStreamCreateWithCUMask, which gives it a hardware queue of its
  // own: the runtime multiplexes ordinary streams onto at most GPU_MAX_HW_QUEUES (default 4) queues, and with
  // five streams in use two of them would share a queue and serialise (measured: -8% throughput).
  // By default the masks enable every CU.  RVT_STAGE2_CUS=k (a multiple of 32, 0 = off) instead partitions the
  // chip: k CUs for the latency-bound stage (assemble / eigen / p-value), the rest for the streaming stage
  // (sufficient statistics, burden collapse).  Mask bit b addresses XCD (b % 8), and inside it shader engine
  // ((b / 8) % 4), CU slot (b / 32) — measured with tools/cu_probe.hip; an XCD whose bits are all zero is NOT
  // restricted.  With the current kernels the partition does not pay (r1 measurements in DESIGN.md), so it is off.
  int stage2_cus = 0;
  if (const char* e = getenv("RVT_STAGE2_CUS")) stage2_cus = atoi(e);
  hipDeviceProp_t prop;
  const bool have_prop = hipGetDeviceProperties(&prop, device_id) == hipSuccess;
  const int ncu = have_prop ? prop.multiProcessorCount : 0;
  bool masked = false;
  if (ncu >= 64 && stage2_cus < ncu && !getenv("RVT_PLAIN_STREAMS")) {
    const int words = (ncu + 31) / 32;
    std::vector<uint32_t> m1(words, 0u), m2(words, 0u);
    stage2_cus = stage2_cus > 0 ? std::max(32, stage2_cus / 32 * 32) : 0;
    for (int b = 0; b < ncu; ++b) {
      if (stage2_cus == 0) {
        m1[b / 32] |= 1u << (b % 32);
        m2[b / 32] |= 1u << (b % 32);
      } else {
        (b < stage2_cus ? m2 : m1)[b / 32] |= 1u << (b % 32);
      }
    }
    // The streaming stage is the critical path of the pipeline: its stream is an ordinary stream at the HIGHEST queue
    // priority, so the dispatcher hands free SIMDs to its workgroups first (+3 %); the batch streams keep dedicated
    // default-priority queues through the CU-mask constructor.  RVT_K2_HIGH=0 restores a CU-mask stream for it too.
    const char* k2h = getenv("RVT_K2_HIGH");
    if ((!k2h || atoi(k2h) != 0) && stage2_cus == 0) {
      int lo = 0, hi = 0;
      hipDeviceGetStreamPriorityRange(&lo, &hi);
      masked = hipStreamCreateWithPriority(&c->k2_stream, hipStreamNonBlocking, hi) == hipSuccess &&
               hipStreamCreateWithPriority(&c->k2b_stream, hipStreamNonBlocking, hi) == hipSuccess;
    } else {
      masked = hipExtStreamCreateWithCUMask(&c->k2_stream, words, m1.data()) == hipSuccess;
    }
    // RVT_TAIL_CUS = k (experimental): the batch streams (flags, assembly, eigen stages, hand-backs) on the first k mask
    // bits only, the streaming stage everywhere
    if (const char* e = getenv("RVT_TAIL_CUS")) {
      const int tk = atoi(e) / 8 * 8;
      if (stage2_cus == 0 && tk >= 8 && tk < ncu) {
        std::fill(m2.begin(), m2.end(), 0u);
        for (int b = 0; b < tk; ++b) m2[b / 32] |= 1u << (b % 32);
      }
    }
    for (int i = 0; masked && i < kSlotsAll; ++i)
      masked = hipExtStreamCreateWithCUMask(&c->slots[i].stream, words, m2.data()) == hipSuccess;
    if (!masked) {
      (void)hipGetLastError();
      if (c->k2_stream) hipStreamDestroy(c->k2_stream);
      c->k2_stream = nullptr;
      if (c->k2b_stream) hipStreamDestroy(c->k2b_stream);
      c->k2b_stream = nullptr;
      for (int i = 0; i < kSlotsAll; ++i) {
        if (c->slots[i].stream) hipStreamDestroy(c->slots[i].stream);
        c->slots[i].stream = nullptr;
      }
    }
  }
  c->cu_partitioned = masked && stage2_cus > 0;
  // The SKAT-O p-values keep a 256-register wave per gene resident for milliseconds (QAGS over ~10^3 Davies evaluations:
  // latency-bound).  Spread over the whole chip those waves sit on every CU, and the workgroup-cooperative streaming
  // kernel (suffstat_hcx.hip.h: eight waves, all the registers of a CU's SIMDs) finds no free CU.  RVT_PV_CUS = k
  // (a multiple of 8; default 64) confines the p-value kernel to k CUs — 8 per XCD for 64: two waves per SIMD hold all
  // 512 genes of a batch — and leaves the others to the streaming stage; 0 = everywhere, as before round 4.
  {
    int pv = 64;
    if (const char* e = getenv("RVT_PV_CUS")) pv = atoi(e);
    pv = pv > 0 ? std::max(8, pv / 8 * 8) : 0;  // (whole CUs per XCD: the mask bits go round the 8 XCDs)
    if (const char* e = getenv("RVT_HCX_FUSED")) c->hcx_fused = atoi(e) != 0;
    if (const char* e = getenv("RVT_SUBMIT_GROUP")) c->submit_group = std::min(256, std::max(1, atoi(e)));
    if (const char* e = getenv("RVT_AS_THREADS")) c->as_threads = std::min(1024, std::max(64, atoi(e) / 64 * 64));
    if (masked && stage2_cus == 0 && pv > 0 && pv < ncu) {
      const int words = (ncu + 31) / 32;
      std::vector<uint32_t> mp(words, 0u);
      for (int b = 0; b < pv; ++b) mp[b / 32] |= 1u << (b % 32);
      if (hipExtStreamCreateWithCUMask(&c->pv_stream[0], words, mp.data()) == hipSuccess &&
          hipExtStreamCreateWithCUMask(&c->pv_stream[1], words, mp.data()) == hipSuccess) {
        c->pv_cus = pv;
        for (int i = 0; i < kSlotsAll; ++i) {
          hipEventCreateWithFlags(&c->ev_pv_in[i], hipEventDisableTiming);
          hipEventCreateWithFlags(&c->ev_pv_out[i], hipEventDisableTiming);
        }
      } else {
        (void)hipGetLastError();
        for (int k = 0; k < 2; ++k) {
          if (c->pv_stream[k]) hipStreamDestroy(c->pv_stream[k]);
          c->pv_stream[k] = nullptr;
        }
      }
    }
  }
  if (!masked) {
    for (int i = 0; i < kSlotsAll; ++i)
      if (hipStreamCreateWithFlags(&c->slots[i].stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        return RVT_E_HIP;
      }
    if (hipStreamCreateWithFlags(&c->k2_stream, hipStreamNonBlocking) != hipSuccess) {
      delete c;
      return RVT_E_HIP;
    }
  }
  if (!c->k2b_stream && hipStreamCreateWithFlags(&c->k2b_stream, hipStreamNonBlocking) != hipSuccess) {
    delete c;
    return RVT_E_HIP;
  }
  c->stream = c->slots[0].stream;
  if (hipStreamCreateWithFlags(&c->io_stream, hipStreamNonBlocking) != hipSuccess ||
      hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking) != hipSuccess) {
    delete c;
    return RVT_E_HIP;
  }
  c->h2d_stream = c->io_stream;
  for (int k = 0; k < rvt_ctx::kPack; ++k) {
    hipEventCreateWithFlags(&c->ev_pack_copied[k], hipEventDisableTiming);
    hipEventCreateWithFlags(&c->ev_pack_free[k], hipEventDisableTiming);
  }
  for (int i = 0; i < kSlotsAll; ++i) {
    hipEventCreateWithFlags(&c->ev_in[i], hipEventDisableTiming);
    hipEventCreateWithFlags(&c->ev_k2[i], hipEventDisableTiming);
    hipEventCreateWithFlags(&c->ev_k2b[i], hipEventDisableTiming);
  }
  if (hipMalloc((void**)&c->d_nc, sizeof(NullConsts)) != hipSuccess) {
    delete c;
    return RVT_E_HIP;
  }
  {  // let the eigen kernel keep matrices up to ~120 x 120 doubles in LDS
    const int want = 128 * 1024;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(gene_tridiag_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, want) == hipSuccess)
      c->eigen_lds_max = want;
    (void)hipGetLastError();
  }
  seed_rand_state(c->rand_state, 1u);
  if (const char* e = getenv("RVT_HARDCALL")) c->hc_enabled = atoi(e) != 0;
  if (const char* e = getenv("RVT_STAGE")) c->stage_on = atoi(e) != 0;
  if (const char* e = getenv("RVT_PERM_EXACT")) c->perm_exact = atoi(e) != 0;
  if (const char* e = getenv("RVT_TRACE_SUBMIT")) c->trace_submit = atoi(e) != 0;
  hipEventCreateWithFlags(&c->ev_io, hipEventDisableTiming);
  *out = c;
  return RVT_OK;
}

static void free_null(rvt_ctx* c) {
  for (double** p : {&c->d_nulltile, &c->d_res, &c->d_v}) {
    if (*p) hipFree(*p);
    *p = nullptr;
  }
  c->d_X = c->d_rr = c->d_zeros = nullptr;  // inside d_nulltile
  if (c->d_hcp_xq) hipFree(c->d_hcp_xq);
  if (c->d_hcp_scale) hipFree(c->d_hcp_scale);
  c->d_hcp_xq = nullptr;
  c->d_hcp_scale = nullptr;
  c->hcp_planes_state = 0;
  if (c->d_nulltile_w) hipFree(c->d_nulltile_w);
  if (c->d_vq) hipFree(c->d_vq);
  if (c->d_dq) hipFree(c->d_dq);
  if (c->d_xq) hipFree(c->d_xq);
  if (c->d_xscale) hipFree(c->d_xscale);
  if (c->d_fxq) hipFree(c->d_fxq);
  c->d_fxq = nullptr;
  c->fdx_ok = false;
  c->d_nulltile_w = nullptr;
  c->d_vq = nullptr;
  c->d_dq = c->d_xq = nullptr;
  c->d_xscale = nullptr;
  c->hcx_ok = false;
  c->have_null = false;
}

void rvt_destroy(rvt_ctx* c) {
  if (!c) return;
  if (c->trace_submit && c->tr_genes > 0)
    fprintf(stderr, "[rvt submit trace] %lld genes: per gene us: total %.1f  copy %.1f  (unused %.1f %.1f)  launch %.1f "
            "| collect total %.1f ms\n", c->tr_genes, 1e6 * c->tr_block / c->tr_genes, 1e6 * c->tr_copy / c->tr_genes,
            1e6 * c->tr_consol / c->tr_genes, 1e6 * c->tr_af / c->tr_genes, 1e6 * c->tr_launch / c->tr_genes,
            1e3 * c->tr_collect);
  hipSetDevice(c->device);
  for (auto& sl : c->slots) sync_stream(sl.stream);
  if (c->io_stream) sync_stream(c->io_stream);
  for (size_t i = 0; i < c->host_reg.size(); ++i)
    if (c->host_reg_owned[i]) (void)hipHostUnregister(const_cast<char*>(c->host_reg[i].first));
  c->host_reg.clear();
  if (c->ev_reg) hipEventDestroy(c->ev_reg);
  if (c->k2b_stream) {
    sync_stream(c->k2b_stream);
    hipStreamDestroy(c->k2b_stream);
  }
  for (int k = 0; k < 2; ++k)
    if (c->pv_stream[k]) {
      sync_stream(c->pv_stream[k]);
      hipStreamDestroy(c->pv_stream[k]);
    }
  for (int i = 0; i < kSlotsAll; ++i) {
    if (c->ev_pv_in[i]) hipEventDestroy(c->ev_pv_in[i]);
    if (c->ev_pv_out[i]) hipEventDestroy(c->ev_pv_out[i]);
  }
  if (c->k2_stream) {
    sync_stream(c->k2_stream);
    hipStreamDestroy(c->k2_stream);
  }
  if (c->io_stream) {
    sync_stream(c->io_stream);
    hipStreamDestroy(c->io_stream);
  }
  for (char* p : c->stage.chunk) hipHostFree(p);
  if (c->h_small) hipHostFree(c->h_small);
  for (hipEvent_t e : c->small_ev)
    if (e) hipEventDestroy(e);
  for (hipEvent_t e : c->stage_ev)
    if (e) hipEventDestroy(e);
  if (c->ev_io) hipEventDestroy(c->ev_io);
  for (int i = 0; i < kSlotsAll; ++i) {
    if (c->ev_in[i]) hipEventDestroy(c->ev_in[i]);
    if (c->ev_k2[i]) hipEventDestroy(c->ev_k2[i]);
    if (c->ev_k2b[i]) hipEventDestroy(c->ev_k2b[i]);
  }
  drain_events(c);
  for (auto e : c->event_pool) hipEventDestroy(e);
  free_null(c);
  for (auto& p : c->queue)
    if (p.dG) hipFree(p.dG);
  for (auto& bp : c->block_pool) hipFree(bp.second);
  for (auto& bp : c->pk_pool) hipFree(bp.second);
  c->pk_pool.clear();
  for (auto& sl : c->slots) {
    if (sl.arena.base) hipFree(sl.arena.base);
    if (sl.h_stage) hipHostFree(sl.h_stage);
    hipStreamDestroy(sl.stream);
  }
  if (c->d_nc) hipFree(c->d_nc);
  for (double* p : {c->d_S, c->d_u1, c->d_uxy, c->d_lmm_part, c->d_fX, c->d_frr, c->d_fv, c->d_fzeros,
                    c->d_fbeta, c->d_Gp, c->d_Gt, c->d_cX, c->d_cv, c->d_cr})
    if (p) hipFree(p);
  if (c->d_famcov_nc) hipFree(c->d_famcov_nc);
  for (void* p : {(void*)c->d_perm_idx, (void*)c->d_perm_states, (void*)c->d_perm_R, (void*)c->d_perm_C,
                  (void*)c->d_perm_Q, (void*)c->d_perm_cur, (void*)c->d_pc_part, (void*)c->d_pc_Q})
    if (p) hipFree(p);
  if (c->d_consol_af) hipFree(c->d_consol_af);
  if (c->d_consol_parts) hipFree(c->d_consol_parts);
  if (c->h_af_ring) hipHostFree(c->h_af_ring);
  if (c->d_consol_i8) hipFree(c->d_consol_i8);
  for (int k = 0; k < rvt_ctx::kPack; ++k) {
    if (c->d_pack[k]) hipFree(c->d_pack[k]);
    if (c->ev_pack_copied[k]) hipEventDestroy(c->ev_pack_copied[k]);
    if (c->ev_pack_free[k]) hipEventDestroy(c->ev_pack_free[k]);
  }
  if (c->copy_stream) {
    sync_stream(c->copy_stream);
    hipStreamDestroy(c->copy_stream);
  }
  for (auto& kv : c->col_kind) kv.second.release();
  if (c->d_cc_part) hipFree(c->d_cc_part);
  for (int k = 0; k < rvt_ctx::kTextBufs; ++k) {
    if (c->text_buf[k]) hipFree(c->text_buf[k]);
    if (c->ev_text_free[k]) hipEventDestroy(c->ev_text_free[k]);
    if (c->ev_text_copied[k]) hipEventDestroy(c->ev_text_copied[k]);
  }
  if (c->d_vcf_rec) hipFree(c->d_vcf_rec);
  if (c->d_vcf_seg) hipFree(c->d_vcf_seg);
  if (c->d_vcf_rows) hipFree(c->d_vcf_rows);
  if (c->h_io_err) hipHostFree(c->h_io_err);
  if (c->d_vcf_sex) hipFree(c->d_vcf_sex);
  if (c->d_fam_list) hipFree(c->d_fam_list);
  if (c->d_bgen_rec) hipFree(c->d_bgen_rec);
  if (c->d_bgen_seg) hipFree(c->d_bgen_seg);
  if (c->d_Uq) hipFree(c->d_Uq);
  if (c->d_uq_range) hipFree(c->d_uq_range);
  for (void* q : {(void*)c->d_csc_ptr, (void*)c->d_csc_rows, (void*)c->d_csc_vals})
    if (q) hipFree(q);
  if (c->d_rotB) hipFree(c->d_rotB);
  if (c->d_rotA) hipFree(c->d_rotA);
  if (c->d_rot_part) hipFree(c->d_rot_part);
  if (c->d_cov_work) hipFree(c->d_cov_work);
  if (c->d_colpack) hipFree(c->d_colpack);
  if (c->d_mu_nan) hipFree(c->d_mu_nan);
  if (c->d_bedbatch) hipFree(c->d_bedbatch);
  for (int i = 0; i < 2; ++i) {
    if (c->colq.h[i]) hipHostFree(c->colq.h[i]);
    if (c->colq.ev[i]) hipEventDestroy(c->colq.ev[i]);
  }
  for (int i = 0; i < 2; ++i) {
    if (c->ev_band_fin[i]) hipEventDestroy(c->ev_band_fin[i]);
    if (c->ev_band_copied[i]) hipEventDestroy(c->ev_band_copied[i]);
  }
  if (c->d_rot_scale) hipFree(c->d_rot_scale);
  if (c->d_rot_sexp) hipFree(c->d_rot_sexp);
  if (c->d_kind) hipFree(c->d_kind);
  if (c->d_fam_nc) hipFree(c->d_fam_nc);
  delete c;
}

// A fixed-point image of a null-model column sits a fixed number of bits below the column's LARGEST entry: are its typical
// entries — the median of the non-zero magnitudes; a root mean square follows a single outlier — within `ratio` of it?
static bool column_scale_ok(const double* col, int64_t N, double mx, double ratio) {
  std::vector<double> mag;
  mag.reserve((size_t)N);
  for (int64_t i = 0; i < N; ++i)
    if (col[i] != 0.0) mag.push_back(std::fabs(col[i]));
  if (mag.empty()) return true;
  std::nth_element(mag.begin(), mag.begin() + mag.size() / 2, mag.end());
  return mx <= ratio * mag[mag.size() / 2];
}

int rvt_set_null(rvt_ctx* c, int trait, int64_t N, int d, const double* X, const double* res, const double* v,
                 double sigma2) {
  if (!c || !X || !res || !v || N < 1 || d < 1 || d > RVT_MAX_COV) return fail(c, RVT_E_INVALID, "bad null model");
  hipSetDevice(c->device);
  for (auto& sl : c->slots) HIP_TRY(c, finish_slot(c, sl) ? hipErrorUnknown : hipSuccess);
  if (!c->queue.empty()) return fail(c, RVT_E_STATE, "collect the submitted genes before changing the null model");
  for (auto& bp : c->block_pool) hipFree(bp.second);  // pooled blocks were laid out for the previous N
  c->block_pool.clear();
  for (auto& bp : c->pk_pool) hipFree(bp.second);
  c->pk_pool.clear();
  free_null(c);
  c->have_null_beta = false;
  const int64_t ld = rvt_padded_ld(N);
  NullConsts& nc = c->nc;
  std::memset(&nc, 0, sizeof(nc));
  nc.N = N;
  nc.ld = ld;
  nc.d = d;
  nc.binary = (trait == RVT_TRAIT_BINARY) ? 1 : 0;
  nc.sigma2 = nc.binary ? 1.0 : sigma2;
  // d x d cross-products of the null model (once per analysis; X'VX for a binary trait, X'X otherwise)
  double rss = 0, rsum = 0;
  for (int64_t i = 0; i < N; ++i) {
    rss += res[i] * res[i];
    rsum += res[i];
  }
  nc.rss = rss;
  nc.rsum = rsum;
  for (int a = 0; a < d; ++a)
    for (int b = a; b < d; ++b) {
      double s = 0;
      const double *xa = X + (size_t)a * N, *xb = X + (size_t)b * N;
      if (nc.binary)
        for (int64_t i = 0; i < N; ++i) s += xa[i] * v[i] * xb[i];
      else
        for (int64_t i = 0; i < N; ++i) s += xa[i] * xb[i];
      nc.C[a * d + b] = nc.C[b * d + a] = s;
    }
  if (!invert_spd(nc.C, d, nc.Cinv)) return fail(c, RVT_E_INVALID, "X'VX is singular");
  // device copies, padded with zeros
  const size_t vb = sizeof(double) * (size_t)ld;
  // [X | rr | zeros] in one allocation: the hard-call kernel reads its null-model tile through one buffer descriptor
  HIP_TRY(c, hipMalloc((void**)&c->d_nulltile, vb * (d + 2)));
  c->d_X = c->d_nulltile;
  c->d_rr = c->d_nulltile + (size_t)ld * d;
  c->d_zeros = c->d_nulltile + (size_t)ld * (d + 1);
  HIP_TRY(c, hipMalloc((void**)&c->d_res, vb));
  HIP_TRY(c, hipMalloc((void**)&c->d_v, vb));
  HIP_TRY(c, hipMemset(c->d_nulltile, 0, vb * (d + 2)));
  HIP_TRY(c, hipMemset(c->d_res, 0, vb));
  HIP_TRY(c, hipMemset(c->d_v, 0, vb));
  HIP_TRY(c, hipMemcpy2D(c->d_X, vb, X, sizeof(double) * (size_t)N, sizeof(double) * (size_t)N, d,
                         hipMemcpyHostToDevice));
  HIP_TRY(c, hipMemcpy(c->d_res, res, sizeof(double) * (size_t)N, hipMemcpyHostToDevice));
  HIP_TRY(c, hipMemcpy(c->d_v, v, sizeof(double) * (size_t)N, hipMemcpyHostToDevice));
  {
    std::vector<double> rr((size_t)N);
    if (nc.binary)
      for (int64_t i = 0; i < N; ++i) rr[i] = res[i] / v[i];
    else
      for (int64_t i = 0; i < N; ++i) rr[i] = res[i];
    HIP_TRY(c, hipMemcpy(c->d_rr, rr.data(), sizeof(double) * (size_t)N, hipMemcpyHostToDevice));
  }
  if (nc.binary && c->hc_enabled && d <= kHcwMaxD) {
    // weighted hard-call kernel (suffstat_hcw.hip.h): v = p (1 - p) <= 1/4 rounded to 7 * kHcwPlanes fractional bits and split
    // into that many balanced base-128 digits (value = sum_p digit_p 128^-(p+1), digits in [-64, 63]: reaches 0.496), stored per
    // four samples as [plane 0..7][4 bytes]; a model with a weight outside [0, 0.49] stays on the fp64 kernel
    bool ok = true;
    for (int64_t i = 0; i < N && ok; ++i) ok = v[i] >= 0.0 && v[i] <= 0.49;
    if (ok) {
      std::vector<unsigned char> vq((size_t)ld * 8, 0);
      for (int64_t i = 0; i < N; ++i) {
        long long q = llrint(std::ldexp(v[i], 7 * kHcwPlanes));
        unsigned char* grp = vq.data() + (size_t)(i >> 2) * 32 + (size_t)(i & 3);
        for (int p = kHcwPlanes - 1; p >= 0; --p) {
          long long r = q & 127;
          if (r >= 64) r -= 128;
          q = (q - r) >> 7;
          grp[p * 4] = (unsigned char)(signed char)r;
        }
      }
      std::vector<double> tile((size_t)ld * (d + 3), 0.0);
      for (int k = 0; k < d; ++k)
        for (int64_t i = 0; i < N; ++i) tile[(size_t)k * ld + i] = v[i] * X[(size_t)k * N + i];
      for (int64_t i = 0; i < N; ++i) {
        tile[(size_t)d * ld + i] = res[i];
        tile[(size_t)(d + 1) * ld + i] = v[i];
      }
      HIP_TRY(c, hipMalloc((void**)&c->d_nulltile_w, sizeof(double) * tile.size()));
      HIP_TRY(c, hipMalloc((void**)&c->d_vq, vq.size()));
      HIP_TRY(c, hipMemcpy(c->d_nulltile_w, tile.data(), sizeof(double) * tile.size(), hipMemcpyHostToDevice));
      HIP_TRY(c, hipMemcpy(c->d_vq, vq.data(), vq.size(), hipMemcpyHostToDevice));
      // The workgroup-cooperative kernel (suffstat_hcx.hip.h) multiplies the null tile on the int8 matrix cores too: every
      // column [vX_k | res | v] as six balanced base-128 digits of its fixed-point value, 42 bits below a power of two above
      // twice the column's largest entry.  A column whose largest entry exceeds 256 x its root mean square would leave
      // its typical entries fewer than 34 bits: such a model stays on the one-wave kernel (fp64 products of the tile).
      const char* ex = getenv("RVT_HCX");
      bool okx = !(ex && atoi(ex) == 0) && d + 2 <= kHcxStageCols;  // (wider models: the LDS stage of the kernel holds 8 columns)
      double scale[kHcxNullCols];
      int shift[kHcxNullCols];
      for (int k = 0; k < kHcxNullCols; ++k) {
        scale[k] = 1.0;
        shift[k] = 0;
      }
      for (int k = 0; k < d + 2 && okx; ++k) {
        double mx = 0.0, ss = 0.0;
        for (int64_t i = 0; i < N; ++i) {
          const double x = tile[(size_t)k * ld + i];
          mx = std::max(mx, std::fabs(x));
          ss += x * x;
        }
        if (!std::isfinite(mx)) okx = false;
        if (mx > 0.0) {
          // (until round 6 the test was 256 x the root mean square, which one outlier drags along with itself; against the
          //  median a rare disease's residual column — cases near 1, controls at minus the prevalence — must still pass: 4 096,
          //  i.e. typical entries keep 30 of the 42 bits)
          if (!column_scale_ok(tile.data() + (size_t)k * ld, N, mx, 4096.0)) okx = false;
          int e;
          std::frexp(mx, &e);        // mx = f 2^e, 0.5 <= f < 1
          shift[k] = 42 - (e + 1);   // |x| 2^shift < 2^41 ...
          // ... but six balanced digits end at 63 (128^5 + .. + 1) = 0.496 2^42: keep the top digit at 62 or below
          if (std::ldexp(mx, shift[k]) > 0.98 * 0x1p41) shift[k] -= 1;
          scale[k] = std::ldexp(1.0, -shift[k]);
        }
      }
      if (okx) {
        const int64_t ngroups = (ld + 63) / 64 + 4;  // (four groups of padding: the kernel fetches an iteration as one range)
        const int ncx = d + 2;
        std::vector<unsigned char> dq((size_t)ngroups * kHcxSliceDg, 0), xq((size_t)ngroups * kHcwPlanes * 4 * ncx * 16, 0);
        auto digits6 = [](long long q, signed char* dg) {
          for (int p = kHcwPlanes - 1; p >= 0; --p) {
            long long r = q & 127;
            if (r >= 64) r -= 128;
            q = (q - r) >> 7;
            dg[p] = (signed char)r;
          }
        };
        for (int64_t i = 0; i < N; ++i) {
          const int64_t g = i >> 6, T = (i >> 4) & 3, q = (i >> 2) & 3, l = i & 3;
          signed char dg[kHcwPlanes];
          digits6(llrint(std::ldexp(v[i], 7 * kHcwPlanes)), dg);  // (the same integer the planes of vq encode)
          for (int j = 0; j < kHcwPairs; ++j) {
            unsigned char* e = dq.data() + (size_t)g * kHcxSliceDg + ((q * kHcwPairs + j) * 4 + T) * 16 + l;
            e[0] = (unsigned char)dg[2 * j];
            e[4] = (unsigned char)dg[2 * j + 1];
            e[8] = (unsigned char)(signed char)(2 * dg[2 * j]);
            e[12] = (unsigned char)(signed char)(2 * dg[2 * j + 1]);
          }
          for (int k = 0; k < d + 2; ++k) {
            digits6(llrint(std::ldexp(tile[(size_t)k * ld + i], shift[k])), dg);
            for (int p = 0; p < kHcwPlanes; ++p)
              xq[(((size_t)(g * kHcwPlanes + p) * 4 + q) * ncx + k) * 16 + T * 4 + l] = (unsigned char)dg[p];
          }
        }
        HIP_TRY(c, hipMalloc((void**)&c->d_dq, dq.size()));
        HIP_TRY(c, hipMalloc((void**)&c->d_xq, xq.size()));
        HIP_TRY(c, hipMalloc((void**)&c->d_xscale, sizeof(scale)));
        HIP_TRY(c, hipMemcpy(c->d_dq, dq.data(), dq.size(), hipMemcpyHostToDevice));
        HIP_TRY(c, hipMemcpy(c->d_xq, xq.data(), xq.size(), hipMemcpyHostToDevice));
        HIP_TRY(c, hipMemcpy(c->d_xscale, scale, sizeof(scale), hipMemcpyHostToDevice));
        c->hcx_tile.dq = c->d_dq;
        c->hcx_tile.xq = c->d_xq;
        for (int k = 0; k < kHcxNullCols; ++k) c->hcx_tile.scale[k] = scale[k];
        c->hcx_tile.ncols = d + 2;
        c->hcx_ok = true;
      }
    }
  }
  if (!nc.binary && c->hc_enabled && 2 * d + 3 <= kFdxStageCols) {
    // float-digit dosage kernel (suffstat_fdx.hip.h): the columns [X_0 .. X_{d-1} | res | 1 | lo ..] as balanced base-256 digits
    // of their fixed-point values Q = 256 hi + lo, 46 bits below a power of two above the column's largest entry (|error| <=
    // 2^-47 of it per entry, unbiased): hi in the five digit planes of column k, lo (one digit) in plane 0 of column d + 2 + k;
    // the column of ones is the integer 1 (its tile column is the exact column sum of K).  Digits of an integer q: the bytes of
    // q + 0x8080808080 with the top bits flipped.
    const char* ef = getenv("RVT_FDX");
    bool okf = !(ef && atoi(ef) == 0);
    const int ncf = 2 * d + 3;
    double scale[16];
    int shift[16];
    for (int k = 0; k < 16; ++k) {
      scale[k] = 1.0;
      shift[k] = 0;
    }
    for (int k = 0; k <= d && okf; ++k) {
      const double* col = (k < d) ? X + (size_t)k * N : res;
      double mx = 0.0;
      for (int64_t i = 0; i < N; ++i) mx = std::max(mx, std::fabs(col[i]));
      if (!std::isfinite(mx)) okf = false;
      if (mx > 0.0 && !column_scale_ok(col, N, mx, 4096.0)) okf = false;  // (typical entries keep 34 of the 46 bits)
      if (mx > 0.0) {
        int e;
        std::frexp(mx, &e);       // mx = f 2^e, 0.5 <= f < 1
        shift[k] = 38 - e;        // |x| 2^shift < 2^38: the top digit of hi stays below 64
        scale[k] = std::ldexp(1.0, -shift[k]);
        scale[d + 2 + k] = std::ldexp(1.0, -shift[k] - 8);
      }
    }
    if (okf) {
      const int64_t ngroups = (ld + 31) / 32 + 8;  // (padding: the kernel fetches the groups of an iteration, also past the end)
      std::vector<unsigned char> fx((size_t)ngroups * kFdxPlanes * 4 * ncf * 8, 0);
      auto put = [&](int64_t i, int k, long long qv) {
        const int64_t g = i >> 5, T = (i >> 4) & 1, q = (i >> 2) & 3, l = i & 3;
        const unsigned long long kb = (unsigned long long)(qv + 0x8080808080ll);
        for (int p = 0; p < kFdxPlanes; ++p)
          fx[(((size_t)(g * kFdxPlanes + p) * 4 + q) * ncf + k) * 8 + T * 4 + l] = (unsigned char)(((kb >> (8 * p)) & 0xffu) ^ 0x80u);
      };
      for (int64_t i = 0; i < N; ++i) {
        for (int k = 0; k <= d; ++k) {
          const long long Q = llrint(std::ldexp((k < d) ? X[(size_t)k * N + i] : res[i], shift[k] + 8));
          long long lo = Q & 255;
          if (lo >= 128) lo -= 256;
          put(i, k, (Q - lo) >> 8);
          put(i, d + 2 + k, lo);
        }
        put(i, d + 1, 1ll);
      }
      HIP_TRY(c, hipMalloc((void**)&c->d_fxq, fx.size()));
      HIP_TRY(c, hipMemcpy(c->d_fxq, fx.data(), fx.size(), hipMemcpyHostToDevice));
      c->fdx_tile.xq = c->d_fxq;
      for (int k = 0; k < 16; ++k) c->fdx_tile.scale[k] = scale[k];
      c->fdx_tile.ncols = ncf;
      c->fdx_ok = true;
    }
  }
  HIP_TRY(c, hipMemcpy(c->d_nc, &nc, sizeof(nc), hipMemcpyHostToDevice));
  c->null_ld = ld;
  c->have_null = true;
  ++c->null_gen;
  return RVT_OK;
}

int rvt_block_alloc(rvt_ctx* c, int M, double** out) {
  if (!c || !out || M < 1) return fail(c, RVT_E_INVALID, "bad block");
  if (!c->have_null && !c->have_fam) return fail(c, RVT_E_STATE, "set the null model first (defines N)");
  hipSetDevice(c->device);
  const size_t bytes = sizeof(double) * (size_t)(c->have_null ? c->null_ld : c->fam_nc.ld) * M;
  HIP_TRY(c, hipMalloc((void**)out, bytes));
  // cleared on the stream the streaming entry points write blocks on, and complete on return: a hipMemset on the null
  // stream is not ordered against that (non-blocking) stream and could land after a decoder had filled the block
  HIP_TRY(c, hipMemsetAsync(*out, 0, bytes, c->io_stream));
  HIP_TRY(c, sync_stream(c->io_stream));
  {  // (flags are allocated by the first column upload; a zeroed block holds hard calls only)
    rvt_ctx::ColKind& ck = c->col_kind[*out];
    ck.release();  // (an earlier block at the same address that was freed behind our back)
    ck.cols = M;
  }
  return RVT_OK;
}

// one streaming pass over a block: 1 when every entry is exactly 0.0, 1.0 or 2.0.  Enqueued on `st`; the flag lands
// in d_flag[0] (set to 1 first).
int enqueue_classify(rvt_ctx* c, const double* dG, int M, int64_t N, int64_t ld, hipStream_t st, int* d_flag) {
  static const int one = 1;
  HIP_TRY(c, hipMemcpyAsync(d_flag, &one, sizeof(int), hipMemcpyHostToDevice, st));
  const long long pairs = ((long long)N + 1) / 2 * M;
  long long blocks = (pairs + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  k2_launch_classify(dim3((unsigned)blocks), st, dG, (long long)N, (long long)ld, M, d_flag);
  HIP_TRY(c, hipGetLastError());
  return RVT_OK;
}

int rvt_block_classify(rvt_ctx* c, const double* dG, int M, int* is_hard_call) {
  if (!c || !dG || M < 1) return fail(c, RVT_E_INVALID, "bad block");
  if (!c->have_null && !c->have_fam) return fail(c, RVT_E_STATE, "set the null model first (defines N)");
  hipSetDevice(c->device);
  const int64_t N = c->have_null ? c->nc.N : c->fam_nc.N, ld = c->have_null ? c->null_ld : c->fam_nc.ld;
  if (!c->d_kind) HIP_TRY(c, hipMalloc((void**)&c->d_kind, sizeof(int)));
  int rc = enqueue_classify(c, dG, M, N, ld, c->io_stream, c->d_kind);
  if (rc) return rc;
  int flag = 0;
  HIP_TRY(c, hipMemcpyAsync(&flag, c->d_kind, sizeof(int), hipMemcpyDeviceToHost, c->io_stream));
  HIP_TRY(c, sync_stream(c->io_stream));
  if (is_hard_call) *is_hard_call = flag ? 1 : 0;  // a query: nothing is remembered about the block
  return RVT_OK;
}

int rvt_set_hardcall(rvt_ctx* c, int on) {
  if (!c) return RVT_E_INVALID;
  // (genes kept as packed rows are waiting for the integer kernels they were submitted for)
  if (!c->queue.empty()) return fail(c, RVT_E_STATE, "collect the submitted genes before switching the hard-call kernels");
  int rc = rvt_sync(c);
  if (rc) return rc;
  c->hc_enabled = on != 0;
  return RVT_OK;
}

int rvt_set_dosage_lattice(rvt_ctx* c, int denominator) {
  if (!c) return RVT_E_INVALID;
  if (denominator < 0 || denominator > kLatMaxDen)
    return fail(c, RVT_E_INVALID, "rvt_set_dosage_lattice: denominator %d outside [0, %d]", denominator, kLatMaxDen);
  c->lattice_den = denominator;
  return RVT_OK;
}

int rvt_set_dosage_float(rvt_ctx* c, int on) {
  if (!c) return RVT_E_INVALID;
  c->dosage_float = on != 0;
  return RVT_OK;
}

int rvt_hardcall_kernel(const rvt_ctx* c) {
  if (!c || !c->have_null || !c->hc_enabled) return 0;
  if (!c->nc.binary) return 1;
  if (c->hcx_ok) return 3;
  return (c->d_nulltile_w && c->d_vq) ? 2 : 0;
}

int rvt_set_content_hint(rvt_ctx* c, int hint) {
  if (!c || hint < -1 || hint > 1) return fail(c, RVT_E_INVALID, "hint must be -1, 0 or 1");
  c->content_hint = hint;
  return RVT_OK;
}

int rvt_block_free(rvt_ctx* c, double* dG) {
  if (!c) return RVT_E_INVALID;
  hipSetDevice(c->device);
  if (c->colq.n > 0) {
    if (c->colq.dG == dG) c->colq.n = 0;  // (queued columns of the block that goes away)
    else if (int rc = flush_col_queue(c)) return rc;
  }
  (void)sync_stream(c->io_stream);
  {
    auto it = c->col_kind.find(dG);
    if (it != c->col_kind.end()) {
      it->second.release();
      c->col_kind.erase(it);
    }
  }
  if (dG) HIP_TRY(c, hipFree(dG));
  return RVT_OK;
}

// ---- host -> device through the pinned staging ring (host_stage.h), on io_stream --------------------------------------
// The calls return when the caller's memory has been READ; the data arrives in stream order behind them.
int stage_ready(rvt_ctx* c) {
  if (!c->stage.chunk.empty()) return RVT_OK;
  for (int k = 0; k < rvt_ctx::kStageChunks; ++k) {
    char* p = nullptr;
    HIP_TRY(c, hipHostMalloc((void**)&p, rvt_ctx::kStageBytes, hipHostMallocDefault));
    c->stage.chunk.push_back(p);
    HIP_TRY(c, hipEventCreateWithFlags(&c->stage_ev[k], hipEventDisableTiming));
  }
  c->stage.chunk_bytes = rvt_ctx::kStageBytes;
  c->stage.wait = [c](int k) {
    hipError_t e;
    int spins = 0;
    while ((e = hipEventQuery(c->stage_ev[k])) == hipErrorNotReady)
      if (++spins > 64) {
        struct timespec ts = {0, 20000};
        nanosleep(&ts, nullptr);
      }
    (void)hipGetLastError();
    return e == hipSuccess ? 0 : 1;
  };
  c->stage.send = [c](int k, size_t off, void* dst, size_t bytes) {
    return hipMemcpyAsync(dst, c->stage.chunk[k] + off, bytes, hipMemcpyHostToDevice, c->h2d_stream) == hipSuccess ? 0 : 1;
  };
  c->stage.send2d = [c](int k, void* dst, size_t dpitch, size_t width, size_t rows) {
    return hipMemcpy2DAsync(dst, dpitch, c->stage.chunk[k], width, width, rows, hipMemcpyHostToDevice, c->h2d_stream) ==
                   hipSuccess
               ? 0
               : 1;
  };
  c->stage.sent = [c](int k) { return hipEventRecord(c->stage_ev[k], c->h2d_stream) == hipSuccess ? 0 : 1; };
  return RVT_OK;
}
// a small table (<= kSmallBytes) through a pinned ring: no host synchronisation, the source may be a local
bool host_registered(const rvt_ctx* c, const void* src, size_t bytes) {
  const char* p = (const char*)src;
  for (const auto& r : c->host_reg)
    if (p >= r.first && p + bytes <= r.first + r.second) return true;
  return false;
}
// a copy out of registered caller memory has been enqueued on the io stream: remember to wait for it before the entry
// point returns (the caller may overwrite its buffer then)
static int reg_mark(rvt_ctx* c) {
  if (!c->ev_reg) HIP_TRY(c, hipEventCreateWithFlags(&c->ev_reg, hipEventDisableTiming));
  // one event stands for every copy marked so far only while they all went to ONE stream (a deferred wait, rvt_submit_genes,
  // marks once per gene): a copy on another stream first waits out what is pending
  if (c->reg_pending && c->reg_stream != c->h2d_stream) {
    const int rc = reg_wait(c);
    if (rc) return rc;
  }
  HIP_TRY(c, hipEventRecord(c->ev_reg, c->h2d_stream));
  c->reg_stream = c->h2d_stream;
  c->reg_pending = true;
  return RVT_OK;
}
int reg_wait(rvt_ctx* c) {
  if (!c->reg_pending) return RVT_OK;
  c->reg_pending = false;
  for (int spins = 0;; ++spins) {  // (copies of 0.1 - 4 ms: poll closely first)
    const hipError_t e = hipEventQuery(c->ev_reg);
    if (e == hipSuccess) return RVT_OK;
    if (e != hipErrorNotReady) return fail(c, RVT_E_HIP, "copy from registered host memory failed: %s", hipGetErrorString(e));
    (void)hipGetLastError();
    if (spins > 200) {
      struct timespec ts = {0, 20000};
      nanosleep(&ts, nullptr);
    }
  }
}

int rvt_host_register(rvt_ctx* c, const void* ptr, size_t bytes) {
  if (!c || !ptr || bytes == 0) return fail(c, RVT_E_INVALID, "rvt_host_register: bad range");
  hipSetDevice(c->device);
  for (const auto& r : c->host_reg)
    if ((const char*)ptr < r.first + r.second && r.first < (const char*)ptr + bytes)
      return fail(c, RVT_E_STATE, "rvt_host_register: the range overlaps a registered one");
  const hipError_t e = hipHostRegister(const_cast<void*>(ptr), bytes, hipHostRegisterPortable);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return fail(c, RVT_E_HIP, "hipHostRegister(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
  }
  c->host_reg.emplace_back((const char*)ptr, bytes);
  c->host_reg_owned.push_back(1);
  return RVT_OK;
}

// group members other than the one that registered: the range is page-locked for every device already (portable)
int rvt_host_adopt(rvt_ctx* c, const void* ptr, size_t bytes) {
  if (!c || !ptr || bytes == 0) return RVT_E_INVALID;
  c->host_reg.emplace_back((const char*)ptr, bytes);
  c->host_reg_owned.push_back(0);
  return RVT_OK;
}

int rvt_host_unregister(rvt_ctx* c, const void* ptr) {
  if (!c || !ptr) return RVT_E_INVALID;
  hipSetDevice(c->device);
  for (size_t i = 0; i < c->host_reg.size(); ++i)
    if (c->host_reg[i].first == (const char*)ptr) {
      int rc = reg_wait(c);
      if (!rc) rc = sync_stream(c->io_stream) == hipSuccess ? RVT_OK : RVT_E_HIP;  // nothing in flight reads the range any more
      const bool owned = c->host_reg_owned[i] != 0;
      c->host_reg.erase(c->host_reg.begin() + (long)i);
      c->host_reg_owned.erase(c->host_reg_owned.begin() + (long)i);
      if (owned && hipHostUnregister(const_cast<void*>(ptr)) != hipSuccess) {
        (void)hipGetLastError();
        return fail(c, RVT_E_HIP, "hipHostUnregister failed");
      }
      return rc;
    }
  return fail(c, RVT_E_INVALID, "rvt_host_unregister: not a registered range");
}

// NUMA node of the page at p (move_pages with no target nodes only reports), -1 when the kernel does not say
static int numa_node_of(void* p) {
  void* page = (void*)((uintptr_t)p & ~(uintptr_t)4095);
  int status = -1;
  if (syscall(SYS_move_pages, 0, 1UL, &page, nullptr, &status, 0) != 0) return -1;
  return status < 0 ? -1 : status;
}

int rvt_host_diagnose(rvt_ctx* c, rvt_host_diag* o) {
  if (!c || !o) return fail(c, RVT_E_INVALID, "bad arguments");
  std::memset(o, 0, sizeof(*o));
  hipSetDevice(c->device);
  o->hardware_threads = (int)std::thread::hardware_concurrency();
  {
    cpu_set_t set;
    CPU_ZERO(&set);
    o->affinity_cpus = sched_getaffinity(0, sizeof(set), &set) == 0 ? CPU_COUNT(&set) : -1;
  }
  o->copy_threads = CopyPool::instance().threads();
  o->pack_threads = CopyPool::pack_instance().threads();
  o->thp = -1;
  if (FILE* f = fopen("/sys/kernel/mm/transparent_hugepage/enabled", "r")) {
    char buf[128] = {0};
    if (fgets(buf, sizeof(buf), f)) {
      if (strstr(buf, "[never]")) o->thp = 0;
      else if (strstr(buf, "[madvise]")) o->thp = 1;
      else if (strstr(buf, "[always]")) o->thp = 2;
    }
    fclose(f);
  }
  o->gpu_numa_node = -1;
  {
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, sizeof(bus), c->device) == hipSuccess) {
      for (char* q = bus; *q; ++q) *q = (char)tolower(*q);
      const std::string path = std::string("/sys/bus/pci/devices/") + bus + "/numa_node";
      if (FILE* f = fopen(path.c_str(), "r")) {
        if (fscanf(f, "%d", &o->gpu_numa_node) != 1) o->gpu_numa_node = -1;
        fclose(f);
      }
    } else {
      (void)hipGetLastError();
    }
  }
  {
    double la[1] = {0};
    o->loadavg1 = getloadavg(la, 1) == 1 ? la[0] : -1.0;
  }
  const size_t B = (size_t)64 << 20;
  std::vector<char> src(B), dst(B);
  for (size_t i = 0; i < B; i += 4096) {  // first touch by the caller, as an adapter's genotype buffer
    src[i] = (char)i;
    dst[i] = 1;
  }
  o->buffer_numa_node = numa_node_of(src.data());
  char* pin = nullptr;
  void* dev = nullptr;
  auto best = [&](const std::function<void()>& fn) {
    double t = 1e30;
    for (int r = 0; r < 3; ++r) {
      const auto a = std::chrono::steady_clock::now();
      fn();
      t = std::min(t, std::chrono::duration<double>(std::chrono::steady_clock::now() - a).count());
    }
    return (double)B / t / 1e9;
  };
  o->memcpy_one_thread = best([&] { std::memcpy(dst.data(), src.data(), B); });
  if (hipHostMalloc((void**)&pin, B, hipHostMallocDefault) == hipSuccess && hipMalloc(&dev, B) == hipSuccess) {
    std::memset(pin, 0, B);
    o->pinned_numa_node = numa_node_of(pin);
    o->stage_pool = best([&] { CopyPool::instance().copy(pin, src.data(), B); });
    o->h2d_pinned = best([&] {
      (void)hipMemcpyAsync(dev, pin, B, hipMemcpyHostToDevice, c->io_stream);
      (void)sync_stream(c->io_stream);
    });
    o->d2h_pinned = best([&] {
      (void)hipMemcpyAsync(pin, dev, B, hipMemcpyDeviceToHost, c->io_stream);
      (void)sync_stream(c->io_stream);
    });
  } else {
    (void)hipGetLastError();
    o->pinned_numa_node = -1;
  }
  if (pin) hipHostFree(pin);
  if (dev) hipFree(dev);
  return RVT_OK;
}

// Bind the CALLING thread (and with it every thread it creates later: the engine's staging pools inherit the mask) to the CPUs of
// the NUMA node the device hangs on.  The hand-offs that pack on the host read the caller's buffers with a few threads and write a
// pinned ring that the runtime allocates on the device's node: with caller, buffers and threads on that node a site's 4 MB
// column packs in 40 us instead of 55-65 (MetaCov adapter 18 -> 24 k sites/s on a two-socket EPYC 9575F), a gene's fp64 block
// 15 % faster.  Returns the node (>= 0), -1 when it is unknown or the mask cannot be set (nothing changed).  No context needed;
// call it first in main() — or run the program under `numactl --cpunodebind=<node> --membind=<node>`.
int rvt_pin_to_device_node(int device_id) {
  char bus[64] = {0};
  if (hipDeviceGetPCIBusId(bus, sizeof(bus), device_id) != hipSuccess) {
    (void)hipGetLastError();
    return -1;
  }
  for (char* p = bus; *p; ++p) *p = (char)std::tolower((unsigned char)*p);
  int node = -1;
  if (FILE* f = fopen((std::string("/sys/bus/pci/devices/") + bus + "/numa_node").c_str(), "r")) {
    if (fscanf(f, "%d", &node) != 1) node = -1;
    fclose(f);
  }
  if (node < 0) return -1;
  char list[4096] = {0};
  if (FILE* f = fopen(("/sys/devices/system/node/node" + std::to_string(node) + "/cpulist").c_str(), "r")) {
    if (!fgets(list, sizeof(list), f)) list[0] = 0;
    fclose(f);
  }
  cpu_set_t set;
  CPU_ZERO(&set);
  int n = 0;
  for (char* p = list; *p && *p != '\n';) {  // "0-63,128-191"
    char* e = nullptr;
    const long a = strtol(p, &e, 10);
    if (e == p) break;
    long b = a;
    p = e;
    if (*p == '-') {
      b = strtol(p + 1, &e, 10);
      p = e;
    }
    for (long k = a; k <= b && k < CPU_SETSIZE; ++k) {
      CPU_SET((int)k, &set);
      ++n;
    }
    if (*p == ',') ++p;
  }
  if (n == 0 || sched_setaffinity(0, sizeof(set), &set) != 0) return -1;
  return node;
}

int small_h2d(rvt_ctx* c, void* dst, const void* src, size_t bytes) {
  if (bytes > rvt_ctx::kSmallBytes) {
    HIP_TRY(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->io_stream));
    HIP_TRY(c, sync_stream(c->io_stream));
    return RVT_OK;
  }
  if (!c->h_small) {
    HIP_TRY(c, hipHostMalloc((void**)&c->h_small, rvt_ctx::kSmallSlots * rvt_ctx::kSmallBytes, hipHostMallocDefault));
    for (auto& e : c->small_ev) HIP_TRY(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
  }
  const int k = c->small_next;
  c->small_next = (k + 1) % rvt_ctx::kSmallSlots;
  while (hipEventQuery(c->small_ev[k]) == hipErrorNotReady) {
    struct timespec ts = {0, 20000};
    nanosleep(&ts, nullptr);
  }
  (void)hipGetLastError();
  char* p = c->h_small + (size_t)k * rvt_ctx::kSmallBytes;
  std::memcpy(p, src, bytes);
  HIP_TRY(c, hipMemcpyAsync(dst, p, bytes, hipMemcpyHostToDevice, c->io_stream));
  HIP_TRY(c, hipEventRecord(c->small_ev[k], c->io_stream));
  return RVT_OK;
}
int staged_h2d(rvt_ctx* c, void* dst, const void* src, size_t bytes) {
  TraceScope ts(c, &c->tr_copy);
  if (host_registered(c, src, bytes)) {  // DMA straight out of the caller's page-locked buffer
    HIP_TRY(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->h2d_stream));
    return reg_mark(c);
  }
  if (!c->stage_on || bytes < ((size_t)256 << 10)) {
    HIP_TRY(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->h2d_stream));
    return RVT_OK;
  }
  int rc = stage_ready(c);
  if (rc) return rc;
  if (c->stage.copy(dst, src, bytes, CopyPool::instance())) return fail(c, RVT_E_HIP, "staged host-to-device copy failed");
  return RVT_OK;
}
int staged_h2d_2d(rvt_ctx* c, void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t rows,
                  bool pad_zero) {
  TraceScope ts(c, &c->tr_copy);
  // Big blocks (the 200 MB of an fp64 gene at N = 500 000) go through the runtime's own pageable path: measured 46-48
  // GB/s of the link's 57 (tools/bench_group_stream.py), which the staged ring does not beat at this size; the call is
  // then synchronous.  The ring is for the packed hand-offs, where returning before the data has crossed matters.
  if (host_registered(c, src, spitch * (rows - 1) + width)) {
    HIP_TRY(c, hipMemcpy2DAsync(dst, dpitch, src, spitch, width, rows, hipMemcpyHostToDevice, c->h2d_stream));
    return reg_mark(c);
  }
  static const bool stage_big = getenv("RVT_STAGE_BIG") && atoi(getenv("RVT_STAGE_BIG")) != 0;
  if (!c->stage_on || width * rows < ((size_t)256 << 10) || (!stage_big && width * rows >= ((size_t)64 << 20))) {
    HIP_TRY(c, hipMemcpy2DAsync(dst, dpitch, src, spitch, width, rows, hipMemcpyHostToDevice, c->h2d_stream));
    HIP_TRY(c, sync_stream(c->h2d_stream));  // (a small pageable copy: the runtime has not necessarily read it yet)
    return RVT_OK;
  }
  int rc = stage_ready(c);
  if (rc) return rc;
  if (c->stage.copy2d(dst, dpitch, src, spitch, width, rows, CopyPool::instance(), pad_zero))
    return fail(c, RVT_E_HIP, "staged host-to-device copy failed");
  return RVT_OK;
}

// (enqueued on io_stream: the block is complete in stream order behind the call, see staged_h2d_2d)
int upload_block_data(rvt_ctx* c, double* dG, int M, const double* G) {
  if (!c || !dG || !G || M < 1) return fail(c, RVT_E_INVALID, "bad upload");
  if (!c->have_null && !c->have_fam) return fail(c, RVT_E_STATE, "set the null model first");
  hipSetDevice(c->device);
  if (c->colq.n > 0) {
    int rcq = flush_col_queue(c);
    if (rcq) return rcq;
  }
  {  // per-column flags of an earlier column-wise fill no longer describe the block
    auto it = c->col_kind.find(dG);
    if (it != c->col_kind.end()) {
      const int cols = it->second.cols;
      it->second.release();
      it->second.cols = cols;
    }
  }
  const size_t N = (size_t)(c->have_null ? c->nc.N : c->fam_nc.N);
  const size_t bld = (size_t)(c->have_null ? c->null_ld : c->fam_nc.ld);
  return staged_h2d_2d(c, dG, sizeof(double) * bld, G, sizeof(double) * N, sizeof(double) * N, (size_t)M);
}

int rvt_block_upload(rvt_ctx* c, double* dG, int M, const double* G) {
  int rc = upload_block_data(c, dG, M, G);
  if (rc) return rc;
  HIP_TRY(c, sync_stream(c->io_stream));  // the block is complete on return
  c->reg_pending = false;
  return RVT_OK;
}

int rvt_set_profiling(rvt_ctx* c, int on) {
  if (!c) return RVT_E_INVALID;
  c->profiling = on != 0;
  return RVT_OK;
}

int rvt_get_timing(rvt_ctx* c, rvt_timing* t, int reset) {
  if (!c || !t) return RVT_E_INVALID;
  hipSetDevice(c->device);
  for (auto& sl : c->slots) HIP_TRY(c, sync_stream(sl.stream));
  HIP_TRY(c, sync_stream(c->k2_stream));
  HIP_TRY(c, sync_stream(c->k2b_stream));
  drain_events(c);
  *t = c->timing;
  if (reset) std::memset(&c->timing, 0, sizeof(c->timing));
  return RVT_OK;
}



// covZZ / covZZInv and the other constants of the MetaCov algebra for the installed null model (or, fam = true,
// for the family set prepared by rvt_fit_fam_null).  zz receives covZZ (cc.d x cc.d).
int cov_constants(rvt_ctx* c, bool fam, CovConsts* ccp, std::vector<double>* zzp) {
  const NullConsts& nc = c->nc;
  const int d = nc.d;
  const int64_t N = nc.N;
  CovConsts& cc = *ccp;
  std::vector<double>& zz = *zzp;
  std::memset(&cc, 0, sizeof(cc));
    cc.d = d;
    cc.binary = nc.binary;
    cc.inv_n = 1.0 / (double)N;
  zz.assign((size_t)d * d, 0.0);
    if (fam) {  // MetaCovFamQtl: constants prepared by rvt_fit_fam_null
      const int du = d - 2;  // U'X columns (the null set carries u1 and the allele-frequency column as well)
      cc.fam = 1;
      cc.d = du;
      cc.inv_sigma2 = 1.0;
      cc.c11 = c->famcov_c11;
      cc.k1r = c->famcov_k1r;
      cc.af_denom = c->fam_nc.rss;
      zz.assign((size_t)du * du, 0.0);
      for (int a = 0; a < du; ++a) {
        cc.zsum[a] = c->famcov_c1x[a];
        for (int b = 0; b < du; ++b) {
          zz[a * du + b] = c->famcov_zz[a * du + b];
          cc.zzinv[a * du + b] = c->famcov_zzinv[a * du + b];
        }
      }
    } else if (nc.binary) {  // covZZ = Z'WZ, covZZInv its inverse (MetaCovUnrelatedBinary::calculateZZ, Model.cpp:748-765)
      cc.inv_sigma2 = 1.0;
      for (int a = 0; a < d * d; ++a) {
        zz[a] = nc.C[a];
        cc.zzinv[a] = nc.Cinv[a];
      }
    } else {
      // covZZ = Zc'Zc / sigma2 with centred columns (MetaCovUnrelatedQtl::calculateZZ, Model.cpp:582-591); the
      // intercept's row/column is exactly zero and CholeskyInverseMatrix (LDLT solve) leaves it zero, so covZZInv is
      // the inverse of the covariate block — a d x d job on null-model constants, done once per call on the host.
      cc.inv_sigma2 = 1.0 / nc.sigma2;
      for (int k = 0; k < d; ++k) cc.zsum[k] = nc.C[k];  // first row of X'X = 1'Z (column 0 is the intercept)
      for (int a = 0; a < d; ++a)
        for (int b = 0; b < d; ++b) zz[a * d + b] = (nc.C[a * d + b] - cc.zsum[a] * cc.zsum[b] / (double)N) / nc.sigma2;
      for (int a = 0; a < d; ++a) zz[a * d + 0] = zz[0 * d + a] = 0.0;
      const int q = d - 1;
      if (q > 0) {
        std::vector<double> A((size_t)q * q), Ai((size_t)q * q);
        for (int a = 0; a < q; ++a)
          for (int b = 0; b < q; ++b) A[a * q + b] = zz[(a + 1) * d + (b + 1)];
        if (!invert_spd(A.data(), q, Ai.data())) return fail(c, RVT_E_INVALID, "covariate covariance is singular");
        for (int a = 0; a < q; ++a)
          for (int b = 0; b < q; ++b) cc.zzinv[(a + 1) * d + (b + 1)] = Ai[a * q + b];
      }
    }
  return RVT_OK;
}

// Arena layout of one gene of a batch (shared by run_batch and rvt_reserve).
struct GeneOff {
  size_t parts, colstat, masks, flags, bparts, scratch, lambda, qags, stats, vt, dbg_flip, dbg_kept, pq, wflags;
};
// hc: 1 = hard-call path (no mask planes, burden records per wave-part), 0 = general path, -1 = either (rvt_reserve)
static void layout_gene(int M, int d, int n_wparts, int64_t nsteps, int n_bparts, bool dbg, int hc, size_t* total,
                        GeneOff* o, bool vt = false) {
  auto add = [&](size_t bytes) {
    *total = (*total + 255) / 256 * 256;
    const size_t at = *total;
    *total += bytes;
    return at;
  };
  const int MT = (M + 15) / 16, CT = (M + d + 1 + 15) / 16, Mp = 16 * MT, Cp = 16 * CT;
  o->parts = add(sizeof(double) * (size_t)n_wparts * Mp * Cp);
  o->colstat = add(sizeof(double) * (size_t)n_wparts * (hc == 0 ? 3 : kHcColstatRows) * Mp);
  o->masks = (hc == 1) ? 0 : add(sizeof(unsigned long long) * (size_t)2 * nsteps * MT * 4);
  o->pq = o->wflags = 0;
  if (hc != 0 && MT <= kHcMaxMT) {  // packed counters of the masked tiles + flags, per wave-part (suffstat_hc.hip.h)
    o->pq = add(sizeof(unsigned) * (size_t)n_wparts * hc_pq_words(MT));
    o->wflags = add(sizeof(unsigned) * (size_t)n_wparts);
  }
  o->flags = add(sizeof(unsigned short) * (2 * MT + 2));
  const int nb = (hc == 1) ? n_wparts : (hc == 0 ? n_bparts : std::max(n_bparts, n_wparts));
  o->bparts = add(sizeof(double) * (size_t)nb * 2 * (3 + d));
  o->scratch = add(sizeof(double) * (gene_scratch_doubles(Mp, Cp) + 8));
  o->lambda = add(sizeof(double) * 2 * M);
  o->qags = add(qags_workspace_bytes(kSkatoLimit));
  o->stats = add(sizeof(GeneStats));
  o->vt = vt ? add(sizeof(double) * (gene_vt_doubles(Mp) + Mp)) : 0;
  o->dbg_flip = o->dbg_kept = 0;
  if (dbg) {
    o->dbg_flip = add(sizeof(int) * M);
    o->dbg_kept = add(sizeof(int) * M);
  }
}

// kind (optional, per gene): what the engine's own decoder wrote into the block — 1 hard calls (+ imputed means),
// 0 dosages, -1 unknown.
// RVT_TRACE_BATCH=1: host time of run_batch's phases to stderr (diagnosis of a host-bound launch path)
struct BatchTrace {
  bool on;
  std::chrono::steady_clock::time_point t;
  double ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  BatchTrace() : on(getenv("RVT_TRACE_BATCH") != nullptr), t(std::chrono::steady_clock::now()) {}
  void mark(int k) {
    if (!on) return;
    const auto now = std::chrono::steady_clock::now();
    ph[k] += std::chrono::duration<double, std::micro>(now - t).count();
    t = now;
  }
  void print(int n) const {
    if (on)
      fprintf(stderr, "run_batch n=%d us: slot %.0f layout %.0f arena+desc %.0f sort+copy %.0f suffstat %.0f tail %.0f\n", n, ph[0],
              ph[1], ph[2], ph[3], ph[4], ph[5]);
  }
};

// The packed-row path of RESIDENT .bed genes forms G'[X | rr] on the int8 matrix cores (gene_tnull_hcp, suffstat_hcp.hip.h): every
// column of [X_0 .. X_{d-1} | rr] as eight balanced base-128 digits of its fixed-point value, 56 bits below a power of two above
// twice the column's largest entry, in operand order.  Built from the device's null tile the first time a batch asks for it
// (c->hcp_planes_state: 0 not built, 1 ready, -1 this model cannot: a column whose largest entry exceeds 2^16 x the median of
// its non-zero magnitudes would leave its typical entries fewer than 39 bits — such a model keeps the fp64 product).
static int ensure_hcp_planes(rvt_ctx* c) {
  if (c->hcp_planes_state != 0) return RVT_OK;
  c->hcp_planes_state = -1;
  const NullConsts& nc = c->nc;
  const int64_t N = nc.N, ld = nc.ld;
  const int d = nc.d, ncx = d + 1;
  if (nc.binary || !c->d_nulltile || ncx > 16) return RVT_OK;
  std::vector<double> tile((size_t)ld * ncx);
  HIP_TRY(c, hipMemcpy(tile.data(), c->d_nulltile, sizeof(double) * tile.size(), hipMemcpyDeviceToHost));  // [X | rr], ld apart
  double scale[16];
  int shift[16];
  for (int k = 0; k < 16; ++k) {
    scale[k] = 1.0;
    shift[k] = 0;
  }
  for (int k = 0; k < ncx; ++k) {
    const double* col = tile.data() + (size_t)k * ld;
    double mx = 0.0;
    for (int64_t i = 0; i < N; ++i) mx = std::max(mx, std::fabs(col[i]));
    if (!std::isfinite(mx)) return RVT_OK;
    if (mx > 0.0) {
      // the fixed point is 56 bits below twice the LARGEST entry: the column's typical entry — the median of its non-zero
      // magnitudes (a root mean square follows a single outlier) — must keep 39 of them
      if (!column_scale_ok(col, N, mx, 0x1p16)) return RVT_OK;
      int e;
      std::frexp(mx, &e);       // mx = f 2^e, 0.5 <= f < 1
      shift[k] = 56 - (e + 1);  // |x| 2^shift < 2^55; eight balanced digits end at 63 (128^7 + .. + 1) = 0.496 2^56
      if (std::ldexp(mx, shift[k]) > 0.98 * 0x1p55) shift[k] -= 1;
      scale[k] = std::ldexp(1.0, -shift[k]);
    }
  }
  const int64_t ngroups = (ld + 63) / 64 + 1;
  std::vector<unsigned char> xq((size_t)ngroups * kHcpPlanes * 4 * ncx * 16, 0);
  for (int k = 0; k < ncx; ++k) {
    const double* col = tile.data() + (size_t)k * ld;
    for (int64_t i = 0; i < N; ++i) {
      const int64_t g = i >> 6, T = (i >> 4) & 3, q = (i >> 2) & 3, l = i & 3;
      long long qv = llrint(std::ldexp(col[i], shift[k]));
      for (int p = kHcpPlanes - 1; p >= 0; --p) {
        long long r = qv & 127;
        if (r >= 64) r -= 128;
        qv = (qv - r) >> 7;
        xq[(((size_t)(g * kHcpPlanes + p) * 4 + q) * ncx + k) * 16 + T * 4 + l] = (unsigned char)(signed char)r;
      }
    }
  }
  HIP_TRY(c, hipMalloc((void**)&c->d_hcp_xq, xq.size()));
  HIP_TRY(c, hipMalloc((void**)&c->d_hcp_scale, sizeof(scale)));
  HIP_TRY(c, hipMemcpy(c->d_hcp_xq, xq.data(), xq.size(), hipMemcpyHostToDevice));
  HIP_TRY(c, hipMemcpy(c->d_hcp_scale, scale, sizeof(scale), hipMemcpyHostToDevice));
  c->hcp_planes_state = 1;
  return RVT_OK;
}

int run_batch(rvt_ctx* c, int n, const double* const* dG, const int* Ms, const double* af,
              const int64_t* ids, uint32_t tests, const rvt_params* prm, rvt_gene_result* out,
              DebugOut* dbg, CovOut* cov, const signed char* kind) {
  BatchTrace bt;
  if (!c || n < 0 || (n > 0 && (!dG || !Ms || !af || !out))) return fail(c, RVT_E_INVALID, "bad batch arguments");
  if (!c->have_null) return fail(c, RVT_E_STATE, "no null model set");
  if (n == 0) return RVT_OK;
  hipSetDevice(c->device);
  if (c->colq.n > 0) {  // (columns queued by rvt_block_upload_columns may belong to a block of this batch)
    int rcq = flush_col_queue(c);
    if (rcq) return rcq;
    HIP_TRY(c, sync_stream(c->io_stream));
  }
  // pick the slot that was launched longest ago; if its batch is still in flight, finish it first
  Slot* slp = &c->slots[0];
  for (int i = 1; i < kSlots; ++i)
    if (c->slots[i].seq < slp->seq) slp = &c->slots[i];
  if (dbg) {  // inspection calls run alone
    int rc = rvt_sync(c);
    if (rc) return rc;
    slp = &c->slots[0];
  }
  Slot& sl = *slp;
  {
    int rc = finish_slot(c, sl);
    if (rc) return rc;
  }
  bt.mark(0);
  sl.seq = ++c->launch_seq;
  hipStream_t st = sl.stream;
  if (c->io_wait_pending) {
    HIP_TRY(c, hipStreamWaitEvent(st, c->ev_io, 0));
    c->io_wait_pending = false;
  }
  rvt_params params;
  if (prm)
    params = *prm;
  else
    params = rvt_params{1.0, 25.0, 1.0, 25.0, 0, 0.05};
  const NullConsts& nc = c->nc;
  const int d = nc.d;
  const int64_t ld = nc.ld, N = nc.N, nsteps = ld >> 4;
  const int n_bparts = (int)((N + kBurdenSPB - 1) / kBurdenSPB);
  // binary trait, gene tests: the hard-call genes of the batch take the workgroup-cooperative integer kernel when the null
  // model's operands exist (rvt_set_null); MetaScore slices and debug runs keep the one-wave kernel
  bool hcx = nc.binary && c->hcx_ok && c->hc_enabled && !cov && !(dbg && dbg->cmc) && !(tests & RVT_TEST_FAMSKAT);
  if (hcx) {  // ... and only if a gene of the batch will start on it (the split of the sample axis follows the kernel: a batch
              // of dosage blocks must be cut exactly as with the hard-call path off — the same records bit for bit)
    bool any = false;
    for (int g = 0; g < n && !any; ++g) {
      const int k = kind ? (kind[g] < 0 ? -1 : (kind[g] & 0xf)) : -1;
      any = (Ms[g] + 15) / 16 <= kHcxMaxMT && (k == 1 || (k < 0 && c->content_hint != 0));
    }
    hcx = any;
  }
  const bool nd_is_default = c->d_nulltile && c->d_X == c->d_nulltile && c->d_rr == c->d_nulltile + (size_t)ld * d &&
                             c->d_zeros == c->d_nulltile + (size_t)ld * (d + 1);
  const bool score_hc = cov && cov->score && cov->slice_hc;
  // (a binary trait takes the weighted hard-call kernel when its digit planes exist: gene tests and MetaScore slices)
  const bool hcw = nc.binary && c->d_nulltile_w != nullptr && c->d_vq != nullptr && (!cov || score_hc);
  const bool hc_possible = c->hc_enabled && (!nc.binary || hcw) && (!cov || score_hc) &&
                           !(dbg && dbg->cmc) && d <= kHcMaxD && !(tests & RVT_TEST_FAMSKAT) &&
                           c->d_nulltile != nullptr && nd_is_default;
  // the per-gene condition of the float-digit dosage kernel (quantitative trait; wave-part length aside).  Round 5: OPT-IN —
  // only when the caller asked for it with rvt_set_dosage_float(1).  It computes G'G of float-precision dosages as an exact
  // integer (bit-reproducible whatever the split of the samples), but it does not pay in speed (0.44 of HBM live against the
  // fp64 kernel's 0.42: +3 %), and BGEN genes with a missing call paid a failed pass on it — so BGEN genes no
  // longer start on it by themselves.  ONE predicate for
  // the batch-level prediction below and the per-gene decision further down (they had drifted apart)
  const bool fdx_model = hc_possible && !nc.binary && !cov && c->fdx_ok && c->lattice_den == 0;
  auto fdx_gene = [&](int g) {
    const int k = kind ? (kind[g] < 0 ? -1 : (kind[g] & 0xf)) : -1;
    return fdx_model && (Ms[g] + 15) / 16 <= kFdxEngineMT && (uint64_t)Ms[g] * (uint64_t)ld * 8ull < (1ull << 31) &&
           c->dosage_float && (k == 0 || (k < 0 && c->content_hint == 0));
  };
  // does a gene of the batch start on it?  (its wave-parts are shorter, and the batch is not split over two streams)
  bool fdx_batch = false;
  for (int g = 0; g < n && !fdx_batch; ++g) fdx_batch = fdx_gene(g);
  int n_wparts, steps_per;
  choose_split(ld, n, nc.binary != 0, &n_wparts, &steps_per, hcx, fdx_batch);
  // ---- sizes ---------------------------------------------------------------------------------------
  std::vector<GeneDesc> desc(n);
  size_t total = 0, af_total = 0;
  auto add = [&](size_t bytes) {
    total = (total + 255) / 256 * 256;
    const size_t o = total;
    total += bytes;
    return o;
  };
  std::vector<GeneOff> offs(n);
  int maxM = 0, n_hc = 0;
  const int hc_max_mt = hcw ? kHcwMaxMT : kHcMaxMT;
  const bool predict_hc = c->content_hint != 0;  // blocks of unknown content (rvt_set_content_hint)
  // dosages on a decimal lattice (rvt_set_dosage_lattice): the caller's doubles when the hint says dosages, and what the
  // VCF dosage decoder wrote — gene_suffstat_lat (hc = 2), which tests every value like the hard-call kernel does
  const bool lat_possible = hc_possible && !nc.binary && !cov && c->lattice_den > 0;
  // float-precision dosages (what the BGEN decoder wrote: kind 0; blocks of unknown content when rvt_set_dosage_float says
  // so): gene_suffstat_fdx (hc = 4), which tests every value as well.  M <= 64; the batch's wave-parts are cut for it.
  const bool fdx_possible = fdx_model && steps_per <= kFdxMaxSteps;

  for (int g = 0; g < n; ++g) {
    const int M = Ms[g];
    if (M < 1) return fail(c, RVT_E_INVALID, "gene %d has M=%d", g, M);
    if (M > RVT_MAX_VARIANTS) return fail(c, RVT_E_TOO_LARGE, "gene %d: M=%d exceeds RVT_MAX_VARIANTS", g, M);
    maxM = std::max(maxM, M);
    GeneDesc& gd = desc[g];
    std::memset(&gd, 0, sizeof(gd));
    gd.G = dG[g];
    gd.M = M;
    gd.MT = (M + 15) / 16;
    gd.CT = (M + d + 1 + 15) / 16;
    gd.Mp = 16 * gd.MT;
    gd.Cp = 16 * gd.CT;
    gd.n_wparts = n_wparts;
    gd.steps_per_wpart = steps_per;
    gd.gene_id = ids ? ids[g] : g;
    // hard-call path: unweighted null model, block known to hold only 0.0 / 1.0 / 2.0, a single-pass tile class
    gd.hc = 0;
    const bool kind_packed = kind && kind[g] >= 0 && (kind[g] & 0xf) == 3;  // (bit 4 of a packed kind: T from digit planes)
    gd.hcp_planes = 0;
    if (hc_possible && gd.MT <= hc_max_mt && (kind_packed || (uint64_t)M * (uint64_t)ld * 8ull < (1ull << 31))) {
      if (score_hc) {
        gd.hc = cov->slice_hc[g] ? 1 : 0;
        if (kind_packed) {  // a slice of a resident .bed matrix (rvt_score_bed_dev): packed rows, gene_suffstat_hcp
          gd.hc = 3;
          gd.pk_pitch = (int)(((size_t)((N + 3) / 4) + 15) / 16 * 16);
          gd.hcp_planes = (kind[g] & 0x10) ? 1 : 0;
        }
      } else {
        const int k = kind ? (kind[g] < 0 ? -1 : (kind[g] & 0xf)) : -1;
        gd.hc = (k == 1 || (k < 0 && predict_hc)) ? 1 : 0;
        if (k == 3) {  // the block holds PLINK 2-bit rows (rvt_submit_gene_bed): gene_suffstat_hcp
          gd.hc = 3;
          gd.hcp_planes = (kind[g] & 0x10) ? 1 : 0;
          gd.pk_pitch = (int)(((size_t)((N + 3) / 4) + 15) / 16 * 16);
        }
        if (lat_possible && gd.MT <= kLatMaxMT && (k == 2 || (k < 0 && !predict_hc))) {
          gd.hc = 2;
          gd.lat_den = (double)c->lattice_den;
        }
        if (fdx_possible && fdx_gene(g)) {
          gd.hc = 4;
          gd.lat_den = 0x1p37;
        }
      }
    }
    if (kind_packed && gd.hc != 3)
      return fail(c, RVT_E_STATE, "gene %d was submitted as packed rows but the batch cannot take the packed kernel", g);
    gd.n_bparts = gd.hc ? n_wparts : n_bparts;
    if (gd.hc) {
      n_hc++;
      for (int j = 0; j < M; ++j)  // predicted flips: column sum > N  <=>  allele frequency > 1/2 (verified on the device)
        if (af[af_total + j] > 0.5) gd.pflip[j >> 4] |= (unsigned short)(1u << (j & 15));
    }
    layout_gene(M, d, n_wparts, nsteps, n_bparts, dbg != nullptr, gd.hc ? 1 : 0, &total, &offs[g],
                (tests & RVT_TEST_ANALYTICVT) != 0);
    af_total += M;
  }
  bt.mark(1);
  const size_t off_af = add(sizeof(double) * af_total);
  const size_t off_desc = add(sizeof(GeneDesc) * n);
  const size_t off_res = add(sizeof(rvt_gene_result) * n);
  // masked-entry tables of the genes on the cooperative weighted kernel (64-bit integers, zeroed before the launch)
  size_t pqw_bytes = 0;
  std::vector<size_t> pqw_at(hcx ? n : 0);
  if (hcx)
    for (int g = 0; g < n; ++g)
      if (desc[g].hc == 1) {
        pqw_at[g] = pqw_bytes;
        pqw_bytes += sizeof(unsigned long long) * hcx_pq_entries(desc[g].Mp);
      }
  const size_t off_pqw = pqw_bytes ? add(pqw_bytes) : 0;
  // device work lists of the hard-call genes (gene_flags_hc_kernel): [0] handed back, [1] burden sums to redo, then
  // the two index lists
  const size_t off_lists = add(sizeof(int) * (4 + 2 * (size_t)std::max(n_hc, 1)));
  size_t off_cov = 0, off_cov_xz = 0, off_cov_cs = 0, off_cov_poly = 0, off_cov_bur = 0, off_cov_ok = 0;
  if (cov && cov->score) {  // one gene per 16-column slice of the block: per-variant records only
    size_t vt = 0;
    for (int g = 0; g < n; ++g) vt += (size_t)Ms[g];
    off_cov_bur = add(sizeof(double) * vt * 5);
    off_cov_ok = add(sizeof(int) * vt);
  } else if (cov) {
    const size_t V = (size_t)Ms[0];
    off_cov = add(sizeof(double) * V * V);
    off_cov_xz = add(sizeof(double) * V * (size_t)d);
    off_cov_cs = add(sizeof(double) * V);
    off_cov_poly = add(sizeof(int) * V);
    off_cov_bur = add(sizeof(double) * V * 4);
  }
  size_t off_dbg_cmc = 0, off_dbg_zeg = 0;
  if (dbg && dbg->cmc) {
    off_dbg_cmc = add(sizeof(double) * N);
    off_dbg_zeg = add(sizeof(double) * N);
  }
  int rc = ensure_arena(c, sl, total + 4096);
  if (rc) return rc;
  char* base = sl.arena.base;
  // ---- host staging: descriptors + af ------------------------------------------------------------------
  const size_t stage_bytes = sizeof(GeneDesc) * n + sizeof(double) * af_total + sizeof(rvt_gene_result) * n + 64;
  rc = ensure_stage(c, sl, stage_bytes);
  if (rc) return rc;
  GeneDesc* h_desc = reinterpret_cast<GeneDesc*>(sl.h_stage);
  double* h_af = reinterpret_cast<double*>(sl.h_stage + sizeof(GeneDesc) * n);
  size_t afpos = 0;
  for (int g = 0; g < n; ++g) {
    GeneDesc& gd = desc[g];
    const GeneOff& o = offs[g];
    gd.parts = reinterpret_cast<double*>(base + o.parts);
    gd.colstat = reinterpret_cast<double*>(base + o.colstat);
    gd.masks = gd.hc ? nullptr : reinterpret_cast<unsigned long long*>(base + o.masks);
    gd.pq = ((gd.hc == 1 || gd.hc == 3) && !hcw && o.pq) ? reinterpret_cast<unsigned*>(base + o.pq) : nullptr;
    gd.pqw = (hcx && gd.hc == 1) ? reinterpret_cast<unsigned long long*>(base + off_pqw + pqw_at[g]) : nullptr;
    gd.wflags = (gd.hc && o.wflags) ? reinterpret_cast<unsigned*>(base + o.wflags) : nullptr;
    gd.flags = reinterpret_cast<unsigned short*>(base + o.flags);
    gd.bparts = reinterpret_cast<double*>(base + o.bparts);
    gd.scratch = reinterpret_cast<double*>(base + o.scratch);
    gd.lambda = reinterpret_cast<double*>(base + o.lambda);
    gd.qags_mem = base + o.qags;
    gd.af = reinterpret_cast<const double*>(base + off_af) + afpos;
    gd.stats = reinterpret_cast<GeneStats*>(base + o.stats);
    gd.vt_mem = (tests & RVT_TEST_ANALYTICVT) ? reinterpret_cast<double*>(base + o.vt) : nullptr;
    gd.result = reinterpret_cast<rvt_gene_result*>(base + off_res) + g;
    if (dbg) {
      gd.dbg_flip = reinterpret_cast<int*>(base + o.dbg_flip);
      gd.dbg_kept = reinterpret_cast<int*>(base + o.dbg_kept);
      if (dbg->cmc && g == 0) {
        gd.dbg_cmc = reinterpret_cast<double*>(base + off_dbg_cmc);
        gd.dbg_zeg = reinterpret_cast<double*>(base + off_dbg_zeg);
      }
    }
    afpos += gd.M;
  }
  bt.mark(2);
  std::memcpy(h_af, af, sizeof(double) * af_total);
  HIP_TRY(c, hipMemcpyAsync(base + off_af, h_af, sizeof(double) * af_total, hipMemcpyHostToDevice, st));
  // widest genes first: their workgroups run longest, so they should not be the tail of the launch
  std::vector<int> order(n);
  for (int g = 0; g < n; ++g) order[g] = g;
  // genes whose tile configuration has no unrolled body (more than 6 row tiles, or — with many covariates — more than
  // one extra column tile) go to the panelled kernel; they come first
  // order: general-path genes first (panelled ones in front), then the hard-call genes; widest first inside each
  auto needs_panel = [&](const GeneDesc& g) { return !g.hc && (g.MT > kMaxMT || g.CT > g.MT + 1); };
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
    if (desc[a].hc != desc[b].hc) return desc[a].hc < desc[b].hc;
    const bool pa = needs_panel(desc[a]), pb = needs_panel(desc[b]);
    if (pa != pb) return pa;
    return desc[a].M > desc[b].M;
  });
  const int n_gen = n - n_hc;  // descriptors [0, n_gen) take the general kernels, [n_gen, n) the hard-call kernels
  for (int k = 0; k < n; ++k) h_desc[k] = desc[order[k]];
  {  // order of the p-value workgroups: workgroup b takes h_desc[h_desc[b].pv_gene] — indices into the SORTED array, by
     // falling M (stable): the longest work first
    std::vector<int>& ord = c->pv_order;
    ord.resize(n);
    for (int k = 0; k < n; ++k) ord[k] = k;
    std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return h_desc[a].M > h_desc[b].M; });
    for (int b = 0; b < n; ++b) h_desc[b].pv_gene = ord[b];
  }
  GeneDesc* d_desc = reinterpret_cast<GeneDesc*>(base + off_desc);
  HIP_TRY(c, hipMemcpyAsync(d_desc, h_desc, sizeof(GeneDesc) * n, hipMemcpyHostToDevice, st));
  NullDev nd{c->d_X, c->d_res, c->d_rr, c->d_v, c->d_zeros};
  // ---- K2: one launch for the whole batch ------------------------------------------------------------------
  // Stage 1 (bandwidth / MFMA bound) runs on the shared K2 stream so that two batches never split the chip
  // between two sufficient-statistics launches; stage 2 (latency bound, few waves) continues on the slot's
  // stream and overlaps the NEXT batch's stage 1.
  bt.mark(3);
  const int slot_idx = (int)(slp - &c->slots[0]);
  HIP_TRY(c, hipEventRecord(c->ev_in[slot_idx], st));
  HIP_TRY(c, hipStreamWaitEvent(c->k2_stream, c->ev_in[slot_idx], 0));
  // general-path genes of a mixed batch (a few genes with imputed values among hard-call ones) go to the second
  // sufficient-statistics stream and run beside the hard-call launches
  // (not beside the float-digit kernel: its whole-CU workgroups find no free CU while one-wave workgroups flood the chip)
  const bool split = n_gen > 0 && n_hc > 0 && !fdx_batch;
  hipStream_t gst = split ? c->k2b_stream : c->k2_stream;
  if (split) HIP_TRY(c, hipStreamWaitEvent(gst, c->ev_in[slot_idx], 0));
  int k0 = 0;
  while (k0 < n_gen && needs_panel(h_desc[k0])) ++k0;  // these need the panelled kernel
  if (k0 > 0) {
    Scope sc(c, 0, gst);
    const int nPR = (h_desc[0].MT + 3) / 4, nPC = (h_desc[0].CT + 3) / 4;
    int npanels = 0;
    for (int pr = 0; pr < nPR; ++pr) npanels += nPC - pr;
    dim3 grid(n_wparts, k0, npanels), block(64);
    if (nc.binary)
      k2_launch_panel_w1(grid, gst, d_desc, nd, (long long)N, (long long)ld, d);
    else
      k2_launch_panel_w0(grid, gst, d_desc, nd, (long long)N, (long long)ld, d);
  }
  for (int k = k0; k < n_gen;) {  // descriptors are sorted by width, so every register-budget group is one contiguous run
    const int grp = suffstat_group(h_desc[k].MT, h_desc[k].CT, nc.binary != 0);
    int e = k;
    while (e < n_gen && suffstat_group(h_desc[e].MT, h_desc[e].CT, nc.binary != 0) == grp) ++e;
    launch_suffstat(c, gst, grp, d_desc + k, e - k, n_wparts, nd);
    k = e;
  }
  if (pqw_bytes) HIP_TRY(c, hipMemsetAsync(base + off_pqw, 0, pqw_bytes, c->k2_stream));
  for (int k = n_gen; k < n;) {  // hard-call genes: one launch per tile class (contiguous runs, widest class first)
    int e = k;
    // (gene_suffstat_hcx_any / gene_suffstat_fdx_any: every class at once)
    const bool one_launch = (hcx && c->hcx_fused && h_desc[k].hc == 1) || (h_desc[k].hc == 4 && c->hcx_fused);
    while (e < n && (one_launch || h_desc[e].MT == h_desc[k].MT) && h_desc[e].hc == h_desc[k].hc &&
           h_desc[e].hcp_planes == h_desc[k].hcp_planes)
      ++e;
    hipStream_t hst = c->k2_stream;
    Scope sc(c, 4, hst);
    if (h_desc[k].hc == 3) {
      // (resident genes: G'[X | rr] from the digit planes of THIS null tile — rvt_set_null drops them with the model; a wave-part
      //  beyond the int32 range of a pair sum, or a model whose columns the planes cannot carry, keeps the fp64 product)
      static const int planes_sw = getenv("RVT_HCP_PLANES") ? atoi(getenv("RVT_HCP_PLANES")) : 1;  // 0 never, 2 every packed gene
      bool planes = nd_is_default && (h_desc[k].hcp_planes || planes_sw == 2) && planes_sw != 0 &&
                    (long long)h_desc[k].steps_per_wpart * 16 <= kHcpPlaneMaxSamples;
      if (planes) {
        const int rcp = ensure_hcp_planes(c);
        if (rcp) return rcp;
        planes = c->hcp_planes_state == 1;
      }
      const HcpPlanes pl{c->d_hcp_xq, c->d_hcp_scale, d + 1};
      k2_launch_hcp(h_desc[k].MT, dim3(n_wparts, e - k), hst, d_desc + k, NullTile{c->d_nulltile, d + 2}, (long long)N,
                    (long long)ld, d, planes ? &pl : nullptr,
                    planes && cov && cov->score && h_desc[k].MT <= 2 && !getenv("RVT_SCORE_BED_FULL"));
    }
    else if (h_desc[k].hc == 4)
      k2_launch_fdx(one_launch ? 0 : h_desc[k].MT, dim3(n_wparts, e - k), hst, d_desc + k, c->fdx_tile, (long long)N, (long long)ld, d);
    else if (h_desc[k].hc == 2)
      k2_launch_lat(h_desc[k].MT, dim3(n_wparts, e - k), hst, d_desc + k, NullTile{c->d_nulltile, d + 2},
                    (double)c->lattice_den, (long long)N, (long long)ld, d);
    else if (hcx)
      k2_launch_hcx(one_launch ? 0 : h_desc[k].MT, dim3(n_wparts, e - k), hst, d_desc + k, c->hcx_tile, (long long)N,
                    (long long)ld, d);
    else if (hcw)
      k2_launch_hcw(h_desc[k].MT, dim3(n_wparts, e - k), hst, d_desc + k, NullTileW{c->d_nulltile_w, d + 3, c->d_vq},
                    (long long)N, (long long)ld, d);
    else
      k2_launch_hc(h_desc[k].MT, dim3(n_wparts, e - k), hst, d_desc + k, NullTile{c->d_nulltile, d + 2}, (long long)N,
                   (long long)ld, d);
    k = e;
  }
  if (split) {
    HIP_TRY(c, hipEventRecord(c->ev_k2b[slot_idx], gst));
    HIP_TRY(c, hipStreamWaitEvent(st, c->ev_k2b[slot_idx], 0));
  }
  HIP_TRY(c, hipEventRecord(c->ev_k2[slot_idx], c->k2_stream));
  HIP_TRY(c, hipStreamWaitEvent(st, c->ev_k2[slot_idx], 0));
  if (n_hc > 0) {
    // What the hard-call kernels assumed is verified now (flip predictions, counted monomorphic columns, and above all
    // that every block held hard calls and at most one other value per column): gene_flags_hc_kernel.  Genes it hands
    // back are computed by the general kernel right here, on the batch's own stream (the shared streaming stream goes
    // straight on to the next batch) — one launch per register-budget group over the device work list the flags kernel
    // writes; its workgroups leave at once when the list is empty.
    int* d_lists = reinterpret_cast<int*>(base + off_lists);
    HIP_TRY(c, hipMemsetAsync(d_lists, 0, 4 * sizeof(int), st));
    hipLaunchKernelGGL(gene_flags_hc_kernel, dim3(n_hc), dim3(64), 0, st, d_desc + n_gen, (long long)N, d_lists,
                       n_hc);
    bool grp_present[3] = {false, false, false};
    for (int k = n_gen; k < n; ++k) grp_present[suffstat_group(h_desc[k].MT, h_desc[k].CT, nc.binary != 0)] = true;
    for (int grp = 0; grp < 3; ++grp)
      if (grp_present[grp]) launch_suffstat(c, st, grp, d_desc + n_gen, n_hc, n_wparts, nd, d_lists);
  }
  bt.mark(4);
  if (cov && cov->score) {  // MetaScore: per-variant statistics of every slice, returned synchronously
    size_t vt = 0;
    for (int g = 0; g < n; ++g) vt += (size_t)Ms[g];
    double* d_bur = reinterpret_cast<double*>(base + off_cov_bur);
    int* d_ok = reinterpret_cast<int*>(base + off_cov_ok);
    hipLaunchKernelGGL(score_finish_kernel, dim3((unsigned)n), dim3(64), 0, st, d_desc, c->d_nc, (int)vt, d_bur, d_ok);
    HIP_TRY(c, hipGetLastError());
    const size_t vb8 = sizeof(double) * vt;
    HIP_TRY(c, hipMemcpyAsync(cov->ustat, d_bur, vb8, hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipMemcpyAsync(cov->vstat, d_bur + vt, vb8, hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipMemcpyAsync(cov->effect, d_bur + 2 * vt, vb8, hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipMemcpyAsync(cov->se, d_bur + 3 * vt, vb8, hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipMemcpyAsync(cov->pval, d_bur + 4 * vt, vb8, hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipMemcpyAsync(cov->ok, d_ok, sizeof(int) * vt, hipMemcpyDeviceToHost, st));
    HIP_TRY(c, sync_stream(st));
    return RVT_OK;
  }
  if (cov) {  // MetaCov: finish the covariance algebra of this one block and return its band synchronously
    const int V = Ms[0];
    CovConsts cc;
    std::vector<double> zz;
    {
      int rcc = cov_constants(c, cov->fam, &cc, &zz);
      if (rcc) return rcc;
      if (cov->fam && cov->uncentred) cc.inv_n = 0.0;  // every centring term carries the column mean s / N
    }
    double* d_xz = reinterpret_cast<double*>(base + off_cov_xz);
    double* d_cs = reinterpret_cast<double*>(base + off_cov_cs);
    int* d_poly = reinterpret_cast<int*>(base + off_cov_poly);
    double* d_cov = reinterpret_cast<double*>(base + off_cov);
    if (cov->fam) {
      HIP_TRY(c, hipMemcpyAsync(d_cs, cov->d_raw_colsum, sizeof(double) * (size_t)V, hipMemcpyDeviceToDevice, st));
      HIP_TRY(c, hipMemcpyAsync(d_poly, cov->d_raw_poly, sizeof(int) * (size_t)V, hipMemcpyDeviceToDevice, st));
    }
    double* d_bur = reinterpret_cast<double*>(base + off_cov_bur);  // ustat | vstat | af | pval
    const bool bur = cov->fam && cov->ustat;
    hipLaunchKernelGGL(cov_prepare_kernel, dim3(1), dim3(256), 0, st, d_desc, cc, d_xz, d_cs, d_poly,
                       bur ? d_bur : (double*)nullptr, bur ? d_bur + 2 * (size_t)V : (double*)nullptr);
    hipLaunchKernelGGL(cov_rows_kernel, dim3(V), dim3(256), 0, st, d_desc, cc, d_xz, d_cs, d_cov);
    if (bur)
      hipLaunchKernelGGL(fam_burden_finish_kernel, dim3((unsigned)((V + 255) / 256)), dim3(256), 0, st, d_cov, V, d_bur,
                         d_bur + (size_t)V, d_bur + 3 * (size_t)V);
    HIP_TRY(c, hipGetLastError());
    const int dz = cc.d;
    if (cov->cov)
      HIP_TRY(c, hipMemcpyAsync(cov->cov, d_cov, sizeof(double) * (size_t)V * V, hipMemcpyDeviceToHost, st));
    if (cov->xz)
      HIP_TRY(c, hipMemcpyAsync(cov->xz, d_xz, sizeof(double) * (size_t)V * dz, hipMemcpyDeviceToHost, st));
    if (cov->poly)
      HIP_TRY(c, hipMemcpyAsync(cov->poly, d_poly, sizeof(int) * (size_t)V, hipMemcpyDeviceToHost, st));
    if (bur) {
      const size_t vb8 = sizeof(double) * (size_t)V;
      HIP_TRY(c, hipMemcpyAsync(cov->ustat, d_bur, vb8, hipMemcpyDeviceToHost, st));
      HIP_TRY(c, hipMemcpyAsync(cov->vstat, d_bur + (size_t)V, vb8, hipMemcpyDeviceToHost, st));
      HIP_TRY(c, hipMemcpyAsync(cov->af, d_bur + 2 * (size_t)V, vb8, hipMemcpyDeviceToHost, st));
      HIP_TRY(c, hipMemcpyAsync(cov->pval, d_bur + 3 * (size_t)V, vb8, hipMemcpyDeviceToHost, st));
    }
    HIP_TRY(c, sync_stream(st));
    if (cov->zz) std::memcpy(cov->zz, zz.data(), sizeof(double) * (size_t)dz * dz);
    return RVT_OK;
  }
  const bool burden = (tests & (RVT_TEST_CMC | RVT_TEST_ZEGGINI)) != 0;
  if (burden || dbg) {
    // on the batch's own stream: the collapse overlaps the next batch's sufficient-statistics launches, which
    // leave about half of the HBM bandwidth unused
    hipStream_t bs = st;
    if (n_gen > 0) {
      {
        Scope sc(c, 1, bs);
        hipLaunchKernelGGL(gene_flags_kernel, dim3(n_gen), dim3(64), 0, bs, d_desc, (long long)N);
      }
      // gene groups of <= 64 keep one workgroup's loop short while X/res/v stay in registers; all groups in one launch
      const int gpg = 64, ngroups = (n_gen + gpg - 1) / gpg;
      Scope sc(c, 1, bs);
      if (d <= 4)
        hipLaunchKernelGGL((burden_collapse_kernel<4>), dim3(n_bparts, ngroups), dim3(256), 0, bs, d_desc, n_gen, gpg,
                           nd, (long long)N, (long long)ld, d, nc.binary, tests);
      else
        hipLaunchKernelGGL((burden_collapse_kernel<RVT_MAX_COV>), dim3(n_bparts, ngroups), dim3(256), 0, bs, d_desc,
                           n_gen, gpg, nd, (long long)N, (long long)ld, d, nc.binary, tests);
    }
    if (n_hc > 0) {
      // hard-call genes carry their burden sums already; where gene_flags_hc_kernel found the in-pass collapse wrong
      // (flip prediction, a counted monomorphic column, an imputed value that counts) or handed the gene back, the sums
      // are redone from the block (a handed-back gene first gets its flags from the general kernel's statistics)
      Scope sc(c, 1, bs);
      const int* d_lists = reinterpret_cast<const int*>(base + off_lists);
      hipLaunchKernelGGL(gene_flags_kernel, dim3(n_hc), dim3(64), 0, bs, d_desc + n_gen, (long long)N);
      if (d <= 4)
        hipLaunchKernelGGL((burden_fallback_kernel<4>), dim3(kFallbackGrid), dim3(256), 0, bs, d_desc + n_gen, d_lists,
                           n_hc, n_wparts, nd, (long long)N, (long long)ld, d, nc.binary);
      else
        hipLaunchKernelGGL((burden_fallback_kernel<kHcMaxD>), dim3(kFallbackGrid), dim3(256), 0, bs, d_desc + n_gen,
                           d_lists, n_hc, n_wparts, nd, (long long)N, (long long)ld, d, nc.binary);
    }
  }
  const unsigned tests_eff = burden ? tests : (tests & ~(RVT_TEST_CMC | RVT_TEST_ZEGGINI));
  {
    Scope sc(c, 2, st);
    if (tests & RVT_TEST_FAMSKAT)
      hipLaunchKernelGGL(fam_assemble_kernel, dim3(n), dim3(1024), 0, st, d_desc, c->d_nc);
    else {
      // the sum of the wave-part images on a grid of its own (RVT_AS_REDUCE=0: inside the assembly kernel, as until round 5)
      static const bool reduce_apart = !(getenv("RVT_AS_REDUCE") && atoi(getenv("RVT_AS_REDUCE")) == 0);
      if (reduce_apart) {
        const int maxMpA = (maxM + 15) / 16 * 16, maxCpA = (maxM + d + 1 + 15) / 16 * 16;
        hipLaunchKernelGGL(gene_reduce_parts_kernel, dim3((unsigned)((maxMpA * maxCpA + kReducePiece - 1) / kReducePiece), (unsigned)n),
                           dim3(256), 0, st, d_desc);
      }
      hipLaunchKernelGGL(gene_assemble_kernel, dim3(n), dim3(c->as_threads), 0, st, d_desc, c->d_nc, params, tests_eff,
                         n_bparts, (const double*)c->d_xscale, reduce_apart ? 1 : 0);
    }
  }
  if ((tests & RVT_TEST_ANALYTICVT) && !(tests & RVT_TEST_FAMSKAT)) {
    Scope sc(c, 2, st);
    hipLaunchKernelGGL(gene_vt_kernel, dim3(n), dim3(256), 0, st, d_desc, c->d_nc);
    for (int stage = 0; stage < 2; ++stage) {  // the band probability: a short first stage, a long one where it is needed
      hipLaunchKernelGGL(vt_integrate_kernel, dim3(n, kMvnShifts), dim3(256), 0, st, d_desc, stage);
      hipLaunchKernelGGL(vt_finish_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_desc, n, stage);
    }
  }
  const unsigned tests_eig = (tests & RVT_TEST_FAMSKAT) ? (unsigned)RVT_TEST_SKAT : tests_eff;
  if (tests & (RVT_TEST_SKAT | RVT_TEST_SKATO | RVT_TEST_FAMSKAT)) {
    const int maxMp = (maxM + 15) / 16 * 16;
    {
      Scope sc(c, 2, st);
      size_t want = sizeof(double) * ((size_t)4 * maxMp + (size_t)maxM * maxM);
      if (want > c->eigen_lds_max) want = sizeof(double) * (size_t)4 * maxMp;  // matrices stay in global scratch
      hipLaunchKernelGGL(gene_tridiag_kernel, dim3(kNTridiag, n), dim3(256), want, st, d_desc, c->d_nc, tests_eig,
                         (int)(want / sizeof(double)));
    }
    {
      Scope sc(c, 2, st);
      // (one workgroup per gene over all 13 problems; RVT_SPECTRUM_PER_PROBLEM=1 or a gene too wide for the LDS: a workgroup
      //  per problem, as until round 4)
      static const bool per_problem = getenv("RVT_SPECTRUM_PER_PROBLEM") != nullptr;
      const size_t lds_all = sizeof(double) * ((size_t)39 * maxMp + 64);
      if (per_problem || lds_all > ((size_t)60 << 10))
        hipLaunchKernelGGL(gene_spectrum_kernel, dim3(kNEigen, n), dim3(128), sizeof(double) * (size_t)4 * maxMp, st,
                           d_desc, c->d_nc, tests_eig);
      else
        hipLaunchKernelGGL(gene_spectrum_all_kernel, dim3(n), dim3(128), lds_all, st, d_desc, c->d_nc, tests_eig);
    }
  }
  {
    // (on CUs of its own when RVT_PV_CUS is in force and the batch is big enough to matter: see rvt_init)
    const bool pv_apart = c->pv_cus > 0 && n >= 64 && (tests & RVT_TEST_SKATO);
    hipStream_t pst = st;
    if (pv_apart) {
      pst = c->pv_stream[c->pv_turn++ & 1];
      HIP_TRY(c, hipEventRecord(c->ev_pv_in[slot_idx], st));
      HIP_TRY(c, hipStreamWaitEvent(pst, c->ev_pv_in[slot_idx], 0));
    }
    {
      Scope sc(c, 3, pst);
      // (four coefficient arrays of an even length, the 42 integrand values, the staging of wave_davies_pvalue)
      const size_t smem = sizeof(double) * 4 * ((maxM + 1) & ~1) + sizeof(double) * 42 +
                          sizeof(double) * (3 * 64 + 2 * kTermCap) + sizeof(int) * (64 + 64 + 66) + 32;
      if (tests & RVT_TEST_EXACT_DAVIES)  // Davies' coefficient sums term by term (verification) / in product form
        hipLaunchKernelGGL((gene_pvalue_kernel<false>), dim3(n), dim3(64), smem, pst, d_desc, tests);
      else
        hipLaunchKernelGGL((gene_pvalue_kernel<true>), dim3(n), dim3(64), smem, pst, d_desc, tests);
    }
    if (pv_apart) {
      HIP_TRY(c, hipEventRecord(c->ev_pv_out[slot_idx], pst));
      HIP_TRY(c, hipStreamWaitEvent(st, c->ev_pv_out[slot_idx], 0));
    }
  }
  HIP_TRY(c, hipGetLastError());
  rvt_gene_result* h_res = reinterpret_cast<rvt_gene_result*>(sl.h_stage + sizeof(GeneDesc) * n +
                                                               (sizeof(double) * af_total + 63) / 64 * 64);
  HIP_TRY(c, hipMemcpyAsync(h_res, base + off_res, sizeof(rvt_gene_result) * n, hipMemcpyDeviceToHost, st));
  bt.mark(5);
  bt.print(n);
  sl.pending_out = out;
  sl.pending_done = c->next_done_flag;
  c->next_done_flag = nullptr;
  sl.h_results = h_res;
  sl.pending_n = n;
  if (c->profiling) {
    c->timing.genes += n;
    c->timing.genes_hard_call += n_hc;
    for (int g = 0; g < n; ++g) {
      if (desc[g].hc == 3)
        c->timing.alg_bytes_hc += (double)desc[g].pk_pitch * Ms[g] + 8.0 * (double)N * (d + 2);
      else if (desc[g].hc) c->timing.alg_bytes_hc += 8.0 * (double)N * Ms[g] + 8.0 * (double)N * (d + 2);  // (SURVEY 8d's d + 2, also for the weighted kernels)
      c->timing.alg_bytes += 8.0 * (double)N * Ms[g] + 8.0 * (double)N * (d + 2);
      c->timing.alg_flops += 2.0 * (double)N * Ms[g] * (Ms[g] + d + 1);
    }
  }
  if (dbg) {
    HIP_TRY(c, sync_stream(st));
    const GeneDesc& g0 = desc[0];
    if (dbg->flip) HIP_TRY(c, hipMemcpy(dbg->flip, g0.dbg_flip, sizeof(int) * g0.M, hipMemcpyDeviceToHost));
    if (dbg->kept) HIP_TRY(c, hipMemcpy(dbg->kept, g0.dbg_kept, sizeof(int) * g0.M, hipMemcpyDeviceToHost));
    if (dbg->cmc) HIP_TRY(c, hipMemcpy(dbg->cmc, g0.dbg_cmc, sizeof(double) * N, hipMemcpyDeviceToHost));
    if (dbg->zeg) HIP_TRY(c, hipMemcpy(dbg->zeg, g0.dbg_zeg, sizeof(double) * N, hipMemcpyDeviceToHost));
    if (dbg->desc0) *dbg->desc0 = g0;
  }
  return RVT_OK;
}

int rvt_sync(rvt_ctx* c) {
  if (!c) return RVT_E_INVALID;
  hipSetDevice(c->device);
  if (c->colq.n > 0) {  // columns queued by rvt_block_upload_columns reach their block before anything reads it
    int rc = flush_col_queue(c);
    if (rc) return rc;
  }
  if (c->colq.used[0] || c->colq.used[1]) HIP_TRY(c, sync_stream(c->io_stream));  // (the uploads' device work is on io_stream)
  // oldest batch first, so records reach the caller in launch order
  Slot* order[kSlots];
  for (int i = 0; i < kSlots; ++i) order[i] = &c->slots[i];
  std::sort(order, order + kSlots, [](const Slot* a, const Slot* b) { return a->seq < b->seq; });
  for (Slot* sl : order) {
    int rc = finish_slot(c, *sl);
    if (rc) return rc;
  }
  return RVT_OK;
}

int rvt_wait_oldest(rvt_ctx* c) {
  if (!c) return RVT_E_INVALID;
  hipSetDevice(c->device);
  Slot* oldest = nullptr;
  for (auto& sl : c->slots)
    if (sl.pending_out && (!oldest || sl.seq < oldest->seq)) oldest = &sl;
  return oldest ? finish_slot(c, *oldest) : RVT_OK;
}

int rvt_reserve(rvt_ctx* c, int n, const int* Ms) {
  if (!c || n < 1 || !Ms) return fail(c, RVT_E_INVALID, "bad reserve arguments");
  if (!c->have_null) return fail(c, RVT_E_STATE, "no null model set");
  hipSetDevice(c->device);
  const NullConsts& nc = c->nc;
  const int d = nc.d;
  const int64_t nsteps = nc.ld >> 4;
  const int n_bparts = (int)((nc.N + kBurdenSPB - 1) / kBurdenSPB);
  const bool hcx = nc.binary && c->hcx_ok && c->hc_enabled;
  int n_wparts, steps_per, n_wparts1, steps_per1;
  choose_split(nc.ld, n, nc.binary != 0, &n_wparts, &steps_per, hcx);
  choose_split(nc.ld, n, nc.binary != 0, &n_wparts1, &steps_per1, false);  // (a batch that cannot take the cooperative kernel)
  n_wparts = std::max(n_wparts, n_wparts1);
  if (!nc.binary && c->fdx_ok) {  // (a batch with float-precision dosages is cut into shorter wave-parts)
    choose_split(nc.ld, n, false, &n_wparts1, &steps_per1, false, true);
    n_wparts = std::max(n_wparts, n_wparts1);
  }
  size_t total = 0, af_total = 0;
  GeneOff o;
  for (int g = 0; g < n; ++g) {
    if (Ms[g] < 1 || Ms[g] > RVT_MAX_VARIANTS) return fail(c, RVT_E_INVALID, "gene %d has M=%d", g, Ms[g]);
    layout_gene(Ms[g], d, n_wparts, nsteps, n_bparts, false, -1, &total, &o);
    af_total += (size_t)Ms[g];
    if (hcx) total += 256 + sizeof(unsigned long long) * hcx_pq_entries(16 * ((Ms[g] + 15) / 16));
  }
  total += sizeof(double) * af_total + sizeof(GeneDesc) * n + sizeof(rvt_gene_result) * n + 4 * 256;
  const size_t stage_bytes = sizeof(GeneDesc) * n + sizeof(double) * af_total + sizeof(rvt_gene_result) * n + 64;
  for (int i = 0; i < kSlots; ++i) {  // (the re-run slot grows on demand)
    Slot& sl = c->slots[i];
    int rc = ensure_arena(c, sl, total + 4096);
    if (rc) return rc;
    rc = ensure_stage(c, sl, stage_bytes);
    if (rc) return rc;
  }
  return RVT_OK;
}


int rvt_run_blocks_async(rvt_ctx* c, int n, const double* const* dG, const int* M, const double* af,
                         const int64_t* ids, uint32_t tests, const rvt_params* prm, rvt_gene_result* out) {
  if (c && prm && prm->skat_nperm > 0 && (tests & RVT_TEST_SKAT))
    return fail(c, RVT_E_STATE, "permutation p-values need the synchronous rvt_run_blocks / rvt_collect");
  if (c && (tests & RVT_TEST_FAMSKAT)) return fail(c, RVT_E_INVALID, "FamSKAT runs through rvt_run_fam_blocks");
  return run_batch(c, n, dG, M, af, ids, tests, prm, out, nullptr);
}

int rvt_run_blocks(rvt_ctx* c, int n, const double* const* dG, const int* M, const double* af, const int64_t* ids,
                   uint32_t tests, const rvt_params* prm, rvt_gene_result* out) {
  if (c && (tests & RVT_TEST_FAMSKAT)) return fail(c, RVT_E_INVALID, "FamSKAT runs through rvt_run_fam_blocks");
  if (c && prm && prm->skat_nperm > 0 && (tests & RVT_TEST_SKAT)) {
    int rc0 = rvt_sync(c);
    if (rc0) return rc0;
    return run_blocks_with_perm(c, n, dG, M, af, ids, tests, prm, out);
  }
  // batches of at most 256 genes (bounded work space per batch); consecutive batches pipeline over the slots
  size_t afo = 0;
  for (int g0 = 0; g0 < n; g0 += 256) {
    const int nb = std::min(256, n - g0);
    int rc = run_batch(c, nb, dG + g0, M + g0, af + afo, ids ? ids + g0 : nullptr, tests, prm, out + g0, nullptr);
    if (rc) return rc;
    for (int g = g0; g < g0 + nb; ++g) afo += (size_t)M[g];
  }
  int rc = rvt_sync(c);
  if (rc) return rc;
  if (!ids)
    for (int g = 0; g < n; ++g) out[g].gene_id = g;
  return RVT_OK;
}

int rvt_debug_collapse(rvt_ctx* c, const double* dG, int M, double* cmc_out, double* zeg_out, int* flipped_out,
                       int* kept_out) {
  if (!c || !dG || !cmc_out || !zeg_out) return fail(c, RVT_E_INVALID, "bad arguments");
  std::vector<double> af(M, 0.01);
  rvt_gene_result r;
  DebugOut dbg;
  dbg.flip = flipped_out;
  dbg.kept = kept_out;
  dbg.cmc = cmc_out;
  dbg.zeg = zeg_out;
  const double* p = dG;
  int rc = run_batch(c, 1, &p, &M, af.data(), nullptr, RVT_TEST_ALL, nullptr, &r, &dbg);
  if (rc) return rc;
  return rvt_sync(c);
}

int rvt_debug_suffstat(rvt_ctx* c, const double* dG, int M, double* S, double* T, double* u, double* colsum,
                       double* cmin, double* cmax) {
  if (!c || !dG) return fail(c, RVT_E_INVALID, "bad arguments");
  std::vector<double> af(M, 0.01);
  rvt_gene_result r;
  DebugOut dbg;
  GeneDesc g0;
  dbg.desc0 = &g0;
  const double* p = dG;
  int rc = run_batch(c, 1, &p, &M, af.data(), nullptr, RVT_TEST_SKAT, nullptr, &r, &dbg);
  if (rc) return rc;
  rc = rvt_sync(c);
  if (rc) return rc;
  const int d = c->nc.d;
  const size_t psz = (size_t)g0.Mp * g0.Cp;
  if (g0.hc) {  // handed back by the hard-call kernel: the general kernel's statistics are what the buffers hold
    unsigned short fl = 0;
    HIP_TRY(c, hipMemcpy(&fl, g0.flags + 2 * g0.MT + 1, sizeof(fl), hipMemcpyDeviceToHost));
    if (fl) g0.hc = 0;
  }
  const int rows = g0.hc ? kHcColstatRows : 3;
  std::vector<double> parts((size_t)g0.n_wparts * psz), cs((size_t)g0.n_wparts * rows * g0.Mp);
  HIP_TRY(c, hipMemcpy(parts.data(), g0.parts, sizeof(double) * parts.size(), hipMemcpyDeviceToHost));
  HIP_TRY(c, hipMemcpy(cs.data(), g0.colstat, sizeof(double) * cs.size(), hipMemcpyDeviceToHost));
  // hard-call kernel (suffstat_hc.hip.h): masked-entry tiles and per-column imputed values, combined as gene_assemble does
  const int Mp = g0.Mp, MT = g0.MT;
  std::vector<double> Pp, Qq, mu((size_t)Mp, 0.0), cm((size_t)Mp, 0.0);
  if (g0.hc && g0.pq && g0.wflags) {
    const int pqw = hc_pq_words(MT);
    std::vector<unsigned> pq((size_t)g0.n_wparts * pqw), wf((size_t)g0.n_wparts);
    HIP_TRY(c, hipMemcpy(pq.data(), g0.pq, sizeof(unsigned) * pq.size(), hipMemcpyDeviceToHost));
    HIP_TRY(c, hipMemcpy(wf.data(), g0.wflags, sizeof(unsigned) * wf.size(), hipMemcpyDeviceToHost));
    Pp.assign((size_t)Mp * Mp, 0.0);
    Qq.assign((size_t)Mp * Mp, 0.0);
    const int ntiles = hc_pq_tiles(MT);
    for (int p2 = 0; p2 < g0.n_wparts; ++p2) {
      if (!(wf[p2] & 1u)) continue;
      for (int tile = 0; tile < ntiles; ++tile)
        for (int ri = 0; ri < 16; ++ri)
          for (int ci = 0; ci < 16; ++ci) {
            const int lane = 16 * (ri >> 2) + ci, reg = ri & 3;
            const unsigned w = pq[(size_t)p2 * pqw + (size_t)(tile * 2 + (reg >> 1)) * 64 + lane];
            const double val = (double)((w >> (16 * (reg & 1))) & 0xffffu);
            if (tile < MT * MT) {
              Pp[(size_t)((tile / MT) * 16 + ri) * Mp + (tile % MT) * 16 + ci] += val;
            } else {
              int t = tile - MT * MT, r = 0;
              while (t >= MT - r) {
                t -= MT - r;
                ++r;
              }
              const int cc = r + t;
              Qq[(size_t)(r * 16 + ri) * Mp + cc * 16 + ci] += val;
              if (cc != r) Qq[(size_t)(cc * 16 + ci) * Mp + r * 16 + ri] += val;
            }
          }
    }
  }
  if (g0.hc)
    for (int j = 0; j < M; ++j) {
      unsigned long long orb = 0ull;
      for (int p2 = 0; p2 < g0.n_wparts; ++p2) {
        const double* cc = cs.data() + (size_t)p2 * rows * Mp;
        cm[j] += cc[3 * Mp + j];
        unsigned long long b;
        std::memcpy(&b, &cc[4 * Mp + j], 8);
        orb |= b;
      }
      if (cm[j] > 0) std::memcpy(&mu[j], &orb, 8);
    }
  // weighted cooperative kernel (suffstat_hcx.hip.h): the gene's integer tables P = m'VH, Q = m'Vm, R = m'V[X | res]
  std::vector<long long> pqx;
  if (g0.hc && g0.pqw) {
    pqx.resize(hcx_pq_entries(Mp));
    HIP_TRY(c, hipMemcpy(pqx.data(), g0.pqw, sizeof(long long) * pqx.size(), hipMemcpyDeviceToHost));
  }
  auto R = [&](int i, int j) {
    if (j < M && j < i) std::swap(i, j);  // the engine uses the upper triangle of G'DG
    double s = 0;
    for (int p2 = 0; p2 < g0.n_wparts; ++p2) s += parts[(size_t)p2 * psz + (size_t)i * g0.Cp + j];
    if (!pqx.empty()) {  // (as gene_assemble combines them)
      if (j < M) {
        const double pij = (double)pqx[(size_t)i * Mp + j] * 0x1p-42, pji = (double)pqx[(size_t)j * Mp + i] * 0x1p-42;
        const double q = (double)pqx[(size_t)Mp * Mp + (size_t)i * Mp + j] * 0x1p-42;  // (i <= j here)
        s += mu[j] * pji + mu[i] * pij + (mu[i] * mu[j]) * q;
      } else if (j - M <= d) {
        s += mu[i] * ((double)pqx[2 * (size_t)Mp * Mp + (size_t)i * kHcxNullCols + (j - M)] * c->hcx_tile.scale[j - M]);
      }
    }
    if (j < M && !Pp.empty()) {
      const double q = Qq[(size_t)i * Mp + j];
      const double pij = Pp[(size_t)i * Mp + j] - 4.0 * q, pji = Pp[(size_t)j * Mp + i] - 4.0 * q;
      const double hh = s - 4.0 * (pij + pji) - 16.0 * q;
      s = hh + mu[j] * pij + mu[i] * pji + (mu[i] * mu[j]) * q;
    }
    if (j < M && (g0.hc == 2 || g0.hc == 4)) s /= g0.lat_den * g0.lat_den;  // lattice dosages: the integer K'K, divided once (gene_assemble)
    return s;
  };
  for (int i = 0; i < M; ++i) {
    if (S)
      for (int j = 0; j < M; ++j) S[(size_t)i * M + j] = R(i, j);
    if (T)
      for (int k = 0; k < d; ++k) T[(size_t)i * d + k] = R(i, M + k);
    if (u) u[i] = R(i, M + d);
    double s = 0, mn = INFINITY, mx = -INFINITY;
    for (int p2 = 0; p2 < g0.n_wparts; ++p2) {
      const double* cc = cs.data() + (size_t)p2 * rows * g0.Mp;
      s += cc[i];
      mn = std::min(mn, cc[g0.Mp + i]);
      mx = std::max(mx, cc[2 * g0.Mp + i]);
    }
    if (g0.hc && cm[i] > 0) {
      s += cm[i] * mu[i];
      mn = std::min(mn, mu[i]);
      mx = std::max(mx, mu[i]);
    }
    if (g0.hc == 2 || g0.hc == 4) s /= g0.lat_den;
    if (colsum) colsum[i] = s;
    if (cmin) cmin[i] = mn;
    if (cmax) cmax[i] = mx;
  }
  return RVT_OK;
}

// ---- unrelated null models on the device -----------------------------------------------------------------------
int rvt_fit_null(rvt_ctx* c, int trait, int64_t N, int d, const double* X, const double* y, double* beta_out,
                 double* sigma2_out) {
  if (!c || !X || !y || N < 1 || d < 1 || d > RVT_MAX_COV) return fail(c, RVT_E_INVALID, "bad null model");
  hipSetDevice(c->device);
  int rc = rvt_sync(c);
  if (rc) return rc;
  hipStream_t st = c->stream;
  const bool binary = trait == RVT_TRAIT_BINARY;
  double *d_xy = nullptr, *d_part = nullptr, *d_beta = nullptr, *d_a = nullptr, *d_b = nullptr, *d_zero = nullptr;
  struct Guard {
    double **a, **b, **c2, **d2, **e, **f;
    ~Guard() {
      for (double** p : {a, b, c2, d2, e, f})
        if (*p) hipFree(*p);
    }
  } guard{&d_xy, &d_part, &d_beta, &d_a, &d_b, &d_zero};
  const int rec = lmm_rec_len(d);
  HIP_TRY(c, hipMalloc((void**)&d_xy, sizeof(double) * (size_t)N * (d + 1)));  // X | y
  HIP_TRY(c, hipMalloc((void**)&d_part, sizeof(double) * (size_t)kLmmBlocks * rec));
  HIP_TRY(c, hipMalloc((void**)&d_beta, sizeof(double) * RVT_MAX_COV));
  HIP_TRY(c, hipMalloc((void**)&d_a, sizeof(double) * (size_t)N));
  HIP_TRY(c, hipMalloc((void**)&d_b, sizeof(double) * (size_t)N));
  HIP_TRY(c, hipMalloc((void**)&d_zero, sizeof(double) * (size_t)N));
  HIP_TRY(c, hipMemcpyAsync(d_xy, X, sizeof(double) * (size_t)N * d, hipMemcpyHostToDevice, st));
  HIP_TRY(c, hipMemcpyAsync(d_xy + (size_t)N * d, y, sizeof(double) * (size_t)N, hipMemcpyHostToDevice, st));
  HIP_TRY(c, hipMemsetAsync(d_zero, 0, sizeof(double) * (size_t)N, st));
  std::vector<double> part((size_t)kLmmBlocks * rec), sums(rec), beta(RVT_MAX_COV, 0.0);
  auto reduce = [&](int n_rec) {
    for (int q = 0; q < n_rec; ++q) {
      double s = 0.0;
      for (int b = 0; b < kLmmBlocks; ++b) s += part[(size_t)b * n_rec + q];
      sums[q] = s;
    }
  };
  std::vector<double> res(N), v(N);
  double sigma2 = 1.0;
  if (!binary) {
    // X'X, X'y with the weighted-sums kernel at unit weights (lambda = 0, delta = 1)
    k_lmm_sums(dim3(kLmmBlocks), st, d_xy, d_zero, (long long)N, d, 1.0, 0, d_part);
    HIP_TRY(c, hipMemcpyAsync(part.data(), d_part, sizeof(double) * part.size(), hipMemcpyDeviceToHost, st));
    HIP_TRY(c, sync_stream(st));
    reduce(rec);
    double Ai[RVT_MAX_COV * RVT_MAX_COV];
    if (!invert_spd(sums.data(), d, Ai)) return fail(c, RVT_E_INVALID, "X'X is singular");
    for (int a = 0; a < d; ++a) {
      double t = 0.0;
      for (int k = 0; k < d; ++k) t += Ai[a * d + k] * sums[d * d + k];
      beta[a] = t;
    }
    HIP_TRY(c, hipMemcpyAsync(d_beta, beta.data(), sizeof(double) * RVT_MAX_COV, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(linear_residual_kernel, dim3(kLmmBlocks), dim3(256), 0, st, d_xy, d_xy + (size_t)N * d, d_beta,
                       (long long)N, (long long)N, d, d_a, d_part);
    HIP_TRY(c, hipMemcpyAsync(part.data(), d_part, sizeof(double) * kLmmBlocks, hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipMemcpyAsync(res.data(), d_a, sizeof(double) * (size_t)N, hipMemcpyDeviceToHost, st));
    HIP_TRY(c, sync_stream(st));
    double rss = 0.0;
    for (int b = 0; b < kLmmBlocks; ++b) rss += part[b];
    sigma2 = rss / (double)N;
    std::fill(v.begin(), v.end(), sigma2);
  } else {
    int rounds = 0;
    double lastDev = -99999, curDev = -9999;
    const int nrrounds = 100;
    while (rounds < nrrounds) {
      HIP_TRY(c, hipMemcpyAsync(d_beta, beta.data(), sizeof(double) * RVT_MAX_COV, hipMemcpyHostToDevice, st));
      hipLaunchKernelGGL(logistic_round_kernel, dim3(kLmmBlocks), dim3(256), sizeof(double) * 256, st, d_xy,
                         d_xy + (size_t)N * d, d_beta, (long long)N, (long long)N, d, d_a, d_b, d_part);
      HIP_TRY(c, hipMemcpyAsync(part.data(), d_part, sizeof(double) * part.size(), hipMemcpyDeviceToHost, st));
      HIP_TRY(c, sync_stream(st));
      reduce(rec);
      double Di[RVT_MAX_COV * RVT_MAX_COV];
      if (!invert_spd(sums.data(), d, Di)) return fail(c, RVT_E_INVALID, "X'VX is singular");
      for (int a = 0; a < d; ++a) {
        double t = 0.0;
        for (int k = 0; k < d; ++k) t += Di[a * d + k] * sums[d * d + k];
        beta[a] += t;
      }
      curDev = -2.0 * sums[d * d + d];  // GetDeviance() on the stored p (LogisticRegression.cpp:75-94)
      if (rounds > 1 && std::fabs(curDev - lastDev) < 1e-3) {
        rounds = 0;
        break;
      }
      if (std::fpclassify(curDev) != FP_NORMAL) return fail(c, RVT_E_INVALID, "logistic deviance is not normal");
      lastDev = curDev;
      ++rounds;
    }
    if (rounds == nrrounds) return fail(c, RVT_E_INVALID, "logistic model did not converge in 100 rounds");
    std::vector<double> p(N);
    HIP_TRY(c, hipMemcpy(p.data(), d_a, sizeof(double) * (size_t)N, hipMemcpyDeviceToHost));
    HIP_TRY(c, hipMemcpy(v.data(), d_b, sizeof(double) * (size_t)N, hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < N; ++i) res[i] = y[i] - p[i];
  }
  if (beta_out)
    for (int a = 0; a < d; ++a) beta_out[a] = beta[a];
  if (sigma2_out) *sigma2_out = sigma2;
  const int rcs = rvt_set_null(c, trait, N, d, X, res.data(), v.data(), sigma2);
  if (rcs == RVT_OK) {  // kept for rvt_null_summary (rvt_set_null alone leaves the estimates unknown)
    for (int a = 0; a < d; ++a) c->null_beta[a] = beta[a];
    c->have_null_beta = true;
  }
  return rcs;
}

}  // extern "C"

// ---- launchers of this unit's kernels for the other units (rvt_engine_int.h, "kernel families") -----------------------------
void k_vt_integrate(dim3 grid, hipStream_t st, const GeneDesc* genes, int stage) {
  hipLaunchKernelGGL(vt_integrate_kernel, grid, dim3(256), 0, st, genes, stage);
}
void k_vt_finish(dim3 grid, hipStream_t st, const GeneDesc* genes, int n, int stage) {
  hipLaunchKernelGGL(vt_finish_kernel, grid, dim3(256), 0, st, genes, n, stage);
}
