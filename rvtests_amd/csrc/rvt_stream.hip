// rvtests_amd — the streaming interface (rvt_submit_gene*, rvt_collect*) and the genotype front ends behind it: packed
// hand-offs (raw / int8 / PLINK 2-bit), VCF record text, BGEN probability blocks — all decoded on the device.  Part of
// librvtests_amd.so; the batch pipeline it feeds is rvt_engine.hip's run_batch.
#define RVT_STREAM_UNIT 1
// this unit compiles (and ships) the STREAM kernel family only: see "kernel families" in rvt_engine_int.h
#define RVT_K_SPLIT
#define RVT_K_STREAM
#include "rvt_engine_int.h"

extern "C" {

// ---- streaming interface --------------------------------------------------------------------------------
namespace {
constexpr int kSubmitGroup = 32;  // genes per asynchronous sub-batch of the streaming interface

bool same_config(const rvt_ctx::Pending& a, const rvt_ctx::Pending& b) {
  return a.tests == b.tests && std::memcmp(&a.prm, &b.prm, sizeof(rvt_params)) == 0;
}

// wait for the consolidation of every submitted gene (blocks written, allele frequencies in the pinned ring) and move
// the frequencies into their queue entries
int resolve_af(rvt_ctx* c) {
  if (c->af_unresolved == 0) return RVT_OK;
  HIP_TRY(c, sync_stream(c->io_stream));
  for (auto& p : c->queue) {
    if (p.af_slot >= 0) {
      const double* h = c->h_af_ring + (size_t)p.af_slot * RVT_MAX_VARIANTS;
      p.af.assign(h, h + p.M);
      if (p.decoded && c->h_io_err && c->h_io_err[p.af_slot]) {  // malformed VCF text / BGEN block of THIS gene
        p.io_error = c->h_io_err[p.af_slot];
        c->h_io_err[p.af_slot] = 0;
      }
      p.af_slot = -1;
    }
  }
  c->af_unresolved = 0;
  return RVT_OK;
}

// launch queue[first, first+n) as one asynchronous batch (analytic tests only)
int launch_group(rvt_ctx* c, size_t first, int n) {
  {
    int rc = resolve_af(c);
    if (rc) return rc;
  }
  std::vector<const double*> ptrs;
  std::vector<int> Ms;
  std::vector<double> af;
  std::vector<int64_t> ids;
  std::vector<signed char> kinds;
  for (int g = 0; g < n; ++g) {
    const rvt_ctx::Pending& p = c->queue[first + g];
    ptrs.push_back(p.dG);
    Ms.push_back(p.M);
    ids.push_back(p.id);
    af.insert(af.end(), p.af.begin(), p.af.end());
    kinds.push_back((signed char)(p.kind == 3 && p.planes ? (3 | 0x10) : p.kind));
  }
  c->launched.emplace_back();
  rvt_ctx::Launched& L = c->launched.back();
  L.first = first;
  L.n = n;
  L.res.resize(n);
  const rvt_ctx::Pending& p0 = c->queue[first];
  // the blocks of these genes may still be crossing the link (staged copies on io_stream): the batch waits for them
  HIP_TRY(c, hipEventRecord(c->ev_io, c->io_stream));
  c->io_wait_pending = true;
  c->next_done_flag = &L.done;
  int rc = run_batch(c, n, ptrs.data(), Ms.data(), af.data(), ids.data(), p0.tests, &p0.prm, L.res.data(), nullptr, nullptr,
                     kinds.data());
  c->next_done_flag = nullptr;
  if (rc) {
    c->launched.pop_back();
    return rc;
  }
  for (int g = 0; g < n; ++g) c->queue[first + g].launched = true;
  return RVT_OK;
}

// launch every still-unlaunched run of equally configured genes among the first `upto` queue entries;
// only_full: launch only complete groups of kSubmitGroup (the submit path), else everything (the collect path)
int launch_pending(rvt_ctx* c, size_t upto, bool only_full) {
  size_t i = 0;
  while (i < upto) {
    if (c->queue[i].launched) {
      ++i;
      continue;
    }
    size_t e = i + 1;
    while (e < upto && !c->queue[e].launched && same_config(c->queue[e], c->queue[i]) &&
           (int)(e - i) < (only_full ? c->submit_group : 256))
      ++e;
    const rvt_ctx::Pending& p0 = c->queue[i];
    const bool perm = p0.prm.skat_nperm > 0 && (p0.tests & RVT_TEST_SKAT);
    if (only_full && ((int)(e - i) < c->submit_group || perm)) return RVT_OK;  // wait for more genes / for collect
    if (perm) {
      // permutation p-values consume one random stream in gene order: synchronous, gene by gene
      int rc = rvt_sync(c);
      if (rc) return rc;
      rc = resolve_af(c);
      if (rc) return rc;
      std::vector<const double*> ptrs;
      std::vector<int> Ms;
      std::vector<double> af;
      std::vector<int64_t> ids;
      for (size_t g = i; g < e; ++g) {
        ptrs.push_back(c->queue[g].dG);
        Ms.push_back(c->queue[g].M);
        ids.push_back(c->queue[g].id);
        af.insert(af.end(), c->queue[g].af.begin(), c->queue[g].af.end());
      }
      std::vector<rvt_gene_result> res(e - i);
      HIP_TRY(c, sync_stream(c->io_stream));  // (the blocks are complete before the synchronous permutation path)
      rc = run_blocks_with_perm(c, (int)(e - i), ptrs.data(), Ms.data(), af.data(), ids.data(), p0.tests, &p0.prm,
                                res.data());
      if (rc) return rc;
      for (size_t g = i; g < e; ++g) {
        c->queue[g].res = res[g - i];
        c->queue[g].launched = true;
      }
    } else {
      int rc = launch_group(c, i, (int)(e - i));
      if (rc) return rc;
    }
    i = e;
  }
  return RVT_OK;
}
}  // namespace

namespace {
int io_err_ready(rvt_ctx* c) {
  if (c->h_io_err) return RVT_OK;
  HIP_TRY(c, hipHostMalloc((void**)&c->h_io_err, sizeof(int) * (rvt_ctx::kAfSlots + 1), hipHostMallocMapped));
  std::memset(c->h_io_err, 0, sizeof(int) * (rvt_ctx::kAfSlots + 1));
  return RVT_OK;
}
void io_err_message(rvt_ctx* c, int k, bool bgen, const char* whose) {
  if (bgen)
    fail(c, RVT_E_INVALID, "BGEN variant %d%s: the block is shorter than its ploidy bytes demand (or a ploidy exceeds the "
         "declared maximum)", k - 1, whose);
  else if (k < 0)
    fail(c, RVT_E_INVALID, "VCF record %d%s holds a dosage the device cannot round exactly (more than 15 digits, "
         "|exponent| > 22, inf / nan / hex)", -k - 1, whose);
  else
    fail(c, RVT_E_INVALID, "VCF record %d%s does not hold %d sample columns", k - 1, whose, c->vcf_n_file);
}

// ---- the text / block buffer of a gene (VCF text, BGEN blocks): a ring of device buffers fed through the copy stream -------
// The copies of gene g + 1 cross the link while the decode kernels of gene g read their own buffer (one buffer on one
// stream serialised them).  text_acquire: next buffer, grown to `total` bytes; the copy stream waits for the kernels that
// read it last.  text_copied: the kernels on `st` wait for the copies.  text_release: the last reader has been enqueued.
static int text_acquire(rvt_ctx* c, size_t total) {
  const int k = c->text_next;
  c->text_next = (k + 1) % rvt_ctx::kTextBufs;
  if (!c->ev_text_free[k]) {
    HIP_TRY(c, hipEventCreateWithFlags(&c->ev_text_free[k], hipEventDisableTiming));
    HIP_TRY(c, hipEventCreateWithFlags(&c->ev_text_copied[k], hipEventDisableTiming));
  }
  if (c->text_buf_cap[k] < total) {
    if (c->text_buf[k]) hipFree(c->text_buf[k]);
    c->text_buf[k] = nullptr;
    c->text_buf_cap[k] = 0;
    HIP_TRY(c, hipMalloc((void**)&c->text_buf[k], total + total / 4));
    c->text_buf_cap[k] = total + total / 4;
  }
  HIP_TRY(c, hipStreamWaitEvent(c->copy_stream, c->ev_text_free[k], 0));
  c->d_vcf_text = c->text_buf[k];
  c->text_cur = k;
  return RVT_OK;
}
static int text_copied(rvt_ctx* c, hipStream_t st) {
  HIP_TRY(c, hipEventRecord(c->ev_text_copied[c->text_cur], c->copy_stream));
  HIP_TRY(c, hipStreamWaitEvent(st, c->ev_text_copied[c->text_cur], 0));
  return RVT_OK;
}
static int text_release(rvt_ctx* c, hipStream_t st) {
  HIP_TRY(c, hipEventRecord(c->ev_text_free[c->text_cur], st));
  return RVT_OK;
}
// the pieces of a gene (records / blocks) from the caller's memory into the acquired buffer, on the copy stream: gathered
// through the pinned ring (one pool batch and one DMA per 32 MB, gaps and `tail_zero` bytes behind every piece zeroed), or
// piece by piece when the staging ring is off or a piece lies in registered memory
static int text_upload(rvt_ctx* c, const std::vector<StageRing::Piece>& pieces, size_t tail_zero) {
  TraceScope ts(c, &c->tr_copy);
  bool gather = c->stage_on;
  for (const auto& pc : pieces)
    if (host_registered(c, pc.src, pc.bytes)) gather = false;
  c->h2d_stream = c->copy_stream;
  int rc = RVT_OK;
  if (gather) {
    rc = stage_ready(c);
    if (!rc && c->stage.copy_gather(c->d_vcf_text, pieces.data(), pieces.size(), tail_zero, CopyPool::instance()))
      rc = fail(c, RVT_E_HIP, "staged host-to-device copy failed");
  } else {
    for (const auto& pc : pieces) {
      if (tail_zero && hipMemsetAsync(c->d_vcf_text + pc.dst_off + pc.bytes, 0, tail_zero, c->copy_stream) != hipSuccess)
        rc = fail(c, RVT_E_HIP, "hipMemsetAsync failed");
      if (!rc) rc = staged_h2d(c, c->d_vcf_text + pc.dst_off, pc.src, pc.bytes);
      if (rc) break;
    }
  }
  c->h2d_stream = c->io_stream;
  return rc;
}

// mode 0: imputed doubles + caller's af; 1: raw doubles (consolidated on the device); 2: packed int8 (ditto);
// 3: PLINK 2-bit codes (ditto)
// VCF text of one gene -> N x M signed bytes in c->d_consol_i8 (vcf_kernels.hip.h), on stream st
struct VcfGene {
  const char* const* text;  // per record: first byte of the first sample column
  const int64_t* len;       // per record: bytes up to (not including) the end of line
  const int* gt_idx;        // per record: FORMAT index of GT (-1: absent -> every genotype missing)
  const int* gd_idx;        // may be NULL (-1)
  const int* gq_idx;        // may be NULL (-1)
};
int vcf_decode_gene(rvt_ctx* c, const VcfGene* vg, int M, int64_t N, hipStream_t st, int err_slot, double* dosage_out = nullptr,
                    int64_t dosage_ld = 0) {
  std::vector<VcfRecord> rec(M);
  size_t total = 0;
  int64_t max_len = 0;
  for (int j = 0; j < M; ++j) {
    rec[j].text_off = (long long)total;
    rec[j].len = vg->len[j];
    rec[j].gt_idx = vg->gt_idx[j];
    rec[j].gd_idx = vg->gd_idx ? vg->gd_idx[j] : -1;
    rec[j].gq_idx = vg->gq_idx ? vg->gq_idx[j] : -1;
    rec[j].alt = ((int)c->vcf_alt.size() == M) ? c->vcf_alt[j] : 0;
    rec[j].hemi = ((int)c->vcf_hemi.size() == M && c->d_vcf_sex) ? c->vcf_hemi[j] : 0;
    total += ((size_t)vg->len[j] + 31) / 16 * 16;  // 16-byte aligned starts, >= 16 readable bytes behind the end
    max_len = std::max<int64_t>(max_len, vg->len[j]);
  }
  if (int rca = text_acquire(c, total)) return rca;
  if (!c->d_vcf_rec) HIP_TRY(c, hipMalloc((void**)&c->d_vcf_rec, sizeof(VcfRecord) * RVT_MAX_VARIANTS));
  const int max_seg = (int)std::max<int64_t>(1, (max_len + kVcfSegBytes - 1) / kVcfSegBytes);
  if (c->vcf_seg_cap < (size_t)max_seg * M) {
    if (c->d_vcf_seg) hipFree(c->d_vcf_seg);
    c->d_vcf_seg = nullptr;
    c->vcf_seg_cap = 0;
    const size_t want = (size_t)max_seg * std::max(M, 64);
    HIP_TRY(c, hipMalloc((void**)&c->d_vcf_seg, sizeof(int) * want));
    c->vcf_seg_cap = want;
  }
  {
    int rce = io_err_ready(c);
    if (rce) return rce;
  }
  c->vcf_alt.clear();  // (one call only)
  c->vcf_hemi.clear();
  {
    std::vector<StageRing::Piece> pieces;
    for (int j = 0; j < M; ++j)
      if (vg->len[j] > 0) pieces.push_back(StageRing::Piece{(size_t)rec[j].text_off, vg->text[j], (size_t)vg->len[j]});
    if (int rcs = text_upload(c, pieces, 0)) return rcs;
    if (int rcs = text_copied(c, st)) return rcs;
  }
  if (int rcs = small_h2d(c, c->d_vcf_rec, rec.data(), sizeof(VcfRecord) * M)) return rcs;  // (`rec` is a local)
  int* d_err = nullptr;
  HIP_TRY(c, hipHostGetDevicePointer((void**)&d_err, c->h_io_err, 0));
  d_err += err_slot;
  const dim3 grid((unsigned)max_seg, (unsigned)M);
  hipLaunchKernelGGL(vcf_tab_count_kernel, grid, dim3(256), 0, st, c->d_vcf_text, c->d_vcf_rec, max_seg, c->d_vcf_seg);
  hipLaunchKernelGGL(vcf_tab_scan_kernel, dim3((unsigned)M), dim3(256), 0, st, c->d_vcf_rec, max_seg, c->vcf_n_file,
                     c->d_vcf_seg, d_err);
  if (dosage_out) {  // --dosage TAG: doubles straight into the gene's block (rows the map never addresses stay missing)
    hipLaunchKernelGGL(vcf_fill_kernel, dim3(1024), dim3(256), 0, st, dosage_out, (long long)N, (long long)dosage_ld, M,
                       (double)kVcfMissing);
    hipLaunchKernelGGL(vcf_decode_dosage_kernel, grid, dim3(256), 0, st, c->d_vcf_text, c->d_vcf_rec, max_seg,
                       c->d_vcf_seg, c->d_vcf_rows, c->d_vcf_sex, c->vcf_n_file, (long long)dosage_ld, c->vcf_flt, dosage_out,
                       d_err);
  } else {
    signed char* out = (signed char*)c->d_consol_i8;
    HIP_TRY(c, hipMemsetAsync(out, 0xF7, (size_t)N * M, st));  // -9: rows the sample map never addresses stay missing
    hipLaunchKernelGGL(vcf_decode_kernel, grid, dim3(256), 0, st, c->d_vcf_text, c->d_vcf_rec, max_seg, c->d_vcf_seg,
                       c->d_vcf_rows, c->d_vcf_sex, c->vcf_n_file, (long long)N, c->vcf_flt, out);
  }
  HIP_TRY(c, hipGetLastError());
  return text_release(c, st);
}

// BGEN probability blocks of one gene -> raw genotype doubles (missing = -9) in out (N rows x M, leading dimension ld)
struct BgenGene {
  const unsigned char* const* block;  // per variant: the uncompressed probability block
  const int64_t* len;
  int layout;                         // 1 (v1.1) or 2 (v1.2 / v1.3)
};
int bgen_decode_gene(rvt_ctx* c, const BgenGene* bg, int M, int64_t N, hipStream_t st, int err_slot, double* out,
                     int64_t ld) {
  const int64_t n_file = c->d_vcf_rows ? c->vcf_n_file : N;
  std::vector<BgenRecord> rec(M);
  size_t total = 0;
  for (int j = 0; j < M; ++j) {
    const unsigned char* b = bg->block[j];
    const int64_t len = bg->len[j];
    BgenRecord& r = rec[j];
    r.layout = bg->layout;
    r.len = len;
    r.alt = ((int)c->vcf_alt.size() == M) ? c->vcf_alt[j] : 0;
    size_t pad = 0;
    if (bg->layout == 1) {
      if (len < 6 * n_file) return fail(c, RVT_E_INVALID, "BGEN variant %d: block of %lld bytes, %lld samples", j, (long long)len, (long long)n_file);
      r.K = 2;
      r.phased = 0;
      r.bits = 16;
      r.scale = 0.0f;
      r.zmax = 2;
    } else {
      if (len < 10 + n_file) return fail(c, RVT_E_INVALID, "BGEN variant %d: block of %lld bytes, %lld samples", j, (long long)len, (long long)n_file);
      uint32_t n_indv;
      uint16_t K;
      std::memcpy(&n_indv, b, 4);
      std::memcpy(&K, b + 4, 2);
      if ((int64_t)n_indv != n_file)
        return fail(c, RVT_E_INVALID, "BGEN variant %d holds %u samples, the sample map %lld", j, n_indv, (long long)n_file);
      const int B = b[8 + n_file + 1];
      if (B < 1 || B > 32 || K < 1) return fail(c, RVT_E_INVALID, "BGEN variant %d: %d bits, %d alleles", j, B, (int)K);
      r.K = K;
      r.phased = b[8 + n_file] != 0;
      r.bits = B;
      float scale = 1.0f;  // BitReader's constructor, in float (libBgen/BitReader.h:17-27)
      for (int i = 0; i < B; ++i) scale *= 2;
      scale -= 1;
      scale = (float)(1.0 / scale);
      r.scale = scale;
      const int zmax = b[7] & 0x3f;  // declared maximum ploidy; the device checks every sample against it
      r.zmax = zmax;
      if (!r.phased) {  // C(Z + K - 1, K - 1) must stay an ordinary int (the reference's choose() overflows silently beyond)
        double cmb = 1.0;
        for (int i = 0; i < K - 1 && cmb < 1e9; ++i) cmb = cmb * (zmax + K - 1 - i) / (i + 1);
        if (cmb > 16777216.0)
          return fail(c, RVT_E_TOO_LARGE, "BGEN variant %d: ploidy %d with %d alleles is not supported", j, zmax, (int)K);
      }
      // The packed values every sample's ploidy byte demands must lie inside the block: the decode kernels index them from
      // those bytes alone (as the reference's BitReader would read on, libBgen/BitReader.h), so a truncated or corrupt
      // block is refused HERE, before anything is enqueued (one pass over the N ploidy bytes of the variant).
      {
        auto choose = [](int n, int m) {  // BGenFile::choose (libBgen/BGenFile.cpp:438-453), int arithmetic
          if (m == 1) return n;
          if (n == 1) return 1;
          int ret = 1;
          for (int i = 0; i < m; ++i) ret *= (n - i);
          for (int i = 0; i < m; ++i) ret /= (i + 1);
          return ret;
        };
        int per_z[64];
        for (int z = 0; z < 64; ++z) per_z[z] = r.phased ? z * ((int)K - 1) : (z == 0 ? 0 : choose(z + (int)K - 1, (int)K - 1) - 1);
        unsigned long long values = 0;
        int worst = 0;
        const unsigned char* pm = b + 8;
        // (nearly every sample has the same ploidy byte: eight at a time while they equal the first one)
        unsigned long long pat;
        std::memset(&pat, pm[0], 8);
        const int z0 = pm[0] & 0x3f;
        int64_t i = 0;
        for (; i + 8 <= n_file; i += 8) {
          unsigned long long w;
          std::memcpy(&w, pm + i, 8);
          if (w == pat) {
            values += 8ull * (unsigned long long)per_z[z0];
          } else {
            for (int t = 0; t < 8; ++t) {
              const int z = pm[i + t] & 0x3f;
              values += (unsigned long long)per_z[z];
              worst = z > worst ? z : worst;
            }
          }
        }
        worst = z0 > worst ? z0 : worst;
        for (; i < n_file; ++i) {
          const int z = pm[i] & 0x3f;
          values += (unsigned long long)per_z[z];
          worst = z > worst ? z : worst;
        }
        const unsigned long long have_bits = (unsigned long long)(len - (10 + n_file)) * 8ull;
        if (worst > zmax || values * (unsigned long long)B > have_bits)
          return fail(c, RVT_E_INVALID, "BGEN variant %d: the block is shorter than its ploidy bytes demand (or a ploidy "
                      "exceeds the declared maximum)", j);
      }
      pad = (size_t)((4 - (10 + n_file) % 4) % 4);  // the packed values start on a 4-byte boundary
    }
    total = (total + 15) / 16 * 16 + pad;
    r.off = (long long)total;
    total += (size_t)len + 16;
  }
  total += 16;
  if (int rca = text_acquire(c, total)) return rca;
  c->vcf_alt.clear();  // (one call only)
  if (!c->d_bgen_rec) HIP_TRY(c, hipMalloc((void**)&c->d_bgen_rec, sizeof(BgenRecord) * RVT_MAX_VARIANTS));
  const int max_seg = (int)((n_file + kBgenSeg - 1) / kBgenSeg);
  if (c->bgen_seg_cap < (size_t)max_seg * M) {
    if (c->d_bgen_seg) hipFree(c->d_bgen_seg);
    c->d_bgen_seg = nullptr;
    c->bgen_seg_cap = 0;
    const size_t want = (size_t)max_seg * std::max(M, 64);
    HIP_TRY(c, hipMalloc((void**)&c->d_bgen_seg, sizeof(long long) * want));
    c->bgen_seg_cap = want;
  }
  {
    int rce = io_err_ready(c);
    if (rce) return rce;
  }
  {  // the bytes behind a block read as zero (BitReader stops at its end)
    std::vector<StageRing::Piece> pieces;
    for (int j = 0; j < M; ++j) pieces.push_back(StageRing::Piece{(size_t)rec[j].off, bg->block[j], (size_t)bg->len[j]});
    if (int rcs = text_upload(c, pieces, 16)) return rcs;
    if (int rcs = text_copied(c, st)) return rcs;
  }
  if (int rcs = small_h2d(c, c->d_bgen_rec, rec.data(), sizeof(BgenRecord) * M)) return rcs;  // (`rec` is a local)
  int* d_err = nullptr;
  HIP_TRY(c, hipHostGetDevicePointer((void**)&d_err, c->h_io_err, 0));
  d_err += err_slot;
  const unsigned char* data = reinterpret_cast<const unsigned char*>(c->d_vcf_text);
  const dim3 grid((unsigned)max_seg, (unsigned)M);
  if (bg->layout == 2) {
    hipLaunchKernelGGL(bgen_count_kernel, grid, dim3(kBgenSeg), 0, st, data, c->d_bgen_rec, (long long)n_file, max_seg,
                       c->d_bgen_seg, d_err);
    hipLaunchKernelGGL(bgen_scan_kernel, dim3((unsigned)M), dim3(256), 0, st, c->d_bgen_rec, (long long)n_file, max_seg,
                       c->d_bgen_seg, d_err);
  }
  hipLaunchKernelGGL(bgen_decode_kernel, grid, dim3(kBgenSeg), 0, st, data, c->d_bgen_rec, (long long)n_file, max_seg,
                     c->d_bgen_seg, c->d_vcf_rows, (long long)ld, out);
  HIP_TRY(c, hipGetLastError());
  return text_release(c, st);
}

// Header of a packed block (suffstat_hcp.hip.h) from the count pass over its 2-bit rows: per column the imputed value,
// whether the column is flipped (sum of the imputed column > N, DataConsolidator.cpp:46-69), polymorphic (min != max,
// DataConsolidator.cpp:94-116) and whether its imputed value counts in the burden collapse ((int)mu' > 0).  One thread per
// column; parts: consolidate_count_kernel<bed2_t>'s records (pad = the number of 2s), fill: consolidate_fill_kernel's.
__global__ __launch_bounds__(128) void hcp_header_kernel(const ConsolPart* __restrict__ parts, int nparts, int M, long long N,
                                                         const double* __restrict__ fill, HcpHeader* __restrict__ hdr) {
  const int j = threadIdx.x;
  bool flip = false, poly = false, cm = false;
  if (j < M) {
    const ConsolPart* p = parts + (long long)j * nparts;
    double ac = 0.0;
    long long nonneg = 0, n2 = 0;
    for (int k = 0; k < nparts; ++k) {
      ac += p[k].ac;
      nonneg += p[k].nonneg;
      n2 += p[k].pad;
    }
    const long long nm = N - nonneg, n1 = (long long)ac - 2 * n2, n0 = nonneg - n1 - n2;
    const double mu = nm > 0 ? fill[j] : 0.0;
    hdr->mu[j] = mu;
    const double s = ac + (double)nm * mu;
    flip = !(s <= (double)N);
    double mn = n0 > 0 ? 0.0 : (n1 > 0 ? 1.0 : (n2 > 0 ? 2.0 : INFINITY));
    double mx = n2 > 0 ? 2.0 : (n1 > 0 ? 1.0 : (n0 > 0 ? 0.0 : -INFINITY));
    if (nm > 0) {
      mn = fmin(mn, mu);
      mx = fmax(mx, mu);
    }
    poly = !(mn == mx);
    cm = nm > 0 && poly && (flip ? mu <= 1.0 : mu >= 1.0);
  }
  const unsigned long long bf = __ballot(flip), bp = __ballot(poly), bc = __ballot(cm);
  if ((j & 15) == 0 && j < 96) {
    const int b = j >> 4, sh = 16 * ((j >> 4) & 3);
    hdr->flip[b] = (unsigned short)((bf >> sh) & 0xffffu);
    hdr->poly[b] = (unsigned short)((bp >> sh) & 0xffffu);
    hdr->cm[b] = (unsigned short)((bc >> sh) & 0xffffu);
  }
}

// The genes rvt_submit_gene_bed may keep as 2-bit rows (gene_suffstat_hcp): the conditions under which run_batch takes the
// hard-call family, and nothing that needs the fp64 block itself (permutations, AnalyticVT).  RVT_PACKED=0: always expand.
static bool null_is_default(const rvt_ctx* c) {
  const int d = c->nc.d;
  const int64_t ld = c->nc.ld;
  return c->d_nulltile && c->d_X == c->d_nulltile && c->d_rr == c->d_nulltile + (size_t)ld * d &&
         c->d_zeros == c->d_nulltile + (size_t)ld * (d + 1);
}
static bool packed_eligible(const rvt_ctx* c, int M, uint32_t tests, const rvt_params* prm) {
  static const bool on = !(getenv("RVT_PACKED") && atoi(getenv("RVT_PACKED")) == 0);
  if (!on || !c->hc_enabled || c->nc.binary || c->nc.d > kHcMaxD || !null_is_default(c)) return false;
  if ((M + 15) / 16 > kHcMaxMT) return false;
  if (tests & (RVT_TEST_FAMSKAT | RVT_TEST_ANALYTICVT | RVT_TEST_FAMCMC | RVT_TEST_FAMZEGGINI)) return false;
  if ((tests & RVT_TEST_SKAT) && prm && prm->skat_nperm > 0) return false;
  return true;
}

// rvt_submit_gene's block of doubles packed on the way (host_stage.h, round 5): when the gene may take the packed kernel
// (packed_eligible) and the content hint does not say dosages, the staging threads turn every column into 2-bit codes
// + its one other value while they read it, 6 MB cross the link instead of 200, and the gene continues exactly as a
// rvt_submit_gene_bed gene whose imputed values are GIVEN (the doubles the caller's consolidate() wrote) instead of
// computed.  Returns 1 = submitted (rc_out holds the result), 0 = not applicable / not representable: the caller goes on with
// the plain copy.  RVT_PACK_FP64=0 switches it off.
int try_submit_packed_f64(rvt_ctx* c, int64_t gene_id, int M, const double* G, const double* af, uint32_t tests,
                          const rvt_params* prm, int* rc_out) {
  const char* sw = getenv("RVT_PACK_FP64");  // (read per call: tests switch it inside one process)
  const bool on = !(sw && atoi(sw) == 0);
  if (!on || !c->stage_on || c->content_hint == 0 || !packed_eligible(c, M, tests, prm)) return 0;
  // (a page-locked caller buffer is packed like any other: the threads read 200 MB of host memory at 200+ GB/s and 6 MB cross
  //  the link, where the DMA of the doubles themselves is bound by the link — 290 against 1 400 gene-sets/s; until round 6 a
  //  registered fp64 block always took the DMA.  It still does when the block cannot be packed: dosages, RVT_PACK_FP64=0.)
  const int64_t N = c->nc.N;
  if (N < 4096) return 0;  // small blocks: nothing to gain
  const size_t pk_pitch = ((size_t)((N + 3) / 4) + 15) / 16 * 16;
  if (stage_ready(c) != RVT_OK || pk_pitch > c->stage.chunk_bytes) return 0;
  const size_t need = (size_t)kHcpHeaderBytes + pk_pitch * M + 16;
  rvt_ctx::Pending p;
  p.id = gene_id;
  p.M = M;
  p.dG = nullptr;
  p.launched = false;
  std::memset(&p.res, 0, sizeof(p.res));
  bool fresh = false;
  int best = -1;
  auto& pool = c->pk_pool;
  for (int i = 0; i < (int)pool.size(); ++i)
    if (pool[i].first >= need && pool[i].first <= 4 * need && (best < 0 || pool[i].first < pool[best].first)) best = i;
  if (best >= 0) {
    p.dG = pool[best].second;
    p.bytes = pool[best].first;
    pool.erase(pool.begin() + best);
  } else {
    if (hipMalloc((void**)&p.dG, need) != hipSuccess) {
      (void)hipGetLastError();
      return 0;
    }
    p.bytes = need;
    fresh = true;
  }
  auto give_back = [&]() { c->pk_pool.emplace_back(p.bytes, p.dG); };
  hipStream_t st = c->io_stream;
  hipError_t e = hipSuccess;
  if (c->consol_af_cap < (size_t)M) {
    if (c->d_consol_af) hipFree(c->d_consol_af);
    c->d_consol_af = nullptr;
    c->consol_af_cap = 0;
    e = hipMalloc((void**)&c->d_consol_af, sizeof(double) * 2 * RVT_MAX_VARIANTS);
    if (e == hipSuccess) c->consol_af_cap = RVT_MAX_VARIANTS;
  }
  const int nparts = (int)((N + kConsolChunk - 1) / kConsolChunk);
  if (e == hipSuccess && c->consol_parts_cap < (size_t)M * nparts) {
    if (c->d_consol_parts) hipFree(c->d_consol_parts);
    c->d_consol_parts = nullptr;
    c->consol_parts_cap = 0;
    const size_t want = (size_t)std::max(M, 128) * nparts;
    e = hipMalloc((void**)&c->d_consol_parts, sizeof(ConsolPart) * want);
    if (e == hipSuccess) c->consol_parts_cap = want;
  }
  if (e != hipSuccess) {
    (void)hipGetLastError();
    give_back();
    return 0;
  }
  unsigned char* rows = reinterpret_cast<unsigned char*>(p.dG) + kHcpHeaderBytes;
  const int ek = c->pack_next;
  c->pack_next = (ek + 1) % rvt_ctx::kPack;
  if (fresh) e = hipMemsetAsync(p.dG, 0, need, c->copy_stream);
  std::vector<PackedColumn> cols((size_t)M);
  int prc = 1;
  if (e == hipSuccess) {
    TraceScope ts(c, &c->tr_copy);
    c->h2d_stream = c->copy_stream;
    prc = c->stage.pack_f64(rows, pk_pitch, G, (size_t)N, (size_t)N, (size_t)M, CopyPool::pack_instance(), cols.data());
    c->h2d_stream = c->io_stream;
  }
  if (prc == 2) {  // dosages, or a column with two other values: the block crosses as doubles
    give_back();
    return 0;
  }
  if (prc != 0 || e != hipSuccess) {
    give_back();
    *rc_out = fail(c, RVT_E_HIP, "packing the fp64 block failed");
    return 1;
  }
  // the other value of every column (0 where it has none) where the header kernel reads the imputed values
  double mu[RVT_MAX_VARIANTS > 96 ? 96 : RVT_MAX_VARIANTS];
  for (int j = 0; j < M; ++j) mu[j] = cols[j].has_mu ? cols[j].mu : 0.0;
  double* d_fill = c->d_consol_af + RVT_MAX_VARIANTS;
  e = hipEventRecord(c->ev_pack_copied[ek], c->copy_stream);
  if (e == hipSuccess) e = hipStreamWaitEvent(st, c->ev_pack_copied[ek], 0);
  if (e == hipSuccess && small_h2d(c, d_fill, mu, sizeof(double) * (size_t)M) != RVT_OK) e = hipErrorUnknown;
  if (e == hipSuccess) {
    const bed2_t* sb = reinterpret_cast<const bed2_t*>(rows);
    const dim3 cgrid((unsigned)nparts, (unsigned)M);
    hipLaunchKernelGGL((consolidate_count_kernel<bed2_t>), cgrid, dim3(256), 0, st, sb, (long long)pk_pitch, (long long)N,
                       c->d_consol_parts);
    hipLaunchKernelGGL(hcp_header_kernel, dim3(1), dim3(128), 0, st, c->d_consol_parts, nparts, M, (long long)N, d_fill,
                       reinterpret_cast<HcpHeader*>(p.dG));
    e = hipGetLastError();
  }
  if (e != hipSuccess) {
    give_back();
    *rc_out = fail(c, RVT_E_HIP, "packed fp64 gene: %s", hipGetErrorString(e));
    return 1;
  }
  p.kind = 3;
  p.decoded = 0;
  p.af.assign(af, af + M);
  p.tests = tests;
  p.prm = prm ? *prm : rvt_params{1.0, 25.0, 1.0, 25.0, 0, 0.05};
  c->queue.push_back(std::move(p));
  if (c->trace_submit) ++c->tr_genes;
  TraceScope ts_l(c, &c->tr_launch);
  *rc_out = launch_pending(c, c->queue.size(), true);
  return 1;
}

int submit_common(rvt_ctx* c, int64_t gene_id, int M, const void* G, int mode, const double* af, double* af_out,
                  uint32_t tests, const rvt_params* prm) {
  RegWait reg_wait_on_return(c);
  if (!c || !G || M < 1 || (mode == 0 && !af)) return fail(c, RVT_E_INVALID, "bad gene");
  if (!c->have_null) return fail(c, RVT_E_STATE, "no null model set");
  if (tests & RVT_TEST_FAMSKAT) return fail(c, RVT_E_INVALID, "FamSKAT runs through rvt_run_fam_blocks");
  if (M > RVT_MAX_VARIANTS) return fail(c, RVT_E_TOO_LARGE, "gene of %d variants exceeds RVT_MAX_VARIANTS", M);
  hipSetDevice(c->device);
  TraceScope ts_all(c, &c->tr_block);  // (the whole call; "block" in the trace line = total per gene)
  if (mode == 0) {  // the fp64 boundary: packed on the way when the block allows it
    int rcp = RVT_OK;
    if (try_submit_packed_f64(c, gene_id, M, (const double*)G, af, tests, prm, &rcp)) return rcp;
  }
  if (c->trace_submit) ++c->tr_genes;
  rvt_ctx::Pending p;
  p.id = gene_id;
  p.M = M;
  p.dG = nullptr;
  p.launched = false;
  std::memset(&p.res, 0, sizeof(p.res));
  // device block from the pool (smallest that fits) or a fresh zeroed allocation; pad rows stay zero because only
  // the N data rows of a column are ever written
  // PLINK 2-bit rows stay packed when the gene's tests allow it (gene_suffstat_hcp): a block of header + M padded rows
  // (7: the rows are on the device already; 2: int8 hard calls, turned into 2-bit rows by the staging threads on the way —
  //  RVT_PACK_I8=0 sends the bytes)
  const bool i8_packs = mode == 2 && !c->no_i8_pack && c->stage_on && c->nc.N >= 4096 &&
                        !(getenv("RVT_PACK_I8") && atoi(getenv("RVT_PACK_I8")) == 0);
  const bool packed = (mode == 3 || mode == 7 || i8_packs) && packed_eligible(c, M, tests, prm);
  const size_t pk_pitch = ((size_t)((c->nc.N + 3) / 4) + 15) / 16 * 16;
  const size_t need = packed ? (size_t)kHcpHeaderBytes + pk_pitch * M + 16 : sizeof(double) * (size_t)c->null_ld * M;
  bool fresh_packed = false;
  int best = -1;
  auto& pool = packed ? c->pk_pool : c->block_pool;
  for (int i = 0; i < (int)pool.size(); ++i)
    if (pool[i].first >= need && (!packed || pool[i].first <= 4 * need) && (best < 0 || pool[i].first < pool[best].first))
      best = i;
  if (best >= 0) {
    p.dG = pool[best].second;
    p.bytes = pool[best].first;
    pool.erase(pool.begin() + best);
  } else if (packed) {
    if (hipMalloc((void**)&p.dG, need) != hipSuccess) {
      (void)hipGetLastError();
      return fail(c, RVT_E_HIP, "hipMalloc(%zu bytes) failed for a packed gene", need);
    }
    p.bytes = need;
    fresh_packed = true;
  } else {
    int rc = rvt_block_alloc(c, M, &p.dG);
    if (rc) return rc;
    p.bytes = need;
  }
  auto give_back = [&]() { (packed ? c->pk_pool : c->block_pool).emplace_back(p.bytes, p.dG); };
  const int64_t N = c->nc.N, ld = c->null_ld;
  // what this entry point writes into the block: hard calls with imputed means (packed / text genotypes), dosages
  // (dosage text, BGEN), or whatever the caller's doubles are
  p.kind = packed ? 3 : ((mode == 2 || mode == 3 || mode == 4 || mode == 7) ? 1 : (mode == 5 ? 2 : (mode == 6 ? 0 : -1)));
  p.planes = packed && mode == 7;
  p.decoded = (mode == 4 || mode == 5) ? 1 : (mode == 6 ? 2 : 0);
  if (mode == 0) {
    int rc = upload_block_data(c, p.dG, M, (const double*)G);  // synchronous copy: the caller may overwrite G on return
    if (rc) {
      give_back();
      return rc;
    }
    p.af.assign(af, af + M);
  } else {
    // DataConsolidator::consolidate's genotype part on the device: counter AF + mean imputation.  On a stream of its
    // own: the set-up stream is also slot 0's batch stream, and waiting on it would wait for a whole batch.
    hipStream_t st = c->io_stream;
    const size_t afb = sizeof(double) * (size_t)M;
    if (c->consol_af_cap < (size_t)M) {
      if (c->d_consol_af) hipFree(c->d_consol_af);
      c->d_consol_af = nullptr;
      c->consol_af_cap = 0;
      if (hipMalloc((void**)&c->d_consol_af, sizeof(double) * 2 * RVT_MAX_VARIANTS) != hipSuccess) {
        give_back();
        return fail(c, RVT_E_HIP, "hipMalloc failed");
      }
      c->consol_af_cap = RVT_MAX_VARIANTS;
    }
    hipError_t e = hipSuccess;
    const int nparts = (int)((N + kConsolChunk - 1) / kConsolChunk);
    if (c->consol_parts_cap < (size_t)M * nparts) {
      if (c->d_consol_parts) hipFree(c->d_consol_parts);
      c->d_consol_parts = nullptr;
      c->consol_parts_cap = 0;
      const size_t want = (size_t)std::max(M, 128) * nparts;
      e = hipMalloc((void**)&c->d_consol_parts, sizeof(ConsolPart) * want);
      if (e == hipSuccess) c->consol_parts_cap = want;
    }
    double* d_fill = c->d_consol_af + RVT_MAX_VARIANTS;
    const dim3 cgrid((unsigned)nparts, (unsigned)M);
    // where this gene's allele frequencies (and, for VCF text / BGEN blocks, its input-error word) come back: through a
    // ring slot when nobody waits for them, through the synchronous word otherwise
    int ring_slot = -1;
    if (!af_out && e == hipSuccess) {
      if (!c->h_af_ring)
        e = hipHostMalloc((void**)&c->h_af_ring, sizeof(double) * rvt_ctx::kAfSlots * RVT_MAX_VARIANTS, hipHostMallocMapped);
      if (e == hipSuccess && c->af_unresolved >= rvt_ctx::kAfSlots && resolve_af(c)) e = hipErrorUnknown;
      if (e == hipSuccess) ring_slot = (int)(c->af_seq++ % rvt_ctx::kAfSlots);
    }
    // the frequencies go straight into the (device-mapped) ring slot, or into the device buffer the caller waits for
    double* d_af_dst = c->d_consol_af;
    if (ring_slot >= 0 && e == hipSuccess) {
      double* mapped = nullptr;
      e = hipHostGetDevicePointer((void**)&mapped, c->h_af_ring, 0);
      if (e == hipSuccess) d_af_dst = mapped + (size_t)ring_slot * RVT_MAX_VARIANTS;
    }
    const int err_slot = ring_slot >= 0 ? ring_slot : rvt_ctx::kAfSlots;
    const bool decodes = mode == 4 || mode == 5 || mode == 6;
    if (decodes && e == hipSuccess) {
      if (io_err_ready(c) != RVT_OK)
        e = hipErrorUnknown;
      else
        c->h_io_err[err_slot] = 0;  // (the slot's previous user has been resolved; nothing on the device refers to it)
    }
    if (e != hipSuccess) {
      // fall through to the error return below
    } else if (packed) {
      // the rows go straight into the gene's own block (copy stream, 2-D: pitch padded to 16 bytes); the count pass and the
      // header follow on the io stream.  Nothing is expanded.
      unsigned char* rows = reinterpret_cast<unsigned char*>(p.dG) + kHcpHeaderBytes;
      const size_t cb = (size_t)((N + 3) / 4);
      const int ek = c->pack_next;
      c->pack_next = (ek + 1) % rvt_ctx::kPack;
      // The pad bytes of a row read as zeros.  Invariant: a block is cleared once, when it is allocated, and the pool hands a
      // block out again only for the same (N, M) — the same pk_pitch — so a later gene's rows land on the same offsets; the
      // staged copy (pad_zero) writes zeros into the pads BETWEEN rows of a chunk and nothing into the pad behind a chunk's last
      // row, which therefore still holds the allocation's zeros.
      if (mode == 7) {
        // rows of a .bed matrix RESIDENT on the device (rvt_submit_gene_bed_dev): the count pass copies them into the gene's
        // block as it reads them — 6 MB of HBM traffic instead of the link, three launches per gene on one stream
        if (fresh_packed) e = hipMemsetAsync(p.dG, 0, need, st);
        if (e == hipSuccess) {
          hipLaunchKernelGGL(bed_count_copy_kernel, cgrid, dim3(256), 0, st, (const unsigned char*)G, (long long)cb, (long long)N,
                             c->d_consol_parts, rows, (long long)pk_pitch);
          const bed2_t* sb = reinterpret_cast<const bed2_t*>(rows);
          hipLaunchKernelGGL((consolidate_fill_kernel<bed2_t>), dim3((unsigned)M), dim3(64), 0, st, sb, (long long)pk_pitch,
                             (long long)N, nparts, c->d_consol_parts, d_af_dst, d_fill);
          hipLaunchKernelGGL(hcp_header_kernel, dim3(1), dim3(128), 0, st, c->d_consol_parts, nparts, M, (long long)N, d_fill,
                             reinterpret_cast<HcpHeader*>(p.dG));
        }
      } else {
      if (fresh_packed) e = hipMemsetAsync(p.dG, 0, need, c->copy_stream);
      if (e == hipSuccess && mode == 2) {
        // int8 hard calls: the staging threads write the gene's .bed rows into the pinned ring while they read its bytes
        int prc = stage_ready(c) == RVT_OK ? 0 : 1;
        if (prc == 0) {
          c->h2d_stream = c->copy_stream;
          prc = c->stage.pack_i8(rows, pk_pitch, (const signed char*)G, (size_t)N, (size_t)N, (size_t)M, CopyPool::pack_instance());
          c->h2d_stream = c->io_stream;
        }
        if (prc == 2) {  // a value above 2 (not a hard call), or rows longer than a staging chunk: this gene crosses as bytes
          (void)hipStreamSynchronize(c->copy_stream);  // (rows already on their way land before the block is handed out again)
          give_back();
          c->no_i8_pack = true;
          const int rc2 = submit_common(c, gene_id, M, G, mode, af, af_out, tests, prm);
          c->no_i8_pack = false;
          return rc2;
        }
        if (prc != 0) e = hipErrorUnknown;
      } else if (e == hipSuccess) {
        c->h2d_stream = c->copy_stream;
        const int rcs = staged_h2d_2d(c, rows, pk_pitch, G, cb, cb, (size_t)M, true);  // (pad bytes of a packed row are zero anyway)
        c->h2d_stream = c->io_stream;
        if (rcs != RVT_OK) e = hipErrorUnknown;
      }
      if (e == hipSuccess) e = hipEventRecord(c->ev_pack_copied[ek], c->copy_stream);
      if (e == hipSuccess) e = hipStreamWaitEvent(st, c->ev_pack_copied[ek], 0);
      if (e == hipSuccess) {
        const bed2_t* sb = reinterpret_cast<const bed2_t*>(rows);
        hipLaunchKernelGGL((consolidate_count_kernel<bed2_t>), cgrid, dim3(256), 0, st, sb, (long long)pk_pitch, (long long)N,
                           c->d_consol_parts);
        hipLaunchKernelGGL((consolidate_fill_kernel<bed2_t>), dim3((unsigned)M), dim3(64), 0, st, sb, (long long)pk_pitch,
                           (long long)N, nparts, c->d_consol_parts, d_af_dst, d_fill);
        hipLaunchKernelGGL(hcp_header_kernel, dim3(1), dim3(128), 0, st, c->d_consol_parts, nparts, M, (long long)N, d_fill,
                           reinterpret_cast<HcpHeader*>(p.dG));
      }
      }
    } else if (mode == 1 || mode == 5 || mode == 6) {
      int rc = mode == 1   ? upload_block_data(c, p.dG, M, (const double*)G)
               : mode == 5 ? vcf_decode_gene(c, (const VcfGene*)G, M, N, st, err_slot, p.dG, ld)  // VCF dosage text -> doubles
                           : bgen_decode_gene(c, (const BgenGene*)G, M, N, st, err_slot, p.dG, ld);  // BGEN blocks -> doubles
      if (rc) {
        give_back();
        return rc;
      }
      hipLaunchKernelGGL((consolidate_count_kernel<double>), cgrid, dim3(256), 0, st, p.dG, (long long)ld, (long long)N,
                         c->d_consol_parts);
      hipLaunchKernelGGL((consolidate_fill_kernel<double>), dim3((unsigned)M), dim3(64), 0, st, p.dG, (long long)ld,
                         (long long)N, nparts, c->d_consol_parts, d_af_dst, d_fill);
      hipLaunchKernelGGL((consolidate_write_kernel<double>), cgrid, dim3(256), 0, st, p.dG, (long long)ld, (long long)N,
                         (long long)ld, d_fill, p.dG);
    } else {
      // packed hard calls: one byte per genotype (mode 2; mode 4 decodes VCF text into that form first) or PLINK's
      // 2-bit codes, ceil(N/4) bytes per variant (mode 3)
      const size_t col_bytes = (mode == 3 || mode == 7) ? (size_t)((N + 3) / 4) : (size_t)N;
      const size_t bytes8 = col_bytes * M;
      const void* d_packed = nullptr;  // where the packed genotypes of this gene are on the device
      int pk = -1;
      if (mode == 4) {
        if (c->consol_i8_cap < bytes8) {
          if (c->d_consol_i8) hipFree(c->d_consol_i8);
          c->d_consol_i8 = nullptr;
          c->consol_i8_cap = 0;
          e = hipMalloc((void**)&c->d_consol_i8, bytes8 + bytes8 / 4);
          if (e == hipSuccess) c->consol_i8_cap = bytes8 + bytes8 / 4;
        }
        if (e == hipSuccess && vcf_decode_gene(c, (const VcfGene*)G, M, N, st, err_slot) != RVT_OK) e = hipErrorUnknown;
        d_packed = c->d_consol_i8;
      } else {
        // host copy on the copy stream into the next landing buffer of the ring (free once the consolidation kernels of
        // the gene that used it last have run); the kernels of THIS gene wait for the copy by event
        pk = c->pack_next;
        c->pack_next = (pk + 1) % rvt_ctx::kPack;
        if (c->pack_cap[pk] < bytes8) {
          if (c->d_pack[pk]) hipFree(c->d_pack[pk]);
          c->d_pack[pk] = nullptr;
          c->pack_cap[pk] = 0;
          e = hipMalloc((void**)&c->d_pack[pk], bytes8 + bytes8 / 4);
          if (e == hipSuccess) c->pack_cap[pk] = bytes8 + bytes8 / 4;
        }
        if (e == hipSuccess) e = hipStreamWaitEvent(c->copy_stream, c->ev_pack_free[pk], 0);
        if (e == hipSuccess && mode == 7) {  // (device-resident rows: into the same landing buffer, whatever their alignment)
          e = hipMemcpyAsync(c->d_pack[pk], G, bytes8, hipMemcpyDeviceToDevice, c->copy_stream);
        } else if (e == hipSuccess) {
          c->h2d_stream = c->copy_stream;
          const int rcs = staged_h2d(c, c->d_pack[pk], G, bytes8);
          c->h2d_stream = c->io_stream;
          if (rcs != RVT_OK) e = hipErrorUnknown;
        }
        if (e == hipSuccess) e = hipEventRecord(c->ev_pack_copied[pk], c->copy_stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(st, c->ev_pack_copied[pk], 0);
        d_packed = c->d_pack[pk];
      }
      if (e == hipSuccess && (mode == 3 || mode == 7)) {
        const bed2_t* sb = (const bed2_t*)d_packed;
        const long long cb = (long long)col_bytes;
        hipLaunchKernelGGL((consolidate_count_kernel<bed2_t>), cgrid, dim3(256), 0, st, sb, cb, (long long)N,
                           c->d_consol_parts);
        hipLaunchKernelGGL((consolidate_fill_kernel<bed2_t>), dim3((unsigned)M), dim3(64), 0, st, sb, cb, (long long)N,
                           nparts, c->d_consol_parts, d_af_dst, d_fill);
        hipLaunchKernelGGL((consolidate_write_kernel<bed2_t>), cgrid, dim3(256), 0, st, sb, cb, (long long)N,
                           (long long)ld, d_fill, p.dG);
      } else if (e == hipSuccess) {
        const signed char* s8 = (const signed char*)d_packed;
        hipLaunchKernelGGL((consolidate_count_kernel<signed char>), cgrid, dim3(256), 0, st, s8, (long long)N,
                           (long long)N, c->d_consol_parts);
        hipLaunchKernelGGL((consolidate_fill_kernel<signed char>), dim3((unsigned)M), dim3(64), 0, st, s8, (long long)N,
                           (long long)N, nparts, c->d_consol_parts, d_af_dst, d_fill);
        hipLaunchKernelGGL((consolidate_write_kernel<signed char>), cgrid, dim3(256), 0, st, s8, (long long)N,
                           (long long)N, (long long)ld, d_fill, p.dG);
      }
      if (pk >= 0 && e == hipSuccess) e = hipEventRecord(c->ev_pack_free[pk], st);  // the landing buffer may be refilled
    }
    p.af.resize(M);
    if (af_out) {
      if (e == hipSuccess) e = hipMemcpyAsync(p.af.data(), c->d_consol_af, afb, hipMemcpyDeviceToHost, st);
      if (e == hipSuccess) e = sync_stream(st);
      if (e == hipSuccess && decodes && c->h_io_err[err_slot]) {  // malformed text / block: this very gene is refused
        io_err_message(c, c->h_io_err[err_slot], mode == 6, "");
        c->h_io_err[err_slot] = 0;
        give_back();
        return RVT_E_INVALID;
      }
    } else if (e == hipSuccess) {
      // nobody waits for the frequencies: the host copy of the block is already consumed (a copy from pageable memory
      // returns once the source has been read), so return now and pick the frequencies up at launch time
      p.af_slot = ring_slot;  // (written by consolidate_fill_kernel through the mapping; read after the stream is waited for)
      ++c->af_unresolved;
    }
    if (e != hipSuccess) {
      give_back();
      return fail(c, RVT_E_HIP, "genotype consolidation failed: %s", hipGetErrorString(e));
    }
    if (af_out) std::memcpy(af_out, p.af.data(), afb);
  }
  p.tests = tests;
  p.prm = prm ? *prm : rvt_params{1.0, 25.0, 1.0, 25.0, 0, 0.05};
  c->queue.push_back(std::move(p));
  // complete groups start computing now and overlap the host-side copies of the following genes
  TraceScope ts_l(c, &c->tr_launch);
  return launch_pending(c, c->queue.size(), true);
}
}  // namespace

// ---- several genes of a device-resident .bed matrix per call (rvt_submit_genes, kind 7) -----------------------------------------
// One gene at a time the hand-off is three small dependent kernels on one stream (~45 us per gene whatever the host does).  A
// call that names n genes runs them as TWO launches: every row of every gene counted and copied into its gene's block (grid
// (sample pieces, rows)), then one workgroup per gene for the allele frequencies, the imputation values and the header of the
// packed-row kernel — the same expressions in the same order as consolidate_count_kernel<bed2_t>, consolidate_fill_kernel and
// hcp_header_kernel: the records are bit-identical.
namespace {
struct BedRowRef {
  const unsigned char* src;
  unsigned char* dst;
};
struct BedGeneRef {
  int row0, M;
  HcpHeader* hdr;
  double* af_dst;
  long long* cnt_dst;  // optional: per row the numbers of 0 / 1 / 2 / missing calls (rvt_score_bed_dev)
};
__global__ __launch_bounds__(256) void bed_count_copy_rows_kernel(const BedRowRef* __restrict__ rows, long long N,
                                                                  ConsolPart* __restrict__ parts) {
  __shared__ unsigned s_n1[256], s_n2[256], s_nm[256];
  const BedRowRef rr = rows[blockIdx.y];
  const long long i0 = (long long)blockIdx.x * kConsolChunk;
  const long long i1 = (i0 + kConsolChunk < N) ? i0 + kConsolChunk : N;
  unsigned n1 = 0, n2 = 0, nm = 0;
  if ((((N + 3) / 4) & 3) == 0 && (((unsigned long long)rr.src | (unsigned long long)rr.dst) & 3ull) == 0ull) {
    // rows of whole dwords that start on a 4-byte boundary (ceil(N/4) a multiple of 4, e.g. N = 500 000): a dword = 16 samples per
    // thread and trip instead of a byte — the pass moved 1.5 TB/s byte by byte.  Nothing outside the row is read; the samples of
    // the last dword beyond N are masked.
    const unsigned* s4 = reinterpret_cast<const unsigned*>(rr.src);
    unsigned* d4 = reinterpret_cast<unsigned*>(rr.dst);
    for (long long w = (i0 >> 4) + threadIdx.x; 16 * w < i1; w += 256) {
      unsigned v = s4[w];
      const long long left = i1 - 16 * w;
      if (left < 16) v &= (1u << (2 * left)) - 1u;
      d4[w] = v;
      const unsigned lo = v & 0x55555555u, hi = (v >> 1) & 0x55555555u;
      n1 += __popc(hi & ~lo);
      n2 += __popc(hi & lo);
      nm += __popc(lo & ~hi);
    }
  } else {
    for (long long b = (i0 >> 2) + threadIdx.x; 4 * b < i1; b += 256) {
      unsigned v = rr.src[b];
      const long long left = i1 - 4 * b;
      if (left < 4) v &= (1u << (2 * left)) - 1u;
      rr.dst[b] = (unsigned char)v;
      const unsigned lo = v & 0x55u, hi = (v >> 1) & 0x55u;
      n1 += __popc(hi & ~lo);
      n2 += __popc(hi & lo);
      nm += __popc(lo & ~hi);
    }
  }
  s_n1[threadIdx.x] = n1;
  s_n2[threadIdx.x] = n2;
  s_nm[threadIdx.x] = nm;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) {
      s_n1[threadIdx.x] += s_n1[threadIdx.x + off];
      s_n2[threadIdx.x] += s_n2[threadIdx.x + off];
      s_nm[threadIdx.x] += s_nm[threadIdx.x + off];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double ac = (double)s_n1[0] + 2.0 * (double)s_n2[0];
    parts[(long long)blockIdx.y * gridDim.x + blockIdx.x] =
        ConsolPart{ac, ac, (long long)(i1 - i0) - (long long)s_nm[0], s_nm[0] ? 3 : 0, (int)s_n2[0]};
  }
}
__global__ __launch_bounds__(128) void bed_fill_header_kernel(const BedGeneRef* __restrict__ genes,
                                                              const ConsolPart* __restrict__ parts, int nparts, long long N) {
  const BedGeneRef g = genes[blockIdx.x];
  const int j = threadIdx.x, M = g.M;
  bool flip = false, poly = false, cm = false;
  if (j < M) {
    const ConsolPart* p = parts + (long long)(g.row0 + j) * nparts;
    double sumAC = 0.0, ac = 0.0;
    long long nonneg = 0, n2 = 0;
    int flags = 0;
    for (int k = 0; k < nparts; ++k) {
      sumAC += p[k].sumAC;
      ac += p[k].ac;
      nonneg += p[k].nonneg;
      flags |= p[k].flags;
      n2 += p[k].pad;
    }
    g.af_dst[j] = N ? 0.5 * sumAC / (double)N : -1.0;  // (consolidate_fill_kernel)
    double fill = 0.0;
    if ((flags & 1) && (flags & 2)) {
      const long long an = 2 * nonneg;
      const int aci = (int)ac;
      fill = (an == 0) ? 0.0 : 2.0 * (1.0 * aci / (double)an);
    }
    const long long nm = N - nonneg, n1 = (long long)ac - 2 * n2, n0 = nonneg - n1 - n2;  // (hcp_header_kernel)
    if (g.cnt_dst) {
      g.cnt_dst[4 * j + 0] = n0;
      g.cnt_dst[4 * j + 1] = n1;
      g.cnt_dst[4 * j + 2] = n2;
      g.cnt_dst[4 * j + 3] = nm;
    }
    const double mu = nm > 0 ? fill : 0.0;
    g.hdr->mu[j] = mu;
    const double s = ac + (double)nm * mu;
    flip = !(s <= (double)N);
    double mn = n0 > 0 ? 0.0 : (n1 > 0 ? 1.0 : (n2 > 0 ? 2.0 : INFINITY));
    double mx = n2 > 0 ? 2.0 : (n1 > 0 ? 1.0 : (n0 > 0 ? 0.0 : -INFINITY));
    if (nm > 0) {
      mn = fmin(mn, mu);
      mx = fmax(mx, mu);
    }
    poly = !(mn == mx);
    cm = nm > 0 && poly && (flip ? mu <= 1.0 : mu >= 1.0);
  }
  const unsigned long long bf = __ballot(flip), bp = __ballot(poly), bc = __ballot(cm);
  if ((j & 15) == 0 && j < 96) {
    const int b = j >> 4, sh = 16 * ((j >> 4) & 3);
    g.hdr->flip[b] = (unsigned short)((bf >> sh) & 0xffffu);
    g.hdr->poly[b] = (unsigned short)((bp >> sh) & 0xffffu);
    g.hdr->cm[b] = (unsigned short)((bc >> sh) & 0xffffu);
  }
}

// returns -1 when the call does not apply (a gene that has to be expanded): the caller submits gene by gene
int submit_bed_dev_batch(rvt_ctx* c, int n, const int64_t* ids, const int* Ms, const void* const* data, uint32_t tests,
                         const rvt_params* prm) {
  if (n < 1 || n > rvt_ctx::kAfSlots / 2) return -1;
  if (!c->have_null || (tests & RVT_TEST_FAMSKAT)) return -1;
  int R = 0;
  for (int g = 0; g < n; ++g) {
    if (!data[g] || Ms[g] < 1 || Ms[g] > RVT_MAX_VARIANTS || !packed_eligible(c, Ms[g], tests, prm)) return -1;
    R += Ms[g];
  }
  hipSetDevice(c->device);
  TraceScope ts_all(c, &c->tr_block);
  hipStream_t st = c->io_stream;
  const int64_t N = c->nc.N;
  const size_t cb = (size_t)((N + 3) / 4), pk_pitch = (cb + 15) / 16 * 16;
  const int nparts = (int)((N + kConsolChunk - 1) / kConsolChunk);
  if (!c->h_af_ring)
    HIP_TRY(c, hipHostMalloc((void**)&c->h_af_ring, sizeof(double) * rvt_ctx::kAfSlots * RVT_MAX_VARIANTS, hipHostMallocMapped));
  if (c->af_unresolved + n > rvt_ctx::kAfSlots) {
    int rc = resolve_af(c);
    if (rc) return rc;
  }
  double* mapped = nullptr;
  HIP_TRY(c, hipHostGetDevicePointer((void**)&mapped, c->h_af_ring, 0));
  // device work space of the call: row and gene references, the rows' partial counts
  const size_t b_rows = (sizeof(BedRowRef) * (size_t)R + 255) / 256 * 256, b_genes = (sizeof(BedGeneRef) * (size_t)n + 255) / 256 * 256;
  const size_t need = b_rows + b_genes + sizeof(ConsolPart) * (size_t)R * nparts;
  if (c->bedbatch_cap < need) {
    HIP_TRY(c, sync_stream(st));
    if (c->d_bedbatch) hipFree(c->d_bedbatch);
    c->d_bedbatch = nullptr;
    c->bedbatch_cap = 0;
    HIP_TRY(c, hipMalloc((void**)&c->d_bedbatch, need + need / 2));
    c->bedbatch_cap = need + need / 2;
  }
  std::vector<BedRowRef> rows((size_t)R);
  std::vector<BedGeneRef> genes((size_t)n);
  std::vector<rvt_ctx::Pending> pend((size_t)n);
  auto give_back = [&](int upto) {
    for (int g = 0; g < upto; ++g) c->pk_pool.emplace_back(pend[(size_t)g].bytes, pend[(size_t)g].dG);
  };
  int r0 = 0;
  for (int g = 0; g < n; ++g) {
    rvt_ctx::Pending& p = pend[(size_t)g];
    const int M = Ms[g];
    p.id = ids[g];
    p.M = M;
    p.launched = false;
    std::memset(&p.res, 0, sizeof(p.res));
    const size_t gene_bytes = (size_t)kHcpHeaderBytes + pk_pitch * M + 16;
    int best = -1;
    for (int i = 0; i < (int)c->pk_pool.size(); ++i)
      if (c->pk_pool[i].first >= gene_bytes && c->pk_pool[i].first <= 4 * gene_bytes &&
          (best < 0 || c->pk_pool[i].first < c->pk_pool[best].first))
        best = i;
    if (best >= 0) {
      p.dG = c->pk_pool[best].second;
      p.bytes = c->pk_pool[best].first;
      c->pk_pool.erase(c->pk_pool.begin() + best);
    } else {
      if (hipMalloc((void**)&p.dG, gene_bytes) != hipSuccess) {
        (void)hipGetLastError();
        give_back(g);
        return fail(c, RVT_E_HIP, "hipMalloc(%zu bytes) failed for a packed gene", gene_bytes);
      }
      p.bytes = gene_bytes;
      if (hipMemsetAsync(p.dG, 0, gene_bytes, st) != hipSuccess) {
        give_back(g + 1);
        return fail(c, RVT_E_HIP, "clearing a packed gene's block failed");
      }
    }
    unsigned char* dst = reinterpret_cast<unsigned char*>(p.dG) + kHcpHeaderBytes;
    const unsigned char* src = static_cast<const unsigned char*>(data[g]);
    for (int j = 0; j < M; ++j) rows[(size_t)(r0 + j)] = BedRowRef{src + cb * (size_t)j, dst + pk_pitch * (size_t)j};
    const int slot = (int)(c->af_seq++ % rvt_ctx::kAfSlots);
    genes[(size_t)g] = BedGeneRef{r0, M, reinterpret_cast<HcpHeader*>(p.dG), mapped + (size_t)slot * RVT_MAX_VARIANTS, nullptr};
    p.af_slot = slot;
    p.af.resize((size_t)M);
    p.kind = 3;
    p.planes = true;  // (resident rows: nothing waits for a link, the sufficient-statistics kernel is the bound)
    p.decoded = 0;
    p.tests = tests;
    p.prm = prm ? *prm : rvt_params{1.0, 25.0, 1.0, 25.0, 0, 0.05};
    r0 += M;
  }
  char* w = c->d_bedbatch;
  BedRowRef* d_rows = reinterpret_cast<BedRowRef*>(w);
  BedGeneRef* d_genes = reinterpret_cast<BedGeneRef*>(w + b_rows);
  ConsolPart* d_parts = reinterpret_cast<ConsolPart*>(w + b_rows + b_genes);
  hipError_t e = hipMemcpyAsync(d_rows, rows.data(), sizeof(BedRowRef) * (size_t)R, hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = hipMemcpyAsync(d_genes, genes.data(), sizeof(BedGeneRef) * (size_t)n, hipMemcpyHostToDevice, st);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(bed_count_copy_rows_kernel, dim3((unsigned)nparts, (unsigned)R), dim3(256), 0, st, d_rows, (long long)N, d_parts);
    hipLaunchKernelGGL(bed_fill_header_kernel, dim3((unsigned)n), dim3(128), 0, st, d_genes, d_parts, nparts, (long long)N);
    e = hipGetLastError();
  }
  if (e != hipSuccess) {
    give_back(n);
    return fail(c, RVT_E_HIP, "resident .bed genes: %s", hipGetErrorString(e));
  }
  for (int g = 0; g < n; ++g) {
    c->queue.push_back(std::move(pend[(size_t)g]));
    ++c->af_unresolved;
    if (c->trace_submit) ++c->tr_genes;
  }
  TraceScope ts_l(c, &c->tr_launch);
  return launch_pending(c, c->queue.size(), true);
}
}  // namespace

// ---- single-variant score tests of a resident .bed matrix ------------------------------------------------------------------------
// LinearRegressionScoreTest::TestCovariate on ONE column per variant (regression/LinearRegressionScoreTest.cpp:173-263; what
// MetaScoreTest::fit prints per site, src/Model.h:3246-3258) for V consecutive rows of a .bed matrix in device memory: slices
// of 32 rows go through the packed-row machinery as "genes" of 32 variants — counts and imputation values per row, the rows
// copied into pitched blocks (two launches per 256 slices), gene_suffstat_hcp + gene_tnull_hcp, then the score finisher that
// rvt_score_block uses — so a site costs N/4 bytes of HBM traffic instead of 8 N.  Quantitative traits (the packed kernel's
// domain); same outputs as rvt_score_block plus, optionally, the genotype counts (n0, n1, n2, missing per variant) the
// adapter's AF / call rate / HWE columns are made of.  Synchronous.
int rvt_score_bed_dev(rvt_ctx* c, const unsigned char* d_rows, int64_t V, int* ok, double* ustat, double* vstat, double* effect,
                      double* effect_se, double* pvalue, long long* counts) {
  if (!c || !d_rows || V < 1 || !ok || !ustat || !vstat || !effect || !effect_se || !pvalue) return fail(c, RVT_E_INVALID, "bad arguments");
  if (!c->have_null) return fail(c, RVT_E_STATE, "no null model set");
  constexpr int kSlice = 32;
  static const int kChunk = getenv("RVT_SCORE_BED_CHUNK") ? std::max(16, std::min(2048, atoi(getenv("RVT_SCORE_BED_CHUNK")))) : 256;
  if (!packed_eligible(c, kSlice, 0u, nullptr)) return fail(c, RVT_E_STATE, "the packed-row kernel does not take this null model (binary trait, too many covariates, hard calls switched off)");
  hipSetDevice(c->device);
  int rc = rvt_sync(c);
  if (rc) return rc;
  hipStream_t st = c->io_stream;
  const int64_t N = c->nc.N;
  const size_t cb = (size_t)((N + 3) / 4), pk_pitch = (cb + 15) / 16 * 16;
  const int nparts = (int)((N + kConsolChunk - 1) / kConsolChunk);
  const size_t gene_bytes = (size_t)kHcpHeaderBytes + pk_pitch * kSlice + 16;
  // work space: row / gene references, partial counts (as submit_bed_dev_batch), a frequency sink, the counts
  const int Rmax = kSlice * kChunk;
  const size_t b_rows = (sizeof(BedRowRef) * (size_t)Rmax + 255) / 256 * 256, b_genes = (sizeof(BedGeneRef) * (size_t)kChunk + 255) / 256 * 256;
  const size_t b_parts = (sizeof(ConsolPart) * (size_t)Rmax * nparts + 255) / 256 * 256, b_af = sizeof(double) * (size_t)Rmax;
  const size_t need = b_rows + b_genes + b_parts + b_af + sizeof(long long) * 4 * (size_t)Rmax;
  if (c->bedbatch_cap < need) {
    HIP_TRY(c, sync_stream(st));
    if (c->d_bedbatch) hipFree(c->d_bedbatch);
    c->d_bedbatch = nullptr;
    c->bedbatch_cap = 0;
    HIP_TRY(c, hipMalloc((void**)&c->d_bedbatch, need + need / 2));
    c->bedbatch_cap = need + need / 2;
  }
  char* w = c->d_bedbatch;
  BedRowRef* d_rref = reinterpret_cast<BedRowRef*>(w);
  BedGeneRef* d_gref = reinterpret_cast<BedGeneRef*>(w + b_rows);
  ConsolPart* d_parts = reinterpret_cast<ConsolPart*>(w + b_rows + b_genes);
  double* d_afsink = reinterpret_cast<double*>(w + b_rows + b_genes + b_parts);
  long long* d_cnt = reinterpret_cast<long long*>(w + b_rows + b_genes + b_parts + b_af);
  // the slices' blocks: kChunk of them, reused chunk after chunk (the batch is synchronous)
  std::vector<double*> blk;
  auto release = [&]() {
    for (double* b : blk) c->pk_pool.emplace_back(gene_bytes, b);
    blk.clear();
  };
  const int64_t n_slices_all = (V + kSlice - 1) / kSlice;
  const int n_blk = (int)std::min<int64_t>(kChunk, n_slices_all);
  for (int g = 0; g < n_blk; ++g) {
    double* b = nullptr;
    int best = -1;
    for (int i = 0; i < (int)c->pk_pool.size(); ++i)
      if (c->pk_pool[i].first >= gene_bytes && c->pk_pool[i].first <= 4 * gene_bytes && (best < 0 || c->pk_pool[i].first < c->pk_pool[best].first)) best = i;
    if (best >= 0) {
      b = c->pk_pool[best].second;
      c->pk_pool.erase(c->pk_pool.begin() + best);
    } else {
      if (hipMalloc((void**)&b, gene_bytes) != hipSuccess) {
        (void)hipGetLastError();
        release();
        return fail(c, RVT_E_HIP, "hipMalloc(%zu bytes) failed for a packed slice", gene_bytes);
      }
      if (hipMemsetAsync(b, 0, gene_bytes, st) != hipSuccess) {
        blk.push_back(b);
        release();
        return fail(c, RVT_E_HIP, "clearing a packed slice failed");
      }
    }
    blk.push_back(b);
  }
  std::vector<BedRowRef> rows;
  std::vector<BedGeneRef> genes;
  std::vector<const double*> ptr;
  std::vector<int> Ms;
  std::vector<int64_t> ids;
  std::vector<signed char> kinds;
  std::vector<unsigned char> shc;
  std::vector<double> af((size_t)kSlice * kChunk, 0.01);
  std::vector<rvt_gene_result> rs(kChunk);
  for (int64_t v0 = 0; v0 < V && !rc; v0 += (int64_t)kSlice * kChunk) {
    const int cols = (int)std::min<int64_t>(V - v0, (int64_t)kSlice * kChunk), n = (cols + kSlice - 1) / kSlice;
    rows.clear();
    genes.clear();
    ptr.clear();
    Ms.clear();
    ids.clear();
    kinds.assign((size_t)n, (signed char)(3 | 0x10));
    shc.assign((size_t)n, 1);
    for (int g = 0; g < n; ++g) {
      const int M = std::min(kSlice, cols - g * kSlice);
      unsigned char* dst = reinterpret_cast<unsigned char*>(blk[(size_t)g]) + kHcpHeaderBytes;
      const unsigned char* src = d_rows + cb * (size_t)(v0 + (int64_t)g * kSlice);
      const int r0 = (int)rows.size();
      for (int j = 0; j < M; ++j) rows.push_back(BedRowRef{src + cb * (size_t)j, dst + pk_pitch * (size_t)j});
      // (a shorter last slice: the rows behind it in the block keep an earlier slice's codes — beyond M, never read)
      genes.push_back(BedGeneRef{r0, M, reinterpret_cast<HcpHeader*>(blk[(size_t)g]), d_afsink + r0, d_cnt + 4 * (size_t)r0});
      ptr.push_back(blk[(size_t)g]);
      Ms.push_back(M);
      ids.push_back((int64_t)g * kSlice);
    }
    const int R = (int)rows.size();
    hipError_t e = hipMemcpyAsync(d_rref, rows.data(), sizeof(BedRowRef) * (size_t)R, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(d_gref, genes.data(), sizeof(BedGeneRef) * (size_t)n, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
      hipLaunchKernelGGL(bed_count_copy_rows_kernel, dim3((unsigned)nparts, (unsigned)R), dim3(256), 0, st, d_rref, (long long)N, d_parts);
      hipLaunchKernelGGL(bed_fill_header_kernel, dim3((unsigned)n), dim3(128), 0, st, d_gref, d_parts, nparts, (long long)N);
      e = hipGetLastError();
    }
    if (e == hipSuccess && counts)
      e = hipMemcpyAsync(counts + 4 * v0, d_cnt, sizeof(long long) * 4 * (size_t)R, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);  // (run_batch launches on its own streams)
    if (e != hipSuccess) {
      release();
      return fail(c, RVT_E_HIP, "resident .bed score: %s", hipGetErrorString(e));
    }
    CovOut co;
    co.score = true;
    co.slice_hc = shc.data();
    co.ok = ok + v0;
    co.ustat = ustat + v0;
    co.vstat = vstat + v0;
    co.effect = effect + v0;
    co.se = effect_se + v0;
    co.pval = pvalue + v0;
    rc = run_batch(c, n, ptr.data(), Ms.data(), af.data(), ids.data(), 0u, nullptr, rs.data(), nullptr, &co, kinds.data());
  }
  release();
  return rc;
}

int rvt_submit_gene(rvt_ctx* c, int64_t gene_id, int M, const double* G, const double* af, uint32_t tests,
                    const rvt_params* prm) {
  return submit_common(c, gene_id, M, G, 0, af, nullptr, tests, prm);
}
int rvt_submit_gene_bed(rvt_ctx* c, int64_t gene_id, int M, const unsigned char* bed, uint32_t tests,
                        const rvt_params* prm, double* af_out) {
  return submit_common(c, gene_id, M, bed, 3, nullptr, af_out, tests, prm);
}
// ---- a .bed matrix resident in HBM ---------------------------------------------------------------------------------------
// 500 000 samples x 2 000 000 variants of PLINK 2-bit rows are 250 GB: a whole exome-scale cohort fits the 288 GB of one
// MI355X.  rvt_bed_alloc / rvt_bed_upload put rows there once (file layout: ceil(N/4) bytes per variant, no padding);
// rvt_submit_gene_bed_dev then names a gene by the device address of its first row.
int rvt_bed_alloc(rvt_ctx* c, int64_t n_variants, unsigned char** d_bed) {
  if (!c || !d_bed || n_variants < 1) return fail(c, RVT_E_INVALID, "bad .bed allocation");
  if (!c->have_null) return fail(c, RVT_E_STATE, "no null model set");
  hipSetDevice(c->device);
  const size_t cb = (size_t)((c->nc.N + 3) / 4);
  *d_bed = nullptr;
  if (hipMalloc((void**)d_bed, cb * (size_t)n_variants + 16) != hipSuccess) {
    (void)hipGetLastError();
    return fail(c, RVT_E_HIP, "hipMalloc(%zu bytes) failed for a resident .bed matrix", cb * (size_t)n_variants);
  }
  return RVT_OK;
}
int rvt_bed_upload(rvt_ctx* c, unsigned char* d_bed, int64_t first_variant, int64_t n_variants, const unsigned char* rows) {
  if (!c || !d_bed || !rows || first_variant < 0 || n_variants < 0) return fail(c, RVT_E_INVALID, "bad .bed upload");
  if (!c->have_null) return fail(c, RVT_E_STATE, "no null model set");
  if (n_variants == 0) return RVT_OK;
  hipSetDevice(c->device);
  const size_t cb = (size_t)((c->nc.N + 3) / 4);
  int rc = staged_h2d(c, d_bed + cb * (size_t)first_variant, rows, cb * (size_t)n_variants);
  if (rc) return rc;
  HIP_TRY(c, sync_stream(c->h2d_stream));  // (the caller may reuse `rows`; genes submitted next read the device copy)
  return RVT_OK;
}
int rvt_bed_free(rvt_ctx* c, unsigned char* d_bed) {
  if (!c) return RVT_E_INVALID;
  if (!d_bed) return RVT_OK;
  hipSetDevice(c->device);
  int rc = rvt_sync(c);  // (queued genes may still read it)
  hipFree(d_bed);
  return rc;
}
int rvt_submit_gene_bed_dev(rvt_ctx* c, int64_t gene_id, int M, const unsigned char* d_rows, uint32_t tests,
                            const rvt_params* prm, double* af_out) {
  return submit_common(c, gene_id, M, d_rows, 7, nullptr, af_out, tests, prm);
}
int rvt_submit_gene_raw(rvt_ctx* c, int64_t gene_id, int M, const double* Graw, uint32_t tests,
                        const rvt_params* prm, double* af_out) {
  return submit_common(c, gene_id, M, Graw, 1, nullptr, af_out, tests, prm);
}
int rvt_submit_gene_i8(rvt_ctx* c, int64_t gene_id, int M, const int8_t* G8, uint32_t tests, const rvt_params* prm,
                       double* af_out) {
  return submit_common(c, gene_id, M, G8, 2, nullptr, af_out, tests, prm);
}

// Several genes per call (the same hand-off as rvt_submit_gene_raw / _i8 / _bed, gene after gene): for callers whose
// per-call cost is not negligible against the ~120 us a packed gene needs on the link (a ctypes / JNI / cgo caller)
int rvt_submit_genes(rvt_ctx* c, int kind, int n, const int64_t* gene_ids, const int* M, const void* const* data,
                     uint32_t tests, const rvt_params* prm) {
  if (!c || n < 0 || (n > 0 && (!gene_ids || !M || !data))) return fail(c, RVT_E_INVALID, "bad gene list");
  if ((kind < 1 || kind > 3) && kind != 7)
    return fail(c, RVT_E_INVALID, "kind %d: 1 = doubles with missing codes, 2 = int8, 3 = PLINK 2-bit rows, 7 = PLINK 2-bit rows on the device", kind);
  // The genes' transfers out of page-locked caller memory (rvt_host_register) are queued back to back and waited for ONCE:
  // the call returns when the last of them has been read (a gene-by-gene submission waits per gene — the buffer may be
  // rewritten on return — which leaves the link idle between two genes).
  if (kind == 7) {
    // resident rows: whole calls at once where every gene may stay packed, groups of up to 256 genes per batch of the pipeline
    // (nothing waits for a link here: the larger the batch, the fuller the chip)
    const int group0 = c->submit_group;
    c->submit_group = std::max(group0, 256);
    int rc = RVT_OK;
    for (int g0 = 0; g0 < n && !rc; g0 += rvt_ctx::kAfSlots / 2) {
      const int ng = std::min(n - g0, rvt_ctx::kAfSlots / 2);
      rc = submit_bed_dev_batch(c, ng, gene_ids + g0, M + g0, data + g0, tests, prm);
      if (rc == -1) {
        rc = RVT_OK;
        for (int g = g0; g < g0 + ng && !rc; ++g)
          rc = submit_common(c, gene_ids[g], M[g], data[g], 7, nullptr, nullptr, tests, prm);
      }
    }
    c->submit_group = group0;
    return rc;
  }
  c->reg_defer = true;
  int rc = RVT_OK;
  for (int g = 0; g < n && !rc; ++g)
    rc = submit_common(c, gene_ids[g], M[g], data[g], kind, nullptr, nullptr, tests, prm);  // (genes [0, g) are queued on failure)
  c->reg_defer = false;
  const int rcw = reg_wait(c);
  return rc ? rc : rcw;
}

// ---- VCF text front end ----------------------------------------------------------------------------------------------
int rvt_vcf_set_samples(rvt_ctx* c, int n_file_samples, const int32_t* row_of_sample) {
  if (!c || n_file_samples < 1 || !row_of_sample) return fail(c, RVT_E_INVALID, "bad sample map");
  hipSetDevice(c->device);
  int64_t rows = 0;
  for (int i = 0; i < n_file_samples; ++i) rows = std::max<int64_t>(rows, (int64_t)row_of_sample[i] + 1);
  std::vector<char> seen((size_t)rows, 0);
  for (int i = 0; i < n_file_samples; ++i) {
    const int r = row_of_sample[i];
    if (r < 0) continue;
    if (seen[r]) return fail(c, RVT_E_INVALID, "sample map: row %d addressed twice", r);
    seen[r] = 1;
  }
  for (int64_t r = 0; r < rows; ++r)
    if (!seen[r]) return fail(c, RVT_E_INVALID, "sample map: row %lld is never addressed", (long long)r);
  int rc = rvt_sync(c);
  if (rc) return rc;
  if (c->d_vcf_rows) hipFree(c->d_vcf_rows);
  c->d_vcf_rows = nullptr;
  if (c->d_vcf_sex) hipFree(c->d_vcf_sex);  // (belongs to the previous file)
  c->d_vcf_sex = nullptr;
  HIP_TRY(c, hipMalloc((void**)&c->d_vcf_rows, sizeof(int) * (size_t)n_file_samples));
  HIP_TRY(c, hipMemcpy(c->d_vcf_rows, row_of_sample, sizeof(int) * (size_t)n_file_samples, hipMemcpyHostToDevice));
  c->vcf_n_file = n_file_samples;
  c->vcf_n_rows = rows;
  return RVT_OK;
}

int rvt_vcf_set_alt_alleles(rvt_ctx* c, int M, const int* alt) {
  if (!c || M < 0 || (M > 0 && !alt)) return RVT_E_INVALID;
  for (int j = 0; j < M; ++j)
    if (alt[j] < 0 || alt[j] > 9) return fail(c, RVT_E_INVALID, "alternative allele index %d (single digits only)", alt[j]);
  c->vcf_alt.assign(alt, alt + M);
  return RVT_OK;
}

int rvt_vcf_set_sex(rvt_ctx* c, int n_file_samples, const int8_t* sex) {
  if (!c || n_file_samples < 1 || !sex) return fail(c, RVT_E_INVALID, "bad sex codes");
  if (!c->d_vcf_rows || c->vcf_n_file != n_file_samples)
    return fail(c, RVT_E_STATE, "rvt_vcf_set_samples first (same number of file samples)");
  hipSetDevice(c->device);
  int rc = rvt_sync(c);
  if (rc) return rc;
  if (c->d_vcf_sex) hipFree(c->d_vcf_sex);
  c->d_vcf_sex = nullptr;
  HIP_TRY(c, hipMalloc((void**)&c->d_vcf_sex, (size_t)n_file_samples));
  HIP_TRY(c, hipMemcpy(c->d_vcf_sex, sex, (size_t)n_file_samples, hipMemcpyHostToDevice));
  return RVT_OK;
}

int rvt_vcf_set_hemi(rvt_ctx* c, int M, const int* hemi) {
  if (!c || M < 0 || (M > 0 && !hemi)) return RVT_E_INVALID;
  if (M > 0 && !c->d_vcf_sex) return fail(c, RVT_E_STATE, "rvt_vcf_set_sex first");
  c->vcf_hemi.assign(hemi, hemi + M);
  return RVT_OK;
}

int rvt_vcf_set_dosage(rvt_ctx* c, int use_dosage) {
  if (!c) return RVT_E_INVALID;
  c->vcf_dosage = use_dosage != 0;
  return RVT_OK;
}

int rvt_vcf_set_filters(rvt_ctx* c, int gd_min, int gd_max, int gq_min, int gq_max) {
  if (!c) return RVT_E_INVALID;
  c->vcf_flt = VcfFilters{gd_min, gd_max, gq_min, gq_max};
  return RVT_OK;
}

int rvt_submit_gene_vcf(rvt_ctx* c, int64_t gene_id, int M, const char* const* sample_text, const int64_t* text_len,
                        const int* gt_index, const int* gd_index, const int* gq_index, uint32_t tests,
                        const rvt_params* prm, double* af_out) {
  if (!c || !sample_text || !text_len || !gt_index || M < 1) return fail(c, RVT_E_INVALID, "bad gene");
  if (!c->d_vcf_rows) return fail(c, RVT_E_STATE, "rvt_vcf_set_samples first");
  if (c->have_null && c->vcf_n_rows != c->nc.N)
    return fail(c, RVT_E_STATE, "the sample map addresses %lld rows, the null model has %lld samples",
                (long long)c->vcf_n_rows, (long long)c->nc.N);
  for (int j = 0; j < M; ++j)
    if (!sample_text[j] || text_len[j] < 0) return fail(c, RVT_E_INVALID, "record %d: no text", j);
  VcfGene vg{sample_text, text_len, gt_index, gd_index, gq_index};
  return submit_common(c, gene_id, M, &vg, c->vcf_dosage ? 5 : 4, nullptr, af_out, tests, prm);
}

// Decode only: the N x M signed bytes (column-major) the device reads out of the text, copied back to the caller.
int rvt_vcf_decode(rvt_ctx* c, int M, const char* const* sample_text, const int64_t* text_len, const int* gt_index,
                   const int* gd_index, const int* gq_index, int8_t* out) {
  if (!c || !sample_text || !text_len || !gt_index || !out || M < 1 || M > RVT_MAX_VARIANTS)
    return fail(c, RVT_E_INVALID, "bad arguments");
  if (!c->d_vcf_rows) return fail(c, RVT_E_STATE, "rvt_vcf_set_samples first");
  hipSetDevice(c->device);
  const int64_t N = c->vcf_n_rows;
  const size_t bytes8 = (size_t)N * M;
  hipStream_t st = c->io_stream;
  HIP_TRY(c, sync_stream(st));
  if (c->consol_i8_cap < bytes8) {
    if (c->d_consol_i8) hipFree(c->d_consol_i8);
    c->d_consol_i8 = nullptr;
    c->consol_i8_cap = 0;
    HIP_TRY(c, hipMalloc((void**)&c->d_consol_i8, bytes8 + bytes8 / 4));
    c->consol_i8_cap = bytes8 + bytes8 / 4;
  }
  VcfGene vg{sample_text, text_len, gt_index, gd_index, gq_index};
  int rc = io_err_ready(c);
  if (rc) return rc;
  c->h_io_err[rvt_ctx::kAfSlots] = 0;
  rc = vcf_decode_gene(c, &vg, M, N, st, rvt_ctx::kAfSlots);
  if (rc) return rc;
  HIP_TRY(c, hipMemcpyAsync(out, c->d_consol_i8, bytes8, hipMemcpyDeviceToHost, st));
  HIP_TRY(c, sync_stream(st));
  if (const int k = c->h_io_err[rvt_ctx::kAfSlots]) {
    c->h_io_err[rvt_ctx::kAfSlots] = 0;
    io_err_message(c, k, false, "");
    return RVT_E_INVALID;
  }
  return RVT_OK;
}

static int bgen_check(rvt_ctx* c, int M, const unsigned char* const* block, const int64_t* len, int layout) {
  if (!c || !block || !len || M < 1 || M > RVT_MAX_VARIANTS || (layout != 1 && layout != 2))
    return fail(c, RVT_E_INVALID, "bad arguments");
  for (int j = 0; j < M; ++j)
    if (!block[j] || len[j] < 0) return fail(c, RVT_E_INVALID, "variant %d: no block", j);
  return RVT_OK;
}

int rvt_submit_gene_bgen(rvt_ctx* c, int64_t gene_id, int M, const unsigned char* const* block, const int64_t* block_len,
                         int layout, uint32_t tests, const rvt_params* prm, double* af_out) {
  int rc = bgen_check(c, M, block, block_len, layout);
  if (rc) return rc;
  if (!c->have_null) return fail(c, RVT_E_STATE, "no null model set");
  if (c->d_vcf_rows && c->vcf_n_rows != c->nc.N)
    return fail(c, RVT_E_STATE, "the sample map addresses %lld rows, the null model has %lld samples",
                (long long)c->vcf_n_rows, (long long)c->nc.N);
  BgenGene bg{block, block_len, layout};
  return submit_common(c, gene_id, M, &bg, 6, nullptr, af_out, tests, prm);
}

// Decode only: the N x M raw genotypes (column-major doubles, -9 = missing) the device reads out of the blocks
int rvt_bgen_decode(rvt_ctx* c, int M, const unsigned char* const* block, const int64_t* block_len, int layout,
                    int64_t n_rows, double* out) {
  int rc = bgen_check(c, M, block, block_len, layout);
  if (rc) return rc;
  if (!out || n_rows < 1) return fail(c, RVT_E_INVALID, "bad arguments");
  if (c->d_vcf_rows && c->vcf_n_rows != n_rows)
    return fail(c, RVT_E_STATE, "the sample map addresses %lld rows, not %lld", (long long)c->vcf_n_rows, (long long)n_rows);
  hipSetDevice(c->device);
  hipStream_t st = c->io_stream;
  HIP_TRY(c, sync_stream(st));
  double* d_out = nullptr;
  HIP_TRY(c, hipMalloc((void**)&d_out, sizeof(double) * (size_t)n_rows * M));
  BgenGene bg{block, block_len, layout};
  rc = io_err_ready(c);
  if (!rc) {
    c->h_io_err[rvt_ctx::kAfSlots] = 0;
    rc = bgen_decode_gene(c, &bg, M, n_rows, st, rvt_ctx::kAfSlots, d_out, n_rows);
  }
  hipError_t e = hipSuccess;
  if (!rc) e = hipMemcpyAsync(out, d_out, sizeof(double) * (size_t)n_rows * M, hipMemcpyDeviceToHost, st);
  if (!rc && e == hipSuccess) e = sync_stream(st);
  hipFree(d_out);
  if (rc) return rc;
  if (e != hipSuccess) return fail(c, RVT_E_HIP, "BGEN decode failed: %s", hipGetErrorString(e));
  if (const int k = c->h_io_err[rvt_ctx::kAfSlots]) {
    c->h_io_err[rvt_ctx::kAfSlots] = 0;
    io_err_message(c, k, true, "");
    return RVT_E_INVALID;
  }
  return RVT_OK;
}

// Host-only helper: the first nine columns of one record.  *sample_off = offset of the first sample column; FORMAT
// indices by VCFRecord::getFormatIndex's rule (libVcf/VCFRecord.h:280-305: the key matches when the FORMAT entry
// STARTS with it).  Returns RVT_E_INVALID when the line has fewer than ten columns.
int rvt_vcf_locate(const char* line, int64_t len, int64_t* sample_off, int* gt_index, int* gd_index, int* gq_index) {
  if (!line || len < 0 || !sample_off) return RVT_E_INVALID;
  int64_t p = 0, fmt_b = -1, fmt_e = -1;
  int tabs = 0;
  for (; p < len && tabs < 9; ++p)
    if (line[p] == '\t') {
      ++tabs;
      if (tabs == 8) fmt_b = p + 1;
      if (tabs == 9) fmt_e = p;
    }
  if (tabs < 9) return RVT_E_INVALID;
  *sample_off = p;
  auto index_of = [&](const char* key) {
    int64_t b = fmt_b;
    int idx = 0;
    while (b < fmt_e) {
      bool match = true;
      for (int i = 0; key[i]; ++i)
        if (b + i >= len || line[b + i] != key[i]) {
          match = false;
          break;
        }
      if (match) return idx;
      ++idx;
      while (line[b++] != ':')
        if (b >= fmt_e) return -1;
    }
    return -1;
  };
  if (gt_index) *gt_index = index_of("GT");
  if (gd_index) *gd_index = index_of("GD");
  if (gq_index) *gq_index = index_of("GQ");
  return RVT_OK;
}

// FORMAT index of an arbitrary key (the dosage tag of --dosage), same prefix rule
int rvt_vcf_format_index(const char* line, int64_t len, const char* key, int* index) {
  if (!line || len < 0 || !key || !index) return RVT_E_INVALID;
  int64_t p = 0, fmt_b = -1, fmt_e = -1;
  int tabs = 0;
  for (; p < len && tabs < 9; ++p)
    if (line[p] == '\t') {
      ++tabs;
      if (tabs == 8) fmt_b = p + 1;
      if (tabs == 9) fmt_e = p;
    }
  if (tabs < 9) return RVT_E_INVALID;
  int64_t b = fmt_b;
  int idx = 0;
  *index = -1;
  while (b < fmt_e) {
    bool match = true;
    for (int i = 0; key[i]; ++i)
      if (b + i >= len || line[b + i] != key[i]) {
        match = false;
        break;
      }
    if (match) {
      *index = idx;
      return RVT_OK;
    }
    ++idx;
    bool more = false;
    while (b < fmt_e)
      if (line[b++] == ':') {
        more = true;
        break;
      }
    if (!more) break;
  }
  return RVT_OK;
}

// Decode only, dosage mode: out = N x M doubles (column-major, leading dimension N), missing = -9
int rvt_vcf_decode_dosage(rvt_ctx* c, int M, const char* const* sample_text, const int64_t* text_len, const int* tag_index,
                          const int* gd_index, const int* gq_index, double* out) {
  if (!c || !sample_text || !text_len || !tag_index || !out || M < 1 || M > RVT_MAX_VARIANTS)
    return fail(c, RVT_E_INVALID, "bad arguments");
  if (!c->d_vcf_rows) return fail(c, RVT_E_STATE, "rvt_vcf_set_samples first");
  hipSetDevice(c->device);
  const int64_t N = c->vcf_n_rows;
  hipStream_t st = c->io_stream;
  HIP_TRY(c, sync_stream(st));
  double* d_out = nullptr;
  HIP_TRY(c, hipMalloc((void**)&d_out, sizeof(double) * (size_t)N * M));
  VcfGene vg{sample_text, text_len, tag_index, gd_index, gq_index};
  int rc = io_err_ready(c);
  if (!rc) {
    c->h_io_err[rvt_ctx::kAfSlots] = 0;
    rc = vcf_decode_gene(c, &vg, M, N, st, rvt_ctx::kAfSlots, d_out, N);
  }
  hipError_t e = rc ? hipSuccess : hipMemcpyAsync(out, d_out, sizeof(double) * (size_t)N * M, hipMemcpyDeviceToHost, st);
  if (!rc && e == hipSuccess) e = sync_stream(st);
  hipFree(d_out);
  if (rc) return rc;
  HIP_TRY(c, e);
  if (const int k = c->h_io_err[rvt_ctx::kAfSlots]) {
    c->h_io_err[rvt_ctx::kAfSlots] = 0;
    io_err_message(c, k, false, "");
    return RVT_E_INVALID;
  }
  return RVT_OK;
}

// hand the first n queue entries (all launched and finished) to the caller, recycle their blocks
static void pop_collected(rvt_ctx* c, int n, rvt_gene_result* out) {
  while (!c->launched.empty() && c->launched.front().first < (size_t)n) {
    rvt_ctx::Launched& L = c->launched.front();
    for (int g = 0; g < L.n; ++g) c->queue[L.first + g].res = L.res[g];
    c->launched.pop_front();
  }
  for (int g = 0; g < n; ++g) {
    out[g] = c->queue[g].res;
    if (c->queue[g].io_error) {
      // the gene's VCF text / BGEN blocks were malformed: what was computed from the partly decoded block is void.  The
      // record keeps its place in the order, says so (RVT_ST_INPUT_ERROR: every test NA) and the message names the record.
      rvt_gene_result r;
      std::memset(&r, 0, sizeof(r));
      r.gene_id = c->queue[g].id;
      r.n_variants = c->queue[g].M;
      r.status = RVT_ST_INPUT_ERROR | RVT_ST_NO_POLY;
      // (every p-value field NaN, not 0: a consumer that prints p-values without looking at the *_ok flags must not read
      //  "most significant")
      r.skat_p = r.skato_p = r.cmc_p = r.zeg_p = r.perm_pvalue = r.famskat_p = r.famcmc_p = r.famzeg_p = r.vt_p = NAN;
      out[g] = r;
      char who[64];
      snprintf(who, sizeof(who), " of gene %lld", (long long)c->queue[g].id);
      io_err_message(c, c->queue[g].io_error, c->queue[g].decoded == 2, who);
    }
    (c->queue[g].kind == 3 ? c->pk_pool : c->block_pool).emplace_back(c->queue[g].bytes, c->queue[g].dG);
  }
  c->queue.erase(c->queue.begin(), c->queue.begin() + n);
  for (auto& L : c->launched) L.first -= (size_t)n;
  {  // keep the free blocks for the next window, bounded by their BYTES (a count bound would free and re-allocate
     // half of a 256-gene window every time: hipMalloc + memset of a 200 MB block is ~3 ms)
    static const size_t kPoolBytes = [] {  // a third of the device memory (96 GB of the MI355X's 288)
      size_t fr = 0, tot = 0;
      return (hipMemGetInfo(&fr, &tot) == hipSuccess && tot > 0) ? tot / 3 : (size_t)16 << 30;
    }();
    size_t total = 0;
    for (auto& bp : c->block_pool) total += bp.first;
    while (!c->block_pool.empty() && (total > kPoolBytes || c->block_pool.size() > 4096)) {
      total -= c->block_pool.back().first;
      hipFree(c->block_pool.back().second);
      c->block_pool.pop_back();
    }
    while (c->pk_pool.size() > 4096) {
      hipFree(c->pk_pool.back().second);
      c->pk_pool.pop_back();
    }
  }
}

int rvt_collect(rvt_ctx* c, rvt_gene_result* out, int cap, int* n_out) {
  if (!c || !out || !n_out) return RVT_E_INVALID;
  *n_out = 0;
  hipSetDevice(c->device);
  const int n = (int)std::min<size_t>(c->queue.size(), (size_t)cap);
  if (n == 0) return RVT_OK;
  int rc = launch_pending(c, (size_t)n, false);
  if (!rc) rc = rvt_sync(c);
  if (rc) return rc;
  pop_collected(c, n, out);
  *n_out = n;
  return RVT_OK;
}

int rvt_collect_ready(rvt_ctx* c, rvt_gene_result* out, int cap, int* n_out) {
  if (!c || !out || !n_out) return RVT_E_INVALID;
  TraceScope ts_c(c, &c->tr_collect);
  *n_out = 0;
  hipSetDevice(c->device);
  // batches whose stream has run dry are finished: take their records without waiting for anything else
  for (auto& sl : c->slots)
    if (sl.pending_out && hipStreamQuery(sl.stream) == hipSuccess) {
      int rc = finish_slot(c, sl);
      if (rc) return rc;
    }
  (void)hipGetLastError();  // hipErrorNotReady of the queries is not an error
  int n = 0;
  for (const auto& L : c->launched) {  // the finished prefix of the submission order
    if (!L.done || L.first != (size_t)n || n + L.n > cap) break;
    n += L.n;
  }
  if (n == 0) return RVT_OK;
  pop_collected(c, n, out);
  *n_out = n;
  return RVT_OK;
}

}  // extern "C"
