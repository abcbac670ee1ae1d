// rvtests_amd — what the engine's translation units share: the context, its pools and streams, the small host helpers
// (error reporting, stream waits, arenas, launch timing, the split of the sample axis) and the kernel headers.  Everything here
// has internal linkage or is a type: each translation unit (rvt_engine.hip, rvt_stream.hip) gets its own copy of the helpers;
// what one unit calls in another is declared at the end.
#pragma once
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <deque>
#include <map>
#include <string>
#include <unordered_map>
#include <thread>
#include <vector>
#include <cmath>
#include <functional>
#include "kernels.hip.h"
#include "suffstat_lat.hip.h"
#include "suffstat_hcp.hip.h"
#include "suffstat_hcx.hip.h"
#include "suffstat_fdx.hip.h"
#include "host_stage.h"

namespace rvt {  // defined in k2_unweighted.hip / k2_weighted.hip
void k2_launch_group_w0(int group, dim3 grid, hipStream_t st, const GeneDesc* d_desc, const int* list, int n_wparts,
                        NullDev nd, long long N, long long ld, int d);
void k2_launch_group_w1(int group, dim3 grid, hipStream_t st, const GeneDesc* d_desc, const int* list, int n_wparts,
                        NullDev nd, long long N, long long ld, int d);
void k2_launch_panel_w0(dim3 grid, hipStream_t st, const GeneDesc* d_desc, NullDev nd, long long N, long long ld,
                        int d);
void k2_launch_panel_w1(dim3 grid, hipStream_t st, const GeneDesc* d_desc, NullDev nd, long long N, long long ld,
                        int d);
void k2_launch_hc(int MT, dim3 grid, hipStream_t st, const GeneDesc* d_desc, NullTile nt, long long N, long long ld,
                  int d);
void k2_launch_hcw(int MT, dim3 grid, hipStream_t st, const GeneDesc* d_desc, NullTileW nt, long long N, long long ld,
                   int d);
void k2_launch_hcx(int MT, dim3 grid, hipStream_t st, const GeneDesc* d_desc, const NullTileX& nt, long long N, long long ld,
                   int d);
void k2_launch_fdx(int MT, dim3 grid, hipStream_t st, const GeneDesc* d_desc, const NullTileF& nt, long long N, long long ld,
                   int d);
void k2_launch_lat(int MT, dim3 grid, hipStream_t st, const GeneDesc* d_desc, NullTile nt, double den, long long N,
                   long long ld, int d);
void k2_launch_hcp(int MT, dim3 grid, hipStream_t st, const GeneDesc* d_desc, NullTile nt, long long N, long long ld,
                   int d, const HcpPlanes* planes = nullptr, bool score_only = false);
void k2_launch_classify(dim3 grid, hipStream_t st, const double* G, long long N, long long ld, int M, int* flag);
}  // namespace rvt
#include "fam_kernels.hip.h"
#include "rot_gemm.hip.h"
#include "perm_kernels.hip.h"
#include "vcf_kernels.hip.h"
#include "bgen_kernels.hip.h"
#include "jacobi_kernels.hip.h"
#include "rvt_hyper.h"

using namespace rvt;

namespace {

constexpr int kMaxMT = 6;  // widest single-pass tile configuration instantiated below

struct Arena {  // grow-only bump allocator over one device allocation
  char* base = nullptr;
  size_t cap = 0, off = 0;
  void reset() { off = 0; }
  void* take(size_t bytes, size_t align = 256) {
    off = (off + align - 1) / align * align;
    void* p = base + off;
    off += bytes;
    return p;
  }
};

struct ProfEvent {
  int family;  // 0 suffstat (general fp64 kernel), 1 burden, 2 stats, 3 pvalue, 4 suffstat (hard-call kernel)
  hipEvent_t a, b;
};

}  // namespace

// One in-flight batch: its stream, workspace, pinned staging and where the results go.  Several slots let the
// latency-bound tail of batch i (eigen / p-value kernels, few waves) overlap the bandwidth-bound head of
// batch i+1 (sufficient statistics) on a second HIP stream.
struct Slot {
  hipStream_t stream = nullptr;
  Arena arena;
  char* h_stage = nullptr;
  size_t h_stage_cap = 0;
  rvt_gene_result* pending_out = nullptr;
  bool* pending_done = nullptr;  // set when the batch's records have been handed over (streaming interface)
  rvt_gene_result* h_results = nullptr;
  int pending_n = 0;
  unsigned long long seq = 0;  // launch order
};
constexpr int kSlots = RVT_MAX_INFLIGHT;
constexpr int kSlotsAll = kSlots;

struct rvt_ctx {
  int device = 0;
  Slot slots[kSlotsAll];
  unsigned long long launch_seq = 0;
  hipStream_t stream = nullptr;  // == slots[0].stream (set-up work, rvt_stream())
  hipStream_t io_stream = nullptr;  // host copies + consolidation of the streaming interface: never behind a batch
  hipStream_t k2_stream = nullptr;  // the sufficient-statistics launches of all slots serialise here
  hipStream_t k2b_stream = nullptr; // ... except the general-path launches of a batch that also has hard-call genes: a
                                    // handful of genes per launch cannot fill the chip, so they run beside the others
  bool cu_partitioned = false;
  // ---- related samples (FastLMM null + FamSKAT) ----
  bool have_kin = false, have_fam = false;
  int64_t kin_N = 0;
  // eigenvectors of the kinship as kRotPlanesU signed base-128 digit planes (rot_gemm.hip.h): plane p at
  // d_Uq + p * uq_plane, row k (= column k of U) at k * uq_ldk; scaled by 2^uq_sexp
  signed char* d_Uq = nullptr;
  // sparse form of U (<= 64 non-zeros per eigenvector on average: families in any sample order), column-compressed;
  // used by the rotation when the K-chunk ranges (d_uq_range) do not apply; null otherwise
  long long* d_csc_ptr = nullptr;
  int* d_csc_rows = nullptr;
  double* d_csc_vals = nullptr;
  int2* d_uq_range = nullptr;  // per 256-row panel of the planes: K chunks [x, y) that hold its non-zeros; null = dense U
  double uq_visit = 1.0;       // fraction of the K chunks the rotation visits (1 = dense)
  size_t uq_plane = 0;
  int64_t uq_ldk = 0, uq_rows_pad = 0;
  int uq_sexp = 0;
  signed char* d_rotB = nullptr;  // digit planes of the columns being rotated (B side of an integer-plane product)
  size_t rotB_cap = 0;
  signed char* d_rotA = nullptr;  // A side of gemm_tn_planes
  size_t rotA_cap = 0;
  double* d_rot_scale = nullptr;  // per-column scale (RVT_ROT_MAXCOLS doubles) | column maxima
  int* d_rot_sexp = nullptr;
  double* d_S = nullptr;   // N raw eigenvalues
  double* d_u1 = nullptr;  // U'1
  std::vector<double> h_S, h_u1;
  double* d_uxy = nullptr;  // N x (d+1): U'X | U'y
  double* d_lmm_part = nullptr;
  NullConsts fam_nc;
  NullConsts* d_fam_nc = nullptr;
  double *d_fX = nullptr, *d_frr = nullptr, *d_fv = nullptr, *d_fzeros = nullptr, *d_fbeta = nullptr;
  // family MetaCov null set + constants
  NullConsts famcov_nc;
  NullConsts* d_famcov_nc = nullptr;
  double *d_cX = nullptr, *d_cv = nullptr;
  double fam_delta = 0.0;   // delta of the fitted FastLMM null (rvt_fam_null_summary)
  double famcov_b2 = 1.0;  // MetaCovFamBinary: b^2
  double famcov_k1r = 0.0; // u1' D uResid
  double* d_cr = nullptr;  // uResid (the rr column of the family-covariance null set)
  double famcov_c11 = 0.0, famcov_c1x[RVT_MAX_COV], famcov_zz[RVT_MAX_COV * RVT_MAX_COV],
         famcov_zzinv[RVT_MAX_COV * RVT_MAX_COV];
  double* d_Gp = nullptr;  // flipped / filtered genotypes of a FamSKAT batch (ld x T)
  double* d_Gt = nullptr;  // ... rotated by U'
  char* d_bedbatch = nullptr;   // rvt_submit_genes kind 7: row / gene references and partial counts of one call
  size_t bedbatch_cap = 0;
  bool no_i8_pack = false;      // submit_common: this int8 gene holds a value above 2 — send its bytes
  // gene_tnull_hcp: eight digit planes of [X | rr] in operand order + the columns' scales (quantitative null models)
  unsigned char* d_hcp_xq = nullptr;
  double* d_hcp_scale = nullptr;
  int hcp_planes_state = 0;     // 0 not built for this null model, 1 ready, -1 the model's columns cannot be carried
  int submit_group = 32;   // genes per asynchronous sub-batch of the streaming interface (RVT_SUBMIT_GROUP, rvt_set_submit_group)
  size_t fam_cols_cap = 0;
  int64_t fam_cols_ld = 0;  // (the leading dimension d_Gp / d_Gt were sized for)
  // raw / packed genotype submission
  double* d_consol_af = nullptr;  // af (RVT_MAX_VARIANTS) | fill values (RVT_MAX_VARIANTS)
  size_t consol_af_cap = 0;
  ConsolPart* d_consol_parts = nullptr;
  size_t consol_parts_cap = 0;
  // allele frequencies of raw / packed submissions whose caller did not ask for them: written into a ring slot and
  // copied back asynchronously; resolved (one stream wait) when the gene's group is launched
  static constexpr int kAfSlots = 128;
  double* h_af_ring = nullptr;  // kAfSlots x RVT_MAX_VARIANTS, pinned and device-mapped: the kernels write it directly
  unsigned long long af_seq = 0;
  int af_unresolved = 0;
  void* d_consol_i8 = nullptr;
  size_t consol_i8_cap = 0;
  // packed hand-offs from the host (int8 / 2-bit): a ring of landing buffers and a copy stream of their own, so that the DMA of
  // gene g + 1 runs while the consolidation kernels of gene g read another buffer (one stream serialised them: 3.3 k
  // 2-bit genes/s where the link carries 8 k)
  static constexpr int kPack = 16;  // landing buffers: a consolidation delayed by a batch launch does not stop the copies
  void* d_pack[kPack] = {};
  size_t pack_cap[kPack] = {};
  hipEvent_t ev_pack_copied[kPack] = {}, ev_pack_free[kPack] = {};
  int pack_next = 0;
  hipStream_t copy_stream = nullptr;
  hipStream_t h2d_stream = nullptr;  // where staged_h2d enqueues: io_stream, or copy_stream for the packed hand-offs
  double* d_rot_part = nullptr;  // split-K partial results of the integer GEMM
  int band_last_path = -1;       // which product the last rvt_cov_band took (rvt_cov_band_last_path)
  double* d_mu_nan = nullptr;    // packed_columns_pass: the other values of up to kColQueue columns (NaN = none)
  char* d_colpack = nullptr;     // rvt_block_upload_columns: the columns as 2-bit rows + their other values, before they are expanded
  size_t colpack_cap = 0;
  // single columns uploaded one call at a time (MetaCovTest / MetaScoreTest: one site per fit()) are packed into pinned memory
  // and QUEUED: the DMA, the expansion and the column pass run once per kColQueue consecutive columns (flush_col_queue) — the
  // per-call device work (eight HIP calls, ~70 us) was what held the adapter at 9 k sites/s.  Everything that reads a block
  // flushes first (rvt_sync, the column operations).
  static constexpr int kColQueue = 32;
  struct ColQueue {
    unsigned char* h[2] = {nullptr, nullptr};  // pinned, kColQueue rows of `pitch` bytes each
    hipEvent_t ev[2] = {nullptr, nullptr};     // recorded behind the DMA that read h[k]
    bool used[2] = {false, false};
    size_t pitch = 0;
    int cur = 0, n = 0, col0 = 0;
    double* dG = nullptr;
    double mu[kColQueue];
    int hard[kColQueue];
  } colq;
  char* d_cov_work = nullptr;    // work space of the MetaCov rectangles (S, T, the band, column statistics): grow-only
  hipEvent_t ev_band_fin[2] = {}, ev_band_copied[2] = {};  // rvt_cov_band: a pass's rows are copied out while the next pass multiplies
  size_t cov_work_cap = 0;
  size_t rot_part_cap = 0;
  // per-column content flags of blocks filled column by column (rvt_block_upload_columns): nonzero = hard calls only
  // Round 5: ... and, for a ring that MetaCov will read (unweighted model), what the column pass of the hard-call band would
  // compute for the column anyway — the int8 copy, the column sum, the polymorphic flag and its row of T = G'X — made by the
  // pass that classifies the column behind its PCIe copy and kept with the block (moved and copied with its columns), so that a
  // flush starts at the integer product.  `valid[j]`: column j's entries were made under null model number `gen`.
  struct ColKind {
    int cols = 0;
    int* d_flags = nullptr;
    signed char* d_i8 = nullptr;  // [cols rounded up + a tile of slack][ldk]
    unsigned char* d_i4 = nullptr;  // [cols][ldk4]: the same hard calls as E2M1 codes, two per byte (band_gemm.hip.h, FP4)
    unsigned char* d_m4 = nullptr;  // [cols][ldk4]: round 6 — the mask of a column's ONE other value (the imputed mean), same codes
    double* d_mu = nullptr;         // [cols]: that other value (0 for a column without one)
    int64_t ldk4 = 0;
    bool cache_failed = false;  // the cache could not be allocated once: not retried for this block
    double* d_cs = nullptr;       // [cols] column sums
    int* d_poly = nullptr;        // [cols]
    double* d_T = nullptr;        // [cols][RVT_MAX_COV]
    int64_t ldk = 0;
    uint64_t gen = 0;
    std::vector<unsigned char> valid;  // per column: 0 nothing known / not usable, 1 hard calls only, 2 hard calls + one other value
    void release() {
      for (void* q : {(void*)d_flags, (void*)d_i8, (void*)d_i4, (void*)d_m4, (void*)d_mu, (void*)d_cs, (void*)d_poly, (void*)d_T})
        if (q) hipFree(q);
      d_flags = nullptr;
      d_i8 = nullptr;
      d_i4 = nullptr;
      d_m4 = nullptr;
      d_mu = nullptr;
      d_cs = nullptr;
      d_poly = nullptr;
      d_T = nullptr;
      valid.clear();
    }
  };
  int handed_back_recently = 0;   // > 0: a hard-call gene was handed back within the last 16 batches (launch_suffstat's list grid)
  uint64_t null_gen = 0;          // counts rvt_set_null / rvt_fit_null: a column cache made under another model is not used
  double* d_cc_part = nullptr;    // slice partials of the one-column pass (64 slices x (RVT_MAX_COV + 3))
  std::unordered_map<const double*, ColKind> col_kind;
  // VCF text front end (vcf_kernels.hip.h)
  char* d_vcf_text = nullptr;   // the text / block buffer of the gene being submitted: text_buf[text_cur]
  static constexpr int kTextBufs = 3;  // ring: the copies of gene g + 1 run while the decode kernels of gene g read theirs
  char* text_buf[kTextBufs] = {};
  size_t text_buf_cap[kTextBufs] = {};
  hipEvent_t ev_text_copied[kTextBufs] = {}, ev_text_free[kTextBufs] = {};
  int text_next = 0, text_cur = 0;
  size_t vcf_text_cap = 0;
  VcfRecord* d_vcf_rec = nullptr;
  int* d_vcf_seg = nullptr;
  size_t vcf_seg_cap = 0;
  int* d_vcf_rows = nullptr;   // output row of every sample column of the file (-1: not analysed)
  int vcf_n_file = 0;
  int64_t vcf_n_rows = 0;      // rows the map addresses (must equal the null model's N)
  VcfFilters vcf_flt{0, 0, 0, 0};
  std::vector<int> vcf_alt;    // rvt_vcf_set_alt_alleles: alternative-allele index per record of the NEXT VCF call
  bool vcf_dosage = false;     // rvt_vcf_set_dosage: the index handed over is a dosage tag's, values through atof
  // BGEN probability blocks (bgen_kernels.hip.h); the blocks are staged in d_vcf_text
  char* d_fam_list = nullptr;  // rvt_run_fam_tests: column pointers + flags of a batch (grow-only)
  size_t fam_list_cap = 0;
  signed char* d_vcf_sex = nullptr;  // PLINK sex code per file sample (rvt_vcf_set_sex); hemizygous records only
  std::vector<int> vcf_hemi;         // per record of the NEXT decode call (rvt_vcf_set_hemi)
  BgenRecord* d_bgen_rec = nullptr;
  long long* d_bgen_seg = nullptr;
  size_t bgen_seg_cap = 0;
  // pinned, device-visible input-error words of the VCF / BGEN decoders: record index + 1 of a record with a wrong column
  // count (negative: a dosage the device cannot round exactly) resp. variant index + 1 of a block shorter than its ploidy
  // bytes demand.  One word per allele-frequency ring slot (the streaming submissions: read when the gene's frequencies
  // are resolved, so the error lands on the gene that caused it) + word kAfSlots for the synchronous calls.
  int* h_io_err = nullptr;
  // ---- SKAT permutations: the emulated glibc rand() stream (TYPE_3), oldest word first ----
  // Permutation mode: exact (the DEFAULT of a single context: `--kernel skat[nPerm=..]` reproduces the reference's
  // ActualPerm / NumGreater / NumEqual / PermPvalue) = the reference's own rand() stream replayed (one sequential stream in
  // gene order: bit-identical counters, ~3 k shuffles/s at N = 500 000); counter-based = permutations keyed by (seed, gene
  // id, shuffle) (perm_counter.h: statistical parity — SURVEY 8e grants it to SHARDED runs only —, any context / device /
  // gene order, ~10^5 shuffles/s): selected by rvt_set_perm_exact(ctx, 0) / RVT_PERM_EXACT=0, and by a device group that
  // deals genes over more than one member (rvt_group_init).
  bool perm_exact = true;
  uint64_t perm_seed = 1;
  double* d_pc_part = nullptr;  // counter mode: partial products [slice][shuffle][variant]
  size_t pc_part_cap = 0;
  double* d_pc_Q = nullptr;
  uint32_t rand_state[31];
  int64_t jump_N = -1;                 // J = A^(jump_N - 1) is cached for this sample count
  std::vector<uint32_t> jump;          // 31 x 31, row-major
  uint32_t* d_perm_idx = nullptr;      // N x B
  uint32_t* d_perm_states = nullptr;   // B x 31
  double *d_perm_R = nullptr, *d_perm_C = nullptr, *d_perm_Q = nullptr, *d_perm_cur = nullptr;
  size_t perm_cap_NB = 0, perm_cap_BM = 0, perm_cap_N = 0;
  int perm_cap_B = 0;
  hipEvent_t ev_in[kSlotsAll] = {}, ev_k2[kSlotsAll] = {}, ev_k2b[kSlotsAll] = {};
  // The p-value kernel on CUs of its own (RVT_PV_CUS, see rvt_init): two streams restricted to the first pv_cus mask bits,
  // used by alternate batches; the batch's stream hands over by event and takes the records back by event.
  hipStream_t pv_stream[2] = {nullptr, nullptr};
  hipEvent_t ev_pv_in[kSlotsAll] = {}, ev_pv_out[kSlotsAll] = {};
  int pv_cus = 0;
  unsigned pv_turn = 0;
  std::vector<int> pv_order;  // scratch of run_batch: the batch's genes by falling M (GeneDesc::pv_gene)
  // host -> device copies of the streaming interface: pinned staging ring filled by the process-wide copy threads
  // (host_stage.h), drained by DMA on io_stream.  RVT_STAGE=0 restores the runtime's own pageable copies.
  static constexpr int kStageChunks = 4;
  static constexpr size_t kStageBytes = (size_t)32 << 20;
  StageRing stage;
  hipEvent_t stage_ev[kStageChunks] = {};
  bool stage_on = true;
  static constexpr int kSmallSlots = 8;
  static constexpr size_t kSmallBytes = (size_t)128 << 10;
  char* h_small = nullptr;      // pinned: kSmallSlots x kSmallBytes (record tables of the VCF / BGEN decoders)
  hipEvent_t small_ev[kSmallSlots] = {};
  int small_next = 0;
  // RVT_TRACE_SUBMIT=1: host seconds spent in the phases of the streaming submissions, printed by rvt_destroy
  bool trace_submit = false;
  double tr_block = 0, tr_copy = 0, tr_consol = 0, tr_af = 0, tr_launch = 0, tr_collect = 0;
  long long tr_genes = 0;
  hipEvent_t ev_io = nullptr;   // recorded on io_stream when a group of submitted genes is launched
  bool io_wait_pending = false;  // the next batch waits for ev_io (its blocks may still be crossing the link)
  std::string err;
  // null model
  bool have_null = false;
  NullConsts nc;
  NullConsts* d_nc = nullptr;
  double *d_X = nullptr, *d_res = nullptr, *d_rr = nullptr, *d_v = nullptr, *d_zeros = nullptr;
  double* d_nulltile = nullptr;  // ONE allocation [X_0 .. X_{d-1} | rr | zeros]: d_X, d_rr and d_zeros point into it
  // binary trait: the weighted hard-call kernel's tile [vX_0 .. vX_{d-1} | res | v | zeros] and the digit planes of v
  double* d_nulltile_w = nullptr;
  unsigned char* d_vq = nullptr;
  // ... and the integer operands of the workgroup-cooperative kernel (suffstat_hcx.hip.h): the digits of v and of the null
  // tile in operand order, the power-of-two scale of every null column (host copy in hcx_tile, device copy for gene_assemble)
  unsigned char *d_dq = nullptr, *d_xq = nullptr;
  double* d_xscale = nullptr;
  NullTileX hcx_tile;
  bool hcx_ok = false;
  // ... and of the float-digit dosage kernel (suffstat_fdx.hip.h; quantitative trait): five base-256 digit planes of [X | res | 1]
  bool reg_defer = false;  // inside rvt_submit_genes: the wait for the DMAs out of registered caller memory comes once, at the end
  unsigned char* d_fxq = nullptr;
  NullTileF fdx_tile;
  bool fdx_ok = false;
  bool dosage_float = false;  // rvt_set_dosage_float: blocks of unknown content hold float-precision dosages
  int as_threads = 256;   // workgroup size of gene_assemble_kernel (RVT_AS_THREADS).  Round 5: 256 instead of 1024 — many of its phases
                          // keep M threads busy while the rest wait at the barriers, and with the eigenvalue stage cut down its
                          // resident waves were 56 % of the per-gene tail (configs[1]: 347 k -> 369 k; the other shapes unchanged)
  bool hcx_fused = true;  // one launch for every tile class of a batch (gene_suffstat_hcx_any); RVT_HCX_FUSED=0: one per class
  int64_t null_ld = 0;
  // Which sufficient-statistics kernel a gene STARTS on is a prediction, never a trust: the hard-call kernel tests every
  // double it loads and hands back genes that hold anything but hard calls and one imputed value per column
  // (gene_flags_hc_kernel); those are computed by the general fp64 kernel in the same batch, through a conditional launch
  // that follows the hard-call launches on the same stream.  The engine's own decoders say what they wrote (hint per
  // gene: 1 hard calls / imputed, 0 dosages); blocks of unknown content (fp64 from the caller) start on the hard-call
  // kernel unless the caller has said they hold dosages (rvt_set_content_hint).  No history: the kernel a block runs on
  // — and with it the last bits of its records — depends on the block and the hint alone.
  int content_hint = -1;
  // rvt_host_register: host ranges of the caller that are page-locked — copies out of them are DMA straight from the
  // caller's memory (no staging copy by the CPU); `reg_pending`: such a copy has been enqueued and not yet waited for
  std::vector<std::pair<const char*, size_t>> host_reg;
  std::vector<char> host_reg_owned;  // (1: this context called hipHostRegister; 0: adopted from another member of a group)
  hipEvent_t ev_reg = nullptr;
  hipStream_t reg_stream = nullptr;  // the stream ev_reg was last recorded on
  bool reg_pending = false;
  int lattice_den = 0;        // rvt_set_dosage_lattice: dosage doubles are multiples of 1 / lattice_den (0: not stated)
  int* d_kind = nullptr;      // device flag of rvt_block_classify (a stateless query)
  bool hc_enabled = true;     // RVT_HARDCALL=0 forces the general kernel (experiments)
  double null_beta[RVT_MAX_COV] = {};  // estimates of the model rvt_fit_null fitted
  bool have_null_beta = false;
  // streaming interface
  struct Pending {
    int64_t id;
    int M;
    double* dG;
    size_t bytes;  // capacity of dG
    std::vector<double> af;
    uint32_t tests;
    rvt_params prm;
    rvt_gene_result res;  // filled by the batch this gene was launched in (the queue is a deque: stable addresses)
    bool launched;
    int af_slot = -1;     // >= 0: the allele frequencies are still on their way back from the device (af_ring slot)
    int io_error = 0;     // != 0: the gene's VCF text / BGEN blocks were malformed (h_io_err): its record is void
    int decoded = 0;      // 1: VCF text, 2: BGEN blocks (the submission has an input-error word in its ring slot)
    bool planes = false;  // kind 3 only: G'[X | rr] on the int8 matrix cores from the null tile's digit planes (resident .bed genes)
    int kind = -1;        // what the engine's decoder wrote: 1 hard calls (+ imputed means), 0 dosages (BGEN), 2 decimal
                          // dosages (VCF text), 3 the block holds PLINK 2-bit rows (not doubles), -1 unknown
  };
  std::deque<Pending> queue;
  std::vector<std::pair<size_t, double*>> block_pool;  // free device blocks of the streaming interface (bytes, ptr)
  std::vector<std::pair<size_t, double*>> pk_pool;     // free PACKED blocks (2-bit rows): never handed out as fp64 blocks,
                                                       // whose pad rows must be zero
  // results of launched sub-batches land in contiguous arrays, then move into Pending::res at collect time
  struct Launched {
    size_t first;  // index into the queue at launch time (adjusted when the queue is popped)
    int n;
    std::vector<rvt_gene_result> res;
    bool done = false;  // the batch has finished and `res` is filled (set by finish_slot)
  };
  bool* next_done_flag = nullptr;  // handed to the slot of the next run_batch call (launch_group)
  std::deque<Launched> launched;
  // profiling
  bool profiling = false;
  std::vector<ProfEvent> events;
  std::vector<hipEvent_t> event_pool;
  rvt_timing timing;
  size_t eigen_lds_max = 48 * 1024;
};

namespace {

int fail(rvt_ctx* c, int code, const char* fmt, ...) {
  if (c) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    c->err = buf;
  }
  return code;
}

#define HIP_TRY(ctx, call)                                                                      \
  do {                                                                                          \
    hipError_t e_ = (call);                                                                     \
    if (e_ != hipSuccess) return fail(ctx, RVT_E_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
  } while (0)

// Wait for a stream by polling.  A blocking hipStreamSynchronize() relies on a completion interrupt; on a fresh box
// one such wake-up was seen to go missing (the kernels had long finished while the host slept for seconds), so the
// engine never blocks in the runtime: it spins briefly, then sleeps in 20 us slices between hipStreamQuery calls.
hipError_t sync_stream(hipStream_t st) {
  hipError_t e;
  int spins = 0;
  while ((e = hipStreamQuery(st)) == hipErrorNotReady) {
    if (++spins > 64) {
      struct timespec ts = {0, 20000};
      nanosleep(&ts, nullptr);
    }
  }
  return e;
}

int ensure_arena(rvt_ctx* c, Slot& sl, size_t bytes) {
  if (sl.arena.cap >= bytes) return RVT_OK;
  if (sl.arena.base) {
    HIP_TRY(c, sync_stream(sl.stream));
    HIP_TRY(c, hipFree(sl.arena.base));
    sl.arena.base = nullptr;
    sl.arena.cap = 0;
  }
  const size_t want = bytes + bytes / 4;
  HIP_TRY(c, hipMalloc((void**)&sl.arena.base, want));
  sl.arena.cap = want;
  // tests: fill the workspace with a byte pattern — no result may depend on what a fresh allocation happens to hold
  if (const char* e = getenv("RVT_POISON")) HIP_TRY(c, hipMemset(sl.arena.base, atoi(e) & 0xff, want));
  return RVT_OK;
}

int ensure_stage(rvt_ctx* c, Slot& sl, size_t bytes) {
  if (sl.h_stage_cap >= bytes) return RVT_OK;
  if (sl.h_stage) {
    HIP_TRY(c, sync_stream(sl.stream));
    HIP_TRY(c, hipHostFree(sl.h_stage));
    sl.h_stage = nullptr;
  }
  const size_t want = bytes * 2;
  HIP_TRY(c, hipHostMalloc((void**)&sl.h_stage, want, hipHostMallocDefault));
  sl.h_stage_cap = want;
  return RVT_OK;
}

// wait for one slot's batch and hand its records to the caller
int finish_slot(rvt_ctx* c, Slot& sl) {
  HIP_TRY(c, sync_stream(sl.stream));
  if (sl.pending_out) {
    // (kStatusHandedBack is bookkeeping: the gene started on the hard-call kernel and was computed by the fp64 kernel)
    bool any_back = false;
    for (int g = 0; g < sl.pending_n; ++g) {
      if (sl.h_results[g].status & kStatusHandedBack) {
        sl.h_results[g].status &= ~kStatusHandedBack;
        any_back = true;
        if (c->profiling) ++c->timing.genes_handed_back;
      }
    }
    // how wide the NEXT batches launch the general kernel over their hand-back lists (launch_suffstat): a stream that has
    // handed nothing back for a while gets a grid of 32 waves instead of 1 024
    if (any_back) c->handed_back_recently = 16;
    else if (c->handed_back_recently > 0) --c->handed_back_recently;
    std::memcpy(sl.pending_out, sl.h_results, sizeof(rvt_gene_result) * sl.pending_n);
    if (sl.pending_done) *sl.pending_done = true;
    sl.pending_done = nullptr;
    sl.pending_out = nullptr;
    sl.pending_n = 0;
  }
  return RVT_OK;
}

hipEvent_t get_event(rvt_ctx* c) {
  if (!c->event_pool.empty()) {
    hipEvent_t e = c->event_pool.back();
    c->event_pool.pop_back();
    return e;
  }
  hipEvent_t e;
  hipEventCreate(&e);
  return e;
}

struct Scope {  // times one kernel launch with HIP events on the launching stream when profiling is on
  rvt_ctx* c;
  int fam;
  hipStream_t st;
  hipEvent_t a = nullptr;
  Scope(rvt_ctx* c_, int fam_, hipStream_t st_) : c(c_), fam(fam_), st(st_) {
    if (c->profiling) {
      a = get_event(c);
      hipEventRecord(a, st);
    }
  }
  ~Scope() {
    if (c->profiling) {
      hipEvent_t b = get_event(c);
      hipEventRecord(b, st);
      c->events.push_back({fam, a, b});
    }
  }
};

void drain_events(rvt_ctx* c) {
  for (auto& e : c->events) {
    float ms = 0.f;
    hipEventElapsedTime(&ms, e.a, e.b);
    switch (e.family) {
      case 0: c->timing.ms_suffstat += ms; c->timing.n_suffstat_launches++; break;
      case 4:
        c->timing.ms_suffstat += ms;
        c->timing.n_suffstat_launches++;
        c->timing.ms_suffstat_hc += ms;
        c->timing.n_suffstat_hc_launches++;
        break;
      case 1: c->timing.ms_burden += ms; c->timing.n_burden_launches++; break;
      case 2: c->timing.ms_stats += ms; c->timing.n_stats_launches++; break;
      case 5: break;  // conditional launches of the general kernel over hard-call genes (mostly empty): not counted
      default: c->timing.ms_pvalue += ms; c->timing.n_pvalue_launches++; break;
    }
    c->event_pool.push_back(e.a);
    c->event_pool.push_back(e.b);
  }
  c->events.clear();
}

// genes [0, n) of one register-budget group (kernels.hip.h: suffstat_group)
// list != nullptr: the genes come from a device work list (handed-back hard-call genes, suffstat_kernels.hip.h); the
// launch is then a fixed small grid whose workgroups loop over the list — empty almost always.
void launch_suffstat(rvt_ctx* c, hipStream_t st, int group, const GeneDesc* d_desc, int n, int max_wparts,
                     const NullDev& nd, const int* list = nullptr) {
  Scope sc(c, list ? 5 : 0, st);
  // Waves are independent (no LDS, no barriers), so a workgroup is ONE wave: the dispatcher can then place the
  // wide classes (one wave fills a SIMD's register file) on any free SIMD, instead of needing four free SIMDs on
  // one CU at once — which a single long-lived p-value wave per CU would block for its whole lifetime.
  dim3 grid(max_wparts, n);
  // (the list is empty almost always, but every wave of these kernels needs a SIMD's whole register file before it can start
  //  and leave: 1 024 of them took 2 ms to trickle through a chip that is busy with the next batch's streaming kernel, and the
  //  batch's tail waited behind them.  While no batch has handed a gene back for 16 batches the grid is 32 waves; a stream that
  //  does hand genes back — dosages the caller did not announce — keeps the wide grid.)
  if (list) grid = dim3(c->handed_back_recently > 0 ? 1024 : 32, 1);
  const long long N = c->nc.N, ld = c->nc.ld;
  const int d = c->nc.d;
  if (c->nc.binary)
    k2_launch_group_w1(group, grid, st, d_desc, list, max_wparts, nd, N, ld, d);
  else
    k2_launch_group_w0(group, grid, st, d_desc, list, max_wparts, nd, N, ld, d);
}

// glibc srandom_r / random_r for the default TYPE_3 generator: r[i] = 16807 r[i-1] mod (2^31 - 1) for the first 31
// words, then 310 outputs are discarded.  The state is kept normalised (oldest word first): a draw is
// x' = (x[1..30], x[0] + x[28]).
void seed_rand_state(uint32_t* x, unsigned seed) {
  int32_t r[34];
  r[0] = (int32_t)(seed == 0 ? 1 : seed);
  for (int i = 1; i < 31; ++i) {
    int64_t v = (16807LL * r[i - 1]) % 2147483647;
    if (v < 0) v += 2147483647;
    r[i] = (int32_t)v;
  }
  std::vector<uint32_t> o(344);
  for (int i = 0; i < 31; ++i) o[i] = (uint32_t)r[i];
  for (int i = 31; i < 34; ++i) o[i] = o[i - 31];
  for (int i = 34; i < 344; ++i) o[i] = o[i - 31] + o[i - 3];
  for (int i = 0; i < 31; ++i) x[i] = o[344 - 31 + i];
}

// inverse of a small (n <= RVT_MAX_COV) nonsingular matrix, row-major, Gauss-Jordan with partial pivoting
bool invert_spd(const double* M, int n, double* Minv) {
  double A[RVT_MAX_COV][2 * RVT_MAX_COV];
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) {
      A[i][j] = M[i * n + j];
      A[i][n + j] = (i == j) ? 1.0 : 0.0;
    }
  for (int k = 0; k < n; ++k) {
    int piv = k;
    for (int i = k + 1; i < n; ++i)
      if (fabs(A[i][k]) > fabs(A[piv][k])) piv = i;
    if (A[piv][k] == 0.0) return false;
    if (piv != k)
      for (int j = 0; j < 2 * n; ++j) std::swap(A[k][j], A[piv][j]);
    const double pv = A[k][k];
    for (int j = 0; j < 2 * n; ++j) A[k][j] /= pv;
    for (int i = 0; i < n; ++i)
      if (i != k) {
        const double f = A[i][k];
        if (f != 0.0)
          for (int j = 0; j < 2 * n; ++j) A[i][j] -= f * A[k][j];
      }
  }
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) Minv[i * n + j] = A[i][n + j];
  return true;
}

// 16-sample steps handled by one wave: large enough to amortise the partial-tile write, small enough
// that a gene still spreads over >= 32..128 waves
// n_genes: genes of the batch — a big batch fills the chip with fewer, longer waves per gene (half the partial tiles
// to write and to reduce: measured +2.7 % at 512 genes)
// hcx: the batch's hard-call genes take the workgroup-cooperative kernel (suffstat_hcx.hip.h): a part is worked by four loader
// waves, so a quarter of the parts gives as many waves per gene; its iteration is 16 steps
void choose_split(int64_t ld, int n_genes, bool weighted, int* n_wparts, int* steps_per, bool hcx = false, bool fdx = false) {
  const int64_t nsteps = ld >> 4;
  const char* fe = getenv("RVT_WPARTS");  // (experiments / tests)
  const int forced = fe ? atoi(fe) : 0;
  // (cooperative kernel: one workgroup per CU — about eight rounds of workgroups over the chip balance a launch; fewer, longer
  //  wave-parts measured better down to that: 47.8 k gene-sets/s with 16 parts, 49.6 k with 8, 50.4-50.9 k with 5-6 at N = 200 000)
  const int coop = std::min(32, std::max(4, (2048 + n_genes - 1) / std::max(n_genes, 1)));
  const int target = forced > 0 ? forced : (hcx ? coop : (n_genes >= 1024 ? 32 : (n_genes >= 128 ? 64 : 128)));
  int64_t spw = (nsteps + target - 1) / target;
  if (spw < 64) spw = 64;
  const int unit = hcx ? 3 * kHcxIterSteps : kHcStepUnit;     // (48: whole iterations of both kernels)
  spw = (spw + unit - 1) / unit * unit;                       // whole ring iterations of the hard-call kernel
  if (hcx && spw > kHcwMaxSteps) spw = kHcwMaxSteps / unit * unit;
  if (spw > kHcwMaxSteps) spw = kHcwMaxSteps;                   // (int32 range of the weighted hard-call kernel's tiles)
  if (!weighted && spw > kHcMaxSteps) spw = kHcMaxSteps;        // (16-bit range of the hard-call kernel's masked-tile counters)
  if (fdx && spw > kFdxMaxSteps / 24 * 24) spw = kFdxMaxSteps / 24 * 24;  // (int32 range of the float-digit kernel's order sums)
  int64_t nw = (nsteps + spw - 1) / spw;
  if (nw < 1) nw = 1;
  *n_wparts = (int)nw;
  *steps_per = (int)spw;
}

}  // namespace

// ---- defined in rvt_engine.hip, called from the other translation units (hidden: not part of the ABI) ------------------------
#define RVT_INTERNAL __attribute__((visibility("hidden")))
// ---- kernel families -----------------------------------------------------------------------------------------------------
// Every non-template kernel of the engine is `static` in a header that all five rvt_*.hip units see through this file; until
// round 6 each unit therefore compiled and shipped its own copy of all of them (76 kernels x 5 code objects).  Now a unit
// defines RVT_K_SPLIT and ONE of RVT_K_ENGINE / RVT_K_STREAM / RVT_K_FAM / RVT_K_PERM / RVT_K_META before including this
// header and gets the bodies of that family alone (tools and the host harness define nothing and get everything).  The few
// kernels that two units launch live in one of them; the other goes through these host launchers (tools/kernel_units.sh
// lists which object carries which kernel).
extern "C" RVT_INTERNAL int flush_col_queue(rvt_ctx* c);                                                      // rvt_meta.hip
RVT_INTERNAL void k_lmm_sums(dim3 grid, hipStream_t st, const double* uxy, const double* lam, long long N, int d, double delta,
                             int take_abs, double* partial);                                                   // rvt_fam.hip
RVT_INTERNAL void k_fam_colstat(dim3 grid, hipStream_t st, const double* const* cols, long long N, int* flags);  // rvt_fam.hip
RVT_INTERNAL void k_fam_flip_compact(dim3 grid, hipStream_t st, const double* const* src_cols, const int* src_flip, long long N,
                                     long long ld, double* dst);                                               // rvt_fam.hip
RVT_INTERNAL void k_raw_colstat(dim3 grid, hipStream_t st, const double* G, long long N, long long ld, double* colsum, int* poly);  // rvt_fam.hip
RVT_INTERNAL void k_rot_reduce_slices(hipStream_t st, const double* part, long long ldc, long long M, long long N, long long stride,
                                      int slices, double* C, int accumulate);                                  // rvt_fam.hip
RVT_INTERNAL void k_vt_integrate(dim3 grid, hipStream_t st, const GeneDesc* genes, int stage);                  // rvt_engine.hip
RVT_INTERNAL void k_vt_finish(dim3 grid, hipStream_t st, const GeneDesc* genes, int n, int stage);              // rvt_engine.hip
extern "C" {
RVT_INTERNAL int stage_ready(rvt_ctx* c);
RVT_INTERNAL bool host_registered(const rvt_ctx* c, const void* src, size_t bytes);
RVT_INTERNAL int reg_wait(rvt_ctx* c);
RVT_INTERNAL int small_h2d(rvt_ctx* c, void* dst, const void* src, size_t bytes);
RVT_INTERNAL int staged_h2d(rvt_ctx* c, void* dst, const void* src, size_t bytes);
RVT_INTERNAL int staged_h2d_2d(rvt_ctx* c, void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t rows,
                               bool pad_zero = false);
RVT_INTERNAL int upload_block_data(rvt_ctx* c, double* dG, int M, const double* G);
struct DebugOut;
struct CovOut;
// kind (optional, per gene): what the engine's own decoder wrote into the block — 1 hard calls (+ imputed means), 0 dosages,
// 2 decimal dosages, 3 packed 2-bit rows, -1 unknown
RVT_INTERNAL int run_batch(rvt_ctx* c, int n, const double* const* dG, const int* Ms, const double* af, const int64_t* ids,
                           uint32_t tests, const rvt_params* prm, rvt_gene_result* out, DebugOut* dbg, CovOut* cov = nullptr,
                           const signed char* kind = nullptr);
// ---- defined in rvt_fam.hip
RVT_INTERNAL int ensure_fam_cols(rvt_ctx* c, size_t T, int64_t ld);
RVT_INTERNAL int rotate_columns(rvt_ctx* c, const double* d_src, int64_t ld_src, int ncols, double* d_dst, int64_t ld_dst,
                                hipStream_t st);
RVT_INTERNAL int gemm_tn_planes(rvt_ctx* c, const double* dA, int64_t ldA, int nA, const double* dB, int64_t ldB, int nB,
                                int64_t n_rows, double* C, int64_t ldc, hipStream_t st);
RVT_INTERNAL int rvt_planes_gemm(rvt_ctx* c, const signed char* A, size_t a_stride, int PA, int nA, const int* row_exp, int a_exp,
                                 const signed char* B, size_t b_stride, int PB, int nB, const int* col_exp, int64_t n_rows,
                                 int64_t ldk, double* C, int64_t ldc, hipStream_t st, const int2* a_krange = nullptr);
// ---- defined in rvt_meta.hip: C = A' D [B | B2] in fp64 on the matrix cores (gemm_f64.hip.h)
RVT_INTERNAL int gemm_tn_f64(rvt_ctx* c, const double* A, int64_t lda, int M, const double* B, int64_t ldb, int Nb, const double* B2,
                             int64_t ldb2, int Nb2, const double* w, int64_t N, double* C, int64_t ldc, bool symmetric,
                             hipStream_t st, bool subtract = false, int halo = -1, int ring = 0, int col0 = 0);
// ---- defined in rvt_perm.hip
RVT_INTERNAL int rvt_kbac_stage(rvt_ctx* c, const double* dG, int M, const double* af, const std::vector<unsigned char>& y,
                                int nPerm, double alpha, rvt_kbac_result* r);
// ---- defined in rvt_engine.hip (continued)
RVT_INTERNAL int enqueue_classify(rvt_ctx* c, const double* dG, int M, int64_t N, int64_t ld, hipStream_t st, int* d_flag);
RVT_INTERNAL int cov_constants(rvt_ctx* c, bool fam, CovConsts* ccp, std::vector<double>* zzp);
RVT_INTERNAL int run_blocks_with_perm(rvt_ctx* c, int n, const double* const* dG, const int* M, const double* af,
                                      const int64_t* ids, uint32_t tests, const rvt_params* prm, rvt_gene_result* out);
}

struct RegWait {  // every entry point that copies out of the caller's memory ends with the wait
  rvt_ctx* c;
  explicit RegWait(rvt_ctx* c_) : c(c_) {}
  ~RegWait() {
    if (c && !c->reg_defer) (void)reg_wait(c);  // (rvt_submit_genes waits ONCE, behind its last gene)
  }
};

static inline double now_s() {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

struct TraceScope {  // RVT_TRACE_SUBMIT: adds the scope's host time to *acc
  double* acc;
  double t0;
  TraceScope(rvt_ctx* c, double* a) : acc(c->trace_submit ? a : nullptr), t0(acc ? now_s() : 0.0) {}
  ~TraceScope() {
    if (acc) *acc += now_s() - t0;
  }
};

struct DebugOut {
  int* flip = nullptr;
  int* kept = nullptr;
  double* cmc = nullptr;
  double* zeg = nullptr;
  double** parts_out = nullptr;  // device pointers of gene 0's partials etc.
  GeneDesc* desc0 = nullptr;
};

struct CovOut {  // rvt_cov_block: host destinations
  double* cov = nullptr;   // V x V
  double* xz = nullptr;    // V x d
  double* zz = nullptr;    // d x d
  int* poly = nullptr;     // V
  bool fam = false;        // family mode: the block is already rotated; raw column sums / flags are supplied
  const double* d_raw_colsum = nullptr;
  const int* d_raw_poly = nullptr;
  // family burden tests: per-column U, V, GLS allele frequency, p-value (host, V entries each); cov/xz may be null
  double *ustat = nullptr, *vstat = nullptr, *af = nullptr, *pval = nullptr;
  // MetaScoreTest (unrelated samples): ustat / vstat / effect / se / pval / ok, V entries each; no covariance rows
  bool score = false;
  const unsigned char* slice_hc = nullptr;  // score mode: per slice, 1 = the slice holds hard calls only
  bool uncentred = false;  // family mode: FastLMM::disableCenterGenotype (MetaFamBinary)
  double *effect = nullptr, *se = nullptr;
  int* ok = nullptr;
};

