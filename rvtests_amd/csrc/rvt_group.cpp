// rvtests_amd — several GPUs behind ONE caller thread (include/rvtests_amd.h, "device groups").
//
// Genes are independent units that share only the null model, so a group is simply one engine context
// per device: the null model (and, for related samples, the kinship decomposition) is installed on every member, the
// gene stream is dealt to the members in runs of kRun genes (the engine starts computing a run as soon as it is
// complete, so one member computes while the caller's thread feeds the next), and rvt_group_collect hands the records
// back in SUBMISSION order whatever member produced them — the reference writes its output files in gene order
// (src/Main.cpp:1221-1254).  No device-to-device traffic at all: the only "collective" is this ordered merge of
// fixed-size records on the host.  Permutation p-values consume one process-wide random stream in gene order
// (src/Permutation.h:69-98), so a gene that asks for them always goes to member 0.
//
// Plain host C++ over the single-device C ABI; nothing here touches HIP directly.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <string>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>
#include "../../include/rvtests_amd.h"
#include "host_stage.h"

// One submit worker per member.  The caller's thread only COPIES a gene's buffer (with the copy pool of host_stage.h: the
// reference reuses its buffer for the next gene) and queues a job; the member's worker thread makes the engine calls —
// the staged host-to-device copy and the ~10 HIP launches a packed submission costs (~300 us) — so the members' PCIe
// links and launch queues are fed concurrently, not one after the other by a single thread.  A context is still used by
// one thread at a time: the worker holds `ctx_mu` during each engine call, rvt_group_collect waits for the queues to
// drain, rvt_group_collect_ready skips a member whose worker is inside a call.
struct MemberWorker {
  struct Job {
    int kind = 0;  // 0 fp64 imputed + af, 1 raw doubles, 2 int8, 3 PLINK 2-bit
    int64_t gene_id = 0;
    int M = 0;
    uint32_t tests = 0;
    rvt_params prm;
    bool has_prm = false;
    int buf = -1;
    std::vector<double> af;
  };
  static constexpr int kDepth = 3;  // copied genes a member may hold before the caller waits
  rvt_ctx* ctx = nullptr;
  std::thread th;
  std::mutex mu;       // queue state
  std::mutex ctx_mu;   // the engine context
  std::condition_variable cv_job, cv_idle;
  std::deque<Job> q;
  std::vector<std::vector<char>> bufs;
  std::vector<int> free_bufs;
  bool stop = false;
  bool busy = false;
  int err = RVT_OK;
  std::string errmsg;

  void start(rvt_ctx* c) {
    ctx = c;
    bufs.resize(kDepth);
    for (int i = 0; i < kDepth; ++i) free_bufs.push_back(i);
    th = std::thread([this] { loop(); });
  }
  void loop() {
    for (;;) {
      Job j;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv_job.wait(lk, [this] { return stop || !q.empty(); });
        if (q.empty()) return;  // stop
        j = std::move(q.front());
        q.pop_front();
        busy = true;
      }
      int rc = RVT_OK;
      {
        std::lock_guard<std::mutex> cl(ctx_mu);
        const rvt_params* prm = j.has_prm ? &j.prm : nullptr;
        const char* data = bufs[j.buf].data();
        switch (j.kind) {
          case 0: rc = rvt_submit_gene(ctx, j.gene_id, j.M, (const double*)data, j.af.data(), j.tests, prm); break;
          case 1: rc = rvt_submit_gene_raw(ctx, j.gene_id, j.M, (const double*)data, j.tests, prm, nullptr); break;
          case 2: rc = rvt_submit_gene_i8(ctx, j.gene_id, j.M, (const int8_t*)data, j.tests, prm, nullptr); break;
          default: rc = rvt_submit_gene_bed(ctx, j.gene_id, j.M, (const unsigned char*)data, j.tests, prm, nullptr); break;
        }
        if (rc && err == RVT_OK) {
          err = rc;
          errmsg = rvt_last_error(ctx);
        }
      }
      {
        std::lock_guard<std::mutex> lk(mu);
        free_bufs.push_back(j.buf);
        busy = false;
      }
      cv_idle.notify_all();
    }
  }
  // caller: a buffer to copy the next gene into (waits while the member holds kDepth genes)
  int take_buffer() {
    std::unique_lock<std::mutex> lk(mu);
    cv_idle.wait(lk, [this] { return !free_bufs.empty(); });
    const int b = free_bufs.back();
    free_bufs.pop_back();
    return b;
  }
  void push(Job&& j) {
    {
      std::lock_guard<std::mutex> lk(mu);
      q.push_back(std::move(j));
    }
    cv_job.notify_one();
  }
  void flush() {
    std::unique_lock<std::mutex> lk(mu);
    cv_idle.wait(lk, [this] { return q.empty() && !busy; });
  }
  void shutdown() {
    {
      std::lock_guard<std::mutex> lk(mu);
      stop = true;
    }
    cv_job.notify_all();
    if (th.joinable()) th.join();
  }
};

struct rvt_group {
  std::vector<rvt_ctx*> member;
  std::vector<std::unique_ptr<MemberWorker>> worker;  // empty (the default): every call on the caller's thread
  std::deque<int> owner;       // member of every submitted, not yet collected gene, in submission order
  std::vector<std::deque<rvt_gene_result>> inbox;  // records already taken from a member, waiting for their turn
  long long submitted = 0;     // genes dealt so far (decides the member of the next one)
  bool perm_exact = true;      // SKAT permutations replay ONE rand() stream (member 0 only) instead of counter-based keys;
                               // the default of a ONE-member group (= the reference's numbers), off when genes are dealt
  std::string err;
  // A submission that failed on a member's worker thread (RVT_GROUP_ASYNC): the gene was already counted in `owner`, so the
  // member now holds fewer records than the group expects.  The group stays FAILED — every call returns this code — until
  // rvt_group_collect has discarded what is pending and left the group empty again.
  int failed = RVT_OK;
};

namespace {
constexpr int kRun = 32;  // genes dealt to one member before moving on (= the engine's asynchronous sub-batch)

int gfail(rvt_group* g, int code, const char* what, rvt_ctx* c) {
  if (g) g->err = std::string(what) + (c ? std::string(": ") + rvt_last_error(c) : std::string());
  return code;
}

int next_member(rvt_group* g, const rvt_params* prm, uint32_t tests) {
  // (counter-based permutations are keyed by gene: any member; the exact mode consumes one stream in gene order)
  if (g->perm_exact && prm && prm->skat_nperm > 0 && (tests & RVT_TEST_SKAT)) return 0;
  return (int)((g->submitted / kRun) % (long long)g->member.size());
}
// every worker idle (their queues drained): the caller's thread may use the contexts
int group_flush(rvt_group* g) {
  for (size_t k = 0; k < g->worker.size(); ++k) {  // (every worker: none keeps an error behind the first one found)
    MemberWorker& w = *g->worker[k];
    w.flush();
    if (w.err) {
      if (!g->failed) {
        g->failed = w.err;
        g->err = "submit (member " + std::to_string(k) + "): " + w.errmsg + " — collect to reset the group";
      }
      w.err = RVT_OK;
    }
  }
  return g->failed;
}

// queue one gene for member k: copy `bytes` of `data` (and the frequencies) now, engine calls on the member's worker
int group_submit_async(rvt_group* g, int k, int kind, int64_t gene_id, int M, const void* data, size_t bytes,
                       const double* af, uint32_t tests, const rvt_params* prm) {
  MemberWorker& w = *g->worker[k];
  if (g->failed) return g->failed;
  if (w.err) {  // an earlier job of this member failed: its gene is counted in `owner` but the member does not hold it
    g->failed = w.err;
    g->err = "submit (member " + std::to_string(k) + "): " + w.errmsg + " — collect to reset the group";
    w.err = RVT_OK;
    return g->failed;
  }
  MemberWorker::Job j;
  j.kind = kind;
  j.gene_id = gene_id;
  j.M = M;
  j.tests = tests;
  j.has_prm = prm != nullptr;
  if (prm) j.prm = *prm;
  if (af) j.af.assign(af, af + M);
  j.buf = w.take_buffer();
  std::vector<char>& b = w.bufs[j.buf];
  if (b.size() < bytes) b.resize(bytes);
  rvt::CopyPool::instance().copy(b.data(), data, bytes);
  w.push(std::move(j));
  g->owner.push_back(k);
  ++g->submitted;
  return RVT_OK;
}
}  // namespace

extern "C" {

int rvt_group_init(rvt_group** out, int n_dev, const int* dev_ids) {
  if (!out || n_dev < 1) return RVT_E_INVALID;
  *out = nullptr;
  rvt_group* g = new rvt_group();
  for (int k = 0; k < n_dev; ++k) {
    rvt_ctx* c = nullptr;
    const int rc = rvt_init(&c, dev_ids ? dev_ids[k] : k);
    if (rc) {
      for (rvt_ctx* m : g->member) rvt_destroy(m);
      delete g;
      return rc;
    }
    g->member.push_back(c);
  }
  g->inbox.resize(g->member.size());
  {
    // Opt-in (RVT_GROUP_ASYNC=1).  Measured with both members on ONE GPU (tools/bench_group_stream.py) the extra host
    // copy costs more than the overlapped engine calls save (fp64 273 -> 140-157, int8 1 510 -> 266-1 084, 2-bit
    // 1 870-3 220 -> 2 950-3 000 gene-sets/s): the members share one PCIe link there, and the blocks are copied twice.
    // Whether separate links change the balance has not been measured (no multi-GPU box in this environment).
    // Members on DIFFERENT devices get their feeder threads by default (each has a PCIe link and a staging ring of its own:
    // the caller's thread only copies the gene into the member's host buffer and moves on; the engine calls — staging copy,
    // DMA enqueue, launches: ~170 us per packed gene, tools/host_feed_bench — run on the member's thread).  RVT_GROUP_ASYNC=0
    // turns them off, =1 forces them on.
    const char* e = getenv("RVT_GROUP_ASYNC");
    bool distinct = n_dev > 1;
    for (int a = 0; a < n_dev && distinct; ++a)
      for (int b = a + 1; b < n_dev; ++b)
        if ((dev_ids ? dev_ids[a] : a) == (dev_ids ? dev_ids[b] : b)) distinct = false;
    if (e ? atoi(e) != 0 : distinct)
      for (rvt_ctx* m : g->member) {
        g->worker.emplace_back(new MemberWorker());
        g->worker.back()->start(m);
      }
  }
  // Permutation mode: one member = one gene stream = the reference's own rand() stream (bit-identical counters).  Genes
  // dealt over several members cannot share one sequential stream without serialising on member 0, so such a group takes
  // the counter-based permutations (statistical parity) unless the caller asks for the exact stream.
  g->perm_exact = g->member.size() == 1;
  if (const char* e = getenv("RVT_PERM_EXACT")) g->perm_exact = atoi(e) != 0;
  for (rvt_ctx* m : g->member) rvt_set_perm_exact(m, g->perm_exact ? 1 : 0);
  *out = g;
  return RVT_OK;
}

int rvt_group_set_perm_exact(rvt_group* g, int on) {
  if (!g) return RVT_E_INVALID;
  if (!g->owner.empty()) return gfail(g, RVT_E_STATE, "collect the submitted genes before changing the permutation mode", nullptr);
  g->perm_exact = on != 0;
  for (rvt_ctx* m : g->member) rvt_set_perm_exact(m, on);
  return RVT_OK;
}

extern "C" int rvt_host_adopt(rvt_ctx* c, const void* ptr, size_t bytes);  // (engine-internal: a range another member registered)

int rvt_group_host_register(rvt_group* g, const void* ptr, size_t bytes) {
  if (!g || g->member.empty()) return RVT_E_INVALID;
  if (int rcf = group_flush(g)) return rcf;
  int rc = rvt_host_register(g->member[0], ptr, bytes);
  if (rc) return gfail(g, rc, "rvt_host_register", g->member[0]);
  for (size_t k = 1; k < g->member.size(); ++k) rvt_host_adopt(g->member[k], ptr, bytes);
  return RVT_OK;
}

int rvt_group_host_unregister(rvt_group* g, const void* ptr) {
  if (!g || g->member.empty()) return RVT_E_INVALID;
  if (int rcf = group_flush(g)) return rcf;
  int rc = RVT_OK;
  for (size_t k = g->member.size(); k-- > 0;) {  // member 0 (the owner of the registration) last
    const int r = rvt_host_unregister(g->member[k], ptr);
    if (r && !rc) rc = gfail(g, r, "rvt_host_unregister", g->member[k]);
  }
  return rc;
}

int rvt_group_set_content(rvt_group* g, int hint, int lattice_denominator) {
  if (!g) return RVT_E_INVALID;
  for (rvt_ctx* m : g->member) {
    int rc = rvt_set_content_hint(m, hint);
    if (!rc) rc = rvt_set_dosage_lattice(m, lattice_denominator);
    if (rc) return gfail(g, rc, "rvt_group_set_content", m);
  }
  return RVT_OK;
}

int rvt_group_set_dosage_float(rvt_group* g, int on) {
  if (!g) return RVT_E_INVALID;
  for (rvt_ctx* m : g->member)
    if (int rc = rvt_set_dosage_float(m, on)) return gfail(g, rc, "rvt_group_set_dosage_float", m);
  return RVT_OK;
}

int rvt_group_rand_seed(rvt_group* g, unsigned seed) {
  if (!g) return RVT_E_INVALID;
  if (int rcf = group_flush(g)) return rcf;
  for (rvt_ctx* m : g->member) rvt_rand_seed(m, seed);
  return RVT_OK;
}

void rvt_group_destroy(rvt_group* g) {
  if (!g) return;
  for (auto& w : g->worker) w->shutdown();
  for (rvt_ctx* m : g->member) rvt_destroy(m);
  delete g;
}

int rvt_group_size(const rvt_group* g) { return g ? (int)g->member.size() : 0; }
rvt_ctx* rvt_group_member(rvt_group* g, int k) {
  return (g && k >= 0 && k < (int)g->member.size()) ? g->member[k] : nullptr;
}
const char* rvt_group_last_error(const rvt_group* g) { return g ? g->err.c_str() : "null group"; }

int rvt_group_set_null(rvt_group* g, int trait, int64_t N, int d, const double* X, const double* res, const double* v,
                       double sigma2) {
  if (!g) return RVT_E_INVALID;
  if (!g->owner.empty()) return gfail(g, RVT_E_STATE, "collect the submitted genes before changing the null model", nullptr);
  for (rvt_ctx* m : g->member) {
    const int rc = rvt_set_null(m, trait, N, d, X, res, v, sigma2);
    if (rc) return gfail(g, rc, "rvt_set_null", m);
  }
  return RVT_OK;
}

int rvt_group_fit_null(rvt_group* g, int trait, int64_t N, int d, const double* X, const double* y, double* beta_out,
                       double* sigma2_out) {
  if (!g) return RVT_E_INVALID;
  if (!g->owner.empty()) return gfail(g, RVT_E_STATE, "collect the submitted genes before changing the null model", nullptr);
  for (size_t k = 0; k < g->member.size(); ++k) {  // every member fits the same model from the same inputs
    const int rc = rvt_fit_null(g->member[k], trait, N, d, X, y, k == 0 ? beta_out : nullptr, k == 0 ? sigma2_out : nullptr);
    if (rc) return gfail(g, rc, "rvt_fit_null", g->member[k]);
  }
  return RVT_OK;
}

#define RVT_GROUP_SUBMIT(call)                                  \
  if (!g) return RVT_E_INVALID;                                 \
  if (g->failed) return g->failed; /* sticky until collected */ \
  const int k = next_member(g, params, tests);                  \
  rvt_ctx* m = g->member[k];                                    \
  if (!g->worker.empty()) {                                     \
    g->worker[k]->flush();                                      \
    if (g->worker[k]->err) return group_flush(g);               \
  }                                                             \
  const int rc = (call);                                        \
  if (rc) return gfail(g, rc, "submit", m);                     \
  g->owner.push_back(k);                                        \
  ++g->submitted;                                               \
  return RVT_OK;

// bytes of one gene's buffer at the boundary (kind as MemberWorker::Job::kind), or 0 when the arguments are unusable
static size_t gene_bytes(rvt_group* g, int kind, int M) {
  int64_t N = 0;
  int d = 0;
  if (M < 1 || M > RVT_MAX_VARIANTS || rvt_null_dims(g->member[0], &N, &d) != RVT_OK) return 0;
  return kind <= 1 ? sizeof(double) * (size_t)N * M : (kind == 2 ? (size_t)N * M : (size_t)((N + 3) / 4) * M);
}

int rvt_group_submit_gene(rvt_group* g, int64_t gene_id, int M, const double* G, const double* af, uint32_t tests,
                          const rvt_params* params) {
  if (g && !g->worker.empty() && G && af) {
    if (const size_t bytes = gene_bytes(g, 0, M))
      return group_submit_async(g, next_member(g, params, tests), 0, gene_id, M, G, bytes, af, tests, params);
  }
  RVT_GROUP_SUBMIT(rvt_submit_gene(m, gene_id, M, G, af, tests, params))
}
int rvt_group_submit_gene_raw(rvt_group* g, int64_t gene_id, int M, const double* Graw, uint32_t tests,
                              const rvt_params* params, double* af_out) {
  if (g && !g->worker.empty() && Graw && !af_out) {  // (somebody waiting for the frequencies: the synchronous call)
    if (const size_t bytes = gene_bytes(g, 1, M))
      return group_submit_async(g, next_member(g, params, tests), 1, gene_id, M, Graw, bytes, nullptr, tests, params);
  }
  RVT_GROUP_SUBMIT(rvt_submit_gene_raw(m, gene_id, M, Graw, tests, params, af_out))
}
int rvt_group_submit_gene_i8(rvt_group* g, int64_t gene_id, int M, const int8_t* G8, uint32_t tests,
                             const rvt_params* params, double* af_out) {
  if (g && !g->worker.empty() && G8 && !af_out) {
    if (const size_t bytes = gene_bytes(g, 2, M))
      return group_submit_async(g, next_member(g, params, tests), 2, gene_id, M, G8, bytes, nullptr, tests, params);
  }
  RVT_GROUP_SUBMIT(rvt_submit_gene_i8(m, gene_id, M, G8, tests, params, af_out))
}
int rvt_group_submit_gene_bed(rvt_group* g, int64_t gene_id, int M, const unsigned char* bed, uint32_t tests,
                              const rvt_params* params, double* af_out) {
  if (g && !g->worker.empty() && bed && !af_out) {
    if (const size_t bytes = gene_bytes(g, 3, M))
      return group_submit_async(g, next_member(g, params, tests), 3, gene_id, M, bed, bytes, nullptr, tests, params);
  }
  RVT_GROUP_SUBMIT(rvt_submit_gene_bed(m, gene_id, M, bed, tests, params, af_out))
}

int rvt_group_vcf_set_samples(rvt_group* g, int n_file_samples, const int32_t* row_of_sample) {
  if (!g) return RVT_E_INVALID;
  if (int rcf = group_flush(g)) return rcf;
  for (rvt_ctx* m : g->member) {
    const int rc = rvt_vcf_set_samples(m, n_file_samples, row_of_sample);
    if (rc) return gfail(g, rc, "rvt_vcf_set_samples", m);
  }
  return RVT_OK;
}
int rvt_group_vcf_set_filters(rvt_group* g, int gd_min, int gd_max, int gq_min, int gq_max) {
  if (!g) return RVT_E_INVALID;
  if (int rcf = group_flush(g)) return rcf;
  for (rvt_ctx* m : g->member) rvt_vcf_set_filters(m, gd_min, gd_max, gq_min, gq_max);
  return RVT_OK;
}
int rvt_group_submit_gene_vcf(rvt_group* g, int64_t gene_id, int M, const char* const* sample_text,
                              const int64_t* text_len, const int* gt_index, const int* gd_index, const int* gq_index,
                              uint32_t tests, const rvt_params* params, double* af_out) {
  RVT_GROUP_SUBMIT(rvt_submit_gene_vcf(m, gene_id, M, sample_text, text_len, gt_index, gd_index, gq_index, tests, params, af_out))
}

int rvt_group_submit_gene_bgen(rvt_group* g, int64_t gene_id, int M, const unsigned char* const* block,
                               const int64_t* block_len, int layout, uint32_t tests, const rvt_params* params,
                               double* af_out) {
  RVT_GROUP_SUBMIT(rvt_submit_gene_bgen(m, gene_id, M, block, block_len, layout, tests, params, af_out))
}

// the ordered merge: each member's records arrive in ITS submission order; hand out the global prefix
static int pop_in_order(rvt_group* g, rvt_gene_result* out, int cap) {
  int n = 0;
  while (n < cap && !g->owner.empty()) {
    std::deque<rvt_gene_result>& box = g->inbox[g->owner.front()];
    if (box.empty()) break;
    out[n++] = box.front();
    box.pop_front();
    g->owner.pop_front();
  }
  return n;
}

int rvt_group_collect(rvt_group* g, rvt_gene_result* out, int cap, int* n_out) {
  if (!g || !out || !n_out) return RVT_E_INVALID;
  *n_out = 0;
  const int n = (int)std::min<size_t>(g->owner.size(), (size_t)std::max(cap, 0));
  // (the flush and the failed-state reset come BEFORE the "nothing to hand back" exit: a failed group is reset by any
  //  collect, also one with cap = 0 or nothing pending — ADVICE r4)
  if (const int rcf = group_flush(g)) {  // every queued gene has reached its member — or one of them failed on the way:
    // the members' records no longer line up with the submission order.  Discard what is pending and start clean.
    std::vector<rvt_gene_result> drop(256);
    for (rvt_ctx* m : g->member)
      for (;;) {
        int nk = 0;
        if (rvt_collect(m, drop.data(), (int)drop.size(), &nk) != RVT_OK || nk == 0) break;
      }
    g->owner.clear();
    for (auto& q : g->inbox) q.clear();
    g->failed = RVT_OK;
    return rcf;
  }
  if (n == 0) return RVT_OK;
  const int nm = (int)g->member.size();
  std::vector<int> want(nm, 0);
  for (int i = 0; i < n; ++i) ++want[g->owner[i]];
  for (int k = 0; k < nm; ++k) {
    const int more = want[k] - (int)g->inbox[k].size();
    if (more <= 0) continue;
    std::vector<rvt_gene_result> got(more);
    int nk = 0;
    const int rc = rvt_collect(g->member[k], got.data(), more, &nk);
    if (rc) return gfail(g, rc, "rvt_collect", g->member[k]);
    if (nk != more) return gfail(g, RVT_E_STATE, "a member returned fewer records than were submitted to it", nullptr);
    g->inbox[k].insert(g->inbox[k].end(), got.begin(), got.end());
  }
  *n_out = pop_in_order(g, out, n);
  return RVT_OK;
}

int rvt_group_collect_ready(rvt_group* g, rvt_gene_result* out, int cap, int* n_out) {
  if (!g || !out || !n_out) return RVT_E_INVALID;
  *n_out = 0;
  if (g->owner.empty() || cap <= 0) return RVT_OK;
  std::vector<rvt_gene_result> got(256);
  for (size_t k = 0; k < g->member.size(); ++k) {
    // (a member whose worker is inside an engine call is left for the next time: the context has one user at a time)
    std::unique_lock<std::mutex> cl;
    if (!g->worker.empty()) {
      cl = std::unique_lock<std::mutex>(g->worker[k]->ctx_mu, std::try_to_lock);
      if (!cl.owns_lock()) continue;
    }
    for (;;) {  // whatever the member has finished, without waiting
      int nk = 0;
      const int rc = rvt_collect_ready(g->member[k], got.data(), (int)got.size(), &nk);
      if (rc) return gfail(g, rc, "rvt_collect_ready", g->member[k]);
      if (nk == 0) break;
      g->inbox[k].insert(g->inbox[k].end(), got.begin(), got.begin() + nk);
    }
  }
  *n_out = pop_in_order(g, out, cap);
  return RVT_OK;
}

// ---- related samples: the kinship decomposition is replicated on every member ------------------------------------------
int rvt_group_set_kinship(rvt_group* g, int64_t N, const float* U, const float* S) {
  if (!g) return RVT_E_INVALID;
  if (int rcf = group_flush(g)) return rcf;
  for (rvt_ctx* m : g->member) {
    const int rc = rvt_set_kinship(m, N, U, S);
    if (rc) return gfail(g, rc, "rvt_set_kinship", m);
  }
  return RVT_OK;
}

int rvt_group_fit_fam_null(rvt_group* g, int64_t N, int d, const double* X, const double* y, rvt_fam_null* out) {
  if (!g) return RVT_E_INVALID;
  if (int rcf = group_flush(g)) return rcf;
  for (size_t k = 0; k < g->member.size(); ++k) {
    rvt_fam_null tmp;
    const int rc = rvt_fit_fam_null(g->member[k], N, d, X, y, k == 0 && out ? out : &tmp);
    if (rc) return gfail(g, rc, "rvt_fit_fam_null", g->member[k]);
  }
  return RVT_OK;
}

// Host genotype blocks (N x M[g] doubles each, imputed, unflipped) dealt to the members in contiguous shares; every
// member uploads and runs its share (rvt_run_fam_tests) at the same time; records come back in the caller's order.
int rvt_group_run_fam_tests_host(rvt_group* g, int n_genes, const double* const* G_host, const int* M,
                                 const int64_t* gene_ids, uint32_t tests, rvt_gene_result* out) {
  if (!g || n_genes < 0 || (n_genes > 0 && (!G_host || !M || !out))) return RVT_E_INVALID;
  if (int rcf = group_flush(g)) return rcf;
  const int nm = (int)g->member.size();
  // contiguous shares balanced by column count (the rotation's cost is proportional to the columns)
  long long total = 0;
  for (int i = 0; i < n_genes; ++i) total += M[i];
  // one host thread per member for the duration of this call (each context is still used by exactly one thread)
  std::vector<int> rcs(nm, RVT_OK);
  std::vector<std::thread> workers;
  int begin = 0;
  long long acc = 0;
  for (int k = 0; k < nm; ++k) {
    int end = begin;
    const long long target = total * (k + 1) / nm;
    while (end < n_genes && (k == nm - 1 || acc + M[end] <= target || end == begin)) acc += M[end++];
    if (end > begin) {
      rvt_ctx* m = g->member[k];
      const int b0 = begin, e0 = end;
      workers.emplace_back([=, &rcs]() {
        std::vector<double*> blocks(e0 - b0, nullptr);
        int rc = RVT_OK;
        for (int i = b0; i < e0 && !rc; ++i) {
          rc = rvt_block_alloc(m, M[i], &blocks[i - b0]);
          if (!rc) rc = rvt_block_upload(m, blocks[i - b0], M[i], G_host[i]);
        }
        if (!rc)
          rc = rvt_run_fam_tests(m, e0 - b0, blocks.data(), M + b0, gene_ids ? gene_ids + b0 : nullptr, tests, out + b0);
        for (double* bl : blocks)
          if (bl) rvt_block_free(m, bl);
        rcs[k] = rc;
      });
    }
    begin = end;
  }
  for (std::thread& t : workers) t.join();
  for (int k = 0; k < nm; ++k)
    if (rcs[k]) return gfail(g, rcs[k], "rvt_run_fam_tests", g->member[k]);
  return RVT_OK;
}

// ---- `--meta score` / `--meta cov` over a device group ("shard by chromosome / chunk with one-window halo
// overlap": every chunk carries the columns its windows reach into, so the chunks are independent and nothing is
// exchanged) ------------------------------------------------------------------------------------------------------------
// G_host: N x V column-major (leading dimension N), the genotype vectors of V consecutive single-variant fit() calls.
// One host thread per member for the duration of the call; each context is used by exactly one thread.

// MetaScore: the V columns are cut into contiguous shares, one per member; outputs as rvt_score_block (V entries each).
int rvt_group_score_block_host(rvt_group* g, int64_t N, int V, const double* G_host, int* ok, double* ustat, double* vstat,
                               double* effect, double* effect_se, double* pvalue) {
  if (!g || N < 1 || V < 1 || !G_host || !ok || !ustat || !vstat || !effect || !effect_se || !pvalue) return RVT_E_INVALID;
  if (int rcf = group_flush(g)) return rcf;
  const int nm = (int)g->member.size();
  std::vector<int> rcs(nm, RVT_OK);
  std::vector<std::thread> workers;
  for (int k = 0; k < nm; ++k) {
    const int c0 = (int)((long long)V * k / nm), c1 = (int)((long long)V * (k + 1) / nm);
    if (c1 <= c0) continue;
    rvt_ctx* m = g->member[k];
    workers.emplace_back([=, &rcs]() {
      int rc = RVT_OK;
      constexpr int kBlock = 4096;  // columns per device block
      for (int b0 = c0; b0 < c1 && !rc; b0 += kBlock) {
        const int nb = std::min(kBlock, c1 - b0);
        double* blk = nullptr;
        rc = rvt_block_alloc(m, nb, &blk);
        if (!rc) rc = rvt_block_upload(m, blk, nb, G_host + (size_t)b0 * (size_t)N);
        if (!rc) rc = rvt_score_block(m, blk, nb, ok + b0, ustat + b0, vstat + b0, effect + b0, effect_se + b0, pvalue + b0);
        if (blk) rvt_block_free(m, blk);
      }
      rcs[k] = rc;
    });
  }
  for (std::thread& t : workers) t.join();
  for (int k = 0; k < nm; ++k)
    if (rcs[k]) return gfail(g, rcs[k], "rvt_score_block", g->member[k]);
  return RVT_OK;
}

// MetaCov: the band of the V variants with up to `halo` following markers per head (the caller's window rule never looks
// further: halo = the largest number of sites one window holds).  band[h * (halo + 1) + t] = the value of head h and
// marker h + t as rvt_cov_block / rvt_cov_rect give it (t = 0 .. halo, NaN beyond the last variant); xz: V x d as
// rvt_cov_block; polymorphic: V.  Heads are dealt in chunks of `chunk` (0 = default) to the members in turn; a chunk's
// device block holds its heads AND the halo behind them, so every chunk is computed on its own.
int rvt_group_cov_band_host(rvt_group* g, int64_t N, int V, const double* G_host, int halo, int chunk, double* band,
                            double* xz, double* zz, int* polymorphic) {
  if (!g || N < 1 || V < 1 || !G_host || halo < 0 || !band || !xz || !polymorphic) return RVT_E_INVALID;
  if (int rcf = group_flush(g)) return rcf;
  const int nm = (int)g->member.size();
  if (chunk <= 0) chunk = std::max(256, std::min(1024, (V + nm - 1) / nm));
  const int n_chunks = (V + chunk - 1) / chunk;
  const size_t bw = (size_t)halo + 1;
  std::vector<int> rcs(nm, RVT_OK);
  std::vector<int> dcov(nm, 0);
  std::vector<std::thread> workers;
  for (int k = 0; k < nm; ++k) {
    rvt_ctx* m = g->member[k];
    workers.emplace_back([=, &rcs, &dcov]() {
      int rc = RVT_OK;
      std::vector<double> cov, xzc;
      std::vector<int> poly;
      for (int ch = k; ch < n_chunks && !rc; ch += nm) {
        const int h0 = ch * chunk, H = std::min(chunk, V - h0), W = std::min(V - h0, H + halo);
        double* blk = nullptr;
        rc = rvt_block_alloc(m, W, &blk);
        if (!rc) rc = rvt_block_upload(m, blk, W, G_host + (size_t)h0 * (size_t)N);
        cov.assign((size_t)H * W, 0.0);
        poly.assign((size_t)W, 0);
        // (covariate count: the xz rows are d wide; rvt_cov_rect writes W x d, so size for the widest d)
        xzc.assign((size_t)W * RVT_MAX_COV, 0.0);
        std::vector<double> zzc((size_t)RVT_MAX_COV * RVT_MAX_COV, 0.0);
        if (!rc) rc = rvt_cov_rect(m, blk, 0, H, W, cov.data(), xzc.data(), zzc.data(), poly.data());
        if (blk) rvt_block_free(m, blk);
        if (rc) break;
        int64_t Nm = 0;
        int d = 0;
        rvt_null_dims(m, &Nm, &d);  // columns of the installed design matrix (xz rows are d wide)
        dcov[k] = d;
        for (int h = 0; h < H; ++h) {
          for (size_t t = 0; t < bw; ++t) {
            const int j = h + (int)t;
            band[(size_t)(h0 + h) * bw + t] = (j < W) ? cov[(size_t)h + (size_t)j * H] : (0.0 / 0.0);
          }
          polymorphic[h0 + h] = poly[h];
          for (int c = 0; c < d; ++c) xz[(size_t)(h0 + h) * d + c] = xzc[(size_t)h * d + c];
        }
        if (zz && ch == 0) std::memcpy(zz, zzc.data(), sizeof(double) * (size_t)d * d);
      }
      rcs[k] = rc;
    });
  }
  for (std::thread& t : workers) t.join();
  for (int k = 0; k < nm; ++k)
    if (rcs[k]) return gfail(g, rcs[k], "rvt_cov_rect", g->member[k]);
  return RVT_OK;
}

}  // extern "C"
