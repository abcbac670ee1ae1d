// rvtests_amd — feeding the device from the caller's PAGEABLE memory at the rate of the PCIe link.
//
// The drop-in boundary hands over host buffers the caller reuses for the next gene (src/Main.cpp:1086,1225: one Matrix
// for every gene), so a submission must have consumed its input when it returns.  A plain hipMemcpy from pageable memory
// does that, but it returns only when the data has crossed the link (measured: 46-48 GB/s for 200 MB blocks, 12-40 GB/s for
// the 6-25 MB of packed genotypes).  Here the copy into pinned memory is done by a small pool of worker threads, in chunks,
// and every chunk is handed to the DMA engine as soon as it is complete — the copies of chunk k + 1 run while chunk k
// crosses the link, and the call returns when the LAST host copy is done, not when the data has arrived (the stream
// orders everything behind it), so the caller's next HIP calls overlap the transfer.
//
// Host-only code (threads, memcpy); the HIP calls are passed in as callbacks so that the test harness (hostcheck.cpp) can
// measure the copy pool without a GPU.
#pragma once
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
#include <immintrin.h>
#endif

namespace rvt {

// Persistent worker threads that execute batches of memcpy tasks.  One pool per process, shared by every context: usually
// one caller thread (src/Main.cpp), but a device group runs one caller per member at once — batches complete independently.  RVT_COPY_THREADS sets the size (default: min(3, hardware threads / 2), at least 1 —
// measured on the MI355X host: one thread copies ~30 GB/s, three keep the link busy, eight only disturb each other).
class CopyPool {
 public:
  struct Task {
    void* dst;
    const void* src;
    size_t bytes;
  };
  explicit CopyPool(int threads) {
    n_ = std::max(1, threads);
    for (int i = 0; i + 1 < n_; ++i) workers_.emplace_back([this] { loop(); });  // the caller's thread is worker n_ - 1
  }
  ~CopyPool() {
    {
      std::lock_guard<std::mutex> lk(m_);
      stop_ = true;
    }
    cv_.notify_all();
    for (auto& t : workers_) t.join();
  }
  int threads() const { return n_; }
  static int default_threads() {
    if (const char* e = getenv("RVT_COPY_THREADS")) return std::max(1, atoi(e));
    const unsigned hw = std::thread::hardware_concurrency();
    return (int)std::max(1u, std::min(3u, hw / 2));
  }
  static CopyPool& instance() {
    static CopyPool pool(default_threads());
    return pool;
  }
  // copy [src, src + bytes) to dst with all threads (pieces of >= 256 KiB); returns when the copy is complete
  void copy(void* dst, const void* src, size_t bytes) {
    Task t{dst, src, bytes};
    run(&t, 1);
  }
  // n independent work items fn(0) .. fn(n - 1) on all threads (the packing of the fp64 boundary: one item per column);
  // returns when all of THIS batch are done
  void run_items(size_t n, const std::function<void(size_t)>& fn) {
    if (n == 0) return;
    if (n_ == 1 || n == 1) {
      for (size_t i = 0; i < n; ++i) fn(i);
      return;
    }
    size_t left = n;
    {
      std::lock_guard<std::mutex> lk(m_);
      for (size_t i = 0; i < n; ++i) q_.push_back(Piece{nullptr, nullptr, i, &left, &fn});
    }
    pending_.fetch_add((int)n, std::memory_order_release);
    cv_.notify_all();
    help();
    wait_batch(&left);
  }
  // a third pool for ONE column at a time (rvt_block_upload_columns: a site of MetaCovTest / MetaScoreTest::fit, 4 MB): eight
  // threads — with sixteen the wake-ups cost more than the second eight save (12.9 k against 20.5 k sites/s, round 6)
  static CopyPool& column_instance() {
    static CopyPool pool([] {
      if (const char* e = getenv("RVT_COLUMN_THREADS")) return std::max(1, atoi(e));
      const unsigned hw = std::thread::hardware_concurrency();
      return (int)std::max(1u, std::min(8u, hw / 2));
    }());
    return pool;
  }
  // the pool of the packing passes: reading 200 MB per gene is bound by memory bandwidth per core, so it takes more threads
  // than the copies into pinned memory do (RVT_PACK_THREADS; default min(16, hardware threads / 2))
  static CopyPool& pack_instance() {
    static CopyPool pool([] {
      if (const char* e = getenv("RVT_PACK_THREADS")) return std::max(1, atoi(e));
      const unsigned hw = std::thread::hardware_concurrency();
      return (int)std::max(1u, std::min(16u, hw / 2));
    }());
    return pool;
  }
  // run a batch of copies, split into pieces so that every thread has work; returns when all of THIS batch are done.
  // Several caller threads may be inside run() at once (one per member of a device group): completion is counted per batch
  // (round 3 kept one process-wide counter: every caller then also waited for the other callers' pieces).
  void run(const Task* tasks, size_t n) {
    size_t total = 0;
    for (size_t i = 0; i < n; ++i) total += tasks[i].bytes;
    if (total == 0) return;
    if (n_ == 1 || total < (size_t)512 << 10) {
      for (size_t i = 0; i < n; ++i) std::memcpy(tasks[i].dst, tasks[i].src, tasks[i].bytes);
      return;
    }
    const size_t piece = std::max<size_t>((size_t)256 << 10, (total / (size_t)(n_ * 2) + 4095) / 4096 * 4096);
    size_t left = 0;  // pieces of this batch not yet copied (guarded by m_)
    {
      std::lock_guard<std::mutex> lk(m_);
      for (size_t i = 0; i < n; ++i)
        for (size_t o = 0; o < tasks[i].bytes; o += piece) {
          q_.push_back(Piece{(char*)tasks[i].dst + o, (const char*)tasks[i].src + o, std::min(piece, tasks[i].bytes - o), &left, nullptr});
          ++left;
        }
    }
    pending_.fetch_add((int)left, std::memory_order_release);
    cv_.notify_all();
    help();  // the calling thread works too
    wait_batch(&left);
  }

 private:
  struct Piece {
    void* dst;
    const void* src;
    size_t bytes;   // (run_items: the item's index)
    size_t* left;   // the batch's counter
    const std::function<void(size_t)>* fn;  // run_items: the work; null = memcpy
  };
  bool take(Piece* t) {
    std::lock_guard<std::mutex> lk(m_);
    if (q_.empty()) return false;
    *t = q_.front();
    q_.pop_front();
    pending_.fetch_sub(1, std::memory_order_relaxed);
    return true;
  }
  // the caller waits for ITS batch: a short spin first — a batch of a few hundred microseconds (one 4 MB column on eight
  // threads takes ~20 us) should not pay a futex sleep and wake-up on top (round 6) — then the condition variable
  static bool spin_enabled() {
    static const bool on = !(getenv("RVT_POOL_SPIN") && atoi(getenv("RVT_POOL_SPIN")) == 0);
    return on;
  }
  void wait_batch(size_t* left) {
    for (int i = 0; spin_enabled() && i < 4000; ++i) {
      {
        std::lock_guard<std::mutex> lk(m_);
        if (*left == 0) return;
      }
      for (int k = 0; k < 16; ++k) cpu_relax();
    }
    std::unique_lock<std::mutex> lk(m_);
    done_.wait(lk, [left] { return *left == 0; });
  }
  static void cpu_relax() {
#if defined(__x86_64__)
    __builtin_ia32_pause();
#else
    std::this_thread::yield();
#endif
  }
  void finish_one(size_t* left) {
    std::lock_guard<std::mutex> lk(m_);
    if (--*left == 0) done_.notify_all();
  }
  void help() {
    Piece t;
    while (take(&t)) {
      if (t.fn)
        (*t.fn)(t.bytes);
      else
        std::memcpy(t.dst, t.src, t.bytes);
      finish_one(t.left);
    }
  }
  void loop() {
    for (;;) {
      // a worker that has just finished looks for the next batch for ~50 us before it sleeps: a caller that hands over one
      // column per call (every ~50 us) then finds its workers awake
      bool found = false;
      for (int i = 0; spin_enabled() && i < 3000 && !found; ++i) {
        if (pending_.load(std::memory_order_acquire) > 0) found = true;
        else
          for (int k = 0; k < 16; ++k) cpu_relax();
      }
      if (!found) {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [this] { return stop_ || !q_.empty(); });
        if (stop_ && q_.empty()) return;
      }
      help();
    }
  }
  int n_ = 1;
  std::vector<std::thread> workers_;
  std::mutex m_;
  std::condition_variable cv_, done_;
  std::deque<Piece> q_;
  std::atomic<int> pending_{0};  // pieces queued and not yet taken (the spinning workers' wake-up word)
  bool stop_ = false;
};

// ---- the fp64 boundary, packed on the way (round 5) ------------------------------------------------------------------------
// rvt_submit_gene is what the unchanged gene loop of the reference calls (src/Main.cpp:1221-1254): the imputed N x M block of
// doubles, 200 MB per gene at N = 500 000, and the PCIe link carried every byte of it (192-294 gene-sets/s).  But what
// consolidate() leaves in such a block is almost always hard calls plus ONE other value per column — the mean it imputed
// (DataConsolidator.cpp:217-245) — and the threads that stage the block touch every double anyway.  pack_column_f64 turns
// a column into PLINK 2-bit codes (00 -> 0, 10 -> 1, 11 -> 2, 01 -> "the column's other value": exactly the rows
// rvt_submit_gene_bed hands over, with the other value where the imputed mean would be computed) — 32x fewer bytes — or
// reports that the column holds a second other value (dosages): the gene then crosses as doubles, as before.
// Values are compared as BIT PATTERNS (-0.0 is not 0.0 here, and is simply "the other value" of its column).
struct PackedColumn {
  bool ok = false;       // false: not representable (two different other values, or an other value outside [0, 2])
  bool has_mu = false;   // the column holds an other value
  double mu = 0.0;
  long long n_other = 0;
};

inline unsigned char pack4_scalar(const double* g, size_t n, unsigned long long* mu_bits, bool* has_mu, bool* ok, long long* n_other) {
  unsigned b = 0;
  for (size_t e = 0; e < n; ++e) {
    unsigned long long u;
    std::memcpy(&u, &g[e], 8);
    unsigned code;
    if (u == 0ull)
      code = 0u;
    else if (u == 0x3FF0000000000000ull)
      code = 2u;
    else if (u == 0x4000000000000000ull)
      code = 3u;
    else {
      if (!*has_mu) {
        *has_mu = true;
        *mu_bits = u;
      }
      if (u != *mu_bits) *ok = false;
      code = 1u;
      ++*n_other;
    }
    b |= code << (2 * e);
  }
  return (unsigned char)b;
}

#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
// 16 doubles -> 4 bytes of codes with AVX2; returns false when a value is neither a hard call nor *mu_bits (the caller
// then looks at the group with the scalar code, which also discovers the column's other value)
__attribute__((target("avx2"))) inline bool pack16_avx2(const double* g, unsigned long long mu_bits, unsigned char* out, int* n_other) {
  const __m256i one = _mm256_set1_epi64x(0x3FF0000000000000ll), two = _mm256_set1_epi64x(0x4000000000000000ll),
                zero = _mm256_setzero_si256(), mu = _mm256_set1_epi64x((long long)mu_bits);
  int others = 0;
  for (int q = 0; q < 4; ++q) {
    const __m256i v = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(g + 4 * q));
    const __m256i is0 = _mm256_cmpeq_epi64(v, zero), is1 = _mm256_cmpeq_epi64(v, one), is2 = _mm256_cmpeq_epi64(v, two),
                  ism = _mm256_cmpeq_epi64(v, mu);
    const int m_ok = _mm256_movemask_pd(_mm256_castsi256_pd(_mm256_or_si256(_mm256_or_si256(is0, is1), _mm256_or_si256(is2, ism))));
    if (m_ok != 15) return false;
    // code bit 0: 2 (11) and other (01); code bit 1: 1 (10) and 2 (11)  [a value equal to a hard call AND to mu is the hard call]
    const __m256i hard = _mm256_or_si256(_mm256_or_si256(is0, is1), is2);
    const int oth = _mm256_movemask_pd(_mm256_castsi256_pd(_mm256_andnot_si256(hard, ism)));
    const int lo = _mm256_movemask_pd(_mm256_castsi256_pd(is2)) | oth;
    const int hi = _mm256_movemask_pd(_mm256_castsi256_pd(_mm256_or_si256(is1, is2)));
    others += __builtin_popcount((unsigned)oth);
    // interleave the two 4-bit masks: bit 2e = lo_e, bit 2e + 1 = hi_e
    unsigned x = (unsigned)lo | ((unsigned)hi << 8);             // .... hhhh .... llll
    x = (x | (x << 2)) & 0x3333u;                                // ..hh..hh ..ll..ll
    x = (x | (x << 1)) & 0x5555u;                                // .h.h.h.h .l.l.l.l
    out[q] = (unsigned char)((x & 0xffu) | ((x >> 8) << 1));
  }
  *n_other += others;
  return true;
}
#endif

#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
// the same 16 doubles with AVX-512 (round 6): two 64-byte loads, four mask compares each — the masks ARE the code bits — and one
// bit deposit per code bit instead of the shift-and-mask interleave: about a third of the instructions of the AVX2 form per
// byte read.  (The packing pass read 200 MB per gene at ~12 GB/s per thread with AVX2: the threads, not the memory, were the
// limit — 16 of them gave 186 GB/s on a two-socket EPYC 9575F and more threads only got in each other's way.)
__attribute__((target("avx512f,bmi2,popcnt"))) inline bool pack16_avx512(const double* g, unsigned long long mu_bits, unsigned char* out,
                                                                         int* n_other) {
  const __m512i one = _mm512_set1_epi64(0x3FF0000000000000ll), two = _mm512_set1_epi64(0x4000000000000000ll),
                zero = _mm512_setzero_si512(), mu = _mm512_set1_epi64((long long)mu_bits);
  unsigned lo = 0, hi = 0, oth = 0;
  for (int q = 0; q < 2; ++q) {
    const __m512i v = _mm512_loadu_si512(reinterpret_cast<const void*>(g + 8 * q));
    const unsigned k0 = _mm512_cmpeq_epi64_mask(v, zero), k1 = _mm512_cmpeq_epi64_mask(v, one), k2 = _mm512_cmpeq_epi64_mask(v, two),
                   km = _mm512_cmpeq_epi64_mask(v, mu);
    const unsigned hard = k0 | k1 | k2;
    if (((hard | km) & 0xffu) != 0xffu) return false;
    const unsigned o = km & ~hard & 0xffu;  // [a value equal to a hard call AND to mu is the hard call]
    lo |= (k2 | o) << (8 * q);              // code bit 0: 2 (11) and other (01)
    hi |= (k1 | k2) << (8 * q);             // code bit 1: 1 (10) and 2 (11)
    oth |= o << (8 * q);
  }
  const unsigned x = _pdep_u32(lo, 0x55555555u) | _pdep_u32(hi, 0xAAAAAAAAu);  // sample e: bits 2 e, 2 e + 1
  std::memcpy(out, &x, 4);
  *n_other += __builtin_popcount(oth);
  return true;
}
#endif

// g[0 .. n) -> out[0 .. ceil(n / 4)), then zeros up to `pitch` bytes.  isa: -1 = the widest the CPU has, 0 scalar, 1 AVX2,
// 2 AVX-512 (tests compare them)
inline PackedColumn pack_column_f64(const double* g, size_t n, unsigned char* out, size_t pitch, const std::atomic<int>* stop,
                                    int isa = -1) {
  PackedColumn r;
  unsigned long long mu_bits = 0x7FF8DEADBEEF0001ull;  // (no double of a genotype block: a NaN payload)
  bool has_mu = false, ok = true;
  long long n_other = 0;
  size_t i = 0, o = 0;
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
  static const bool avx2 = __builtin_cpu_supports("avx2");
  static const bool avx512 = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("bmi2");
  if ((isa < 0 || isa == 2) && avx512) {
    while (i + 16 <= n && ok) {
      if ((o & 0xffff) == 0 && stop && stop->load(std::memory_order_relaxed)) return r;  // another column already failed
      int no = 0;
      if (pack16_avx512(g + i, mu_bits, out + o, &no)) {
        n_other += no;
      } else {  // a first other value, or a second one
        for (int q = 0; q < 4; ++q) out[o + q] = pack4_scalar(g + i + 4 * q, 4, &mu_bits, &has_mu, &ok, &n_other);
      }
      i += 16;
      o += 4;
    }
  } else if ((isa < 0 || isa == 1) && avx2) {
    while (i + 16 <= n && ok) {
      if ((o & 0xffff) == 0 && stop && stop->load(std::memory_order_relaxed)) return r;  // another column already failed
      int no = 0;
      if (pack16_avx2(g + i, mu_bits, out + o, &no)) {
        n_other += no;
      } else {  // a first other value, or a second one
        for (int q = 0; q < 4; ++q) out[o + q] = pack4_scalar(g + i + 4 * q, 4, &mu_bits, &has_mu, &ok, &n_other);
      }
      i += 16;
      o += 4;
    }
  }
#endif
  for (; i < n && ok; i += 4, ++o) out[o] = pack4_scalar(g + i, std::min<size_t>(4, n - i), &mu_bits, &has_mu, &ok, &n_other);
  if (!ok) return r;
  if (o < pitch) std::memset(out + o, 0, pitch - o);
  double mu = 0.0;
  if (has_mu) {
    std::memcpy(&mu, &mu_bits, 8);
    if (!(mu >= 0.0 && mu <= 2.0)) return r;  // (NaN, negative, > 2: the device's packed kernel does not take it)
  }
  r.ok = true;
  r.has_mu = has_mu;
  r.mu = mu;
  r.n_other = n_other;
  return r;
}

// ---- the int8 boundary, packed on the way (round 6) -------------------------------------------------------------------------
// rvt_submit_gene_i8 hands over hard calls 0 / 1 / 2 with negative = missing, one byte per genotype: exactly what PLINK's 2-bit
// codes say (00 -> 0, 10 -> 1, 11 -> 2, 01 -> missing).  The staging threads that would copy the 25 MB of a gene into the pinned
// ring write the 6 MB of its .bed rows there instead, and the gene continues as a rvt_submit_gene_bed gene: a quarter of the
// bytes on the link (the int8 feed ran at the link's rate, 1.4-2.0 k gene-sets/s at N = 500 000).  A value above 2 is not a
// hard call: the gene then crosses as bytes, as before.
inline bool pack_i8_scalar(const signed char* g, size_t n, unsigned char* out) {
  for (size_t i = 0; i < n; i += 4) {
    unsigned b = 0;
    for (size_t e = 0; e < 4 && i + e < n; ++e) {
      const int v = g[i + e];
      if (v > 2) return false;
      const unsigned code = v < 0 ? 1u : (v == 0 ? 0u : (v == 1 ? 2u : 3u));
      b |= code << (2 * e);
    }
    out[i >> 2] = (unsigned char)b;
  }
  return true;
}
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
// 32 bytes -> 8 bytes of codes: the two code bits are two byte masks, deposited into the even and the odd bit positions
__attribute__((target("avx2,bmi2"))) inline bool pack32_i8_avx2(const signed char* g, unsigned char* out) {
  const __m256i v = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(g));
  const __m256i two = _mm256_set1_epi8(2), one = _mm256_set1_epi8(1);
  if (_mm256_movemask_epi8(_mm256_cmpgt_epi8(v, two))) return false;
  const __m256i is2 = _mm256_cmpeq_epi8(v, two), is1 = _mm256_cmpeq_epi8(v, one);
  const unsigned neg = (unsigned)_mm256_movemask_epi8(v);  // sign bits: missing
  const unsigned m2 = (unsigned)_mm256_movemask_epi8(is2), m1 = (unsigned)_mm256_movemask_epi8(is1);
  const unsigned long long x = _pdep_u64((unsigned long long)(m2 | neg), 0x5555555555555555ull) |
                               _pdep_u64((unsigned long long)(m1 | m2), 0xAAAAAAAAAAAAAAAAull);
  std::memcpy(out, &x, 8);
  return true;
}
#endif
// g[0 .. n) -> out[0 .. ceil(n / 4)), zeros up to `pitch` bytes; false = a value above 2.  isa: -1 widest, 0 scalar, 1 AVX2
inline bool pack_column_i8(const signed char* g, size_t n, unsigned char* out, size_t pitch, int isa = -1) {
  size_t i = 0;
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
  static const bool avx2 = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("bmi2");
  if ((isa < 0 || isa == 1) && avx2)
    for (; i + 32 <= n; i += 32)
      if (!pack32_i8_avx2(g + i, out + i / 4)) return false;
#endif
  if (i < n && !pack_i8_scalar(g + i, n - i, out + i / 4)) return false;
  const size_t o = (n + 3) / 4;
  if (o < pitch) std::memset(out + o, 0, pitch - o);
  return true;
}

// A ring of pinned staging chunks.  The owner supplies the pinned memory and three callbacks:
//   wait(k)                       block until the DMA that last read chunk k has finished
//   send(k, off, dst, bytes)      enqueue the DMA of chunk k's bytes [off, off + bytes) to device address dst
//   send2d(k, dst, dpitch, width, rows)  enqueue a 2-D DMA of `rows` packed rows of `width` bytes from the start of chunk k
//   sent(k)                       the DMAs of chunk k have been enqueued (record its event)
struct StageRing {
  std::vector<char*> chunk;  // pinned
  size_t chunk_bytes = 0;
  int next = 0;
  std::function<int(int)> wait;
  std::function<int(int, size_t, void*, size_t)> send;
  std::function<int(int, void*, size_t, size_t, size_t)> send2d;
  std::function<int(int)> sent;

  // contiguous host range -> contiguous device range
  int copy(void* dst, const void* src, size_t bytes, CopyPool& pool) {
    for (size_t o = 0; o < bytes; o += chunk_bytes) {
      const size_t n = std::min(chunk_bytes, bytes - o);
      const int k = next;
      next = (next + 1) % (int)chunk.size();
      if (int rc = wait(k)) return rc;
      pool.copy(chunk[k], (const char*)src + o, n);
      if (int rc = send(k, 0, (char*)dst + o, n)) return rc;
      if (int rc = sent(k)) return rc;
    }
    return 0;
  }
  // Many separate host pieces (the text of a gene's VCF records, its BGEN blocks) into ONE device range: piece i goes to
  // dst_base + dst_off[i] (offsets increasing, pieces not overlapping; the gaps between them and `tail_zero` bytes behind
  // the last one arrive as zeros).  The image of the device range is assembled chunk by chunk in pinned memory — one pool
  // batch and one DMA per chunk instead of one of each per piece (a 2 MB piece costs more in hand-offs than in bytes).
  struct Piece {
    size_t dst_off;
    const void* src;
    size_t bytes;
  };
  int copy_gather(void* dst_base, const Piece* t, size_t n, size_t tail_zero, CopyPool& pool) {
    std::vector<CopyPool::Task> tasks;
    size_t i = 0;
    while (i < n) {
      if (t[i].bytes + tail_zero > chunk_bytes) {  // a piece longer than a chunk: contiguous pieces of its own
        if (int rc = copy((char*)dst_base + t[i].dst_off, t[i].src, t[i].bytes, pool)) return rc;
        const size_t pend = t[i].dst_off + t[i].bytes;
        const size_t zlen = std::min(tail_zero, (i + 1 < n) ? t[i + 1].dst_off - pend : tail_zero);
        if (zlen > 0) {  // the zeros behind it
          const int kz = next;
          next = (next + 1) % (int)chunk.size();
          if (int rc = wait(kz)) return rc;
          std::memset(chunk[kz], 0, zlen);
          if (int rc = send(kz, 0, (char*)dst_base + pend, zlen)) return rc;
          if (int rc = sent(kz)) return rc;
        }
        ++i;
        continue;
      }
      const int k = next;
      next = (next + 1) % (int)chunk.size();
      if (int rc = wait(k)) return rc;
      const size_t base = t[i].dst_off;
      size_t end = base;
      tasks.clear();
      while (i < n && t[i].bytes + tail_zero <= chunk_bytes && t[i].dst_off + t[i].bytes + tail_zero - base <= chunk_bytes) {
        if (t[i].dst_off > end) std::memset(chunk[k] + (end - base), 0, t[i].dst_off - end);
        tasks.push_back(CopyPool::Task{chunk[k] + (t[i].dst_off - base), t[i].src, t[i].bytes});
        end = t[i].dst_off + t[i].bytes;
        ++i;
      }
      // zeros behind the last piece of this chunk: up to the next piece's start, or tail_zero behind the very last one
      const size_t zend = (i < n) ? std::min(t[i].dst_off, end + tail_zero) : end + tail_zero;
      if (zend > end) {
        std::memset(chunk[k] + (end - base), 0, zend - end);
        end = zend;
      }
      pool.run(tasks.data(), tasks.size());
      if (int rc = send(k, 0, (char*)dst_base + base, end - base)) return rc;
      if (int rc = sent(k)) return rc;
    }
    return 0;
  }
  // `nc` columns packed into rows of `dpitch` bytes at `base` (host memory) by the pool's threads; out[j] describes column j.
  // Returns false when a column is not representable (a second other value, an other value outside [0, 2]).
  static bool pack_columns_to(char* base, size_t dpitch, const double* src, size_t spitch_doubles, size_t n, size_t nc,
                              CopyPool& pool, PackedColumn* out) {
    std::atomic<int> stop{0};
    // a column is cut into `segs` runs of rows (multiples of 1 024 samples = 256 bytes of codes) so that the pool's threads
    // stay busy to the end — 50 columns on 16 threads were four rounds, the last one with two columns (round 6) —; a
    // column's runs must agree on its other value
    const size_t segs = std::max<size_t>(1, std::min<size_t>(8, (8 * (size_t)pool.threads() + nc - 1) / nc));
    const size_t seg = std::max<size_t>(1024, ((n + segs - 1) / segs + 1023) / 1024 * 1024), nseg = (n + seg - 1) / seg;
    std::vector<PackedColumn> part(nc * nseg);
    const std::function<void(size_t)> fn = [&](size_t item) {
      if (stop.load(std::memory_order_relaxed)) return;
      const size_t j = item / nseg, q = item % nseg, r0 = q * seg, len = std::min(seg, n - r0);
      unsigned char* o = reinterpret_cast<unsigned char*>(base + j * dpitch) + r0 / 4;
      part[item] = pack_column_f64(src + j * spitch_doubles + r0, len, o, q + 1 == nseg ? dpitch - r0 / 4 : len / 4, &stop);
      if (!part[item].ok) stop.store(1, std::memory_order_relaxed);
    };
    pool.run_items(nc * nseg, fn);
    if (stop.load()) return false;
    for (size_t j = 0; j < nc; ++j) {
      PackedColumn r;
      r.ok = true;
      for (size_t q = 0; q < nseg; ++q) {
        const PackedColumn& t = part[j * nseg + q];
        if (t.has_mu) {
          if (r.has_mu && std::memcmp(&r.mu, &t.mu, sizeof(double)) != 0) return false;  // two other values in one column
          r.has_mu = true;
          r.mu = t.mu;
        }
        r.n_other += t.n_other;
      }
      out[j] = r;
    }
    return true;
  }
  // The columns of an fp64 block packed on the way: column j = src + j * spitch_doubles, n doubles -> device row j of
  // `dpitch` bytes (2-bit codes, zero padded).  out[j] describes every column.  Returns 0 = sent, 1 = a HIP call failed,
  // 2 = not representable (nothing useful was sent; the caller sends the doubles).
  int pack_f64(void* dst, size_t dpitch, const double* src, size_t spitch_doubles, size_t n, size_t cols, CopyPool& pool,
               PackedColumn* out) {
    if (dpitch > chunk_bytes || cols == 0) return 2;
    const size_t per = std::max<size_t>(1, chunk_bytes / dpitch);
    for (size_t c0 = 0; c0 < cols; c0 += per) {
      const size_t nc = std::min(per, cols - c0);
      const int k = next;
      next = (next + 1) % (int)chunk.size();
      if (int rc = wait(k)) return rc;
      if (!pack_columns_to(chunk[k], dpitch, src + c0 * spitch_doubles, spitch_doubles, n, nc, pool, out + c0)) return 2;
      if (int rc = send(k, 0, (char*)dst + c0 * dpitch, nc * dpitch)) return rc;
      if (int rc = sent(k)) return rc;
    }
    return 0;
  }
  // The columns of an int8 block packed on the way (pack_column_i8): column j = src + j * spitch bytes, n samples -> device row j
  // of `dpitch` bytes.  Returns 0 = sent, 1 = a HIP call failed, 2 = a value above 2 (the caller sends the bytes).
  int pack_i8(void* dst, size_t dpitch, const signed char* src, size_t spitch, size_t n, size_t cols, CopyPool& pool) {
    if (dpitch > chunk_bytes || cols == 0) return 2;
    const size_t per = std::max<size_t>(1, chunk_bytes / dpitch);
    for (size_t c0 = 0; c0 < cols; c0 += per) {
      const size_t nc = std::min(per, cols - c0);
      const int k = next;
      next = (next + 1) % (int)chunk.size();
      if (int rc = wait(k)) return rc;
      // columns cut into runs of rows (multiples of 1 024 samples) so that all threads stay busy, as for the doubles
      const size_t segs = std::max<size_t>(1, std::min<size_t>(8, (4 * (size_t)pool.threads() + nc - 1) / nc));
      const size_t seg = std::max<size_t>(4096, ((n + segs - 1) / segs + 1023) / 1024 * 1024), nseg = (n + seg - 1) / seg;
      std::atomic<int> bad{0};
      char* base = chunk[k];
      const std::function<void(size_t)> fn = [&](size_t item) {
        if (bad.load(std::memory_order_relaxed)) return;
        const size_t j = item / nseg, q = item % nseg, r0 = q * seg, len = std::min(seg, n - r0);
        unsigned char* o = reinterpret_cast<unsigned char*>(base + j * dpitch) + r0 / 4;
        if (!pack_column_i8(src + (c0 + j) * spitch + r0, len, o, q + 1 == nseg ? dpitch - r0 / 4 : len / 4))
          bad.store(1, std::memory_order_relaxed);
      };
      pool.run_items(nc * nseg, fn);
      if (bad.load()) return 2;
      if (int rc = send(k, 0, (char*)dst + c0 * dpitch, nc * dpitch)) return rc;
      if (int rc = sent(k)) return rc;
    }
    return 0;
  }
  // `rows` rows of `width` bytes, spitch apart on the host, dpitch apart on the device (hipMemcpy2D's meaning)
  // pad_zero: the destination's pad bytes between rows may be written (with zeros): rows narrower than the device pitch by a
  // few bytes are then staged AT the device pitch and cross as ONE contiguous DMA
  int copy2d(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t rows, CopyPool& pool,
             bool pad_zero = false) {
    if (width == 0 || rows == 0) return 0;
    if (width > chunk_bytes) {  // a row is longer than a chunk: row by row, each in contiguous pieces
      for (size_t r = 0; r < rows; ++r)
        if (int rc = copy((char*)dst + r * dpitch, (const char*)src + r * spitch, width, pool)) return rc;
      return 0;
    }
    // Rows whose device pitch is only a few bytes wider than the rows (the 16-byte padding of packed genotype rows) are
    // staged AT THE DEVICE PITCH, pad bytes zero, and cross as ONE contiguous DMA: a 2-D copy of fifty 125 KB rows is fifty
    // DMA commands and ran at 31-35 GB/s of the link's 57 (tools/host_feed_bench: 2-bit genes 5.6 k/s -> see DESIGN 6).
    if (pad_zero && dpitch > width && dpitch - width <= 64 && dpitch <= chunk_bytes) {
      const size_t perp = std::max<size_t>(1, chunk_bytes / dpitch);
      std::vector<CopyPool::Task> tasks;
      for (size_t r0 = 0; r0 < rows; r0 += perp) {
        const size_t nr = std::min(perp, rows - r0);
        const int k = next;
        next = (next + 1) % (int)chunk.size();
        if (int rc = wait(k)) return rc;
        tasks.clear();
        for (size_t r = 0; r < nr; ++r) {
          tasks.push_back(CopyPool::Task{chunk[k] + r * dpitch, (const char*)src + (r0 + r) * spitch, width});
          if (r + 1 < nr) std::memset(chunk[k] + r * dpitch + width, 0, dpitch - width);
        }
        pool.run(tasks.data(), tasks.size());
        if (int rc = send(k, 0, (char*)dst + r0 * dpitch, (nr - 1) * dpitch + width)) return rc;
        if (int rc = sent(k)) return rc;
      }
      return 0;
    }
    const size_t per = std::max<size_t>(1, chunk_bytes / width);  // whole rows per chunk, packed
    std::vector<CopyPool::Task> tasks;
    for (size_t r0 = 0; r0 < rows; r0 += per) {
      const size_t nr = std::min(per, rows - r0);
      const int k = next;
      next = (next + 1) % (int)chunk.size();
      if (int rc = wait(k)) return rc;
      if (spitch == width) {
        pool.copy(chunk[k], (const char*)src + r0 * spitch, nr * width);
      } else {
        tasks.clear();
        for (size_t r = 0; r < nr; ++r)
          tasks.push_back(CopyPool::Task{chunk[k] + r * width, (const char*)src + (r0 + r) * spitch, width});
        pool.run(tasks.data(), tasks.size());
      }
      if (int rc = send2d(k, (char*)dst + r0 * dpitch, dpitch, width, nr)) return rc;
      if (int rc = sent(k)) return rc;
    }
    return 0;
  }
};

}  // namespace rvt
