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
          q_.push_back(Piece{(char*)tasks[i].dst + o, (const char*)tasks[i].src + o, std::min(piece, tasks[i].bytes - o), &left});
          ++left;
        }
    }
    cv_.notify_all();
    help();  // the calling thread works too
    std::unique_lock<std::mutex> lk(m_);
    done_.wait(lk, [&left] { return left == 0; });
  }

 private:
  struct Piece {
    void* dst;
    const void* src;
    size_t bytes;
    size_t* left;  // the batch's counter
  };
  bool take(Piece* t) {
    std::lock_guard<std::mutex> lk(m_);
    if (q_.empty()) return false;
    *t = q_.front();
    q_.pop_front();
    return true;
  }
  void finish_one(size_t* left) {
    std::lock_guard<std::mutex> lk(m_);
    if (--*left == 0) done_.notify_all();
  }
  void help() {
    Piece t;
    while (take(&t)) {
      std::memcpy(t.dst, t.src, t.bytes);
      finish_one(t.left);
    }
  }
  void loop() {
    for (;;) {
      {
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
  bool stop_ = false;
};

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
