// tools/host_feed_bench.cpp — the streaming hand-off measured from a C++ caller (what the drop-in adapters are): one thread
// submits genes from HOST memory through the C ABI exactly as ModelFitterGpu.cpp does (rvt_submit_gene_* per gene,
// rvt_collect_ready while the stream runs), no Python in the loop.  Prints one JSON object per mode: gene-sets/s, the host
// bytes per second that implies, and the caller thread's own microseconds per gene (the time inside rvt_submit_*: what one
// feeding thread spends per gene, i.e. the bound of a single-thread feed of several devices).
//   host_feed_bench [--samples N] [--m M] [--genes G] [--modes fp64,int8,bed] [--registered] [--batch B]
// build: g++ -std=c++17 -O2 tools/host_feed_bench.cpp -Iinclude -Lrvtests_amd/csrc -lrvtests_amd -Wl,-rpath,$ORIGIN/../rvtests_amd/csrc
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <sys/resource.h>

#include "rvtests_amd.h"

static uint64_t mix(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
// what the kernel did to this process's memory meanwhile (/proc/vmstat is machine-wide, getrusage is ours): automatic NUMA
// balancing unmaps pages to sample accesses (numa_hint_faults) and migrates them (numa_pages_migrated) — a staging copy whose
// source pages are being migrated under it runs at a fraction of its rate
struct VmSnap {
  long long hint_faults = -1, pages_migrated = -1, pgmigrate = -1, thp_collapse = -1;
  long minflt = 0, majflt = 0, nvcsw = 0, nivcsw = 0;
};
static VmSnap vm_snap() {
  VmSnap v;
  if (FILE* f = fopen("/proc/vmstat", "r")) {
    char key[128];
    long long val;
    while (fscanf(f, "%127s %lld", key, &val) == 2) {
      if (!strcmp(key, "numa_hint_faults")) v.hint_faults = val;
      else if (!strcmp(key, "numa_pages_migrated")) v.pages_migrated = val;
      else if (!strcmp(key, "pgmigrate_success")) v.pgmigrate = val;
      else if (!strcmp(key, "thp_collapse_alloc")) v.thp_collapse = val;
    }
    fclose(f);
  }
  struct rusage ru;
  if (getrusage(RUSAGE_SELF, &ru) == 0) {
    v.minflt = ru.ru_minflt;
    v.majflt = ru.ru_majflt;
    v.nvcsw = ru.ru_nvcsw;
    v.nivcsw = ru.ru_nivcsw;
  }
  return v;
}
static int numa_balancing_setting() {
  int v = -1;
  if (FILE* f = fopen("/proc/sys/kernel/numa_balancing", "r")) {
    if (fscanf(f, "%d", &v) != 1) v = -1;
    fclose(f);
  }
  return v;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
  if (!getenv("RVT_NO_PIN")) (void)rvt_pin_to_device_node(0);  // (as a main() built on the library does first)
  long long N = 500000;
  int M = 50, genes = 1536, window = 64, registered = 0, batch = 1;
  std::string modes = "int8,bed,fp64";
  for (int a = 1; a < argc; ++a) {
    if (!strcmp(argv[a], "--samples") && a + 1 < argc) N = atoll(argv[++a]);
    else if (!strcmp(argv[a], "--m") && a + 1 < argc) M = atoi(argv[++a]);
    else if (!strcmp(argv[a], "--genes") && a + 1 < argc) genes = atoi(argv[++a]);
    else if (!strcmp(argv[a], "--modes") && a + 1 < argc) modes = argv[++a];
    else if (!strcmp(argv[a], "--registered")) registered = 1;
    else if (!strcmp(argv[a], "--batch") && a + 1 < argc) batch = atoi(argv[++a]);  // genes per rvt_submit_genes call (1: per-gene calls)
  }
  rvt_ctx* ctx = nullptr;
  if (rvt_init(&ctx, 0)) { fprintf(stderr, "rvt_init failed\n"); return 2; }
  const int d = 3;
  std::vector<double> X((size_t)N * d), y(N);
  for (long long i = 0; i < N; ++i) {
    X[i] = 1.0;
    for (int k = 1; k < d; ++k) X[(size_t)k * N + i] = (double)(mix(i * 7 + k) >> 11) / 9007199254740992.0 - 0.5;
    y[i] = 0.3 * X[(size_t)N + i] + (double)(mix(i * 13 + 5) >> 11) / 9007199254740992.0;
  }
  if (rvt_fit_null(ctx, RVT_TRAIT_QUANTITATIVE, N, d, X.data(), y.data(), nullptr, nullptr)) {
    fprintf(stderr, "rvt_fit_null: %s\n", rvt_last_error(ctx));
    return 2;
  }
  const int K = batch > 4 ? batch : 4;  // distinct host buffers, reused in turn (the reference refills ONE matrix per gene)
  std::vector<std::vector<int8_t>> g8(K);
  for (int k = 0; k < K; ++k) {
    g8[k].resize((size_t)N * M);
    for (size_t i = 0; i < g8[k].size(); ++i) {
      const uint64_t h = mix(i * 31 + k);
      g8[k][i] = (int8_t)(((h & 0xffff) < 1300) + (((h >> 16) & 0xffff) < 1300));
      if (((h >> 40) & 0xfffff) < 1000) g8[k][i] = -9;  // ~0.1 % missing calls
    }
  }
  {  // what this machine gives the hand-off to work with: the figures below are bounded by stage_pool / h2d_pinned
    rvt_host_diag dg;
    if (rvt_host_diagnose(ctx, &dg) == 0)
      printf("{\"diag\": {\"hardware_threads\": %d, \"affinity_cpus\": %d, \"copy_threads\": %d, \"pack_threads\": %d, \"thp\": %d, "
             "\"gpu_numa_node\": %d, \"buffer_numa_node\": %d, \"pinned_numa_node\": %d, \"memcpy_one_thread_GBps\": %.1f, "
             "\"stage_pool_GBps\": %.1f, \"h2d_pinned_GBps\": %.1f, \"d2h_pinned_GBps\": %.1f, \"loadavg1\": %.1f, "
             "\"numa_balancing\": %d}}\n",
             dg.hardware_threads, dg.affinity_cpus, dg.copy_threads, dg.pack_threads, dg.thp, dg.gpu_numa_node, dg.buffer_numa_node,
             dg.pinned_numa_node, dg.memcpy_one_thread, dg.stage_pool, dg.h2d_pinned, dg.d2h_pinned, dg.loadavg1,
             numa_balancing_setting());
    fflush(stdout);
  }
  {  // Is the link ours yet?  As bench.py's child this tool starts right after the parent released ~100 GB of device memory; on
     // one run in six the first mode then ran at 8.7 GB/s of DMA (351 gene-sets/s in round 5's driver run, 349 in one of this
     // round's) while the second mode, seconds later, ran at the link's rate — the copies share the SDMA engines with whatever
     // the driver does to freed VRAM.  So: measure the pinned -> device rate until two measurements in a row reach 45 GB/s (at
     // most 12 tries of ~0.3 s), and print the series — a slow start is then visible instead of folded into the first mode.
    printf("{\"link_settle\": {\"h2d_pinned_GBps_series\": [");
    int good = 0;
    const double t_s = now();
    for (int k = 0; k < 12 && good < 2; ++k) {
      rvt_host_diag dg;
      if (rvt_host_diagnose(ctx, &dg) != 0) break;
      printf("%s%.1f", k ? ", " : "", dg.h2d_pinned);
      good = dg.h2d_pinned >= 45.0 ? good + 1 : 0;
    }
    printf("], \"seconds\": %.2f}}\n", now() - t_s);
    fflush(stdout);
  }
  rvt_params prm{1.0, 25.0, 1.0, 25.0, 0, 0.05};
  std::vector<rvt_gene_result> out(4096);
  size_t pos = 0;
  while (pos < modes.size()) {
    size_t e = modes.find(',', pos);
    if (e == std::string::npos) e = modes.size();
    const std::string mode = modes.substr(pos, e - pos);
    pos = e + 1;
    std::vector<std::vector<unsigned char>> buf(K);
    size_t bytes = 0;
    for (int k = 0; k < K; ++k) {
      if (mode == "int8") {
        buf[k].assign((const unsigned char*)g8[k].data(), (const unsigned char*)g8[k].data() + g8[k].size());
      } else if (mode == "bed") {  // PLINK 2-bit, SNP-major: 00 hom A1 (2), 01 missing, 10 het, 11 hom A2 (0)
        const size_t cb = (size_t)((N + 3) / 4);
        buf[k].assign(cb * M, 0);
        for (int j = 0; j < M; ++j)
          for (long long i = 0; i < N; ++i) {
            const int g = g8[k][(size_t)j * N + i];
            const unsigned code = g < 0 ? 1u : (g == 0 ? 3u : (g == 1 ? 2u : 0u));
            buf[k][(size_t)j * cb + (i >> 2)] |= (unsigned char)(code << (2 * (i & 3)));
          }
      } else {  // fp64 with missing codes
        buf[k].resize(sizeof(double) * (size_t)N * M);
        double* dd = (double*)buf[k].data();
        for (size_t i = 0; i < (size_t)N * M; ++i) dd[i] = (double)g8[k][i];
      }
      bytes = buf[k].size();
      if (registered && rvt_host_register(ctx, buf[k].data(), buf[k].size())) {
        fprintf(stderr, "rvt_host_register: %s\n", rvt_last_error(ctx));
        return 2;
      }
    }
    const int n_timed = (mode == "fp64") ? genes / 8 : genes;
    double t0 = 0, in_submit = 0;
    long long done = 0;
    VmSnap vm0;
    for (int g = -window; g < n_timed; ++g) {  // one untimed window first
      if (g == 0) {
        int n = 0;
        rvt_collect(ctx, out.data(), (int)out.size(), &n);
        t0 = now();
        in_submit = 0;
        done = 0;
        vm0 = vm_snap();
      }
      const unsigned char* b = buf[(g + window) % K].data();
      const double ts = now();
      int rc;
      if (batch > 1) {  // rvt_submit_genes: `batch` genes per call, each out of its own buffer
        if ((g + window) % batch != batch - 1 && g + 1 < n_timed) continue;
        const int nb = ((g + window) % batch) + 1;
        std::vector<int64_t> ids(nb);
        std::vector<int> Ms(nb, M);
        std::vector<const void*> ptrs(nb);
        for (int k = 0; k < nb; ++k) {
          ids[k] = g - (nb - 1) + k;
          ptrs[k] = buf[(g - (nb - 1) + k + window) % K].data();
        }
        rc = rvt_submit_genes(ctx, mode == "int8" ? 2 : (mode == "bed" ? 3 : 1), nb, ids.data(), Ms.data(), ptrs.data(), RVT_TEST_ALL, &prm);
      } else if (mode == "int8") rc = rvt_submit_gene_i8(ctx, g, M, (const int8_t*)b, RVT_TEST_ALL, &prm, nullptr);
      else if (mode == "bed") rc = rvt_submit_gene_bed(ctx, g, M, b, RVT_TEST_ALL, &prm, nullptr);
      else rc = rvt_submit_gene_raw(ctx, g, M, (const double*)b, RVT_TEST_ALL, &prm, nullptr);
      in_submit += now() - ts;
      if (rc) { fprintf(stderr, "submit: %s\n", rvt_last_error(ctx)); return 2; }
      if ((g + 1) % window == 0) {
        int n = 0;
        if (rvt_collect_ready(ctx, out.data(), (int)out.size(), &n)) return 2;
        done += n;
      }
    }
    for (;;) {
      int n = 0;
      if (rvt_collect(ctx, out.data(), (int)out.size(), &n)) return 2;
      if (n == 0) break;
      done += n;
    }
    const double dt = now() - t0;
    const VmSnap vm1 = vm_snap();
    printf("{\"mode\": \"%s%s\", \"N\": %lld, \"M\": %d, \"genes\": %lld, \"gene_sets_per_s\": %.1f, \"host_GBps\": %.2f, "
           "\"caller_us_per_gene\": %.1f, \"caller\": \"C++, one thread, %s\", \"during\": {\"numa_hint_faults\": %lld, "
           "\"numa_pages_migrated\": %lld, \"pgmigrate_success\": %lld, \"thp_collapse_alloc\": %lld, \"minor_faults\": %ld, "
           "\"major_faults\": %ld, \"voluntary_ctx_switches\": %ld, \"involuntary_ctx_switches\": %ld}}\n",
           mode.c_str(), registered ? "_registered" : "", N, M, done, done / dt, (double)bytes * done / dt / 1e9,
           1e6 * in_submit / n_timed, batch > 1 ? "rvt_submit_genes (batched)" : "rvt_submit_gene_* per gene",
           vm1.hint_faults - vm0.hint_faults, vm1.pages_migrated - vm0.pages_migrated, vm1.pgmigrate - vm0.pgmigrate,
           vm1.thp_collapse - vm0.thp_collapse, vm1.minflt - vm0.minflt, vm1.majflt - vm0.majflt, vm1.nvcsw - vm0.nvcsw,
           vm1.nivcsw - vm0.nivcsw);
    fflush(stdout);
    if (registered)
      for (int k = 0; k < K; ++k) rvt_host_unregister(ctx, buf[k].data());
  }
  rvt_destroy(ctx);
  return 0;
}
