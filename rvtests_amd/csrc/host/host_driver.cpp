// rvtests_amd — minimal host driver reproducing the reference's group-mode gene loop for the GPU models
// (src/Main.cpp:1207-1256: for each gene { consolidate; for each model { reset(); fit(&dc); writeOutput(); } })
// on pre-consolidated inputs read from a binary file.  Used by tests/test_host_driver.py; the full VCF front
// end is out of scope (SURVEY §8f "next" #1).
//
//   host_driver <input.bin> <kernel list, e.g. "skat[nPerm=0],skato" or "-"> <burden list, e.g. "cmc,zeggini" or "-">
//               [<meta list, e.g. "cov[windowSize=3000]"> <sites.txt: one "chrom pos" line per variant, file order>]
//               [... <kinship.bin: int64 N; float U[N*N] (column-major); float S[N]>]   (7th argument, for famSkat)
// With a meta list the driver runs the reference's single-variant loop instead (src/Main.cpp:1010-1078): every
// column of every block is one fit() call with CHROM / POS in the site record.
//
// input.bin: int64 N; int32 ncov, binary, ngenes; double y[N]; double cov[N*ncov] (column-major);
//            per gene: int32 M; double af[M]; double G[N*M] (column-major, imputed, unflipped)
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "ModelFitterGpu.h"

#include <execinfo.h>
#include <memory>
#include <signal.h>
#include <unistd.h>
using namespace rvt_host;

static void on_fatal_signal(int sig) {  // test driver: say where it happened
  void* frames[64];
  const int n = backtrace(frames, 64);
  fprintf(stderr, "host_driver: signal %d\n", sig);
  backtrace_symbols_fd(frames, n, 2);
  _exit(128 + sig);
}

int main(int argc, char** argv) {
  signal(SIGSEGV, on_fatal_signal);
  signal(SIGABRT, on_fatal_signal);
  if (argc < 4) {
    fprintf(stderr, "usage: host_driver input.bin <kernel list|-> <burden list|->\n");
    return 2;
  }
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 2;
  int64_t N;
  int32_t ncov, binary, ngenes;
  if (fread(&N, 8, 1, f) != 1 || fread(&ncov, 4, 1, f) != 1 || fread(&binary, 4, 1, f) != 1 ||
      fread(&ngenes, 4, 1, f) != 1)
    return 2;
  std::vector<double> y(N), cov((size_t)N * ncov);
  if (fread(y.data(), 8, N, f) != (size_t)N) return 2;
  if (ncov && fread(cov.data(), 8, (size_t)N * ncov, f) != (size_t)N * ncov) return 2;

  std::unique_ptr<ModelManager> mmp(new ModelManager());  // destroyed BEFORE the broker's contexts (models free device blocks)
  ModelManager& mm = *mmp;
  if (std::string(argv[2]) != "-" && mm.create("kernel", argv[2])) {
    fprintf(stderr, "%s\n", mm.lastError.c_str());
    return 1;
  }
  if (std::string(argv[3]) != "-" && mm.create("burden", argv[3])) {
    fprintf(stderr, "%s\n", mm.lastError.c_str());
    return 1;
  }
  if (const char* vt = getenv("RVT_DRIVER_VT")) {  // `--vt analytic`
    if (mm.create("vt", vt)) {
      fprintf(stderr, "%s\n", mm.lastError.c_str());
      return 1;
    }
  }
  const bool metaMode = argc >= 6 && std::string(argv[4]) != "-";
  std::vector<std::pair<std::string, std::string>> sitePos;
  if (metaMode) {
    if (mm.create("meta", argv[4])) {
      fprintf(stderr, "%s\n", mm.lastError.c_str());
      return 1;
    }
    FILE* sf = fopen(argv[5], "r");
    if (!sf) {
      fprintf(stderr, "cannot open %s\n", argv[5]);
      return 1;
    }
    char cbuf[64], pbuf[64];
    while (fscanf(sf, "%63s %63s", cbuf, pbuf) == 2) sitePos.emplace_back(cbuf, pbuf);
    fclose(sf);
  }
  if (binary)
    mm.setBinaryOutcome();
  else
    mm.setQuantitativeOutcome();
  const auto& models = mm.getModel();
  for (auto* m : models)
    if (auto* ms = dynamic_cast<MetaScoreTest*>(m))
      for (int k = 0; k < ncov; ++k) ms->covLabel.push_back("cov" + std::to_string(k + 1));
  std::vector<TextSink> outs(models.size());
  SiteInfo site;
  site.kv = {{"Range", ""}, {"N_INFORMATIVE", std::to_string(N)}, {"NumVar", ""}, {"NumPolyVar", ""}};
  for (size_t m = 0; m < models.size(); ++m) models[m]->writeHeader(&outs[m], site);

  std::vector<float> kinU, kinS;
  if (argc >= 7) {
    FILE* kf = fopen(argv[6], "rb");
    int64_t kn = 0;
    if (!kf || fread(&kn, 8, 1, kf) != 1 || kn != N) {
      fprintf(stderr, "bad kinship file %s\n", argv[6]);
      return 1;
    }
    kinU.resize((size_t)N * N);
    kinS.resize(N);
    if (fread(kinU.data(), 4, kinU.size(), kf) != kinU.size() || fread(kinS.data(), 4, N, kf) != (size_t)N) return 2;
    fclose(kf);
  }
  GeneData dc;
  dc.N = N;
  if (!kinU.empty()) {
    dc.kinshipU = kinU.data();
    dc.kinshipS = kinS.data();
  }
  dc.phenotype = y.data();
  dc.covariate = cov.data();
  dc.ncov = ncov;
  std::vector<double> G;
  size_t variantIndex = 0;
  for (int g = 0; g < ngenes; ++g) {
    int32_t M;
    if (fread(&M, 4, 1, f) != 1) return 2;
    dc.markerFrequency.resize(M);
    G.resize((size_t)N * M);
    if (fread(dc.markerFrequency.data(), 8, M, f) != (size_t)M) return 2;
    if (fread(G.data(), 8, (size_t)N * M, f) != (size_t)N * M) return 2;
    if (metaMode) {  // single-variant loop: one fit() per column
      for (int j = 0; j < M; ++j) {
        if (variantIndex >= sitePos.size()) return 2;
        SiteInfo vs;
        vs.kv = {{"CHROM", sitePos[variantIndex].first}, {"POS", sitePos[variantIndex].second}};
        dc.M = 1;
        dc.genotype = G.data() + (size_t)j * N;
        dc.serial = (int64_t)(++variantIndex);
        dc.site = &vs;
        // stand-in for dc->countRawGenotype(0, &counter) (+ case / control counters): this harness's genotypes are
        // already imputed, the HWE exact test is the caller's and is not run here
        for (int grp = 0; grp < (binary ? 3 : 1); ++grp) {
          SiteCounts sc;
          double sum = 0.0;
          int64_t n = 0;
          for (int64_t i = 0; i < N; ++i) {
            if (grp == 1 && y[i] != 1.0) continue;
            if (grp == 2 && y[i] != 0.0) continue;
            const double gv = dc.genotype[i];
            sum += gv;
            ++n;
            const long r = std::lround(gv);
            if (r <= 0) ++sc.nHomRef;
            else if (r == 1) ++sc.nHet;
            else ++sc.nHomAlt;
          }
          sc.ac = sum;
          sc.af = n ? sum / (2.0 * n) : -1.0;
          sc.callRate = 1.0;
          sc.hwe = 1.0;
          (grp == 0 ? dc.counter : grp == 1 ? dc.caseCounter : dc.ctrlCounter) = sc;
        }
        for (size_t m = 0; m < models.size(); ++m) {
          models[m]->reset();
          models[m]->fit(&dc);
          models[m]->writeOutput(&outs[m], vs);
        }
        dc.site = nullptr;
      }
      continue;
    }
    dc.M = M;
    dc.genotype = G.data();
    dc.serial = g + 1;  // dc.consolidate(...) happened
    site.kv[0].second = "gene" + std::to_string(g);
    site.kv[2].second = std::to_string(M);
    for (size_t m = 0; m < models.size(); ++m) {
      models[m]->reset();
      models[m]->fit(&dc);
      models[m]->writeOutput(&outs[m], site);
    }
  }
  fclose(f);
  for (size_t m = 0; m < models.size(); ++m) models[m]->writeFootnote(&outs[m]);  // ModelManager::close, :304-314
  const auto names = mm.outputNames("out");
  for (size_t m = 0; m < models.size(); ++m) printf("== %s\n%s", names[m].c_str(), outs[m].text.c_str());
  mmp.reset();
  GpuBroker::instance().shutdown();
  return 0;
}
