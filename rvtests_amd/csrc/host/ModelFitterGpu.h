// rvtests_amd — host side (C++), mirror of the reference's plugin surface for the hot path.
//
// Same class names, constructor arguments, method names and output formats as the reference, so that these
// classes can be dropped into src/ModelManager.cpp in place of the CPU ones:
//   ModelFitter   src/ModelFitter.h:17-75     fit / writeHeader / writeOutput / writeFootnote / reset / setParameter
//   ModelParser   src/ModelParser.{h,cpp}     "name[k=v:k2=v2]", case-folded, ':' or ',' separated
//   ModelManager  src/ModelManager.cpp:26-44,99-103,168-198,273-297   create(type, "a[..],b")
//   SkatTest      src/Model.h:2612-2772       "Q\tPvalue"            %g
//   SkatOTest     src/Model.h:2774-2889       "Q\trho\tPvalue"       %g
//   CMCTest       src/Model.h:807-907         "NonRefSite\tPvalue"   Result / floatToString (6 significant digits)
//   ZegginiTest   src/Model.h:1170-1242       "Pvalue"
// Every fit() goes through the C ABI of include/rvtests_amd.h; nothing here computes statistics on the CPU.
//
// Inside the real rvtests tree the adapters read the reference's own `DataConsolidator`, `Matrix`, `FileWriter`
// and `Result`.  To keep this repository self-contained (and testable without Eigen) the few members the hot
// path touches are abstracted behind `GeneData` / `TextSink` below; INTEGRATION.md lists the one-line mapping
// of each onto the reference types.
#pragma once
#include <cstdint>
#include <cstdio>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#include "../../../include/rvtests_amd.h"

namespace rvt_host {

struct SiteInfo;

// ---- what fit() may read: DataConsolidator getters (src/DataConsolidator.h:126-137,223-224) -------------
// What MetaScoreTest reads from the caller's GenotypeCounter after dc->countRawGenotype(0, &counter)
// (src/Model.h:3211-3230; libsrc/GenotypeCounter.h): the counting and the exact HWE test stay with the caller
// (DataConsolidator / GenotypeCounter are not on the accelerated path), the adapter only prints them.
struct SiteCounts {
  double af = -1.0;       // getAF(); < 0 prints NA
  double ac = 0.0;        // getAC()
  double callRate = 0.0;  // getCallRate()
  double hwe = 0.0;       // getHWE()
  int nHomRef = 0, nHet = 0, nHomAlt = 0;
};

struct GeneData {
  int64_t N = 0;
  int M = 0;
  const double* genotype = nullptr;   // dc->getGenotype(): imputed, unflipped, N x M column-major
  // Optional, INSTEAD of `genotype` for the gene tests (SKAT / SKAT-O / CMC / Zeggini): the gene as the extractor of a PLINK
  // file holds it BEFORE consolidation — M rows of ceil(N / 4) bytes, SNP-major 2-bit codes (libVcf/PlinkInputFile.cpp:24-47;
  // 00 -> 0, 10 -> 1, 11 -> 2, 01 -> missing).  The device then does what DataConsolidator::consolidate does (allele
  // frequencies, mean imputation: rvt_submit_gene_bed); markerFrequency is not needed.  1/32 of the bytes of `genotype`.
  const unsigned char* bed = nullptr;
  const double* phenotype = nullptr;  // dc->getPhenotype(): N
  const double* covariate = nullptr;  // dc->getCovariate(): N x ncov column-major, NO intercept
  int ncov = 0;
  std::vector<double> markerFrequency;  // dc->getMarkerFrequency(col) for col < M
  bool phenotypeUpdated = false, covariateUpdated = false;  // dc->isPhenotypeUpdated() / isCovariateUpdated()
  int64_t serial = 0;                 // increases with every dc.consolidate() (new gene)
  const SiteInfo* site = nullptr;     // dc->getResult(): CHROM / POS of the current site (single-variant models)
  // dc->hasKinship(), getKinshipUForAuto() / getKinshipSForAuto() (src/DataConsolidator.h:236-258): EigenMatrix holds
  // Eigen::MatrixXf, i.e. float, column-major N x N and N x 1
  const float* kinshipU = nullptr;
  const float* kinshipS = nullptr;
  // MetaScoreTest: raw-genotype counters of the current site — all samples, cases, controls (binary traits only)
  SiteCounts counter, caseCounter, ctrlCounter;
};

// ---- FileWriter stand-in (base/IO.h FileWriter::write / printf) -------------------------------------------
struct TextSink {
  std::string text;
  virtual ~TextSink() {}
  virtual void write(const std::string& s) { text += s; }  // in_tree/GpuModelFitter.h forwards to FileWriter::write
  void write(const char* s) { write(std::string(s)); }
};

// site columns the caller passes to writeHeader/writeOutput (Result::writeHeaderTab / writeValueTab)
struct SiteInfo {
  std::vector<std::pair<std::string, std::string>> kv;
  // set by the in-tree binding, which takes both lines from the reference's Result (joinValue): used verbatim
  bool verbatim = false;
  std::string headerLine, valueLine;
  std::string headerTab() const {
    if (verbatim) return headerLine;
    std::string s;
    for (auto& p : kv) s += p.first + "\t";
    return s;
  }
  std::string valueTab() const {
    if (verbatim) return valueLine;
    std::string s;
    for (auto& p : kv) s += p.second + "\t";
    return s;
  }
  std::string get(const std::string& key) const {  // Result::operator[]
    for (auto& p : kv)
      if (p.first == key) return p.second;
    return "";
  }
};

std::string floatToString(double v);  // base/TypeConversion.h:100-105 (6 significant digits)
std::string formatG(double v);        // printf("%g")
size_t formatG(double v, char* out);  // the same characters into out (at least 32 bytes, not terminated); returns their number

// ---- ModelParser --------------------------------------------------------------------------------------------
class ModelParser {
 public:
  int parse(const std::string& s);
  const std::string& getName() const { return name; }
  bool hasTag(const std::string& tag) const;
  const char* value(const std::string& tag) const;
  size_t size() const { return param.size(); }
  const ModelParser& assign(const std::string& tag, double* v, double def) const;
  const ModelParser& assign(const std::string& tag, int* v, int def) const;
  const ModelParser& assign(const std::string& tag, bool* v, bool def) const;
  void set(const std::string& tag, const std::string& value);  // (in-tree binding: copy a tag of the reference's parser)

 private:
  std::string name;
  std::map<std::string, std::string> param;
};

class ModelFitter;

// ---- the engine shared by all GPU-backed models of one run ---------------------------------------------------
// One device group per process (rvt_group_*: RVT_DEVICES=0,1,... lists the GPUs, default device 0): the gene tests'
// stream is dealt to the members and collected in submission order; the models that drive a context themselves
// (MetaCov, MetaScore, the related-sample tests) use member 0.  The null model is installed once on every member (and
// again when the caller flags an updated phenotype/covariate); each new gene is submitted ONCE with the union of the
// registered tests, whichever model's fit() sees it first.
class GpuBroker {
 public:
  static GpuBroker& instance();
  int ensureContext(int device);
  void registerTests(uint32_t mask, const rvt_params& p);
  // Deferred, batched execution.  fit() only SUBMITS the gene (rvt_submit_gene copies the caller's buffer, which
  // the next consolidate() overwrites); writeOutput() only records (sink, site columns); rows are written in call
  // order by flush(), which runs one rvt_collect over everything pending.  flush() is triggered when `window` genes
  // are pending and a new one arrives, by writeFootnote() and by ~ModelManager — the reference's own MetaCovTest
  // defers its rows the same way (src/Model.cpp:828-834), and `main` ignores fit()'s return value
  // (src/Main.cpp:1251).  Default window: 64 genes or 64 GB of genotype blocks, whichever fills first (RVT_ADAPTER_BATCH /
  // RVT_ADAPTER_BATCH_GB override; window = 1 keeps at most one gene in flight).
  void setBatchWindow(int k) { window = k < 1 ? 1 : k; }
  void setBatchBytes(size_t b) { windowBytes = b; }
  // `--dosage TAG` (src/Main.cpp FLAG_dosageTag): the blocks fit() receives hold dosages, not hard calls; `decimals` = the
  // number of decimals the tag is printed with (3 for an imputation server's DS; -1 = unknown / not decimal text, e.g.
  // BGEN): states the lattice 10^decimals to the engine (rvt_group_set_content).  Call before the first fit(); RVT_DOSAGE
  // (1 / the number of decimals as "d3") does the same from the environment.  Never affects the records.
  void setDosage(int decimals) {
    dosage = true;
    dosageDecimals = decimals;
  }
  int submit(const GeneData& gd, bool binary, std::string* err);
  void enqueue(ModelFitter* m, TextSink* fp, const std::string& siteTab, int64_t serial);
  int flush();       // wait for everything pending and write all rows
  int drainReady();  // take the finished prefix without waiting (rvt_collect_ready) and write the rows it completes
  void shutdown();
  // context + null model for models that drive the C ABI themselves (MetaCovTest)
  rvt_ctx* contextWithNull(const GeneData& gd, bool binary, std::string* err);
  // context + kinship + FastLMM null for FamSkatTest (refitted when the caller flags new phenotype / covariates)
  rvt_ctx* contextWithFamNull(const GeneData& gd, std::string* err);
  // the related-sample gene tests (FamSkat, FamCMC, FamZeggini) share one rotation per gene: the first model whose
  // fit() sees a gene runs the union of the registered tests, the others read the cached record
  void registerFamTests(uint32_t mask) { famTests |= mask; }
  const rvt_gene_result* famResultFor(const GeneData& gd, std::string* err);
  // null model: fitted on the device (rvt_fit_null) unless the caller installs its own routine (e.g. the
  // reference's LinearRegression / LogisticRegression inside the rvtests tree); see INTEGRATION.md
  typedef int (*NullFitter)(bool binary, int64_t N, int d, const double* X, const double* y, double* res, double* v,
                            double* sigma2);
  void setNullFitter(NullFitter f) { fitter = f; }
  const rvt_fam_null& familyNull() const { return famNull; }  // estimates of the FastLMM null last fitted

 private:
  rvt_group* grp = nullptr;
  rvt_ctx* ctx = nullptr;  // member 0
  bool dosage = false;
  int dosageDecimals = -1;
  uint32_t tests = 0;
  rvt_params params{1.0, 25.0, 1.0, 25.0, 0, 0.05};
  bool haveNull = false;
  rvt_fam_null famNull{};
  int64_t curSerial = -1;
  bool curOk = false;
  int window = 64;                              // genes in flight (RVT_ADAPTER_BATCH)
  size_t windowBytes = (size_t)64 << 30;        // ... and the bytes of their device blocks (RVT_ADAPTER_BATCH_GB)
  size_t pendingBytes = 0;
  struct Row {
    ModelFitter* model;
    TextSink* fp;
    std::string siteTab;
    int64_t serial;
  };
  std::vector<Row> rows;               // in writeOutput() order
  std::vector<int64_t> pendingSerial;  // genes submitted and not yet collected, submission order
  std::map<int64_t, rvt_gene_result> ready;  // records collected, waiting for their rows to be written
  std::map<int64_t, size_t> recBytes;        // device bytes of each pending gene
  void writeReadyRows(bool all);
  std::vector<int64_t> failedSerial;   // genes whose submission failed: NA rows
  NullFitter fitter = nullptr;
  const float* kinU = nullptr;
  bool haveFamNull = false;
  uint32_t famTests = 0;
  int64_t famSerial = -1;
  bool famOk = false;
  rvt_gene_result famRec{};
  int installNull(const GeneData& gd, bool binary, std::string* err);
};

// ---- ModelFitter ---------------------------------------------------------------------------------------------------
class ModelFitter {
 public:
  virtual int fit(GeneData* dc) = 0;
  virtual void writeHeader(TextSink* fp, const SiteInfo& siteInfo) = 0;
  virtual void writeOutput(TextSink* fp, const SiteInfo& siteInfo) = 0;
  virtual void writeFootnote(TextSink*) {}
  virtual int setParameter(const ModelParser&) { return 0; }
  virtual void reset() {}
  // one output row (without the site columns, with the newline) from a collected record; r == nullptr -> NA row
  virtual std::string formatRow(const rvt_gene_result* r) const {
    (void)r;
    return "\n";
  }
  virtual ~ModelFitter() {}
  const std::string& getModelName() const { return modelName; }
  bool isBinaryOutcome() const { return binaryOutcome; }
  void setBinaryOutcome() { binaryOutcome = true; }
  void setQuantitativeOutcome() { binaryOutcome = false; }

 protected:
  std::string modelName = "UninitializedModel";
  bool binaryOutcome = false;
  int64_t curSerial = -1;  // gene handed to the last fit()
  std::string lastError;
  // shared by the gene-level GPU models: fit() = submit, writeOutput() = enqueue, writeFootnote() = flush
  int deferredFit(GeneData* dc);
  void deferredOutput(TextSink* fp, const SiteInfo& siteInfo);
};

class SkatTest : public ModelFitter {
 public:
  std::string formatRow(const rvt_gene_result* r) const override;
  void writeFootnote(TextSink* fp) override;
  SkatTest(int nPerm, double alpha, double beta1, double beta2);
  int fit(GeneData* dc) override;
  void writeHeader(TextSink* fp, const SiteInfo& siteInfo) override;
  void writeOutput(TextSink* fp, const SiteInfo& siteInfo) override;

 private:
  bool usePermutation;
};

class SkatOTest : public ModelFitter {
 public:
  std::string formatRow(const rvt_gene_result* r) const override;
  void writeFootnote(TextSink* fp) override;
  SkatOTest(double beta1, double beta2);
  int fit(GeneData* dc) override;
  void writeHeader(TextSink* fp, const SiteInfo& siteInfo) override;
  void writeOutput(TextSink* fp, const SiteInfo& siteInfo) override;

 private:
};

class CMCTest : public ModelFitter {
 public:
  std::string formatRow(const rvt_gene_result* r) const override;
  void writeFootnote(TextSink* fp) override;
  CMCTest();
  int fit(GeneData* dc) override;
  void writeHeader(TextSink* fp, const SiteInfo& siteInfo) override;
  void writeOutput(TextSink* fp, const SiteInfo& siteInfo) override;

 private:
};

class ZegginiTest : public ModelFitter {
 public:
  std::string formatRow(const rvt_gene_result* r) const override;
  void writeFootnote(TextSink* fp) override;
  ZegginiTest();
  int fit(GeneData* dc) override;
  void writeHeader(TextSink* fp, const SiteInfo& siteInfo) override;
  void writeOutput(TextSink* fp, const SiteInfo& siteInfo) override;

 private:
};

// `--vt analytic` (src/ModelManager.cpp:158-159; AnalyticVT(UNRELATED), src/Model.h:2105-2259): quantitative traits only,
// columns MinMAF MaxMAF OptimMAF OptimNumVar U V Stat Pvalue.  The reference's p-value comes from a randomised rule at
// absolute accuracy 1e-3 and the row is NA when that rule's error estimate exceeds it (MvtNorm::compute_Band); here the
// integral is evaluated deterministically and the row is NA when ITS error estimate exceeds 1e-3.
// `--vt famanalytic` (AnalyticVT(RELATED), :160-161): the same columns from FastLMM's frequencies, scores and variances
// (rvt_fam_analytic_vt); synchronous, one gene per call.
class AnalyticVTTest : public ModelFitter {
 public:
  std::string formatRow(const rvt_gene_result* r) const override;
  void writeFootnote(TextSink* fp) override;
  explicit AnalyticVTTest(bool related = false);
  int fit(GeneData* dc) override;
  void writeHeader(TextSink* fp, const SiteInfo& siteInfo) override;
  void writeOutput(TextSink* fp, const SiteInfo& siteInfo) override;

 private:
  bool related;
  bool fitOK = false;
  rvt_gene_result rec{};
};

// `--kernel kbac[nPerm=10000:alpha=0.05]` (src/ModelManager.cpp kernel switch; KBACTest, src/Model.h:2891-3045): binary
// traits without covariates, one column "Pvalue" printed with %f.  Synchronous: the permutations consume the process-wide
// random stream gene by gene.
class KbacTest : public ModelFitter {
 public:
  KbacTest(int nPerm, double alpha);
  int fit(GeneData* dc) override;
  void writeHeader(TextSink* fp, const SiteInfo& siteInfo) override;
  void writeOutput(TextSink* fp, const SiteInfo& siteInfo) override;

 private:
  int nPerm;
  double alpha;
  bool fitOK = false;
  rvt_kbac_result rec{};
};

// `--kernel famSkat[beta1:beta2]` (src/Model.h:3048-3145).  The reference ignores beta1 / beta2 for this model
// (FamSkat.cpp:129-137 always uses Beta(1, 25)); so does this adapter.
class FamSkatTest : public ModelFitter {
 public:
  FamSkatTest(double beta1, double beta2);
  int fit(GeneData* dc) override;
  void writeHeader(TextSink* fp, const SiteInfo& siteInfo) override;
  void writeOutput(TextSink* fp, const SiteInfo& siteInfo) override;

 private:
  bool fitOK = false;
  rvt_gene_result rec{};
};

// `--burden famcmc` / `--burden famzeggini` (src/Model.h:2261-2492): collapse + FastLMM score test.
class FamBurdenTest : public ModelFitter {
 public:
  explicit FamBurdenTest(bool zeggini);
  int fit(GeneData* dc) override;
  void writeHeader(TextSink* fp, const SiteInfo& siteInfo) override;
  void writeOutput(TextSink* fp, const SiteInfo& siteInfo) override;

 private:
  bool zeggini;
  bool fitOK = false;
  double effect = -1.0;  // the reference leaves the previous value when V == 0 (src/Model.h:2338-2340)
  rvt_gene_result rec{};
};

// `--meta cov[windowSize=..:gwama]` for unrelated samples.  fit() is called once per variant (genotype.cols == 1,
// src/Model.cpp:844-858) and only copies the column into a device-resident ring; covariance rows are produced block
// by block on the GPU (rvt_cov_block) and written in the reference's order and format when their window is complete
// — the reference itself defers each row until its head is evicted (src/Model.h:3956-3968), so deferring changes
// when a row reaches the file, not what the file holds.  A window that holds more sites than the ring — and, since round 5,
// a flush that could emit less than half of the ring — makes the ring grow (up to RVT_METACOV_MAX_COLUMNS and 96 GB of columns,
// RVT_METACOV_RING_GB): with at least two windows in the ring every flush emits half of what it reads.  Rings wider than one
// block of the symmetric kernel are processed as heads-by-window rectangles of up to 1 024 heads (rvt_cov_rect; the exact int8
// product for hard calls like the symmetric block).  Rows still pending are flushed by writeFootnote() / the
// destructor, as the reference's destructor does (src/Model.cpp:828-834).
class MetaCovTest : public ModelFitter {
 public:
  explicit MetaCovTest(int windowSize);
  ~MetaCovTest() override;
  int setParameter(const ModelParser& parser) override;
  int fit(GeneData* dc) override;
  void writeHeader(TextSink* fp, const SiteInfo& siteInfo) override;
  void writeOutput(TextSink* fp, const SiteInfo& siteInfo) override;
  void writeFootnote(TextSink* fp) override;

 private:
  struct Site {
    std::string chrom;
    int pos;
  };
  int flush(bool final);
  int grow();
  int windowSize;
  int capacity = 4096;                 // columns of the device ring (grows until it holds four windows)
  int head = 0;                        // physical column of sites[0]: site k lives in column (head + k) mod capacity
  int maxColumns = 65536;              // RVT_METACOV_MAX_COLUMNS
  bool canGrow = true;                 // false once a non-mandatory grow() failed (not retried on every fill)
  int formatThreads = 1;               // threads that turn the band of a flush into text (RVT_METACOV_FORMAT_THREADS; default up to 8)
  std::vector<float> bandBuf;          // where the band of a flush lands (page-locked: rvt_host_register)
  float* bandReg = nullptr;
  bool outputGwama = false;
  bool fitOK = false;
  bool useFamilyModel = false;
  int64_t nSample = -1;
  int nCovariate = 0;
  rvt_ctx* ctx = nullptr;
  double* block = nullptr;   // the device ring: `capacity` columns, never compacted
  std::vector<Site> sites;   // variants currently in the ring, file order
  TextSink* fout = nullptr;
};

// `--meta score` (src/Model.h:3155-3398): MetaUnrelatedQtl / MetaUnrelatedBinary, and MetaFamQtl / MetaFamBinary when the
// caller hands over a kinship decomposition (the BOLT variants are not provided).  Sites are copied
// into a device block as fit() sees them; a full block (or writeFootnote / the destructor) runs ONE rvt_score_block
// over all of them and writes their rows in file order, after the summary header with the null-model estimates.
class MetaScoreTest : public ModelFitter {
 public:
  MetaScoreTest();
  ~MetaScoreTest() override;
  int setParameter(const ModelParser& parser) override;
  int fit(GeneData* dc) override;
  void writeHeader(TextSink* fp, const SiteInfo& siteInfo) override;
  void writeOutput(TextSink* fp, const SiteInfo& siteInfo) override;
  void writeFootnote(TextSink* fp) override;
  std::vector<std::string> covLabel;  // g_SummaryHeader->getCovLabel() (src/Model.h:3287-3289): set by the caller

 private:
  struct Row {
    std::string siteTab;            // filled by writeOutput
    SiteCounts all, cases, ctrls;
    bool tested = false;            // fit() reached the score test (false: the row prints the counters only)
    bool written = false;           // writeOutput was called for this site
    int column = -1;                // column in the device block
  };
  int flush();
  int capacity = 1024;              // RVT_METASCORE_BLOCK
  bool outputSE = false;
  bool useFamilyModel = false;
  double famB = 1.0;                // MetaFamBinary: b
  bool headerOutputted = false;
  std::string siteHeaderTab;
  int64_t nSample = -1;
  int nCovariate = 0;
  int used = 0;                     // columns of the block in use
  rvt_ctx* ctx = nullptr;
  double* block = nullptr;
  std::vector<Row> rows;
  TextSink* fout = nullptr;
};

// ---- ModelManager::create -----------------------------------------------------------------------------------------
class ModelManager {
 public:
  ~ModelManager();
  // type: "burden" | "kernel" | "meta"; modelList: "cmc,zeggini", "skat[nPerm=0:beta1=1],skato" or "cov[windowSize=500000]"
  int create(const std::string& type, const std::string& modelList);
  const std::vector<ModelFitter*>& getModel() const { return model; }
  void setBinaryOutcome();
  void setQuantitativeOutcome();
  // "<prefix>.<ModelName>.assoc" names (src/ModelManager.cpp:285-297)
  std::vector<std::string> outputNames(const std::string& prefix) const;
  std::string lastError;

 private:
  std::vector<ModelFitter*> model;
};

}  // namespace rvt_host
