// rvtests_amd — host side (C++): GPU-backed ModelFitter adapters, ModelParser, ModelManager::create.
// See ModelFitterGpu.h for the reference lines each class mirrors.
#include "ModelFitterGpu.h"

#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <iterator>
#include <thread>

namespace rvt_host {

std::string floatToString(double v) {
  std::stringstream ss;
  ss.precision(6);
  ss << std::noshowpoint << v;
  return ss.str();
}

// printf("%g") without printf: a row of MetaCov's window is a thousand of these, and glibc's conversion (arbitrary precision,
// locale, stream state) costs ~200 ns each — the formatting threads of the adapter spent more on a flush than the device.
// The six significant digits are round-to-nearest of v x 10^(5 - k) computed in double: v x 10^p is ONE rounding of an exact
// product for |p| <= 22 (10^p is a double), i.e. off by at most 1.2e-10 at 10^6, so the digits are those of the exact decimal
// expansion unless the fraction lies within 1e-7 of one half — those (and everything outside the range, infinities, NaN) go
// to snprintf.  Same characters as "%g" in the C locale: fixed notation for exponents -4 .. 5, otherwise d.ddddde+XX, trailing
// zeros and a trailing point removed, "-0" for a negative zero.  tests/test_host_format_cpu.py: 3 million values against printf.
size_t formatG(double v, char* out) {
  static const double p10[23] = {1e0,  1e1,  1e2,  1e3,  1e4,  1e5,  1e6,  1e7,  1e8,  1e9,  1e10, 1e11,
                                 1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
  char* o = out;
  double a = v;
  if (std::signbit(v)) {
    a = -v;
    *o++ = '-';
  }
  if (a == 0.0) {
    *o++ = '0';
    return (size_t)(o - out);
  }
  if (!(a >= 1e-17 && a < 1e22)) return (size_t)snprintf(out, 32, "%g", v);  // (also NaN, infinities)
  int ex;
  (void)std::frexp(a, &ex);                       // a in [2^(ex-1), 2^ex)
  int k = (int)std::floor((ex - 1) * 0.30102999566398120);  // floor(log10 a) or one less
  long long digits = 0;
  for (int attempt = 0; attempt < 3; ++attempt) {
    const int p = 5 - k;
    if (p > 22 || p < -22) return (size_t)snprintf(out, 32, "%g", v);
    const double x = p >= 0 ? a * p10[p] : a / p10[-p];
    if (x < 1e5) {
      --k;
      continue;
    }
    if (x >= 1e6) {
      ++k;
      continue;
    }
    const double r = std::floor(x), f = x - r;
    if (std::fabs(f - 0.5) < 1e-7) return (size_t)snprintf(out, 32, "%g", v);  // a tie, or too close to call in double
    digits = (long long)r + (f > 0.5 ? 1 : 0);
    if (digits == 1000000) {
      digits = 100000;
      ++k;
    }
    break;
  }
  if (digits == 0) return (size_t)snprintf(out, 32, "%g", v);
  char d[6];
  for (int i = 5; i >= 0; --i) {
    d[i] = (char)('0' + digits % 10);
    digits /= 10;
  }
  int nd = 6;
  while (nd > 1 && d[nd - 1] == '0') --nd;  // trailing zeros go
  if (k < -4 || k >= 6) {
    *o++ = d[0];
    if (nd > 1) {
      *o++ = '.';
      for (int i = 1; i < nd; ++i) *o++ = d[i];
    }
    *o++ = 'e';
    int e = k;
    if (e < 0) {
      *o++ = '-';
      e = -e;
    } else {
      *o++ = '+';
    }
    if (e >= 100) *o++ = (char)('0' + e / 100);
    *o++ = (char)('0' + (e / 10) % 10);
    *o++ = (char)('0' + e % 10);
  } else if (k >= 0) {
    for (int i = 0; i <= k; ++i) *o++ = i < nd ? d[i] : '0';
    if (nd > k + 1) {
      *o++ = '.';
      for (int i = k + 1; i < nd; ++i) *o++ = d[i];
    }
  } else {
    *o++ = '0';
    *o++ = '.';
    for (int i = 0; i < -k - 1; ++i) *o++ = '0';
    for (int i = 0; i < nd; ++i) *o++ = d[i];
  }
  return (size_t)(o - out);
}
std::string formatG(double v) {
  char buf[40];
  return std::string(buf, formatG(v, buf));
}

static std::string lower(std::string s) {
  for (auto& c : s) c = (char)std::tolower((unsigned char)c);
  return s;
}

// ---- ModelParser (src/ModelParser.cpp:10-42, 106-151) --------------------------------------------------------
int ModelParser::parse(const std::string& s) {
  std::string arg = lower(s);
  param.clear();
  size_t l = arg.find('[');
  if (l == std::string::npos) {
    name = arg;
    return 0;
  }
  name = arg.substr(0, l);
  if (arg[arg.size() - 1] != ']') return -1;  // "Please use this format: model(model_param1=v1)"
  std::string all = arg.substr(l + 1, arg.size() - 1 - 1 - l);
  size_t pos = 0;
  while (pos <= all.size()) {
    size_t e = all.find_first_of(":,", pos);
    if (e == std::string::npos) e = all.size();
    std::string tok = all.substr(pos, e - pos);
    {  // (empty pieces are kept, as stringTokenize keeps them, base/Utils.h:194-218: "cmc[]" has ONE parameter, the tag "")
      size_t q = tok.find('=');
      if (q == std::string::npos)
        param[tok] = "";
      else
        param[tok.substr(0, q)] = tok.substr(q + 1);
    }
    pos = e + 1;
  }
  return 0;
}
void ModelParser::set(const std::string& tag, const std::string& value) { param[lower(tag)] = value; }
bool ModelParser::hasTag(const std::string& tag) const { return param.find(lower(tag)) != param.end(); }
const char* ModelParser::value(const std::string& tag) const {
  auto it = param.find(lower(tag));
  return it == param.end() ? nullptr : it->second.c_str();
}
const ModelParser& ModelParser::assign(const std::string& tag, double* v, double def) const {
  *v = hasTag(tag) ? atof(value(tag)) : def;
  return *this;
}
const ModelParser& ModelParser::assign(const std::string& tag, int* v, int def) const {
  // (through a double, as src/ModelParser.cpp:130-146: "nPerm=1e4" is 10 000)
  *v = hasTag(tag) ? (int)atof(value(tag)) : def;
  return *this;
}
const ModelParser& ModelParser::assign(const std::string& tag, bool* v, bool def) const {
  *v = hasTag(tag) ? true : def;
  return *this;
}

// ---- GpuBroker ----------------------------------------------------------------------------------------------------
GpuBroker& GpuBroker::instance() {
  static GpuBroker b;
  static bool init = false;
  if (!init) {
    init = true;
    if (const char* e = getenv("RVT_ADAPTER_BATCH")) b.setBatchWindow(atoi(e));
    if (const char* e = getenv("RVT_ADAPTER_BATCH_GB")) b.setBatchBytes((size_t)std::max(1, atoi(e)) << 30);
    if (const char* e = getenv("RVT_DOSAGE")) {  // "1": dosages; "d3": dosages printed with three decimals
      if (e[0] == 'd')
        b.setDosage(atoi(e + 1));
      else if (atoi(e) != 0)
        b.setDosage(-1);
    }
  }
  return b;
}

int GpuBroker::ensureContext(int device) {
  if (ctx) return 0;
  std::vector<int> ids;
  if (const char* e = getenv("RVT_DEVICES")) {  // e.g. RVT_DEVICES=0,1,2,3
    for (const char* p = e; *p;) {
      char* end = nullptr;
      const long v = strtol(p, &end, 10);
      if (end == p) break;
      ids.push_back((int)v);
      p = (*end == ',') ? end + 1 : end;
    }
  }
  if (ids.empty()) ids.push_back(device);
  const int rc = rvt_group_init(&grp, (int)ids.size(), ids.data());
  if (rc) return rc;
  ctx = rvt_group_member(grp, 0);
  if (dosage) {
    int den = 0;
    if (dosageDecimals >= 0 && dosageDecimals <= 3) {
      den = 1;
      for (int k = 0; k < dosageDecimals; ++k) den *= 10;
    }
    return rvt_group_set_content(grp, 0, den);
  }
  return 0;
}

void GpuBroker::registerTests(uint32_t mask, const rvt_params& p) {
  tests |= mask;
  if (mask & RVT_TEST_SKAT) {
    params.skat_beta1 = p.skat_beta1;
    params.skat_beta2 = p.skat_beta2;
    params.skat_nperm = p.skat_nperm;
    params.skat_alpha = p.skat_alpha;
  }
  if (mask & RVT_TEST_SKATO) {
    params.skato_beta1 = p.skato_beta1;
    params.skato_beta2 = p.skato_beta2;
  }
}

void GpuBroker::shutdown() {
  if (grp) rvt_group_destroy(grp);
  grp = nullptr;
  ctx = nullptr;
  rows.clear();
  pendingSerial.clear();
  pendingBytes = 0;
  recBytes.clear();
  ready.clear();
  failedSerial.clear();
  haveNull = false;
  haveFamNull = false;
  famSerial = -1;
  famTests = 0;
  kinU = nullptr;
  curSerial = -1;
  tests = 0;
}

// copyCovariateAndIntercept (src/ModelUtil.h:102-130) + the null fit SkatTest::fit caches (src/Model.h:2672-2699)
int GpuBroker::installNull(const GeneData& gd, bool binary, std::string* err) {
  const int d = 1 + gd.ncov;
  std::vector<double> X((size_t)gd.N * d);
  for (int64_t i = 0; i < gd.N; ++i) X[i] = 1.0;
  if (gd.ncov) std::memcpy(X.data() + gd.N, gd.covariate, sizeof(double) * (size_t)gd.N * gd.ncov);
  const int trait = binary ? RVT_TRAIT_BINARY : RVT_TRAIT_QUANTITATIVE;
  int rc;
  if (!fitter) {
    // default: LinearRegression::FitLinearModel / LogisticRegression::FitLogisticModel(cov, y, 100) on the device
    rc = rvt_group_fit_null(grp, trait, gd.N, d, X.data(), gd.phenotype, nullptr, nullptr);
  } else {
    // a caller-supplied fitter (e.g. the reference's own regression classes inside the rvtests tree)
    std::vector<double> res(gd.N), v(gd.N);
    double sigma2 = 1.0;
    if (fitter(binary, gd.N, d, X.data(), gd.phenotype, res.data(), v.data(), &sigma2)) {
      *err = binary ? "failed in fitting null model (logistic model)." : "failed in fitting null model (linear model).";
      return -1;
    }
    rc = rvt_group_set_null(grp, trait, gd.N, d, X.data(), res.data(), v.data(), sigma2);
  }
  if (rc) {
    *err = binary ? "failed in fitting null model (logistic model)." : "failed in fitting null model (linear model).";
    *err += std::string(" [") + rvt_group_last_error(grp) + "]";
    return -1;
  }
  haveNull = true;
  return 0;
}

rvt_ctx* GpuBroker::contextWithNull(const GeneData& gd, bool binary, std::string* err) {
  if (ensureContext(0)) {
    *err = "no MI355X device: the GPU models have no CPU fallback";
    return nullptr;
  }
  if (!haveNull || gd.phenotypeUpdated || gd.covariateUpdated)
    if (installNull(gd, binary, err)) return nullptr;
  return ctx;
}

rvt_ctx* GpuBroker::contextWithFamNull(const GeneData& gd, std::string* err) {
  if (ensureContext(0)) {
    *err = "no MI355X device: the GPU models have no CPU fallback";
    return nullptr;
  }
  if (kinU != gd.kinshipU) {  // a new decomposition (autosomes vs X region in the reference): install it once
    if (rvt_set_kinship(ctx, gd.N, gd.kinshipU, gd.kinshipS)) {
      *err = rvt_last_error(ctx);
      return nullptr;
    }
    kinU = gd.kinshipU;
    haveFamNull = false;
  }
  if (!haveFamNull || gd.phenotypeUpdated || gd.covariateUpdated) {
    const int d = 1 + gd.ncov;  // copyCovariateAndIntercept (src/ModelUtil.h:102-130)
    std::vector<double> X((size_t)gd.N * d);
    for (int64_t i = 0; i < gd.N; ++i) X[i] = 1.0;
    if (gd.ncov) std::memcpy(X.data() + gd.N, gd.covariate, sizeof(double) * (size_t)gd.N * gd.ncov);
    if (rvt_fit_fam_null(ctx, gd.N, d, X.data(), gd.phenotype, &famNull)) {
      *err = "SKAT test (for related individuals) failed in fitting null model (SKAT)";
      return nullptr;
    }
    haveFamNull = true;
  }
  return ctx;
}

const rvt_gene_result* GpuBroker::famResultFor(const GeneData& gd, std::string* err) {
  if (gd.serial == famSerial) return famOk ? &famRec : nullptr;
  famSerial = gd.serial;
  famOk = false;
  rvt_ctx* cx = contextWithFamNull(gd, err);
  if (!cx) return nullptr;
  double* block = nullptr;
  if (rvt_block_alloc(cx, gd.M, &block) || rvt_block_upload(cx, block, gd.M, gd.genotype)) {
    *err = rvt_last_error(cx);
    if (block) rvt_block_free(cx, block);
    return nullptr;
  }
  const double* p = block;
  const int rc = rvt_run_fam_tests(cx, 1, &p, &gd.M, &gd.serial, famTests, &famRec);
  rvt_block_free(cx, block);
  if (rc) {
    *err = rvt_last_error(cx);
    return nullptr;
  }
  famOk = true;
  return &famRec;
}

int GpuBroker::submit(const GeneData& gd, bool binary, std::string* err) {
  if (gd.serial == curSerial) return curOk ? 0 : -1;  // another model of the same gene already submitted it
  // genes in flight are bounded by count and by the bytes of their device blocks (a 1024-variant gene is 4 GB at
  // N = 500 000)
  const size_t geneBytes = sizeof(double) * (size_t)gd.N * (size_t)gd.M;
  // a full window first takes what has finished (no waiting: the device keeps its batches in flight); only when
  // nothing comes back and twice the window is pending does the caller wait
  if ((int)pendingSerial.size() >= window || (!pendingSerial.empty() && pendingBytes + geneBytes > windowBytes)) {
    drainReady();
    if ((int)pendingSerial.size() >= 2 * window || (!pendingSerial.empty() && pendingBytes + geneBytes > 2 * windowBytes))
      flush();
  }
  curSerial = gd.serial;
  curOk = false;
  auto failed = [&]() {
    failedSerial.push_back(gd.serial);
    return -1;
  };
  if (ensureContext(0)) {
    *err = "no MI355X device: the GPU models have no CPU fallback";
    return failed();
  }
  if (!haveNull || gd.phenotypeUpdated || gd.covariateUpdated) {
    flush();  // pending genes belong to the previous null model
    if (installNull(gd, binary, err)) return failed();
  }
  if (gd.bed) {  // 2-bit rows before consolidation: frequencies and imputation happen on the device
    if (rvt_group_submit_gene_bed(grp, gd.serial, gd.M, gd.bed, tests, &params, nullptr)) {
      *err = rvt_group_last_error(grp);
      return failed();
    }
  } else {
    if ((int)gd.markerFrequency.size() < gd.M) {
      *err = "marker frequencies missing";
      return failed();
    }
    if (rvt_group_submit_gene(grp, gd.serial, gd.M, gd.genotype, gd.markerFrequency.data(), tests, &params)) {
      *err = rvt_group_last_error(grp);
      return failed();
    }
  }
  pendingSerial.push_back(gd.serial);
  pendingBytes += geneBytes;
  recBytes[gd.serial] = geneBytes;
  curOk = true;
  return 0;
}

void GpuBroker::enqueue(ModelFitter* m, TextSink* fp, const std::string& siteTab, int64_t serial) {
  rows.push_back(Row{m, fp, siteTab, serial});
}

// Write every row whose gene has its record (or failed at submission), in enqueue order, and stop at the first row that
// has to wait — rows reach each file in the order the reference writes them.
void GpuBroker::writeReadyRows(bool all) {
  size_t k = 0;
  for (; k < rows.size(); ++k) {
    const Row& r = rows[k];
    auto it = ready.find(r.serial);
    const bool failed = std::find(failedSerial.begin(), failedSerial.end(), r.serial) != failedSerial.end();
    if (it == ready.end() && !failed && !all) break;
    r.fp->write(r.siteTab + r.model->formatRow(it == ready.end() ? nullptr : &it->second));
  }
  rows.erase(rows.begin(), rows.begin() + k);
  // records nobody waits for any more: every remaining row belongs to a later gene (serials only grow)
  const int64_t keep = rows.empty() ? (pendingSerial.empty() ? INT64_MAX : pendingSerial.front()) : rows.front().serial;
  for (auto it = ready.begin(); it != ready.end();) it = (it->first < keep) ? ready.erase(it) : std::next(it);
  failedSerial.erase(std::remove_if(failedSerial.begin(), failedSerial.end(), [&](int64_t s) { return s < keep; }),
                     failedSerial.end());
}

// take what the device has finished, without waiting (rvt_collect_ready): the pipeline keeps running
int GpuBroker::drainReady() {
  if (!grp || pendingSerial.empty()) return 0;
  std::vector<rvt_gene_result> recs(pendingSerial.size());
  int n = 0;
  const int rc = rvt_group_collect_ready(grp, recs.data(), (int)recs.size(), &n);
  if (rc) return rc;
  for (int i = 0; i < n; ++i) {
    ready[recs[i].gene_id] = recs[i];
    pendingBytes -= std::min(pendingBytes, recBytes[recs[i].gene_id]);
    recBytes.erase(recs[i].gene_id);
  }
  pendingSerial.erase(pendingSerial.begin(), pendingSerial.begin() + n);
  writeReadyRows(false);
  return 0;
}

int GpuBroker::flush() {
  int rc = 0;
  if (ctx && !pendingSerial.empty()) {
    std::vector<rvt_gene_result> recs(pendingSerial.size());
    int n = 0;
    rc = rvt_group_collect(grp, recs.data(), (int)recs.size(), &n);
    if (!rc)
      for (int i = 0; i < n; ++i) ready[recs[i].gene_id] = recs[i];
  }
  pendingSerial.clear();
  recBytes.clear();
  pendingBytes = 0;
  writeReadyRows(true);
  ready.clear();
  failedSerial.clear();
  return rc;
}

int ModelFitter::deferredFit(GeneData* dc) {
  curSerial = dc->serial;
  return GpuBroker::instance().submit(*dc, isBinaryOutcome(), &lastError);
}
void ModelFitter::deferredOutput(TextSink* fp, const SiteInfo& siteInfo) {
  GpuBroker::instance().enqueue(this, fp, siteInfo.valueTab(), curSerial);
}

// ---- SkatTest ----------------------------------------------------------------------------------------------------------
SkatTest::SkatTest(int nPerm, double alpha, double beta1, double beta2) : usePermutation(nPerm > 0) {
  modelName = "Skat";
  rvt_params p{beta1, beta2, 1.0, 25.0, nPerm, alpha};
  GpuBroker::instance().registerTests(RVT_TEST_SKAT, p);
}
int SkatTest::fit(GeneData* dc) { return deferredFit(dc); }
void SkatTest::writeHeader(TextSink* fp, const SiteInfo& siteInfo) {
  fp->write(siteInfo.headerTab());
  if (!usePermutation)
    fp->write("Q\tPvalue\n");
  else  // Permutation::writeHeader (src/Permutation.h:51-56,99-104)
    fp->write("Q\tPvalue\tNumPerm\tActualPerm\tStat\tNumGreater\tNumEqual\tPermPvalue\n");
}
void SkatTest::writeOutput(TextSink* fp, const SiteInfo& siteInfo) { deferredOutput(fp, siteInfo); }
void SkatTest::writeFootnote(TextSink*) { GpuBroker::instance().flush(); }
std::string SkatTest::formatRow(const rvt_gene_result* res) const {
  // fitOK: genotype.cols == 0 after filtering -> NA row (src/Model.h:2665-2668)
  if (!res || !res->skat_ok || (usePermutation && !res->perm_ok))
    return usePermutation ? "NA\tNA\tNA\tNA\tNA\tNA\tNA\tNA\n" : "NA\tNA\n";
  std::string line = formatG(res->skat_Q) + "\t" + formatG(res->skat_p);
  if (usePermutation)  // Permutation::updateValue: ints via toString, doubles via floatToString (src/Result.h:52-63)
    line += "\t" + std::to_string(res->perm_num_perm) + "\t" + std::to_string(res->perm_actual_perm) + "\t" +
            floatToString(res->skat_Q) + "\t" + std::to_string(res->perm_num_greater) + "\t" +
            std::to_string(res->perm_num_equal) + "\t" + floatToString(res->perm_pvalue);
  return line + "\n";
}

// ---- SkatOTest -----------------------------------------------------------------------------------------------------------
SkatOTest::SkatOTest(double beta1, double beta2) {
  modelName = "SkatO";
  rvt_params p{1.0, 25.0, beta1, beta2, 0, 0.05};
  GpuBroker::instance().registerTests(RVT_TEST_SKATO, p);
}
int SkatOTest::fit(GeneData* dc) { return deferredFit(dc); }
void SkatOTest::writeHeader(TextSink* fp, const SiteInfo& siteInfo) {
  fp->write(siteInfo.headerTab());
  fp->write("Q\trho\tPvalue\n");
}
void SkatOTest::writeOutput(TextSink* fp, const SiteInfo& siteInfo) { deferredOutput(fp, siteInfo); }
void SkatOTest::writeFootnote(TextSink*) { GpuBroker::instance().flush(); }
std::string SkatOTest::formatRow(const rvt_gene_result* res) const {
  // fitOK = (skato.Fit(...) == 0) (src/Model.h:2853-2859)
  if (!res || res->n_poly == 0 || !res->skato_ok) return "NA\tNA\tNA\n";
  return formatG(res->skato_Q) + "\t" + formatG(res->skato_rho) + "\t" + formatG(res->skato_p) + "\n";
}

// ---- CMCTest / ZegginiTest -------------------------------------------------------------------------------------------------
CMCTest::CMCTest() {
  modelName = "CMC";
  GpuBroker::instance().registerTests(RVT_TEST_CMC, rvt_params{1.0, 25.0, 1.0, 25.0, 0, 0.05});
}
int CMCTest::fit(GeneData* dc) { return deferredFit(dc); }
void CMCTest::writeHeader(TextSink* fp, const SiteInfo& siteInfo) {
  fp->write(siteInfo.headerTab());
  fp->write("NonRefSite\tPvalue\n");
}
void CMCTest::writeOutput(TextSink* fp, const SiteInfo& siteInfo) { deferredOutput(fp, siteInfo); }
void CMCTest::writeFootnote(TextSink*) { GpuBroker::instance().flush(); }
std::string CMCTest::formatRow(const rvt_gene_result* res) const {
  if (!res || !res->cmc_ok) return "NA\tNA\n";
  return std::to_string(res->cmc_nonref) + "\t" + floatToString(res->cmc_p) + "\n";
}

ZegginiTest::ZegginiTest() {
  modelName = "Zeggini";
  GpuBroker::instance().registerTests(RVT_TEST_ZEGGINI, rvt_params{1.0, 25.0, 1.0, 25.0, 0, 0.05});
}
int ZegginiTest::fit(GeneData* dc) { return deferredFit(dc); }
void ZegginiTest::writeHeader(TextSink* fp, const SiteInfo& siteInfo) {
  fp->write(siteInfo.headerTab());
  fp->write("Pvalue\n");
}
void ZegginiTest::writeOutput(TextSink* fp, const SiteInfo& siteInfo) { deferredOutput(fp, siteInfo); }
void ZegginiTest::writeFootnote(TextSink*) { GpuBroker::instance().flush(); }
std::string ZegginiTest::formatRow(const rvt_gene_result* res) const {
  return (res && res->zeg_ok) ? floatToString(res->zeg_p) + "\n" : std::string("NA\n");
}

// ---- AnalyticVTTest ---------------------------------------------------------------------------------------------------------
AnalyticVTTest::AnalyticVTTest(bool related_) : related(related_) {
  modelName = related ? "FamAnalyticVT" : "AnalyticVT";
  if (!related) GpuBroker::instance().registerTests(RVT_TEST_ANALYTICVT, rvt_params{1.0, 25.0, 1.0, 25.0, 0, 0.05});
}
int AnalyticVTTest::fit(GeneData* dc) {
  if (!related) return deferredFit(dc);
  fitOK = false;
  if (isBinaryOutcome()) {  // src/Model.h:2143-2149
    lastError = "Analytic VT test does not support binary outcomes. Results will be all NAs.";
    return -1;
  }
  if (!dc->kinshipU || !dc->kinshipS) {  // model and data do not match (src/Model.h:2156-2160)
    lastError = "Analytic VT test has internal error!";
    return -1;
  }
  rvt_ctx* ctx = GpuBroker::instance().contextWithFamNull(*dc, &lastError);
  if (!ctx) return -1;
  double* block = nullptr;
  if (rvt_block_alloc(ctx, dc->M, &block) || rvt_block_upload(ctx, block, dc->M, dc->genotype)) {
    lastError = rvt_last_error(ctx);
    if (block) rvt_block_free(ctx, block);
    return -1;
  }
  const double* bp = block;
  const int M = dc->M;
  const int rc = rvt_fam_analytic_vt(ctx, 1, &bp, &M, &rec);
  rvt_block_free(ctx, block);
  if (rc) {
    lastError = rvt_last_error(ctx);
    return -1;
  }
  fitOK = rec.vt_ok != 0;
  return fitOK ? 0 : -1;
}
void AnalyticVTTest::writeHeader(TextSink* fp, const SiteInfo& siteInfo) {
  fp->write(siteInfo.headerTab());
  fp->write("MinMAF\tMaxMAF\tOptimMAF\tOptimNumVar\tU\tV\tStat\tPvalue\n");  // result.addHeader order, src/Model.h:2123-2130
}
void AnalyticVTTest::writeOutput(TextSink* fp, const SiteInfo& siteInfo) {
  if (!related) {
    deferredOutput(fp, siteInfo);
    return;
  }
  fp->write(siteInfo.valueTab());
  fp->write(formatRow(fitOK ? &rec : nullptr));
}
void AnalyticVTTest::writeFootnote(TextSink*) {
  if (!related) GpuBroker::instance().flush();
}
std::string AnalyticVTTest::formatRow(const rvt_gene_result* r) const {
  // not fitted (binary trait, no polymorphic site, no usable threshold, integral not converged to 1e-3): the Result keeps
  // its cleared values (src/Model.h:2233-2246 after ModelFitter::reset)
  if (!r || !r->vt_ok || r->vt_p_error > 1e-3) return "NA\tNA\tNA\tNA\tNA\tNA\tNA\tNA\n";
  return floatToString(r->vt_minmaf) + "\t" + floatToString(r->vt_maxmaf) + "\t" + floatToString(r->vt_optmaf) + "\t" +
         std::to_string(r->vt_optnum) + "\t" + floatToString(r->vt_U) + "\t" + floatToString(r->vt_V) + "\t" +
         floatToString(r->vt_stat) + "\t" + floatToString(r->vt_p) + "\n";
}

// ---- KbacTest ---------------------------------------------------------------------------------------------------------------
KbacTest::KbacTest(int nPerm_, double alpha_) : nPerm(nPerm_), alpha(alpha_) { modelName = "Kbac"; }
int KbacTest::fit(GeneData* dc) {
  fitOK = false;
  if (!isBinaryOutcome()) {  // src/Model.h:2930-2936
    lastError = "KBAC test does not support continuous outcomes. Results will be all NAs.";
    return -1;
  }
  if (dc->ncov != 0) {  // src/Model.h:2937-2942
    lastError = "KBAC test does not support covariates. Results will be all NAs.";
    return -1;
  }
  rvt_ctx* ctx = GpuBroker::instance().contextWithNull(*dc, true, &lastError);  // (the null model defines N on the device)
  if (!ctx) return -1;
  double* block = nullptr;
  if (rvt_block_alloc(ctx, dc->M, &block) || rvt_block_upload(ctx, block, dc->M, dc->genotype)) {
    lastError = rvt_last_error(ctx);
    if (block) rvt_block_free(ctx, block);
    return -1;
  }
  const double* bp = block;
  const int M = dc->M;
  const int rc = rvt_kbac_blocks(ctx, 1, &bp, &M, dc->markerFrequency.data(), dc->phenotype, nPerm, alpha, &rec);
  rvt_block_free(ctx, block);
  if (rc) {
    lastError = rvt_last_error(ctx);
    return -1;
  }
  fitOK = rec.fit_ok != 0;
  return fitOK ? 0 : -1;
}
void KbacTest::writeHeader(TextSink* fp, const SiteInfo& siteInfo) {
  fp->write(siteInfo.headerTab());
  fp->write("Pvalue\n");
}
void KbacTest::writeOutput(TextSink* fp, const SiteInfo& siteInfo) {
  fp->write(siteInfo.valueTab());
  if (!fitOK) {
    fp->write("NA\n");
  } else {
    char buf[64];
    snprintf(buf, sizeof(buf), "%f\n", rec.pvalue);  // fp->printf("%f\n", this->pValue), src/Model.h:3005
    fp->write(buf);
  }
}

// ---- FamSkatTest ----------------------------------------------------------------------------------------------------------
FamSkatTest::FamSkatTest(double, double) {
  modelName = "FamSkat";
  GpuBroker::instance().registerFamTests(RVT_TEST_FAMSKAT);
}
int FamSkatTest::fit(GeneData* dc) {
  fitOK = false;
  if (isBinaryOutcome()) {  // src/Model.h:3071-3078
    lastError = "SKAT test (for related individuals) does not support binary outcomes. Results will be all NAs.";
    return -1;
  }
  if (!dc->kinshipU || !dc->kinshipS) {  // src/Model.h:3079-3087
    lastError = "SKAT test (for related individuals) cannot find kinship. Results will be all NAs.";
    return -1;
  }
  const rvt_gene_result* r = GpuBroker::instance().famResultFor(*dc, &lastError);
  if (!r) return -1;
  rec = *r;
  if (!rec.famskat_ok) return -1;  // genotype.cols == 0 after filtering -> NA row (src/Model.h:3066-3069)
  fitOK = true;
  return 0;
}
void FamSkatTest::writeHeader(TextSink* fp, const SiteInfo& siteInfo) {
  fp->write(siteInfo.headerTab());
  fp->write("Q\tPvalue\n");
}
void FamSkatTest::writeOutput(TextSink* fp, const SiteInfo& siteInfo) {
  fp->write(siteInfo.valueTab());
  if (!fitOK)
    fp->write("NA\tNA\n");
  else
    fp->write(formatG(rec.famskat_Q) + "\t" + formatG(rec.famskat_p) + "\n");
}

// ---- FamCMC / FamZeggini ---------------------------------------------------------------------------------------------------
FamBurdenTest::FamBurdenTest(bool zeg) : zeggini(zeg) {
  modelName = zeg ? "FamZeggini" : "FamCMC";
  GpuBroker::instance().registerFamTests(zeg ? RVT_TEST_FAMZEGGINI : RVT_TEST_FAMCMC);
}
int FamBurdenTest::fit(GeneData* dc) {
  fitOK = false;
  if (isBinaryOutcome()) {  // src/Model.h:2284-2291
    lastError = "burden test (for related individuals) does not support binary outcomes. Results will be all NAs.";
    return -1;
  }
  if (!dc->kinshipU || !dc->kinshipS) {
    lastError = "burden test (for related individuals) cannot find kinship.";
    return -1;
  }
  const rvt_gene_result* r = GpuBroker::instance().famResultFor(*dc, &lastError);
  if (!r) return -1;
  rec = *r;
  if (!(zeggini ? rec.famzeg_ok : rec.famcmc_ok)) return -1;  // genotype.cols == 0 (src/Model.h:2297-2301)
  const double u = zeggini ? rec.famzeg_U : rec.famcmc_U, v = zeggini ? rec.famzeg_V : rec.famcmc_V;
  if (v != 0) effect = u / v;
  fitOK = true;
  return 0;
}
void FamBurdenTest::writeHeader(TextSink* fp, const SiteInfo& siteInfo) {
  fp->write(siteInfo.headerTab());
  fp->write(zeggini ? "NumSite\tMeanBurden\tU\tV\tEffect\tPvalue\n" : "NumSite\tAF\tU\tV\tEffect\tPvalue\n");
}
void FamBurdenTest::writeOutput(TextSink* fp, const SiteInfo& siteInfo) {
  fp->write(siteInfo.valueTab());
  if (!fitOK) {  // Result keeps "NA" for values that were not updated (src/Model.h:2344-2357)
    fp->write("NA\tNA\tNA\tNA\tNA\tNA\n");
    return;
  }
  const double af = zeggini ? rec.famzeg_af : rec.famcmc_af, u = zeggini ? rec.famzeg_U : rec.famcmc_U,
               v = zeggini ? rec.famzeg_V : rec.famcmc_V, p = zeggini ? rec.famzeg_p : rec.famcmc_p;
  fp->write(std::to_string(rec.n_poly) + "\t" + floatToString(af) + "\t" + floatToString(u) + "\t" +
            floatToString(v) + "\t" + floatToString(effect) + "\t" + floatToString(p) + "\n");
}

// ---- MetaScoreTest (src/Model.h:3155-3398), unrelated samples ------------------------------------------------------------
MetaScoreTest::MetaScoreTest() {
  modelName = "MetaScore";
  if (const char* e = getenv("RVT_METASCORE_BLOCK")) capacity = std::max(1, std::min(65536, atoi(e)));
}
MetaScoreTest::~MetaScoreTest() {
  if (fout) flush();
  if (ctx && block) rvt_block_free(ctx, block);
}
int MetaScoreTest::setParameter(const ModelParser& parser) {
  outputSE = parser.hasTag("se");  // src/Model.h:3177-3182 ("gwama" / "bolt" are not provided)
  return 0;
}
int MetaScoreTest::fit(GeneData* dc) {
  useFamilyModel = dc->kinshipU != nullptr;  // dc->hasKinship(): MetaFamQtl / MetaFamBinary (src/Model.h:3398-3668)
  if ((int)rows.size() >= capacity && used > 0 && flush()) return -1;
  if (used >= capacity) {  // flush() could not run: no writeOutput() has named the output sink yet
    lastError = "MetaScore: the device block is full and no output was requested for its sites";
    return -1;
  }
  rows.emplace_back();
  Row& row = rows.back();
  row.all = dc->counter;  // site statistics are printed whether or not the test runs (src/Model.h:3211-3230)
  if (isBinaryOutcome()) {
    row.cases = dc->caseCounter;
    row.ctrls = dc->ctrlCounter;
  }
  if (dc->N == 0) return -1;
  if (nSample >= 0 && nSample != dc->N) {
    lastError = "Sample size changed";
    return -1;
  }
  if ((nSample >= 0) && (dc->phenotypeUpdated || dc->covariateUpdated) && used > 0) {
    // rows tested so far belong to the previous null model: finish them before it is replaced
    Row keep = row;
    rows.pop_back();
    if (flush()) return -1;
    rows.push_back(keep);
  }
  ctx = useFamilyModel ? GpuBroker::instance().contextWithFamNull(*dc, &lastError)
                       : GpuBroker::instance().contextWithNull(*dc, isBinaryOutcome(), &lastError);
  if (!ctx) return -1;
  if (useFamilyModel && isBinaryOutcome() && (nSample < 0 || dc->phenotypeUpdated)) {
    // MetaFamBinary::FitNullModel (src/Model.h:3566-3581): alpha = log(nCase / nCtrl), b = calculateB()
    int64_t nCase = 0, nCtrl = 0;
    for (int64_t i = 0; i < dc->N; ++i) {
      if (dc->phenotype[i] == 1) ++nCase;
      else if (dc->phenotype[i] == 0) ++nCtrl;
    }
    if (rvt_fam_binary_scale(ctx, nCase, nCtrl, nullptr, &famB)) {
      lastError = rvt_last_error(ctx);
      return -1;
    }
  }
  if (nSample < 0) {
    nSample = dc->N;
    nCovariate = dc->ncov + 1;
    if (rvt_block_alloc(ctx, capacity, &block)) {
      lastError = rvt_last_error(ctx);
      return -1;
    }
  }
  if (dc->M != 1) return -1;  // "sanity check, this should not happen" (src/Model.h:3241-3244)
  // the caller overwrites the genotype buffer for the next site: copy the column into the device block now; whether
  // the site is monomorphic (src/Model.h:3246-3250) is decided on the device when the block is processed
  if (rvt_block_upload_columns(ctx, block, used, 1, dc->genotype)) {
    lastError = rvt_last_error(ctx);
    return -1;
  }
  rows.back().column = used++;
  rows.back().tested = true;
  return 0;
}
void MetaScoreTest::writeHeader(TextSink*, const SiteInfo&) {
  // the header is deferred until the null model is known and is printed with the site columns writeOutput is given
  // (src/Model.h:3262-3264, 3283-3303)
}
void MetaScoreTest::writeOutput(TextSink* fp, const SiteInfo& siteInfo) {
  fout = fp;
  if (siteHeaderTab.empty()) siteHeaderTab = siteInfo.headerTab();
  if (rows.empty() || rows.back().written) rows.emplace_back();  // writeOutput without a fit(): counters unknown
  rows.back().siteTab = siteInfo.valueTab();
  rows.back().written = true;
}
void MetaScoreTest::writeFootnote(TextSink* fp) {
  fout = fp;
  flush();
}

namespace {
std::string triple(const char* fmt, double a, double b, double c) {
  char buf[128];
  snprintf(buf, sizeof(buf), fmt, a, b, c);
  return buf;
}
std::string tripleInt(int a, int b, int c) {
  char buf[128];
  snprintf(buf, sizeof(buf), "%d:%d:%d", a, b, c);
  return buf;
}
}  // namespace

int MetaScoreTest::flush() {
  if (!fout) return 0;
  std::vector<int> ok(std::max(used, 1));
  std::vector<double> u(ok.size()), v(ok.size()), eff(ok.size()), se(ok.size()), pv(ok.size());
  bool scored = false;
  std::vector<double> famAf(ok.size());
  if (used > 0) {
    const int rc = useFamilyModel
                       ? rvt_score_block_fam(ctx, block, used, isBinaryOutcome() ? 1 : 0, ok.data(), u.data(), v.data(),
                                             famAf.data(), pv.data())
                       : rvt_score_block(ctx, block, used, ok.data(), u.data(), v.data(), eff.data(), se.data(),
                                         pv.data());
    if (rc) {
      lastError = rvt_last_error(ctx);
    } else {
      scored = true;
      if (useFamilyModel) {
        // MetaFamQtl::GetEffect / FastLMM::GetSE (src/Model.h:3491-3496, FastLMM.cpp:452-455); MetaFamBinary: U b and
        // V b^2 are returned, effect = U / V / b = (U b) / (V b^2), SE = 1 / sqrt(V b^2) / b (:3647-3662)
        const double bdiv = isBinaryOutcome() ? famB : 1.0;
        for (int k = 0; k < used; ++k) {
          const bool nz = v[k] != 0.0 && bdiv != 0.0;
          eff[k] = nz ? u[k] / v[k] : 0.0;
          se[k] = nz ? 1.0 / std::sqrt(v[k]) / bdiv : 0.0;
        }
      }
    }
  }
  if (!headerOutputted && (scored || used == 0)) {
    // writeSummaryAndHeader (src/Model.h:3283-3297): g_SummaryHeader->outputHeader is the caller's; PrintNullModel:
    if (scored) {
      std::vector<double> beta(nCovariate), covb(nCovariate);
      double sigma2 = 0.0;
      int rcs;
      const rvt_fam_null& fnull = GpuBroker::instance().familyNull();
      if (useFamilyModel) {
        rcs = rvt_fam_null_summary(ctx, covb.data());
        for (int k = 0; k < nCovariate; ++k) beta[k] = fnull.beta[k];
      } else {
        rcs = rvt_null_summary(ctx, beta.data(), covb.data(), &sigma2);
      }
      if (rcs == RVT_OK) {
        fout->write("##NullModelEstimates\n");
        fout->write("## - Name\tBeta\tSD\n");
        fout->write("## - Intercept\t" + formatG(beta[0]) + "\t" + formatG(covb[0]) + "\n");
        for (size_t i = 0; i < covLabel.size(); ++i) {
          if ((int)i + 1 >= nCovariate) break;
          fout->write("## - " + covLabel[i] + "\t" + formatG(beta[i + 1]) + "\t" + formatG(covb[i + 1]) + "\n");
        }
        if (useFamilyModel) {  // GetSigmaG2 = sigma2, GetSigmaE2 = sigma2 * delta (FastLMM.cpp:456-457)
          fout->write("## - SigmaG2\t" + formatG(fnull.sigma2_g) + "\tNA\n");
          fout->write("## - SigmaE2\t" + formatG(fnull.sigma2_g * fnull.delta) + "\tNA\n");
        } else if (isBinaryOutcome()) {  // MetaUnrelatedBinary::PrintNullModel (src/Model.h:3751)
          fout->write("## - Sigma2\tNA\tNA\n");
        } else {
          fout->write("## - Sigma2\t" + formatG(sigma2) + "\tNA\n");
        }
      }
    }
    fout->write(siteHeaderTab);
    fout->write(std::string("AF\tINFORMATIVE_ALT_AC\tCALL_RATE\tHWE_PVALUE\tN_REF\tN_HET\tN_ALT\tU_STAT\tSQRT_V_STAT\t"
                            "ALT_EFFSIZE\t") +
                (outputSE ? "ALT_EFFSIZE_SE\t" : "") + "PVALUE\n");
    headerOutputted = true;
  }
  for (const Row& r : rows) {
    if (!r.written) continue;  // main calls writeOutput after every fit(); a row never written is never printed
    std::string line = r.siteTab;
    const SiteCounts &a = r.all, &ca = r.cases, &ct = r.ctrls;
    if (!isBinaryOutcome()) {  // src/Model.h:3307-3325
      // with kinship a tested site prints FastGetAF instead of the counter's frequency (src/Model.h:3255-3257)
      const double afv = (useFamilyModel && r.tested && scored && r.column >= 0 && ok[r.column]) ? famAf[r.column] : a.af;
      line += (afv >= 0.0 ? floatToString(afv) : std::string("NA")) + "\t";
      line += floatToString(a.ac) + "\t" + floatToString(a.callRate) + "\t" + floatToString(a.hwe) + "\t";
      line += std::to_string(a.nHomRef) + "\t" + std::to_string(a.nHet) + "\t" + std::to_string(a.nHomAlt) + "\t";
    } else {  // src/Model.h:3310-3351
      const double afb = (useFamilyModel && r.tested && scored && r.column >= 0 && ok[r.column]) ? famAf[r.column] : a.af;
      line += (afb >= 0.0 ? triple("%g:%g:%g", afb, ca.af, ct.af) : std::string("NA")) + "\t";
      line += triple("%g:%g:%g", a.ac, ca.ac, ct.ac) + "\t";
      line += triple("%g:%g:%g", a.callRate, ca.callRate, ct.callRate) + "\t";
      line += triple("%g:%g:%g", a.hwe, ca.hwe, ct.hwe) + "\t";
      line += tripleInt(a.nHomRef, ca.nHomRef, ct.nHomRef) + "\t" + tripleInt(a.nHet, ca.nHet, ct.nHet) + "\t" +
              tripleInt(a.nHomAlt, ca.nHomAlt, ct.nHomAlt) + "\t";
    }
    const int k = r.column;
    if (r.tested && scored && k >= 0 && ok[k]) {  // src/Model.h:3353-3364
      line += floatToString(u[k]) + "\t" + floatToString(std::sqrt(v[k])) + "\t" + floatToString(eff[k]) + "\t";
      if (outputSE) line += (v[k] > 0.0 ? floatToString(se[k]) : std::string("NA")) + "\t";
      line += floatToString(pv[k]) + "\n";
    } else {
      line += std::string("NA\tNA\tNA\t") + (outputSE ? "NA\t" : "") + "NA\n";
    }
    fout->write(line);
  }
  rows.clear();
  used = 0;
  return scored || lastError.empty() ? 0 : -1;
}

// ---- MetaCovTest ---------------------------------------------------------------------------------------------------------
MetaCovTest::MetaCovTest(int windowSize_) : windowSize(windowSize_) {
  modelName = "MetaCov";
  capacity = 4096;  // (columns; shrunk to the memory budget at the first fit(), when N is known)
  // columns of the device ring at the start; RVT_METACOV_BLOCK lowers it (tests exercise the mid-stream flush and the wrap with it)
  if (const char* e = getenv("RVT_METACOV_BLOCK")) capacity = std::max(2, std::min(65536, atoi(e)));
  if (const char* e = getenv("RVT_METACOV_MAX_COLUMNS")) maxColumns = std::max(capacity, atoi(e));
  formatThreads = (int)std::max(1u, std::min(8u, std::thread::hardware_concurrency() / 2));
  if (const char* e = getenv("RVT_METACOV_FORMAT_THREADS")) formatThreads = std::max(1, atoi(e));
}
MetaCovTest::~MetaCovTest() {
  if (fout) flush(true);
  if (ctx && bandReg) rvt_host_unregister(ctx, bandReg);
  if (ctx && block) rvt_block_free(ctx, block);
}
int MetaCovTest::setParameter(const ModelParser& parser) {
  outputGwama = parser.hasTag("gwama");
  return 0;
}
int MetaCovTest::fit(GeneData* dc) {
  fitOK = false;
  if (dc->M != 1 || dc->N == 0) return -1;  // src/Model.cpp:851-858
  if (!dc->site) {
    lastError = "MetaCov needs the site's CHROM/POS";
    return -1;
  }
  if (nSample >= 0 && nSample != dc->N) {
    lastError = "Sample size changed at [ " + dc->site->get("CHROM") + ":" + dc->site->get("POS") + " ]";
    return -1;
  }
  useFamilyModel = dc->kinshipU != nullptr;  // dc->hasKinship(): MetaCovFamQtl instead of MetaCovUnrelatedQtl
  ctx = useFamilyModel ? GpuBroker::instance().contextWithFamNull(*dc, &lastError)
                       : GpuBroker::instance().contextWithNull(*dc, isBinaryOutcome(), &lastError);
  if (!ctx) return -1;
  if (useFamilyModel && isBinaryOutcome() && (nSample < 0 || dc->phenotypeUpdated)) {
    // MetaCovFamBinary::FitNullModel (src/Model.cpp:598-640): alpha = log(nCase / nCtrl), b = obtainB(alpha)
    int64_t nCase = 0, nCtrl = 0;
    for (int64_t i = 0; i < dc->N; ++i) {
      if (dc->phenotype[i] == 1) ++nCase;
      else if (dc->phenotype[i] == 0) ++nCtrl;
    }
    if (rvt_fam_binary_scale(ctx, nCase, nCtrl, nullptr, nullptr)) {
      lastError = rvt_last_error(ctx);
      return -1;
    }
  }
  if (nSample < 0) {
    nSample = dc->N;
    nCovariate = dc->ncov + 1;
    // the ring may grow to 96 GB of columns (RVT_METACOV_RING_GB; the int8 copy and the two 4-bit stores the engine keeps per
    // column counted in: 10 bytes per genotype): 19 000 columns at N = 500 000
    {
      double gb = 96.0;
      if (const char* e = getenv("RVT_METACOV_RING_GB")) gb = std::max(1.0, atof(e));
      const double cols = gb * 1e9 / (10.0 * (double)std::max<int64_t>(dc->N, 1));
      if (cols < (double)maxColumns) maxColumns = std::max(std::min(capacity, 1024), (int)cols);
      if (capacity > maxColumns) capacity = maxColumns;
    }
    if (rvt_block_alloc(ctx, capacity, &block)) {
      lastError = rvt_last_error(ctx);
      return -1;
    }
  }
  if ((int)sites.size() == capacity) {
    const int before = (int)sites.size();
    if (flush(false)) return -1;
    // One window holds more sites than the ring: enlarge it.  ALSO when the flush could emit less than three quarters of the
    // ring — a window of 1 000 markers in a ring of 1 024 would read 1 000 columns to write 24 rows; with a ring of at least
    // four windows every flush emits three quarters of what it reads and is long enough to hide its fixed cost (measured at
    // N = 500 000, windows of 200 / 1 000 / 3 000 markers: rings of 4 096 / 4 096 / 16 384 columns run 2.0 / 1.3 / 1.1 times
    // as fast as rings of two windows).
    const bool full = (int)sites.size() == capacity;
    if (full || (canGrow && 4 * (before - (int)sites.size()) < 3 * before)) {
      if (grow()) {
        if (full) return -1;
        canGrow = false;     // (a ring that cannot grow any further just stays as efficient as it was: do not retry every fill)
        lastError.clear();
      }
    }
  }
  // the caller overwrites the genotype buffer for the next site: copy the column into the device ring now — into the physical
  // column behind the tail; nothing in the ring ever moves (RingMemoryPool::allocate, base/RingMemoryPool.cpp:31-47)
  if (rvt_block_upload_columns(ctx, block, (head + (int)sites.size()) % capacity, 1, dc->genotype)) {
    lastError = rvt_last_error(ctx);
    return -1;
  }
  sites.push_back(Site{dc->site->get("CHROM"), atoi(dc->site->get("POS").c_str())});
  fitOK = true;  // whether the site is monomorphic (and therefore skipped) is decided on the device at flush time
  return 0;
}
void MetaCovTest::writeHeader(TextSink* fp, const SiteInfo&) {
  fp->write("CHROM\tSTART_POS\tEND_POS\tNUM_MARKER\tMARKER_POS\tCOV\n");
}
void MetaCovTest::writeOutput(TextSink* fp, const SiteInfo&) { fout = fp; }
void MetaCovTest::writeFootnote(TextSink* fp) {
  fout = fp;
  flush(true);
}

// Double the device ring (a window can hold more sites than the current block): the one occasion on which columns are copied —
// the old ring's two runs, head first, to the front of the new one.
int MetaCovTest::grow() {
  const int want = std::min(maxColumns, capacity * 2);
  if (want <= capacity) {
    lastError = "MetaCov: one window holds more sites than RVT_METACOV_MAX_COLUMNS allows";
    return -1;
  }
  double* bigger = nullptr;
  const int V = (int)sites.size(), first = std::min(V, capacity - head);
  if (rvt_block_alloc(ctx, want, &bigger) || rvt_block_copy_columns(ctx, bigger, 0, block, head, first) ||
      (V > first && rvt_block_copy_columns(ctx, bigger, first, block, 0, V - first))) {
    lastError = rvt_last_error(ctx);
    if (bigger) rvt_block_free(ctx, bigger);
    return -1;
  }
  rvt_block_free(ctx, block);
  block = bigger;
  capacity = want;
  head = 0;
  return 0;
}

// Emit the rows of every head whose window is complete (all of them when `final`) and drop those heads: the ring's head index
// advances, no column moves (RingMemoryPool::deallocate, base/RingMemoryPool.cpp:49-63).
int MetaCovTest::flush(bool final) {
  const int V = (int)sites.size();
  if (V == 0 || !fout) return 0;
  const int d = nCovariate;
  auto outOfWindow = [&](int h, int j) {  // getWindowSize(queue, loci) > windowSize, src/Model.h:3974-3990
    return sites[j].chrom != sites[h].chrom || std::abs(sites[j].pos - sites[h].pos) > windowSize;
  };
  int H = 0;  // heads [0, H) are complete
  for (; H < V; ++H) {
    bool complete = final;
    for (int j = H + 1; j < V && !complete; ++j) complete = outOfWindow(H, j);
    if (!complete) break;
  }
  if (H == 0) return 0;  // nothing can be written yet: the caller enlarges the ring
  // whole row panels of the band kernel (256 heads): a panel of 24 heads costs the tiles of a full one; the rest waits in the
  // ring — nothing moves, so keeping them costs nothing
  if (!final && H > 256) H -= H % 256;
  const float scale = (float)(1.0 / (double)nSample);
  std::vector<double> zz((size_t)d * d);
  // last[h] = the last site printed in head h's row (the scan of printCovariance stops at the first site out of the window)
  std::vector<int> last((size_t)H);
  for (int h = 0; h < H; ++h) {
    int j = h;
    while (j + 1 < V && !outOfWindow(h, j + 1)) ++j;
    last[(size_t)h] = j;
  }
  // chunks of heads: one device call each — the band of the chunk's heads against everything up to the end of their windows,
  // read from the ring where it lies (the call addresses the columns modulo the capacity)
  const int Hc = 4096;
  std::vector<double> xz;
  std::vector<int> poly;
  for (int h0 = 0; h0 < H; h0 += Hc) {
    const int h1 = std::min(H, h0 + Hc), nh = h1 - h0;
    int halo = 0, jmax = h1 - 1;
    for (int h = h0; h < h1; ++h) {
      halo = std::max(halo, last[(size_t)h] - h);
      jmax = std::max(jmax, last[(size_t)h]);
    }
    const int W = jmax - h0 + 1;
    const size_t need = (size_t)nh * ((size_t)halo + 1);
    if (bandBuf.size() < need) {
      // the band lands by DMA in a page-locked buffer (a pageable target went through the runtime's staging copies: a third of
      // the device-side time of a flush)
      if (bandReg) rvt_host_unregister(ctx, bandReg);
      bandReg = nullptr;
      bandBuf.assign(need + need / 4, 0.0f);
      if (rvt_host_register(ctx, bandBuf.data(), sizeof(float) * bandBuf.size()) == 0) bandReg = bandBuf.data();
    }
    xz.assign((size_t)W * d, 0.0);
    poly.assign((size_t)W, 0);
    const int col0 = (head + h0) % capacity;
    if (useFamilyModel ? rvt_cov_band_fam(ctx, block, capacity, col0, nh, W, halo, scale, bandBuf.data(), xz.data(), zz.data(), poly.data())
                       : rvt_cov_band(ctx, block, capacity, col0, nh, W, halo, scale, bandBuf.data(), xz.data(), zz.data(), poly.data())) {
      lastError = rvt_last_error(ctx);
      return -1;
    }
    // the rows of the chunk as text: a row of a 1 000-marker window is 1 000 "%g" conversions (~0.2 ms on one core, more than
    // the device spends on it by two orders of magnitude), and rows are independent — formatted by a few threads, written in
    // order by this one
    std::vector<std::string> rowText((size_t)nh);
    auto formatRow = [&](int h) {
      if (!poly[(size_t)(h - h0)]) return;  // monomorphic sites never entered the reference's queue (src/Model.cpp:879-884)
      const float* row = bandBuf.data() + (size_t)(h - h0) * ((size_t)halo + 1);
      std::string positions, values;
      int lastPrinted = h, num = 0;
      for (int j = h; j <= last[(size_t)h]; ++j) {
        if (!poly[(size_t)(j - h0)]) continue;
        if (num) {
          positions += ',';
          values += ',';
        }
        positions += std::to_string(sites[j].pos);
        char num_buf[40];
        values.append(num_buf, formatG((double)row[j - h], num_buf));  // (the device applied the float cast and the float 1/N,
                                                                      //  src/Model.cpp:975-984)
        lastPrinted = j;
        ++num;
      }
      if (outputGwama || isBinaryOutcome()) {  // src/Model.cpp:992-1000
        const double* xzHead = xz.data() + (size_t)(h - h0) * d;
        values += ':';
        for (int k = 0; k < d; ++k) {
          if (k) values += ',';
          values += formatG((double)((float)xzHead[k] * scale));
        }
        values += ':';
        for (int a = 0; a < d; ++a)
          for (int b = 0; b <= a; ++b) {
            if (a || b) values += ',';
            values += floatToString(zz[(size_t)a * d + b] * (double)scale);
          }
      }
      rowText[(size_t)(h - h0)] = sites[h].chrom + "\t" + std::to_string(sites[h].pos) + "\t" + std::to_string(sites[lastPrinted].pos) +
                                  "\t" + std::to_string(num) + "\t" + positions + "\t" + values + "\n";
    };
    {
      const size_t work = (size_t)nh * ((size_t)halo + 1);
      int T = (int)std::min<size_t>(formatThreads, work / 20000 + 1);  // (a thread is worth its start from ~20 000 values on)
      if (T <= 1) {
        for (int h = h0; h < h1; ++h) formatRow(h);
      } else {
        std::vector<std::thread> pool;
        for (int t = 0; t < T; ++t)
          pool.emplace_back([&, t] {
            for (int h = h0 + t; h < h1; h += T) formatRow(h);  // (interleaved: rows near the end of the ring are shorter)
          });
        for (auto& th : pool) th.join();
      }
    }
    for (int h = h0; h < h1; ++h)
      if (!rowText[(size_t)(h - h0)].empty()) fout->write(rowText[(size_t)(h - h0)]);
  }
  head = (head + H) % capacity;
  sites.erase(sites.begin(), sites.begin() + H);
  return 0;
}

// ---- ModelManager (src/ModelManager.cpp:26-44 tokeniser, :46-271 switch) --------------------------------------------------------
ModelManager::~ModelManager() {
  GpuBroker::instance().flush();  // rows still pending refer to the models
  for (auto* m : model) delete m;
}

int ModelManager::create(const std::string& type, const std::string& modelList) {
  if (modelList.empty()) return 0;
  std::string modelType = lower(type);
  // split on ',' outside []
  std::vector<std::string> argModelName;
  std::string s;
  int depth = 0;
  for (char c : modelList) {
    if (c == '[') depth++;
    if (c == ']') depth--;
    if (c == ',' && depth == 0) {
      argModelName.push_back(s);
      s.clear();
    } else
      s.push_back(c);
  }
  argModelName.push_back(s);
  for (auto& a : argModelName) {
    ModelParser parser;
    if (parser.parse(a)) {
      lastError = "Please use this format: model(model_param1=v1)";
      return -1;
    }
    const std::string modelName = parser.getName();
    int nPerm;
    double alpha;
    if (modelType == "burden") {
      if (modelName == "cmc")
        model.push_back(new CMCTest);
      else if (modelName == "zeggini")
        model.push_back(new ZegginiTest);
      else if (modelName == "famcmc")  // src/ModelManager.cpp:133-136
        model.push_back(new FamBurdenTest(false));
      else if (modelName == "famzeggini")
        model.push_back(new FamBurdenTest(true));
      else {
        lastError = "Unknown model name: " + modelName + " .";
        return -1;
      }
    } else if (modelType == "kernel") {
      if (modelName == "skat") {
        double beta1, beta2;
        parser.assign("nPerm", &nPerm, 10000).assign("alpha", &alpha, 0.05).assign("beta1", &beta1, 1.0).assign(
            "beta2", &beta2, 25.0);
        model.push_back(new SkatTest(nPerm, alpha, beta1, beta2));
      } else if (modelName == "skato") {
        double beta1, beta2;
        parser.assign("beta1", &beta1, 1.0).assign("beta2", &beta2, 25.0);
        model.push_back(new SkatOTest(beta1, beta2));
      } else if (modelName == "kbac") {
        parser.assign("nPerm", &nPerm, 10000).assign("alpha", &alpha, 0.05);
        model.push_back(new KbacTest(nPerm, alpha));
      } else if (modelName == "famskat") {
        double beta1, beta2;
        parser.assign("beta1", &beta1, 1.0).assign("beta2", &beta2, 25.0);  // src/ModelManager.cpp:188-193
        model.push_back(new FamSkatTest(beta1, beta2));
      } else {
        lastError = "Unknown model name: " + modelName + " .";
        return -1;
      }
    } else if (modelType == "vt") {
      if (modelName == "analytic")  // src/ModelManager.cpp:158-161
        model.push_back(new AnalyticVTTest(false));
      else if (modelName == "famanalytic")
        model.push_back(new AnalyticVTTest(true));
      else {
        lastError = "Unknown model name: " + modelName + " .";
        return -1;
      }
    } else if (modelType == "meta") {
      if (modelName == "score") {  // src/ModelManager.cpp:209-210
        model.push_back(new MetaScoreTest());
      } else if (modelName == "cov") {
        int windowSize;
        parser.assign("windowSize", &windowSize, 1000000);  // src/ModelManager.cpp:227-233
        model.push_back(new MetaCovTest(windowSize));
      } else {
        lastError = "Unknown model name: " + modelName + " .";
        return -1;
      }
    } else {
      lastError = "Unrecognized model type: " + type;
      return -1;
    }
    model.back()->setParameter(parser);  // src/ModelManager.cpp:273-275
  }
  return 0;
}

void ModelManager::setBinaryOutcome() {
  for (auto* m : model) m->setBinaryOutcome();
}
void ModelManager::setQuantitativeOutcome() {
  for (auto* m : model) m->setQuantitativeOutcome();
}
std::vector<std::string> ModelManager::outputNames(const std::string& prefix) const {
  std::vector<std::string> v;
  for (auto* m : model) v.push_back(prefix + "." + m->getModelName() + ".assoc");
  return v;
}

}  // namespace rvt_host
