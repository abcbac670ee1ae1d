// rvtests_amd — host side (C++): GPU-backed ModelFitter adapters, ModelParser, ModelManager::create.
// See ModelFitterGpu.h for the reference lines each class mirrors.
#include "ModelFitterGpu.h"

#include <algorithm>
#include <cctype>
#include <cstdlib>
#include <cstring>

namespace rvt_host {

std::string floatToString(double v) {
  std::stringstream ss;
  ss.precision(6);
  ss << std::noshowpoint << v;
  return ss.str();
}

std::string formatG(double v) {
  char buf[64];
  snprintf(buf, sizeof(buf), "%g", v);
  return buf;
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
    if (!tok.empty()) {
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
  *v = hasTag(tag) ? atoi(value(tag)) : def;
  return *this;
}
const ModelParser& ModelParser::assign(const std::string& tag, bool* v, bool def) const {
  *v = hasTag(tag) ? true : def;
  return *this;
}

// ---- GpuBroker ----------------------------------------------------------------------------------------------------
GpuBroker& GpuBroker::instance() {
  static GpuBroker b;
  return b;
}

int GpuBroker::ensureContext(int device) {
  if (ctx) return 0;
  return rvt_init(&ctx, device);
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
  if (ctx) rvt_destroy(ctx);
  ctx = nullptr;
  haveNull = false;
  curSerial = -1;
  tests = 0;
}

// copyCovariateAndIntercept (src/ModelUtil.h:102-130) + the null fit SkatTest::fit caches (src/Model.h:2672-2699)
int GpuBroker::installNull(const GeneData& gd, bool binary, std::string* err) {
  if (!fitter) {
    *err = "no null-model fitter installed";
    return -1;
  }
  const int d = 1 + gd.ncov;
  std::vector<double> X((size_t)gd.N * d), res(gd.N), v(gd.N);
  for (int64_t i = 0; i < gd.N; ++i) X[i] = 1.0;
  if (gd.ncov) std::memcpy(X.data() + gd.N, gd.covariate, sizeof(double) * (size_t)gd.N * gd.ncov);
  double sigma2 = 1.0;
  if (fitter(binary, gd.N, d, X.data(), gd.phenotype, res.data(), v.data(), &sigma2)) {
    *err = binary ? "failed in fitting null model (logistic model)." : "failed in fitting null model (linear model).";
    return -1;
  }
  int rc = rvt_set_null(ctx, binary ? RVT_TRAIT_BINARY : RVT_TRAIT_QUANTITATIVE, gd.N, d, X.data(), res.data(),
                        v.data(), sigma2);
  if (rc) {
    *err = rvt_last_error(ctx);
    return -1;
  }
  haveNull = true;
  return 0;
}

const rvt_gene_result* GpuBroker::resultFor(const GeneData& gd, bool binary, std::string* err) {
  if (gd.serial == curSerial) return curOk ? &cur : nullptr;
  curSerial = gd.serial;
  curOk = false;
  if (ensureContext(0)) {
    *err = "no MI355X device: the GPU models have no CPU fallback";
    return nullptr;
  }
  if (!haveNull || gd.phenotypeUpdated || gd.covariateUpdated)
    if (installNull(gd, binary, err)) return nullptr;
  if ((int)gd.markerFrequency.size() < gd.M) {
    *err = "marker frequencies missing";
    return nullptr;
  }
  int rc = rvt_submit_gene(ctx, gd.serial, gd.M, gd.genotype, gd.markerFrequency.data(), tests, &params);
  int n = 0;
  if (!rc) rc = rvt_collect(ctx, &cur, 1, &n);
  if (rc || n != 1) {
    *err = rvt_last_error(ctx);
    return nullptr;
  }
  curOk = true;
  return &cur;
}

// ---- SkatTest ----------------------------------------------------------------------------------------------------------
SkatTest::SkatTest(int nPerm, double alpha, double beta1, double beta2) : usePermutation(nPerm > 0) {
  modelName = "Skat";
  rvt_params p{beta1, beta2, 1.0, 25.0, nPerm, alpha};
  GpuBroker::instance().registerTests(RVT_TEST_SKAT, p);
}
int SkatTest::fit(GeneData* dc) {
  fitOK = false;
  if (usePermutation) {  // permutation p-values are not part of the device path yet
    lastError = "skat: use skat[nPerm=0] with the GPU backend";
    return -1;
  }
  res = GpuBroker::instance().resultFor(*dc, isBinaryOutcome(), &lastError);
  if (!res || !res->skat_ok) return -1;  // genotype.cols == 0 after filtering -> NA row
  fitOK = true;
  return 0;
}
void SkatTest::writeHeader(TextSink* fp, const SiteInfo& siteInfo) {
  fp->write(siteInfo.headerTab());
  fp->write("Q\tPvalue\n");
}
void SkatTest::writeOutput(TextSink* fp, const SiteInfo& siteInfo) {
  fp->write(siteInfo.valueTab());
  if (!fitOK)
    fp->write("NA\tNA\n");
  else
    fp->write(formatG(res->skat_Q) + "\t" + formatG(res->skat_p) + "\n");
}

// ---- SkatOTest -----------------------------------------------------------------------------------------------------------
SkatOTest::SkatOTest(double beta1, double beta2) {
  modelName = "SkatO";
  rvt_params p{1.0, 25.0, beta1, beta2, 0, 0.05};
  GpuBroker::instance().registerTests(RVT_TEST_SKATO, p);
}
int SkatOTest::fit(GeneData* dc) {
  fitOK = false;
  res = GpuBroker::instance().resultFor(*dc, isBinaryOutcome(), &lastError);
  if (!res || res->n_poly == 0) return -1;
  fitOK = res->skato_ok != 0;  // fitOK = (skato.Fit(...) == 0); fit() itself returns 0 (src/Model.h:2853-2859)
  return 0;
}
void SkatOTest::writeHeader(TextSink* fp, const SiteInfo& siteInfo) {
  fp->write(siteInfo.headerTab());
  fp->write("Q\trho\tPvalue\n");
}
void SkatOTest::writeOutput(TextSink* fp, const SiteInfo& siteInfo) {
  fp->write(siteInfo.valueTab());
  if (!fitOK)
    fp->write("NA\tNA\tNA\n");
  else
    fp->write(formatG(res->skato_Q) + "\t" + formatG(res->skato_rho) + "\t" + formatG(res->skato_p) + "\n");
}

// ---- CMCTest / ZegginiTest -------------------------------------------------------------------------------------------------
CMCTest::CMCTest() {
  modelName = "CMC";
  GpuBroker::instance().registerTests(RVT_TEST_CMC, rvt_params{1.0, 25.0, 1.0, 25.0, 0, 0.05});
}
int CMCTest::fit(GeneData* dc) {
  fitOK = false;
  res = GpuBroker::instance().resultFor(*dc, isBinaryOutcome(), &lastError);
  if (!res || !res->cmc_ok) return -1;
  fitOK = true;
  return 0;
}
void CMCTest::writeHeader(TextSink* fp, const SiteInfo& siteInfo) {
  fp->write(siteInfo.headerTab());
  fp->write("NonRefSite\tPvalue\n");
}
void CMCTest::writeOutput(TextSink* fp, const SiteInfo& siteInfo) {
  fp->write(siteInfo.valueTab());
  if (fitOK)
    fp->write(std::to_string(res->cmc_nonref) + "\t" + floatToString(res->cmc_p) + "\n");
  else
    fp->write("NA\tNA\n");
}

ZegginiTest::ZegginiTest() {
  modelName = "Zeggini";
  GpuBroker::instance().registerTests(RVT_TEST_ZEGGINI, rvt_params{1.0, 25.0, 1.0, 25.0, 0, 0.05});
}
int ZegginiTest::fit(GeneData* dc) {
  fitOK = false;
  res = GpuBroker::instance().resultFor(*dc, isBinaryOutcome(), &lastError);
  if (!res || !res->zeg_ok) return -1;
  fitOK = true;
  return 0;
}
void ZegginiTest::writeHeader(TextSink* fp, const SiteInfo& siteInfo) {
  fp->write(siteInfo.headerTab());
  fp->write("Pvalue\n");
}
void ZegginiTest::writeOutput(TextSink* fp, const SiteInfo& siteInfo) {
  fp->write(siteInfo.valueTab());
  fp->write(fitOK ? floatToString(res->zeg_p) + "\n" : std::string("NA\n"));
}

// ---- ModelManager (src/ModelManager.cpp:26-44 tokeniser, :46-271 switch) --------------------------------------------------------
ModelManager::~ModelManager() {
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
      } else {
        lastError = "Unknown model name: " + modelName + " .";
        return -1;
      }
    } else {
      lastError = "Unrecognized model type: " + type;
      return -1;
    }
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
