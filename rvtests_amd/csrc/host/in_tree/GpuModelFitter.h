// rvtests_amd — the binding a maintainer adds INSIDE the rvtests source tree (src/): GPU-backed models that derive from
// the reference's own ModelFitter (src/ModelFitter.h:17-75) and are constructed exactly where src/ModelManager.cpp
// constructs the CPU ones (:99-103, 168-198, 238-247).  This header is compiled against the REAL plugin headers
// (src/ModelFitter.h, src/Result.h, base/IO.h, base/MathMatrix.h, src/ModelParser.h, src/DataConsolidator.h): the
// repository's CPU test-suite does that syntax-only when the reference tree is present (tests/test_in_tree_binding.py),
// which is what proves that fit(DataConsolidator*), writeHeader / writeOutput(FileWriter*, const Result&) and the
// ModelManager constructor calls bind.
//
// Each model forwards to the adapter of the same name in ../ModelFitterGpu.h (namespace rvt_host), which talks to the
// C ABI (include/rvtests_amd.h).  What is translated here:
//   DataConsolidator*  -> rvt_host::GeneData       GpuDcShim.cpp (getters of src/DataConsolidator.h:126-137,185-186,223-258)
//   FileWriter*        -> rvt_host::TextSink       FileWriterSink below (FileWriter::write, base/IO.h:228-229)
//   const Result&      -> rvt_host::SiteInfo       header / value lines taken verbatim (Result::writeHeaderTab,
//                                                  Result::joinValue, src/Result.h:131-143,221-232); CHROM / POS by key
//   const ModelParser& -> rvt_host::ModelParser    the tags the GPU models read
#ifndef RVT_GPU_MODEL_FITTER_H_
#define RVT_GPU_MODEL_FITTER_H_

#include <map>
#include <memory>
#include <string>
#include <utility>

#include "ModelFitter.h"  // the reference's: class ModelFitter, Result, FileWriter (via Result.h -> base/IO.h)
#include "ModelParser.h"  // the reference's

#include "../ModelFitterGpu.h"

class DataConsolidator;

namespace rvt_intree {

// Fills what the GPU models read from the caller (src/DataConsolidator.h getters).  Defined in GpuDcShim.cpp.
void fillGeneData(DataConsolidator* dc, bool familyModel, const void* who, rvt_host::GeneData* gd);

class FileWriterSink : public rvt_host::TextSink {
 public:
  explicit FileWriterSink(FileWriter* f) : fp(f) {}
  void write(const std::string& s) override { fp->write(s); }

 private:
  FileWriter* fp;
};

inline rvt_host::SiteInfo siteInfoOf(const Result& r) {
  rvt_host::SiteInfo s;
  s.verbatim = true;
  s.valueLine = r.joinValue('\t') + "\t";  // what Result::writeValueTab would write
  // the single-variant models look their position up by key (src/Model.cpp:861-866: siteInfo["CHROM"], ["POS"])
  s.kv.push_back(std::make_pair(std::string("CHROM"), r["CHROM"]));
  s.kv.push_back(std::make_pair(std::string("POS"), r["POS"]));
  return s;
}

// One GPU-backed model: a reference ModelFitter whose work is done by the rvt_host adapter `Impl`.
template <class Impl>
class GpuModel : public ::ModelFitter {
 public:
  template <class... A>
  explicit GpuModel(A&&... a) : impl(std::forward<A>(a)...) {
    this->modelName = impl.getModelName();  // file name <prefix>.<modelName>.assoc (src/ModelManager.cpp:285-297)
  }
  int fit(DataConsolidator* dc) {
    if (isBinaryOutcome())
      impl.setBinaryOutcome();
    else
      impl.setQuantitativeOutcome();
    rvt_host::GeneData gd;
    fillGeneData(dc, familyAware, this, &gd);
    gd.site = &lastSite;
    return impl.fit(&gd);
  }
  void writeHeader(FileWriter* fp, const Result& siteInfo) {
    siteInfo.writeHeaderTab(fp);
    rvt_host::SiteInfo none;
    none.verbatim = true;  // the site columns were written above; the adapter adds its own columns
    impl.writeHeader(sinkFor(fp), none);
  }
  void writeOutput(FileWriter* fp, const Result& siteInfo) {
    lastSite = siteInfoOf(siteInfo);
    impl.writeOutput(sinkFor(fp), lastSite);
  }
  void writeFootnote(FileWriter* fp) { impl.writeFootnote(sinkFor(fp)); }
  int setParameter(const ::ModelParser& parser) {
    rvt_host::ModelParser p;
    static const char* const tags[] = {"windowSize", "gwama", "se", "nPerm", "alpha", "beta1", "beta2"};
    for (const char* t : tags)
      if (parser.hasTag(t)) p.set(t, parser.value(t));
    return impl.setParameter(p);
  }
  void reset() {
    ::ModelFitter::reset();
    impl.reset();
  }
  GpuModel& related() {  // the models that read the kinship decomposition (FamSkat, FamCMC, ...; meta models decide at fit)
    this->familyModel = true;
    familyAware = true;
    return *this;
  }

 private:
  rvt_host::TextSink* sinkFor(FileWriter* fp) {
    std::unique_ptr<FileWriterSink>& s = sinks[fp];
    if (!s) s.reset(new FileWriterSink(fp));
    return s.get();
  }
  Impl impl;
  bool familyAware = false;
  rvt_host::SiteInfo lastSite;
  std::map<FileWriter*, std::unique_ptr<FileWriterSink> > sinks;  // one per output file, alive until the model dies
                                                                 // (ModelManager::close deletes models before writers)
};

// the names src/ModelManager.cpp constructs (the reference's CPU classes of the same names live in src/Model.h)
typedef GpuModel<rvt_host::SkatTest> SkatTest;        // new SkatTest(nPerm, alpha, beta1, beta2)   ModelManager.cpp:175
typedef GpuModel<rvt_host::SkatOTest> SkatOTest;      // new SkatOTest(beta1, beta2)                :183
typedef GpuModel<rvt_host::CMCTest> CMCTest;          // new CMCTest()                              :99-103
typedef GpuModel<rvt_host::ZegginiTest> ZegginiTest;  // new ZegginiTest()
typedef GpuModel<rvt_host::FamSkatTest> FamSkatTest;  // new FamSkatTest(beta1, beta2)              :198
typedef GpuModel<rvt_host::FamBurdenTest> FamBurdenTest;
typedef GpuModel<rvt_host::MetaCovTest> MetaCovTest;      // new MetaCovTest(windowSize)            :238-247
typedef GpuModel<rvt_host::MetaScoreTest> MetaScoreTest;  // new MetaScoreTest()
typedef GpuModel<rvt_host::KbacTest> KBACTest;            // new KBACTest(nPerm, alpha)
// new AnalyticVT(AnalyticVT::UNRELATED) / new AnalyticVT(AnalyticVT::RELATED)   ModelManager.cpp:158-161
class AnalyticVT : public GpuModel<rvt_host::AnalyticVTTest> {
 public:
  typedef enum { UNRELATED = 0, RELATED = 1 } Type;
  explicit AnalyticVT(Type t) : GpuModel<rvt_host::AnalyticVTTest>(t == RELATED) {
    if (t == RELATED) related();
  }
};

}  // namespace rvt_intree
#endif  // RVT_GPU_MODEL_FITTER_H_
