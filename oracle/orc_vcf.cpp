// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_api.h).  CPU restatement of the VCF genotype front end:
//   * column -> subfields at ':'                 VCFIndividual::parse / justGet   libVcf/VCFIndividual.h:27-58,88-93
//   * GT text -> 0 / 1 / 2 / MISSING_GENOTYPE    VCFValue::getGenotype            libVcf/VCFValue.h:74-117
//   * depth / quality filters                    checkGD / checkGQ                src/VCFGenotypeExtractor.cpp:304-317
//   * the per-sample loop and its -9 on failure  getGenotype                      src/VCFGenotypeExtractor.cpp:397-439
//   * FORMAT key -> subfield index (prefix match) VCFRecord::getFormatIndex       libVcf/VCFRecord.h:280-305
// Pinned against the reference's own libVcf/VCFIndividual + VCFValue compiled where they lie (oracle/_ref/libref_vcf.so,
// oracle/ref_vcf_shim.cpp) in tests/test_vcf_cpu.py.  Not covered (malformed input on which the reference itself
// misbehaves): an empty sample column and a column ending in ':'.
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {
const int kMissing = -9;  // libVcf/VCFConstant.h:4

// subfields of one column, as VCFIndividual::parse leaves them (each NUL-terminated in the reference)
std::vector<std::string> split_column(const char* col, int64_t len) {
  std::vector<std::string> fd;
  int64_t b = 0;
  for (;;) {
    int64_t e = b;
    while (e < len && col[e] != ':') ++e;
    fd.emplace_back(col + b, col + e);
    if (e >= len) break;
    b = e + 1;
  }
  return fd;
}

int gt_code(const std::string& s) {  // VCFValue::getGenotype; s.c_str() supplies the terminator the reference reads
  const char* line = s.c_str();
  const int end = (int)s.size();
  int g = 0, p = 0;
  if (line[p] == '.') return kMissing;
  if (line[p] < '0') return kMissing;
  g += line[p] - '0';
  if (g > 1) return kMissing;
  p++;
  if (p == end) return g;
  if (line[p] != '|' && line[p] != '/') return kMissing;
  p++;
  if (p == end) return kMissing;
  if (line[p] == '.') return kMissing;
  if (line[p] < '0') {
    // only reported
  } else {
    const int a2 = line[p] - '0';
    if (a2 > 1) return kMissing;
    g += a2;
  }
  p++;
  if (p != end) return kMissing;
  return g;
}

int alt_code(const std::string& s, int alt) {  // VCFValue::countAltAllele (libVcf/VCFValue.h:180-213)
  const char* line = s.c_str();
  const int end = (int)s.size();
  int g = 0, p = 0;
  if (line[p] == '.') return kMissing;
  g += (line[p] - '0' == alt ? 1 : 0);
  p++;
  if (p == end) return g;
  if (line[p] != '|' && line[p] != '/') return kMissing;
  p++;
  if (p == end) return kMissing;
  if (line[p] == '.') return kMissing;
  if (line[p] < '0' || line[p] > '9') {
    // only reported
  } else {
    g += (line[p] - '0' == alt ? 1 : 0);
  }
  p++;
  if (p != end) return kMissing;
  return g;
}

int column_code(const char* col, int64_t len, int gt_idx, int gd_idx, int gq_idx, const int* flt, int alt = 0) {
  const std::vector<std::string> fd = split_column(col, len);
  auto just_get = [&](int i) -> std::string {  // index past the end (or negative: unsigned wrap) -> default "."
    if (i < 0 || i >= (int)fd.size()) return std::string(".");
    return fd[i];
  };
  if (gt_idx < 0) return kMissing;  // "Cannot find GT field!"
  int ret = alt > 0 ? alt_code(just_get(gt_idx), alt) : gt_code(just_get(gt_idx));
  if (flt) {
    if (flt[0] > 0 || flt[1] > 0) {
      const int gd = atoi(just_get(gd_idx).c_str());
      if ((flt[0] > 0 && gd < flt[0]) || (flt[1] > 0 && gd > flt[1])) return kMissing;
    }
    if (flt[2] > 0 || flt[3] > 0) {
      const int gq = atoi(just_get(gq_idx).c_str());
      if ((flt[2] > 0 && gq < flt[2]) || (flt[3] > 0 && gq > flt[3])) return kMissing;
    }
  }
  return ret;
}
}  // namespace

extern "C" {

int orc_vcf_column_genotype(const char* col, int64_t len, int gt_idx) { return column_code(col, len, gt_idx, -1, -1, nullptr); }
int orc_vcf_column_alt(const char* col, int64_t len, int gt_idx, int alt) {
  return column_code(col, len, gt_idx, -1, -1, nullptr, alt);
}

// text = the sample columns of one record (tab separated, no newline).  out[row_of_sample[s]] = code of column s.
// Returns the number of columns found.
int orc_vcf_decode_record(const char* text, int64_t len, int n_file_samples, const int32_t* row_of_sample, int gt_idx,
                          int gd_idx, int gq_idx, const int* filters, int8_t* out) {
  int s = 0;
  int64_t b = 0;
  for (;;) {
    int64_t e = b;
    while (e < len && text[e] != '\t') ++e;
    if (s < n_file_samples && row_of_sample[s] >= 0)
      out[row_of_sample[s]] = (int8_t)column_code(text + b, e - b, gt_idx, gd_idx, gq_idx, filters);
    ++s;
    if (e >= len) break;
    b = e + 1;
  }
  return s;
}

// dosage mode (useDosage: ret = indv.justGet(genoIdx).toDouble(), src/VCFGenotypeExtractor.cpp:403-414; toDouble = atof of
// the NUL-terminated subfield, libVcf/VCFValue.h:38-41; the default value "." reads 0.0); then the GD / GQ filters
int orc_vcf_decode_record_dosage(const char* text, int64_t len, int n_file_samples, const int32_t* row_of_sample,
                                 int tag_idx, int gd_idx, int gq_idx, const int* flt, double* out) {
  int s = 0;
  int64_t b = 0;
  for (;;) {
    int64_t e = b;
    while (e < len && text[e] != '\t') ++e;
    if (s < n_file_samples && row_of_sample[s] >= 0) {
      const std::vector<std::string> fd = split_column(text + b, e - b);
      auto just_get = [&](int i) -> std::string {
        if (i < 0 || i >= (int)fd.size()) return std::string(".");
        return fd[i];
      };
      double g = tag_idx < 0 ? (double)kMissing : atof(just_get(tag_idx).c_str());
      if (flt) {
        if (flt[0] > 0 || flt[1] > 0) {
          const int gd = atoi(just_get(gd_idx).c_str());
          if ((flt[0] > 0 && gd < flt[0]) || (flt[1] > 0 && gd > flt[1])) g = kMissing;
        }
        if (flt[2] > 0 || flt[3] > 0) {
          const int gq = atoi(just_get(gq_idx).c_str());
          if ((flt[2] > 0 && gq < flt[2]) || (flt[3] > 0 && gq > flt[3])) g = kMissing;
        }
      }
      out[row_of_sample[s]] = g;
    }
    ++s;
    if (e >= len) break;
    b = e + 1;
  }
  return s;
}

// VCFRecord::getFormatIndex: the FORMAT entry at the current position only has to START with the key
int orc_vcf_format_index(const char* format, int64_t len, const char* key) {
  std::string f(format, format + len);
  f.push_back('\t');  // the reference's buffer continues with the rest of the line
  int64_t b = 0;
  const int64_t e = len;
  int idx = 0;
  while (b < e) {
    bool match = true;
    for (int i = 0; key[i] != '\0'; i++)
      if (f[b + i] != key[i]) {
        match = false;
        break;
      }
    if (match) return idx;
    idx++;
    while (f[b++] != ':')
      if (b >= e) return -1;
  }
  return -1;
}

}  // extern "C"
