// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_api.h).  CPU restatement of the VCF genotype front end:
//   * column -> subfields at ':'                 VCFIndividual::parse / justGet   libVcf/VCFIndividual.h:27-58,88-93
//   * GT text -> 0 / 1 / 2 / MISSING_GENOTYPE    VCFValue::getGenotype            libVcf/VCFValue.h:74-117
//   * depth / quality filters                    checkGD / checkGQ                src/VCFGenotypeExtractor.cpp:304-317
//   * the per-sample loop and its -9 on failure  getGenotype                      src/VCFGenotypeExtractor.cpp:397-439
//   * hemizygous regions: males through          getMaleNonParGenotype02 /        libVcf/VCFValue.h:125-142,214-236
//     (unknown sex -> missing, dosage x 2)       countMaleNonParAltAllele2
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

// VCFValue::getAllele1 / getAllele2 / isHaploid (libVcf/VCFValue.h:159-179,242)
int allele1(const std::string& s) {
  const char* line = s.c_str();
  if (line[0] == '.') return kMissing;
  if (line[0] < '0') return 0;  // only reported
  return line[0] - '0';
}
int allele2(const std::string& s) {
  const char* line = s.c_str();
  if (2 >= (int)s.size()) return kMissing;
  if (line[2] == '.') return kMissing;
  if (line[2] < '0') return 0;  // only reported
  return line[2] - '0';
}
// VCFValue::getMaleNonParGenotype02 (libVcf/VCFValue.h:125-142): a male in a hemizygous region
int male02_code(const std::string& s) {
  const int g = allele1(s);
  if (g == kMissing) return kMissing;
  if (s.size() == 1) return g == 0 ? 0 : (g == 1 ? 2 : kMissing);
  const int g2 = allele2(s);
  if (g2 == kMissing) return kMissing;
  if (g == g2) {
    if (g == 0) return 0;
    if (g == 1) return 2;
  }
  return kMissing;
}
// VCFValue::countMaleNonParAltAllele2 (libVcf/VCFValue.h:214-236; release build: its range assert is compiled out)
int male_alt_code(const std::string& s, int alt) {
  const int g = allele1(s);
  if (g == kMissing) return kMissing;
  if (s.size() == 1) return g == alt ? 2 : 0;
  const int g2 = allele2(s);
  if (g2 == kMissing) return kMissing;
  if (g == g2) return (g == alt ? 1 : 0) + (g2 == alt ? 1 : 0);
  return kMissing;
}

// hemi: the record lies in a hemizygous region (ParRegion::isHemiRegion); sex: PLINK code of the sample (1 male,
// 2 female, anything else unknown) — src/VCFGenotypeExtractor.cpp:397-439 (getGenotype), :441-484 (...ForAltAllele)
int column_code(const char* col, int64_t len, int gt_idx, int gd_idx, int gq_idx, const int* flt, int alt = 0,
                int hemi = 0, int sex = 0) {
  const std::vector<std::string> fd = split_column(col, len);
  auto just_get = [&](int i) -> std::string {  // index past the end (or negative: unsigned wrap) -> default "."
    if (i < 0 || i >= (int)fd.size()) return std::string(".");
    return fd[i];
  };
  if (gt_idx < 0) return kMissing;  // "Cannot find GT field!"
  int ret;
  if (!hemi || sex == 2)
    ret = alt > 0 ? alt_code(just_get(gt_idx), alt) : gt_code(just_get(gt_idx));
  else if (sex == 1)
    ret = alt > 0 ? male_alt_code(just_get(gt_idx), alt) : male02_code(just_get(gt_idx));
  else
    ret = kMissing;
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

int orc_vcf_column_male02(const char* col, int64_t len, int gt_idx) {
  return column_code(col, len, gt_idx, -1, -1, nullptr, 0, 1, 1);
}
int orc_vcf_column_male_alt(const char* col, int64_t len, int gt_idx, int alt) {
  return column_code(col, len, gt_idx, -1, -1, nullptr, alt, 1, 1);
}

// as orc_vcf_decode_record with the alternative allele of multi-allelic mode (0 = bi-allelic coding), the hemizygous
// flag of the record and the PLINK sex code of every FILE sample (may be NULL when hemi = 0)
int orc_vcf_decode_record_sex(const char* text, int64_t len, int n_file_samples, const int32_t* row_of_sample, int gt_idx,
                              int gd_idx, int gq_idx, const int* filters, int alt, int hemi, const int8_t* sex,
                              int8_t* out) {
  int s = 0;
  int64_t b = 0;
  for (;;) {
    int64_t e = b;
    while (e < len && text[e] != '\t') ++e;
    if (s < n_file_samples && row_of_sample[s] >= 0)
      out[row_of_sample[s]] = (int8_t)column_code(text + b, e - b, gt_idx, gd_idx, gq_idx, filters, alt, hemi,
                                                  (hemi && sex) ? sex[s] : 0);
    ++s;
    if (e >= len) break;
    b = e + 1;
  }
  return s;
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
int orc_vcf_decode_record_dosage_sex(const char* text, int64_t len, int n_file_samples, const int32_t* row_of_sample,
                                     int tag_idx, int gd_idx, int gq_idx, const int* flt, int hemi, const int8_t* sex,
                                     double* out);
int orc_vcf_decode_record_dosage(const char* text, int64_t len, int n_file_samples, const int32_t* row_of_sample,
                                 int tag_idx, int gd_idx, int gq_idx, const int* flt, double* out) {
  return orc_vcf_decode_record_dosage_sex(text, len, n_file_samples, row_of_sample, tag_idx, gd_idx, gq_idx, flt, 0,
                                          nullptr, out);
}
// ... in a hemizygous region a male's dosage is doubled (src/VCFGenotypeExtractor.cpp:407-414); sex is not looked at otherwise
int orc_vcf_decode_record_dosage_sex(const char* text, int64_t len, int n_file_samples, const int32_t* row_of_sample,
                                     int tag_idx, int gd_idx, int gq_idx, const int* flt, int hemi, const int8_t* sex,
                                     double* out) {
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
      if (tag_idx >= 0 && hemi && sex && sex[s] == 1) g = g * 2.0;
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
