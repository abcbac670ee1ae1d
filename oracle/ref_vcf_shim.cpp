// ORACLE/_ref — TEST INFRASTRUCTURE ONLY.
// Thin extern "C" shim over the REAL reference sources for the VCF genotype text rules, compiled where they lie under
// /root/reference (never copied): libVcf/VCFIndividual.{h,cpp}, libVcf/VCFValue.{h,cpp}, base/Utils.cpp (ssechr).  Their
// headers include GSL headers by the path inside the reference's own vendored tarball (third/gsl-1.16.tar.gz), which the
// Makefile unpacks into oracle/_ref/gslinc.  Output: oracle/_ref/libref_vcf.so (git-ignored).
// Used to validate oracle/orc_vcf.cpp (tests/test_vcf_cpu.py).
#include <string>
#include "libVcf/VCFIndividual.h"

extern "C" int ref_vcf_column_genotype(const char* column, int len, int gt_idx) {
  std::string buf(column, len);
  buf.append(32, '\0');  // ssechr reads 16 bytes at a time
  VCFValue v(&buf[0], 0, len);
  VCFIndividual indv;
  indv.parse(v);
  return indv.justGet(gt_idx).getGenotype();
}

extern "C" int ref_vcf_column_int(const char* column, int len, int idx) {
  std::string buf(column, len);
  buf.append(32, '\0');
  VCFValue v(&buf[0], 0, len);
  VCFIndividual indv;
  indv.parse(v);
  return indv.justGet(idx).toInt();
}

extern "C" int ref_vcf_column_alt(const char* column, int len, int gt_idx, int alt) {
  std::string buf(column, len);
  buf.append(32, '\0');
  VCFValue v(&buf[0], 0, len);
  VCFIndividual indv;
  indv.parse(v);
  return indv.justGet(gt_idx).countAltAllele(alt);
}

extern "C" int ref_vcf_column_male02(const char* column, int len, int gt_idx) {
  std::string buf(column, len);
  buf.append(32, '\0');
  VCFValue v(&buf[0], 0, len);
  VCFIndividual indv;
  indv.parse(v);
  return indv.justGet(gt_idx).getMaleNonParGenotype02();
}

extern "C" int ref_vcf_column_male_alt(const char* column, int len, int gt_idx, int alt) {
  std::string buf(column, len);
  buf.append(32, '\0');
  VCFValue v(&buf[0], 0, len);
  VCFIndividual indv;
  indv.parse(v);
  return indv.justGet(gt_idx).countMaleNonParAltAllele2(alt);
}
