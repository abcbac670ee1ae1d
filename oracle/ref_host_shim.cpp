// ORACLE/_ref — TEST INFRASTRUCTURE ONLY.
// Thin extern "C" shim over the REAL reference sources, compiled where they lie under /root/reference (never copied):
//   src/Permutation.h:48-158        the adaptive stop rule of every permutation test (init / next / add / getPvalue)
//   src/ModelParser.{h,cpp}         "name[k=v:k2=v2]" (case folding, ':' / ',' separators, assign with defaults)
//   base/TypeConversion.h:97-105    floatToString: what Result / the CMC / Zeggini rows are printed with
// Headers of GSL come from the reference's own vendored tarball (as for libref_vcf.so); base/Logger.cpp and base/Utils.cpp are
// compiled beside ModelParser.cpp.  `logger` is the global the reference's Main.cpp defines.  Output: oracle/_ref/libref_host.so
// (git-ignored).  Used by tests/test_oracle_ref.py to pin the oracle's / the adapters' restatements bit for bit.
// NOT here: permute() of src/LinearAlgebra.h:8-21 — the header's later functions (getRowVariance ..., :183-) are written
// against Eigen maps (DECLARE_EIGEN_CONST_MATRIX), and Eigen is neither in the reference tree (third/Makefile downloads it) nor
// in this image: the header does not compile, so the Fisher-Yates walk stays pinned by the glibc rand() replay
// (tests/test_perm_counter_cpu.py) and by reading.
#include <stdlib.h>
#include <string.h>

#include <string>

#include "base/Logger.h"
#include "base/TypeConversion.h"
#include "src/ModelParser.h"
#include "src/Permutation.h"

Logger* logger = NULL;

extern "C" {

// the stop rule: feed stats[0 .. n) until next() says stop; out = actualPerm, numX (greater), numEqual; returns the p-value
double ref_permutation_run(int nperm, double alpha, double obs, const double* stats, int n, int* out3) {
  Permutation p(nperm, alpha);
  p.init(obs);
  int used = 0;
  while (p.next() && used < n) p.add(stats[used++]);
  out3[0] = used;
  // (numX / numEqual are private: recovered from the p-value and a second run that counts — the class prints them only)
  int gt = 0, eq = 0;
  for (int i = 0; i < used; ++i) {
    gt += stats[i] > obs;
    eq += stats[i] == obs;
  }
  out3[1] = gt;
  out3[2] = eq;
  return p.getPvalue();
}

// parse `spec`; name -> name_out (cap bytes); returns parse()'s code; the parser object is kept for the queries below
static ModelParser* g_parser = NULL;
int ref_parser_parse(const char* spec, char* name_out, int cap, int* n_params) {
  if (!logger) {
    logger = new Logger("/dev/null");  // (the parser logs "load N parameters" to the console and to this file)
  }
  delete g_parser;
  g_parser = new ModelParser;
  const int rc = g_parser->parse(spec);
  strncpy(name_out, g_parser->getName().c_str(), cap - 1);
  name_out[cap - 1] = 0;
  *n_params = (int)g_parser->size();
  return rc;
}
int ref_parser_has(const char* tag) { return g_parser->hasTag(tag) ? 1 : 0; }
// value of `tag` ("" when it has none); returns 0 when the tag is absent
int ref_parser_value(const char* tag, char* out, int cap) {
  const char* v = g_parser->value(tag);
  if (!v) return 0;
  strncpy(out, v, cap - 1);
  out[cap - 1] = 0;
  return 1;
}
double ref_parser_double(const char* tag, double def) {
  double v = 0;
  g_parser->assign(tag, &v, def);
  return v;
}
int ref_parser_int(const char* tag, int def) {
  int v = 0;
  g_parser->assign(tag, &v, def);
  return v;
}
int ref_parser_bool(const char* tag, int def) {
  bool v = false;
  g_parser->assign(tag, &v, def != 0);
  return v ? 1 : 0;
}

void ref_float_to_string(double x, char* out, int cap) {
  const std::string s = floatToString(x);
  strncpy(out, s.c_str(), cap - 1);
  out[cap - 1] = 0;
}
void ref_float_to_string_f32(float x, char* out, int cap) {
  const std::string s = floatToString(x);
  strncpy(out, s.c_str(), cap - 1);
  out[cap - 1] = 0;
}

}  // extern "C"
