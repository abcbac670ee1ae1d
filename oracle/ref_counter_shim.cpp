// ORACLE/_ref — TEST INFRASTRUCTURE ONLY.
// Thin extern "C" shim over the REAL reference sources, compiled where they lie under /root/reference (never copied):
//   src/GenotypeCounter.{h,cpp}   the per-variant counter behind getAF() / getHWE() (weights of every kernel test, quirk #4)
//   libsrc/snp_hwe.cpp            the exact HWE test getHWE() calls
//   base/RingMemoryPool.{h,cpp}   the ring of float chunks MetaCovTest keeps its window in (src/Model.h:3915-4096, quirk #20)
// None of them has a third-party dependency.  Output: oracle/_ref/libref_counter.so (git-ignored).
// Used by tests/test_oracle_ref.py to pin orc_counter_af and the window walk of orc_metacov.
#include "src/GenotypeCounter.h"
#include "base/RingMemoryPool.h"

extern "C" {

// out[8] = AF, AC, call rate, HWE p-value, nHomRef, nHet, nHomAlt, nMissing
void ref_counter(const double* g, long long n, double* out) {
  GenotypeCounter c;
  for (long long i = 0; i < n; ++i) c.add(g[i]);
  out[0] = c.getAF();
  out[1] = c.getAC();
  out[2] = c.getCallRate();
  out[3] = c.getHWE();
  out[4] = c.getNumHomRef();
  out[5] = c.getNumHet();
  out[6] = c.getNumHomAlt();
  out[7] = c.getNumMissing();
}

void* ref_ring_new(int elements_per_chunk, int chunks) { return new RingMemoryPool(elements_per_chunk, chunks); }
void ref_ring_delete(void* p) { delete static_cast<RingMemoryPool*>(p); }
int ref_ring_allocate(void* p) { return static_cast<RingMemoryPool*>(p)->allocate(); }
void ref_ring_deallocate(void* p, int idx) { static_cast<RingMemoryPool*>(p)->deallocate(idx); }
float* ref_ring_chunk(void* p, int idx) { return static_cast<RingMemoryPool*>(p)->chunk(idx); }
long long ref_ring_size(void* p) { return (long long)static_cast<RingMemoryPool*>(p)->size(); }
long long ref_ring_capacity(void* p) { return (long long)static_cast<RingMemoryPool*>(p)->capacity(); }

}  // extern "C"
