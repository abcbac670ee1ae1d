#!/bin/bash
# Profiling build of the engine (gene_pvalue_kernel cycle counters, RVT_PROF_K4): only rvt_engine.hip is recompiled;
# links the regular objects of the other translation units.  -> rvtests_amd/csrc/librvtests_amd_prof.so
set -e
cd "$(dirname "$0")/../rvtests_amd/csrc"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -DRVT_PROF_K4 -c rvt_engine.hip -o rvt_engine_prof.o
hipcc --offload-arch=gfx950 -shared -o librvtests_amd_prof.so rvt_engine_prof.o rvt_stream.o rvt_fam.o rvt_perm.o rvt_meta.o k2_unweighted.o k2_weighted.o k2_hardcall.o k2_hardcall_w.o k2_hardcall_x.o k2_lattice.o k2_packed.o k2_floatdigit.o rvt_group.o -lpthread
ls -la librvtests_amd_prof.so
