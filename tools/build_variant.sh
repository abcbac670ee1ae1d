#!/bin/bash
# Experimental build of the whole engine with extra compiler flags -> rvtests_amd/csrc/variant_<tag>/librvtests_amd.so
# usage: tools/build_variant.sh <tag> <flags...>     e.g.  tools/build_variant.sh inflight16 -DRVT_MAX_INFLIGHT=16
# run with RVT_LIBRARY=rvtests_amd/csrc/variant_<tag>/librvtests_amd.so (+ RVT_PY_MAX_INFLIGHT=16 for bench.py's batch ring)
set -e
TAG=$1; shift
cd "$(dirname "$0")/../rvtests_amd/csrc"
OUT=variant_$TAG
mkdir -p $OUT
UNITS="rvt_engine rvt_stream rvt_fam rvt_perm rvt_meta k2_unweighted k2_weighted k2_hardcall k2_hardcall_w k2_hardcall_x k2_lattice k2_packed k2_floatdigit"
for u in $UNITS; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-unused-function "$@" -c $u.hip -o $OUT/$u.o &
done
g++ -std=c++17 -O2 -fPIC "$@" -c rvt_group.cpp -o $OUT/rvt_group.o &
wait
hipcc --offload-arch=gfx950 -shared -o $OUT/librvtests_amd.so $OUT/*.o -lpthread
ls -la $OUT/librvtests_amd.so
