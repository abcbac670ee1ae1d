#!/bin/bash
# Register / spill / LDS usage of every kernel in the built library (reads the code-object metadata).
# usage: tools/kernel_regs.sh [path/to/librvtests_amd.so]
set -e
SO=$(readlink -f ${1:-rvtests_amd/csrc/librvtests_amd.so})
TMP=$(mktemp -d)
cd "$TMP"
# the library is linked from several objects: inspect the code object of each (rvt_engine.o, k2_*.o next to the .so)
for O in "$(dirname "$SO")"/*.o; do
objcopy -O binary --only-section=.hip_fatbin "$O" fatbin
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=fatbin --output=k.co
/opt/rocm/lib/llvm/bin/llvm-readelf --notes k.co | python3 -c '
import sys, re
txt = sys.stdin.read()
for blk in txt.split("- .agpr_count:")[1:]:
    g = lambda k: (re.search(r"\." + k + r":\s*(\S+)", blk) or [None, "?"])[1]
    agpr = blk.split("\n")[0].strip()
    name = g("name")
    print("%-70s vgpr %4s agpr %4s spill %4s sgpr %4s lds %6s scratch %6s" % (name[:70], g("vgpr_count"), agpr, g("vgpr_spill_count"), g("sgpr_count"), g("group_segment_fixed_size"), g("private_segment_fixed_size")))
'
done
rm -rf "$TMP"
