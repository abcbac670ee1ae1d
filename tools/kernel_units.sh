#!/bin/bash
# Which device kernels does every object of the library carry?  Reads the code-object metadata of each .o (as
# tools/kernel_regs.sh) and prints, per kernel, the objects that hold a copy — a kernel that is `static` in a header several
# translation units include is compiled (and shipped) once per unit.
# usage: tools/kernel_units.sh [dir with the .o files]   -> "<copies> <kernel> : <objects>" for kernels with more than one copy,
#        then the totals
set -e
DIR=$(readlink -f ${1:-rvtests_amd/csrc})
TMP=$(mktemp -d)
cd "$TMP"
: > all.txt
for O in "$DIR"/*.o; do
  objcopy -O binary --only-section=.hip_fatbin "$O" fatbin 2>/dev/null || continue
  [ -s fatbin ] || continue
  /opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=fatbin --output=k.co 2>/dev/null || continue
  /opt/rocm/lib/llvm/bin/llvm-readelf --notes k.co | grep -E "^\s+\.name:" | sed 's/.*\.name:\s*//' | sort -u | while read -r K; do echo "$K $(basename "$O")"; done >> all.txt
  echo "$(basename "$O"): $(grep -c " $(basename "$O")\$" all.txt) kernels, code object $(stat -c %s k.co) bytes"
done
python3 - <<'PY'
import collections
d = collections.defaultdict(list)
for ln in open("all.txt"):
    k, o = ln.rsplit(" ", 1)
    d[k].append(o.strip())
dup = {k: v for k, v in d.items() if len(v) > 1}
for k, v in sorted(dup.items(), key=lambda kv: -len(kv[1]))[:400]:
    print(len(v), k[:90], ":", " ".join(v))
print("kernels: %d distinct, %d copies in all, %d kernels in more than one object" % (len(d), sum(len(v) for v in d.values()), len(dup)))
PY
rm -rf "$TMP"
