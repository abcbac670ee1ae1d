#!/bin/bash
# Run on the GPU box (via gpurun): kernel trace of a short bench run + the timeline summary.
# usage: tools/prof_trace.sh <tag> [bench args...]   -> gpurun_out/trace_<tag>/{kernel_stats.csv,timeline.txt,bench.json}
set -u
TAG=${1:-t}
shift
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/trace_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -o k -- python3 bench.py --steps 10 --warmup 5 --no-cpu-baseline "$@" > "$OUT/bench.json" 2> "$OUT/bench.err"
find "$OUT/kt" -name '*kernel_stats.csv' -exec cp {} "$OUT/kernel_stats.csv" \;
TR=$(find "$OUT/kt" -name '*kernel_trace.csv' | head -1)
python3 tools/timeline.py "$TR" --last-ms 200 --rows 120 > "$OUT/timeline.txt" 2>&1
rm -rf "$OUT/kt"
head -40 "$OUT/timeline.txt"
tail -c 1200 "$OUT/bench.json"
