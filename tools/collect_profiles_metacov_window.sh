#!/bin/bash
# Run on the GPU box (via gpurun): MetaCov's sliding window on the circular ring (tools/bench_metacov.py --skip-blocks) — the
# tool's JSON lines, the rocprofv3 kernel statistics of the same command and the two PMC passes (FETCH_SIZE, WRITE_SIZE;
# separate runs, kernel trace only) behind the HBM traffic per evicted variant.
# usage: tools/collect_profiles_metacov_window.sh <tag> [windows]  -> gpurun_out/prof_<tag>/metacov_window_*
set -u
TAG=${1:-r6}
WIN=${2:-1000,3000}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
python3 tools/bench_metacov.py --skip-blocks --window "$WIN" > "$OUT/metacov_window_result.txt" 2> "$OUT/metacov_window.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_mw" -o k -- python3 tools/bench_metacov.py --skip-blocks --no-cpu --window-dosage "" --window-imputed "" --window "$WIN" > "$OUT/metacov_window_kt.log" 2>&1
find "$OUT/kt_mw" -name '*kernel_stats.csv' -exec cp {} "$OUT/metacov_window_kernel_stats.csv" \;
rm -rf "$OUT/kt_mw"
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/pmc_mw_$C" -o p -- python3 tools/bench_metacov.py --skip-blocks --no-cpu --window-dosage "" --window-imputed "" --window "$WIN" > "$OUT/metacov_window_pmc_$C.log" 2>&1
  F=$(find "$OUT/pmc_mw_$C" -name '*counter_collection.csv' | head -1)
  python3 tools/pmc_summary.py "$F" "$OUT/metacov_window_pmc_$C.csv" > /dev/null
  rm -rf "$OUT/pmc_mw_$C"
done
python3 - "$OUT" <<'PY'
import csv, json, sys
out = sys.argv[1]
# (the counter passes run the hard-call windows only: the figure is §8(d)'s bytes per evicted variant of the path the bench names)
lines = [json.loads(l) for l in open(out + "/metacov_window_result.txt") if l.startswith("{") and "window_markers" in l]
lines = [l for l in lines if "hard calls, circular" in l["workload"]]
N = lines[0]["N"]
evicted = sum(l["variants"] * (l["flushes"] + 1) / l["flushes"] for l in lines)      # (the warm-up flush of every width counts)
def load(path, col):
    return {r["kernel"]: (int(r["dispatches"]), float(r[col])) for r in csv.DictReader(open(path))}
f, w = load(out + "/metacov_window_pmc_FETCH_SIZE.csv", "FETCH_SIZE"), load(out + "/metacov_window_pmc_WRITE_SIZE.csv", "WRITE_SIZE")
kern, tot = {}, 0.0
for k in f:
    name = k.split("rvt::")[1].split("(")[0].split("<")[0] if "rvt::" in k else k[:40]
    if not name.startswith("band_"):
        continue          # (the column passes behind the 1 024 uploads of the source block are input delivery)
    b = 2.0 * 1024.0 * f[k][1] + 1024.0 * w.get(k, (0, 0.0))[1]
    kern[name] = {"dispatches": f[k][0], "fetch_bytes": 2.0 * 1024.0 * f[k][1], "write_bytes": 1024.0 * w.get(k, (0, 0.0))[1]}
    tot += b
res = {"workload": "MetaCov sliding window on the circular ring, N=%d, windows %s" % (N, [l["window_markers"] for l in lines]),
       "correction": "FETCH_SIZE x2 (gfx950), x1024 B; WRITE_SIZE x1024 B", "evicted_variants": evicted,
       "hbm_bytes_per_evicted_variant": tot / evicted, "algorithmic_bytes_per_evicted_variant": 8.0 * N,
       "ratio": tot / evicted / (8.0 * N), "kernels": kern}
json.dump(res, open(out + "/pmc_traffic_metacov_window.json", "w"), indent=1)
print(json.dumps({k: res[k] for k in ("evicted_variants", "hbm_bytes_per_evicted_variant", "algorithmic_bytes_per_evicted_variant", "ratio")}))
PY
ls -la "$OUT"
