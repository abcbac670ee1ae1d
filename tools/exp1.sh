run() { echo "== $*"; env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2), round(d['roofline']['frac'],3), {k:round(v,3) for k,v in d['kernel_time_share'].items()})"; }
run A=1
run RVT_K2_ALT=1
run RVT_BENCH_INFLIGHT=3
run RVT_BENCH_INFLIGHT=4
run RVT_BENCH_INFLIGHT=6
run RVT_BENCH_INFLIGHT=4 RVT_K2_ALT=1
