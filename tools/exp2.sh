run() { echo "== $*"; env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2), round(d['roofline']['frac'],3), {k:round(v,3) for k,v in d['kernel_time_share'].items()})"; }
run RVT_WPARTS=128
run RVT_WPARTS=64
run RVT_WPARTS=96
run RVT_WPARTS=192
