mkdir -p gpurun_out/r3
run() { python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
for i in 1 2 3 4; do
run base
RUNET_CONV_X3_MIN_K=64 run min_k64
RUNET_CONV_X3_MIN_K=64 RUNET_CONV_X3_WIDE_N=64 run both64
done
