mkdir -p gpurun_out/r3
timeout -k 10 900 python -m pytest tests/test_gpu_blocks.py -x -q > gpurun_out/r3/blocks.log 2>&1; tail -5 gpurun_out/r3/blocks.log
for i in 1 2 3; do
RUNET_NO_FUSED_BN_INPUT=1 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('two-step', d['value'], d['ms_per_step'])"
python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fused   ', d['value'], d['ms_per_step'])"
done
for i in 1 2 3; do
for a in "--batch 2" ; do
RUNET_NO_DERIVE_MULTI=1 python bench.py $a --steps 60 --warmup 15 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$a per-tensor', d['value'], d['ms_per_step'])"
python bench.py $a --steps 60 --warmup 15 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$a multi     ', d['value'], d['ms_per_step'])"
done
done
