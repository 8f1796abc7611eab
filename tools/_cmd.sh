mkdir -p gpurun_out/r3
timeout -k 10 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_blocks.py tests/test_gpu_model.py tests/test_gpu_fullsize.py -x -q > gpurun_out/r3/t.log 2>&1; tail -3 gpurun_out/r3/t.log
for i in 1 2 3; do
RUNET_NO_EPILOGUE_STATS=1 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('no epilogue stats', d['value'], d['ms_per_step'])"
python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('all              ', d['value'], d['ms_per_step'])"
done
