mkdir -p gpurun_out/r3
timeout -k 10 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_prefetch.py tests/test_gpu_blocks.py -x -q > gpurun_out/r3/t.log 2>&1; tail -2 gpurun_out/r3/t.log
cd /tmp && export TMPDIR=/tmp
RUNET_NO_WGRAD_STREAM=1 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r3/kn -- python3 /root/repo/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline > /dev/null 2>&1
cd /root/repo
f=$(ls gpurun_out/r3/kn/*/*kernel_stats.csv | head -1); grep -E "derive_multi" $f | cut -c1-160; rm -rf gpurun_out/r3/kn
for i in 1 2 3; do python bench.py --batch 2 --graph --steps 60 --warmup 15 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('b2 graph', d['value'], d['ms_per_step'])"; done
python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('n1', d['value'], d['ms_per_step'])"
