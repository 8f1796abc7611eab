mkdir -p gpurun_out/r3
for i in 1 2 3; do
RUNET_SIDE_RECORD_STREAM=1 python bench.py --steps 60 --warmup 8 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('record_stream', d['value'], d['ms_per_step'])"
python bench.py --steps 60 --warmup 8 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('held refs    ', d['value'], d['ms_per_step'])"
done
RUNET_SIDE_RECORD_STREAM=1 python tools/host_lead.py 16 256 30 2>&1 | tail -2 | cut -c1-200
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r3/gpu_all.log 2>&1; tail -2 gpurun_out/r3/gpu_all.log
