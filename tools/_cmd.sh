mkdir -p gpurun_out/r3
for i in 1 2 3; do
for a in "--batch 2" ; do
RUNET_NO_DERIVE_MULTI=1 python bench.py $a --steps 60 --warmup 15 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$a per-tensor', d['value'], d['ms_per_step'])"
python bench.py $a --steps 60 --warmup 15 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$a multi     ', d['value'], d['ms_per_step'])"
done
done
for a in "--config 1" "--config 5"; do
RUNET_NO_DERIVE_MULTI=1 python bench.py $a --steps 40 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$a per-tensor', d['value'], d['ms_per_step'])"
python bench.py $a --steps 40 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$a multi     ', d['value'], d['ms_per_step'])"
done
