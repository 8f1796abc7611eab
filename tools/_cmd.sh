mkdir -p gpurun_out/r3
P=$PWD/eusipco-2026-robust-unet_amd/csrc/librunet_hip_prev.so
timeout -k 10 600 python -m pytest tests/test_gpu_conv.py -x -q > gpurun_out/r3/conv.log 2>&1; tail -3 gpurun_out/r3/conv.log
for i in 1 2 3; do
RUNET_HIP_LIB=$P python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('prev', d['value'], d['ms_per_step'])"
python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('new ', d['value'], d['ms_per_step'])"
done
RUNET_HIP_LIB=$P RUNET_NO_WGRAD_STREAM=1 python tools/conv_launches.py > gpurun_out/r3/cl_prev.txt 2>&1; RUNET_NO_WGRAD_STREAM=1 python tools/conv_launches.py > gpurun_out/r3/cl_new.txt 2>&1
python - <<'P'
a=open("gpurun_out/r3/cl_prev.txt").read().splitlines(); b=open("gpurun_out/r3/cl_new.txt").read().splitlines()
tp=tn=0
for x,y in zip(a,b):
    if "x3" in x or "gemm" in x:
        try:
            u0=float(x[95:104]); u1=float(y[95:104])
        except ValueError:
            continue
        tp+=u0; tn+=u1
        print(x[:104], f"{u1:8.1f}")
print("sum prev", tp, "new", tn)
P
