mkdir -p gpurun_out/r3
P=$PWD/eusipco-2026-robust-unet_amd/csrc/librunet_hip_prev.so
timeout -k 10 600 python -m pytest tests/test_gpu_conv.py -x -q > gpurun_out/r3/conv.log 2>&1; tail -2 gpurun_out/r3/conv.log
for i in 1 2 3; do
RUNET_HIP_LIB=$P python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('prev', d['value'], d['ms_per_step'])"
python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('new ', d['value'], d['ms_per_step'])"
done
cd /tmp && export TMPDIR=/tmp
RUNET_HIP_LIB=$P RUNET_NO_WGRAD_STREAM=1 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r3/kp -- python3 /root/repo/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline > /dev/null 2>&1
RUNET_NO_WGRAD_STREAM=1 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r3/kn -- python3 /root/repo/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline > /dev/null 2>&1
cd /root/repo
for d in kp kn; do f=$(ls gpurun_out/r3/$d/*/*kernel_stats.csv | head -1); echo $d; grep -E "wino4_output_adj|wino4_wgrad_out|wino_conv_x3|gemm_nn_x3|gemm_tn_x3" $f | cut -d, -f1-4 | cut -c1-120; done
rm -rf gpurun_out/r3/kp gpurun_out/r3/kn
