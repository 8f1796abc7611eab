mkdir -p gpurun_out/r3
timeout -k 10 900 python -m pytest tests/test_gpu_graph.py -x -q > gpurun_out/r3/t.log 2>&1; tail -3 gpurun_out/r3/t.log
