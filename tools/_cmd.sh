mkdir -p gpurun_out/r3
export MASTER_ADDR=127.0.0.1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 HSA_ENABLE_IPC_MODE_LEGACY=0 GPU_MAX_HW_QUEUES=8
for i in 1 2 3 4 5 6 7 8 9 10; do
MASTER_PORT=$((29500+i)) timeout -k 10 120 python tests/graph_ddp_child.py > gpurun_out/r3/child_$i.out 2> gpurun_out/r3/child_$i.err; echo "run $i rc=$? $(grep -c 'step' gpurun_out/r3/child_$i.err) $(tail -1 gpurun_out/r3/child_$i.out | cut -c1-60)"
done
