
"""Does a single-rank RCCL all_reduce block the HOST until the waited-for streams catch up?  (diagnostic for ddp.GradAllReducer)"""
import os
import time

import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29541")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
a = torch.randn(8192, 8192, device=dev)
buf = torch.randn(8 << 20, device=dev)
side, comm = torch.cuda.Stream(), torch.cuda.Stream()
dist.all_reduce(buf)
torch.cuda.synchronize()
for trial in range(3):
    with torch.cuda.stream(side):
        for _ in range(20):
            b = a @ a                      # ~20 x 1.1 TFLOP of fp32 work: tens of ms
    t0 = time.perf_counter()
    comm.wait_stream(side)
    with torch.cuda.stream(comm):
        dist.all_reduce(buf)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"trial {trial}: all_reduce call returned after {1e3 * (t1 - t0):.2f} ms; queued GPU work finished after {1e3 * (t2 - t0):.2f} ms")
dist.destroy_process_group()
