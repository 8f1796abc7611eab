"""What the data-parallel wiring costs a step when the collective itself is free: ONE rank with an RCCL process group.
Times the train step (a) without a GradAllReducer, (b) with it, (c) with it but torch.distributed.all_reduce replaced by a no-op, and
prints the host-side profile of (b).  Every millisecond of (b) - (a) is lost on each rank of an N-GPU run before any link is involved.
    python tools/ddp_overhead.py [--batch 2] [--size 256] [--dtype f32]
"""
import argparse
import cProfile
import importlib
import os
import pstats
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def timed(step, x, y, steps):
    for _ in range(6):
        step(x, y)
    torch.cuda.synchronize()
    best, host = 1e9, 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(steps):
            step(x, y)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / steps)
        host = min(host, (t1 - t0) / steps)
    return best * 1e3, host * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--steps", type=int, default=20)
    args = ap.parse_args()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", RANK="0", WORLD_SIZE="1")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=dev)
    pkg = importlib.import_module("eusipco-2026-robust-unet_amd")
    x, y = pkg.synthetic_batch(args.batch, args.size, seed=1)
    x, y = x.to(dev), y.to(dev)

    def build(with_sync):
        torch.manual_seed(0)
        m = pkg.RobustUNet(3, 1, 64).to(dev).train().set_precision(args.dtype)
        sync = pkg.GradAllReducer(m) if with_sync else None
        return pkg.TrainStep(m, grad_sync=sync, loss_scale=1024.0 if args.dtype == "fp16" else None), sync

    step, _ = build(False)
    a = timed(step, x, y, args.steps)
    print(f"(a) no GradAllReducer            {a[0]:7.3f} ms/step, host enqueue {a[1]:7.3f} ms")
    step, sync = build(True)
    b = timed(step, x, y, args.steps)
    print(f"(b) GradAllReducer, 1-rank RCCL  {b[0]:7.3f} ms/step, host enqueue {b[1]:7.3f} ms   buckets {sync.buckets_last_step}")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(10):
        step(x, y)
    pr.disable()
    torch.cuda.synchronize()
    real = dist.all_reduce
    dist.all_reduce = lambda *a_, **k_: None
    c = timed(step, x, y, args.steps)
    dist.all_reduce = real
    print(f"(c) the same, all_reduce a no-op {c[0]:7.3f} ms/step, host enqueue {c[1]:7.3f} ms")
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(14)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
