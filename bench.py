#!/usr/bin/env python3
"""bench.py - training images/sec of the Robust U-Net hot path on N MI355X (one process per GPU).

Workload (BASELINE.json metric / configs[1]): Robust U-Net (base 64, 40.9 M parameters), 256x256 RGB tiles,
batch 16 PER GPU (weak scaling), fp32, synthetic data resident in HBM; one step = zero_grad + forward + BCE +
backward (+ RCCL gradient all-reduce for N > 1, overlapped with backward) + fused Adam, Dropout2d and batch-stat
BatchNorm on (train mode).  W untimed warm-up steps, then exactly K timed steps between barrier + device
synchronisation; the maximum over ranks is reported by rank 0 as one JSON line.

Extra objects on that line:
  roofline     - the dominant convolution kernel (by total time: the fused Winograd F(2x2,3x3) kernel `wino_conv_kernel`,
                 forward + data-gradient launches): ALGORITHMIC FLOPs per launch (the direct convolution's 2*9*Cin*Cout per
                 pixel) / average launch duration measured with HIP events on the launch stream during
                 the timed steps, against the 157.3 TFLOP/s dense fp32 matrix peak of gfx950.  Winograd executes 16/36 of
                 those multiplies, so `frac` can exceed 1; `mfma_frac` is the matrix-pipe utilisation of the work actually
                 executed.  `by_kernel` lists every convolution kernel family [launches, ms, algorithmic TFLOP/s].
  cpu_baseline - the oracle (stock torch CPU ops, same train step) timed on this host's cores on a bounded sample
                 (rank 0, N = 1 only).
"""
import argparse
import importlib
import json
import os

# More hardware queues than HIP's default of 4: main, weight-gradient, communication and RCCL-internal streams otherwise share
# queues and serialise on each other's event waits (measured with a single-rank process group: 380 img/s at the default, 414 with 8).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PKG = "eusipco-2026-robust-unet_amd"
FP32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: dense fp32 matrix peak (v_mfma_f32_32x32x2_f32)


def usable_cores():
    """Cores this process may actually use: scheduler affinity, capped by the cgroup CPU quota when there is one."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 64))


def log(msg):
    print(f"[bench] {msg}", file=sys.stderr, flush=True)


def cpu_baseline(size, seed):
    """Oracle train step on the host cores; bounded sample (about 10-30 s)."""
    oracle = importlib.import_module("oracle.robust_unet_ref")
    data = importlib.import_module(PKG + ".data")
    cores = usable_cores()
    torch.set_num_threads(cores)
    n = 4
    net = oracle.OracleNet(3, 1, 64, seed=seed).train()
    opt = torch.optim.Adam(net.parameters(), lr=1e-4, weight_decay=1e-4)
    x, y = data.synthetic_batch(n, size, seed=seed)
    masks = oracle.dropout_masks(n, 64, seed=seed)

    def step():
        opt.zero_grad()
        loss = oracle.bce_mean(net(x, masks), y)
        loss.backward()
        opt.step()
        return loss.item()

    log(f"cpu_baseline: oracle on {cores} host threads, warm-up step ...")
    step()                                   # warm-up
    log("cpu_baseline: timing ...")
    t0 = time.time()
    steps = 0
    while steps < 2 or (time.time() - t0 < 10.0 and steps < 8):
        step()
        steps += 1
    dt = time.time() - t0
    return {"value": round(n * steps / dt, 3), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{steps} train steps of {n} images {size}x{size} (fwd+BCE+bwd+Adam, fp32, torch CPU ops) after 1 warm-up"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=16, help="images per GPU")
    ap.add_argument("--global-batch", type=int, default=0, help="strong scaling: this many images split across the ranks (overrides --batch)")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--base", type=int, default=64)
    ap.add_argument("--sync-bn", action="store_true", help="cross-rank BatchNorm statistics (results equal the global-batch step)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    strong = args.global_batch > 0
    if strong:
        if args.global_batch % world:
            raise SystemExit("--global-batch must be divisible by the number of ranks")
        args.batch = args.global_batch // world
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    # rehearsal knobs for a 1-GPU box (never used by the driver): all ranks on cuda:0, collectives over gloo
    one_device = os.environ.get("RUNET_BENCH_ONE_DEVICE") == "1"
    backend = os.environ.get("RUNET_BENCH_BACKEND", "nccl")
    dev_index = 0 if one_device else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    # RUNET_BENCH_FORCE_DIST=1: exercise the RCCL path (process group, bucketed all-reduce on the side stream) with a single rank
    use_dist = world > 1 or os.environ.get("RUNET_BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    pkg = importlib.import_module(PKG)
    ops = importlib.import_module(PKG + ".ops")
    torch.manual_seed(1234 + rank)             # dropout draws differ per rank
    model = pkg.RobustUNet(3, 1, args.base).to(dev).train()
    sync = None
    if use_dist:
        sync = pkg.GradAllReducer(model, sync_bn=args.sync_bn)
        sync.broadcast_parameters(0)
    if os.environ.get("RUNET_BENCH_ARENA") == "1":      # diagnostic: gradient arena without a process group
        model.grad_arena()
    step = pkg.TrainStep(model, lr=1e-4, weight_decay=1e-4, grad_sync=sync)
    x, y = pkg.synthetic_batch(args.batch, args.size, seed=1234 + rank)
    x, y = x.to(dev), y.to(dev)

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    if rank == 0:
        log(f"model on {dev}, {args.warmup} warm-up + {args.steps} timed steps of {args.batch} x {args.size}x{args.size} per GPU")
    for _ in range(args.warmup):
        step(x, y)
    torch.cuda.synchronize()
    if rank == 0:
        log("warm-up done")
    prof = None if args.no_roofline else ops.start_conv_profile()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step(x, y)
    barrier()
    dt = time.perf_counter() - t0
    roof = None if prof is None else ops.stop_conv_profile(prof)
    alone = None
    if roof is not None and ops.USE_WGRAD_STREAM:
        # outside the timed region: three more steps with the weight gradients back on the main stream, so that the dominant kernel's
        # rate can also be quoted without a concurrent kernel sharing the GPU with it
        ops.USE_WGRAD_STREAM = False
        step(x, y)
        p2 = ops.start_conv_profile()
        for _ in range(3):
            step(x, y)
        torch.cuda.synchronize()
        alone = ops.stop_conv_profile(p2)["by_kernel"].get(roof["kernel"])
        ops.USE_WGRAD_STREAM = True
    t = torch.tensor([dt], device=dev, dtype=torch.float64)
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    final_loss = float(loss.item())
    if rank == 0:
        log(f"timed region {dt:.3f} s")

    if rank == 0:
        imgs = world * args.batch * args.steps
        out = {
            "metric": "train images/sec Robust U-Net 256x256 bs16", "value": round(imgs / dt, 2), "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3),
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"Robust U-Net base{args.base} (40.9M params) train step, {args.size}x{args.size} RGB+mask tiles, "
                                   f"batch {args.batch}/GPU, fp32, BCE + Adam(lr 1e-4, wd 1e-4), dropout + batch-stat BN on",
                       "global_batch": world * args.batch, "image_size": args.size,
                       "parallelism": f"dp{world}" + (f" (RCCL grad all-reduce overlapped with backward, {'SyncBN' if args.sync_bn else 'per-rank BN'})" if world > 1 else "")},
            "final_loss": round(final_loss, 5),
        }
        if roof is not None:
            traffic = None
            try:   # HBM bytes per launch of the dominant kernel from the committed PMC passes (rocprofv3 cannot run inside bench.py)
                with open(os.path.join(ROOT, "profiles", "round1_pmc_traffic.json")) as f:
                    for name, rec in json.load(f)["kernels"].items():      # rocprof prints template arguments, the live name may not
                        if name == roof["kernel"] or name.startswith(roof["kernel"] + "<"):
                            traffic = rec.get("hbm_bytes_per_launch_corrected")
            except (OSError, ValueError, KeyError):
                pass
            out["roofline"] = {"bound": "mfma", "achieved": round(roof["tflops"], 2), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                               "frac": round(roof["tflops"] / FP32_MFMA_PEAK_TFLOPS, 4), "traffic": traffic,
                               "kernel": roof["kernel"], "launches_per_step": roof["launches"] // args.steps,
                               "avg_launch_us": round(roof["avg_us"], 2), "share_of_step": round(roof["time_s"] / dt, 3),
                               "by_kernel": roof["by_kernel"]}
            out["roofline"]["concurrency"] = ("weight-gradient kernels run on a second HIP stream during backward: launch durations (live events and rocprofv3 "
                                              "alike) include time shared with them, so per-kernel rates read lower than standalone (tools/bench_conv.py, "
                                              "tools/bench_gemm.py) while the step is faster (RUNET_NO_WGRAD_STREAM=1: 393 img/s, all rates standalone)")
            if alone is not None:      # [launches, total ms, TFLOP/s] of the same kernel in three single-stream steps after the timed region
                out["roofline"]["standalone"] = {"achieved": alone[2], "frac": round(alone[2] / FP32_MFMA_PEAK_TFLOPS, 4),
                                                 "avg_launch_us": round(1e3 * alone[1] / alone[0], 2)}
            if roof["kernel"].startswith("gemm_"):
                out["roofline"]["note"] = ("position-GEMMs of the unfused Winograd F(4x4,3x3) path (deep 3x3 convolutions): achieved = the GEMM's own "
                                           "2*36*tiles*K*N FLOPs / its launch time; the convolution it implements is 4x that in direct-conv FLOPs")
            if roof["kernel"].startswith("wino"):
                out["roofline"]["mfma_work_fraction"] = round(16.0 / 36.0, 4)
                out["roofline"]["mfma_frac"] = round(roof["tflops"] * 16.0 / 36.0 / FP32_MFMA_PEAK_TFLOPS, 4)
                out["roofline"]["note"] = "achieved = algorithmic (direct-conv) FLOP rate; Winograd F(2x2,3x3) runs 16/36 of the multiplies on the fp32 MFMA pipe"
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.size, 1234)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
