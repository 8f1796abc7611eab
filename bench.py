"""bench.py - training images/sec of the Robust U-Net hot path on N MI355X (one process per GPU).

Workload (BASELINE.json metric / configs[1]): Robust U-Net (base 64, 40.9 M parameters), 256x256 RGB tiles, batch 16 PER GPU
(weak scaling: the per-GPU work is fixed as N grows), fp32, synthetic data resident in HBM; one step = zero_grad + forward + BCE +
backward (+ RCCL gradient all-reduce for N > 1, overlapped with backward) + fused Adam, Dropout2d and batch-stat BatchNorm on
(train mode).  W untimed warm-up steps, then exactly K timed steps between barrier + device synchronisation; the maximum over
ranks is reported by rank 0 as ONE JSON line.

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts its own N ranks: the parent (which never touches
the GPU) runs `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py <same args>` as a
child process and relays rank 0's line.  Launched under torch.distributed.run it is a rank.

For N > 1 the line also carries `strong` (the BASELINE metric's global batch of 16 split over the ranks, per-rank BatchNorm) and
`sync_bn` (the weak run with cross-rank BatchNorm statistics): SURVEY.md section 8(d).

`--config K` selects one of BASELINE.json's five configurations (1: 2x64x64; 2: 16x256x256 - the default; 3: 4x512x512 per GPU in
bf16; 4: DeepLabV3+ 16x256x256; 5: 1x1024x1024 per GPU, fp16 operands + loss scaling, hipGraph-captured step incl. the RCCL all-reduces for N > 1).

Extra objects on the line:
  roofline     - the dominant matrix-core kernel of the step (largest accumulated launch time).  `frac` = EXECUTED multiply-add FLOPs
                 per launch / average launch duration / dense MFMA peak of the arithmetic's type (fp32 157.3 TFLOP/s; bf16 2500; the
                 split-operand "x3" kernels do fp32 arithmetic with six bf16 MFMAs per product: `bf16_matrix_pipe` has that view), the
                 duration measured with HIP events on the launch stream inside the timed steps.  Winograd kernels execute 16/36
                 (F(2x2)) or 36/144 (F(4x4)) of the direct convolution's multiplies: `algorithmic` is the direct-conv rate (SURVEY
                 section 8d), `frac` the matrix-pipe utilisation of the work actually issued.  `step_mfma_frac` = executed FLOPs of
                 ALL matrix-core launches of a step / step time / peak.  `traffic` = HBM bytes per launch of that kernel from the
                 committed rocprofv3 PMC passes named in `traffic_source` (rocprofv3 cannot run inside bench.py); `hbm` = the
                 whole step's PMC byte count from the same file.
  cpu_baseline - the oracle (stock torch CPU ops, same train step, same batch) timed on this host's cores (rank 0, N = 1 only).
"""
import argparse
import importlib
import json
import os
import socket
import subprocess
import sys
import time

# More hardware queues than HIP's default of 4: main, weight-gradient, communication and RCCL-internal streams otherwise share
# queues and serialise on each other's event waits (measured with a single-rank process group: 380 img/s at the default, 414 with 8).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PKG = "eusipco-2026-robust-unet_amd"
PEAK_TFLOPS = {"f32": 157.3, "bf16": 2500.0, "fp16": 2500.0}   # MI355X_MICROARCH.md: dense matrix peaks (v_mfma_f32_32x32x2_f32 / v_mfma_f32_32x32x16_bf16)
HBM_PEAK_GBS = 8000.0                          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured with a float4 copy)


def traffic_file(model, dtype, batch, size):
    """Committed rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE) of THIS workload, or None: the file is keyed by model, operand type and
    per-GPU batch shape, so a line never quotes another configuration's bytes (tools/collect_profiles.sh writes them)."""
    name = f"profiles/round3_pmc_traffic_{model}_{dtype}_{batch}x{size}.json"
    return name if os.path.exists(os.path.join(ROOT, name)) else None


CONFIGS = {   # BASELINE.json `configs`, per-GPU shard
    1: dict(model="runet", batch=2, size=64, dtype="f32"),
    2: dict(model="runet", batch=16, size=256, dtype="f32"),
    3: dict(model="runet", batch=4, size=512, dtype="bf16"),
    4: dict(model="deeplab", batch=16, size=256, dtype="f32"),
    5: dict(model="runet", batch=1, size=1024, dtype="fp16", graph=True),
}


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


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", type=int, default=0, choices=[0, 1, 2, 3, 4, 5], help="BASELINE.json configuration (default: 2)")
    ap.add_argument("--batch", type=int, default=None, help="images per GPU")
    ap.add_argument("--global-batch", type=int, default=0, help="strong scaling: this many images split across the ranks (overrides --batch)")
    ap.add_argument("--size", type=int, default=None)
    ap.add_argument("--base", type=int, default=64)
    ap.add_argument("--dtype", choices=["f32", "bf16", "fp16"], default=None, help="operand type of the matrix-core kernels (accumulation, master weights, "
                    "BatchNorm statistics and the optimizer stay fp32)")
    ap.add_argument("--model", choices=["runet", "deeplab"], default=None)
    ap.add_argument("--graph", action="store_true", help="hipGraph-captured step (with N > 1 the RCCL gradient all-reduces are captured with it)")
    ap.add_argument("--loss-scale", type=float, default=1024.0, help="static loss scale of the fp16 mode (ignored otherwise)")
    ap.add_argument("--sync-bn", action="store_true", help="primary run with cross-rank BatchNorm statistics (results equal the global-batch step)")
    ap.add_argument("--no-extra-runs", action="store_true", help="N > 1: skip the `strong` and `sync_bn` sub-runs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--selftest-launch", action="store_true", help=argparse.SUPPRESS)   # CPU test of the self-launcher (gloo, no GPU)
    args = ap.parse_args(argv)
    cfg = CONFIGS[args.config or 2]
    for k in ("model", "batch", "size", "dtype"):
        if getattr(args, k) is None:
            setattr(args, k, cfg[k])
    if cfg.get("graph"):
        args.graph = True
    return args


# ------------------------------------------------------------------------------------------------ self-launch (parent never touches the GPU)
def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(nproc, argv):
    """Run `nproc` ranks of this script as children (torch.distributed.run); relay their stderr and rank 0's JSON line.  -> exit code"""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, usable_cores() // nproc)))
    log(f"starting {nproc} ranks: {' '.join(cmd[1:])}")
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for out in proc.stdout:
        s = out.strip()
        if s.startswith("{") and s.endswith("}"):
            line = s
        elif s:
            print(s, file=sys.stderr, flush=True)
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    elif rc == 0:
        rc = 1
        log("no JSON line came back from rank 0")
    return rc


# ------------------------------------------------------------------------------------------------ CPU baseline
def cpu_baseline(batch, size, seed):
    """Oracle train step on the host cores: one warm-up + at least three timed steps on a bounded sample (<= 8 images per step, ~25 s)."""
    import torch
    oracle = importlib.import_module("oracle.robust_unet_ref")
    data = importlib.import_module(PKG + ".data")
    cores = usable_cores()
    torch.set_num_threads(cores)
    # ~0.75 s per 256x256 image and step on 16 cores: half the benchmarked batch (8 images) keeps 1 warm-up + 3 timed steps near 25 s
    n = min(batch, 8) if batch * (size / 256.0) ** 2 <= 16 else max(1, int(8 / (size / 256.0) ** 2))
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

    log(f"cpu_baseline: oracle on {cores} host threads, {n} images per step, warm-up step ...")
    t0 = time.time()
    step()
    warm = time.time() - t0
    log(f"cpu_baseline: warm-up {warm:.1f} s; timing ...")
    times = []
    while len(times) < 3 or (sum(times) + warm < 20.0 and len(times) < 8):
        t0 = time.time()
        step()
        times.append(time.time() - t0)
    dt, steps = sum(times), len(times)
    return {"value": round(n * steps / dt, 3), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "spread": {"steps": steps, "images_per_s_min": round(n / max(times), 3), "images_per_s_max": round(n / min(times), 3)},
            "sample": f"{steps} train steps of {n} images {size}x{size} (fwd+BCE+bwd+Adam, fp32, torch CPU ops; the benchmarked batch is {batch}) after 1 warm-up step"}


# ------------------------------------------------------------------------------------------------ one rank
def selftest_rank():
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo")
    t = torch.ones(1) * (dist.get_rank() + 1)
    dist.all_reduce(t)
    if dist.get_rank() == 0:
        print(json.dumps({"selftest": True, "world": dist.get_world_size(), "sum": float(t.item())}), flush=True)
    dist.destroy_process_group()


def run_rank(args):
    # ONE line on stdout: RCCL prints a version banner to stdout when its communicator comes up, libraries may print more.  Everything but
    # rank 0's JSON line goes to stderr: fd 1 is pointed at stderr for the whole run and the line is written to the saved descriptor.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    strong = args.global_batch > 0
    if strong:
        if args.global_batch % world:
            raise SystemExit("--global-batch must be divisible by the number of ranks")
        args.batch = args.global_batch // world
    # rehearsal knob for a 1-GPU box (never used by the driver): all ranks on cuda:0, collectives over gloo
    one_device = os.environ.get("RUNET_BENCH_ONE_DEVICE") == "1"
    backend = os.environ.get("RUNET_BENCH_BACKEND", "gloo" if one_device else "nccl")
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
    comm = {"nccl": "RCCL", "gloo": "gloo (host-staged rehearsal, not RCCL)"}.get(backend, backend)

    pkg = importlib.import_module(PKG)
    ops = importlib.import_module(PKG + ".ops")
    torch.manual_seed(1234 + rank)             # dropout draws differ per rank
    if args.model == "deeplab":
        if use_dist:
            raise SystemExit("the DeepLabV3+ baseline (config 4) is single-process")
        model = pkg.DeepLabV3Plus(n_classes=1).to(dev).train()
    else:
        model = pkg.RobustUNet(3, 1, args.base).to(dev).train()
    if args.dtype != "f32":
        if not hasattr(model, "set_precision"):
            raise SystemExit(f"--dtype {args.dtype} is not available for --model {args.model}")
        model.set_precision(args.dtype)
    sync = None
    if use_dist:
        sync = pkg.GradAllReducer(model, sync_bn=False)
        sync.broadcast_parameters(0)
    graph = bool(args.graph)          # under a process group the captured step carries the RCCL all-reduces (trainer.TrainStep)
    step = pkg.TrainStep(model, lr=1e-4, weight_decay=1e-4, grad_sync=sync, graph=graph, loss_scale=args.loss_scale if args.dtype == "fp16" else None)

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    host_ms = [0.0]

    def timed_run(batch, sync_bn, profile, warmup, steps):
        """-> (seconds for `steps` steps: max over ranks, last loss, conv profile or None)"""
        if sync is not None:
            sync.set_sync_bn(sync_bn)
        x, y = pkg.synthetic_batch(batch, args.size, seed=1234 + rank)
        x, y = x.to(dev), y.to(dev)
        for _ in range(warmup + (3 if graph else 0)):      # a captured step needs its eager warm-up steps + the capture itself
            step(x, y)
        torch.cuda.synchronize()
        prof = ops.start_conv_profile() if profile else None
        barrier()
        t0 = time.perf_counter()
        per_step, ts = [], t0
        for _ in range(steps):
            loss = step(x, y)
            te = time.perf_counter()
            per_step.append(te - ts)
            ts = te
        # time the host needs to ENQUEUE a step (diagnostic: host- vs GPU-bound): the lower quartile of the per-step host times - once the
        # host is a queue's depth ahead of the GPU it blocks inside launches, so the mean would just repeat the GPU's step time
        host_ms[0] = 1e3 * sorted(per_step)[len(per_step) // 4]
        barrier()
        dt = time.perf_counter() - t0
        roof = ops.stop_conv_profile(prof) if prof is not None else None
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        if use_dist:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item()), float(loss.item()), roof

    if rank == 0:
        log(f"{args.model} on {dev}, {args.warmup} warm-up + {args.steps} timed steps of {args.batch} x {args.size}x{args.size} per GPU, "
            f"{args.dtype}" + (", hipGraph step" if graph else ""))
    want_roof = not args.no_roofline and args.model == "runet" and not graph
    # One rank: the per-launch HIP events of the roofline run inside the timed region (0.2 ms of a 33 ms step).  Under a process group they
    # cost the step 5 % (35.4 against 33.5 ms with one RCCL rank, tools/ddp_overhead.py: 33.4 without them), which would be charged to
    # every rank of a scaling run: there the timed region runs clean and the roofline comes from a pass of its own right after it.
    roof_in_region = want_roof and not use_dist
    dt, final_loss, roof = timed_run(args.batch, args.sync_bn, roof_in_region, args.warmup, args.steps)
    enqueue_ms = host_ms[0]
    if rank == 0:
        log(f"timed region {dt:.3f} s")
    roof_steps, roof_dt = args.steps, dt
    if want_roof and not roof_in_region:
        roof_steps = max(3, min(args.steps, 10))
        roof_dt, _, roof = timed_run(args.batch, args.sync_bn, True, 1, roof_steps)

    alone = None
    if roof is not None and ops.USE_WGRAD_STREAM:
        # outside the timed region: three more steps with the weight gradients back on the main stream, so that the dominant kernel's
        # rate can also be quoted without a concurrent kernel sharing the GPU with it
        ops.USE_WGRAD_STREAM = False
        _, _, p2 = timed_run(args.batch, args.sync_bn, True, 1, 3)
        alone = p2["by_kernel"].get(roof["kernel"])
        ops.USE_WGRAD_STREAM = True

    extra = {}
    if world > 1 and not args.no_extra_runs and not strong:
        # secondary measurements: a failure here must not cost the primary line (every rank takes the same branch: the exceptions these
        # calls can raise - allocation, launch errors - are raised on all ranks alike or abort the job)
        try:
            if 16 % world == 0:
                sdt, _, _ = timed_run(16 // world, False, False, args.warmup, args.steps)
                extra["strong"] = {"global_batch": 16, "images_per_gpu": 16 // world, "value": round(16 * args.steps / sdt, 2), "unit": "images/s",
                                   "ms_per_step": round(1e3 * sdt / args.steps, 3), "batchnorm": "per-rank statistics"}
            if not args.sync_bn:
                bdt, _, _ = timed_run(args.batch, True, False, args.warmup, args.steps)
                extra["sync_bn"] = {"global_batch": world * args.batch, "value": round(world * args.batch * args.steps / bdt, 2), "unit": "images/s",
                                    "ms_per_step": round(1e3 * bdt / args.steps, 3),
                                    "batchnorm": "cross-rank statistics (equals the single-process global-batch step)"}
        except RuntimeError as e:
            extra["extra_runs_error"] = str(e)[:300]

    if rank == 0:
        imgs = world * args.batch * args.steps
        peak = PEAK_TFLOPS[args.dtype]
        name = "Robust U-Net" if args.model == "runet" else "DeepLabV3+"
        out = {
            "metric": f"train images/sec {name} {args.size}x{args.size} bs{args.batch}/GPU", "value": round(imgs / dt, 2), "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3),
            "host_enqueue_ms_per_step": round(enqueue_ms, 3),
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"{name} " + (f"base{args.base} (40.9M params) " if args.model == "runet" else "") +
                                   f"train step, {args.size}x{args.size} RGB+mask tiles, batch {args.batch}/GPU, "
                                   + ("fp32" if args.dtype == "f32" else f"{args.dtype} matrix-core operands, fp32 accumulation / master weights / BatchNorm / Adam"
                                      + (f", loss scale {args.loss_scale:g}" if args.dtype == "fp16" else "")) +
                                   ", BCE + Adam(lr 1e-4, wd 1e-4), dropout + batch-stat BN on" + (", hipGraph-captured step" if graph else ""),
                       "baseline_config": args.config or 2, "global_batch": world * args.batch, "image_size": args.size,
                       "parallelism": f"dp{world}" + (f" ({comm} gradient all-reduce overlapped with backward, "
                                                      f"{'cross-rank' if args.sync_bn else 'per-rank'} BatchNorm statistics)" if use_dist else "")},
            "final_loss": round(final_loss, 5),
        }
        out.update(extra)
        if roof is not None:
            dom = roof["by_kernel"][roof["kernel"]]        # [launches, ms, algorithmic TFLOP/s, executed TFLOP/s]
            traffic, hbm, src = None, None, traffic_file(args.model, args.dtype, args.batch, args.size)
            try:   # HBM bytes from the committed PMC passes of this workload
                if src is not None:
                    with open(os.path.join(ROOT, src)) as f:
                        pmc = json.load(f)
                    for kname, rec in pmc["kernels"].items():      # rocprof prints template arguments, the live name may not
                        if kname == roof["kernel"] or kname.startswith(roof["kernel"] + "<"):
                            traffic = rec.get("hbm_bytes_per_launch_corrected")
                    hbm = pmc.get("step")
            except (OSError, ValueError, KeyError):
                src = None
            exec_frac = dom[3] / peak
            out["roofline"] = {"bound": "mfma", "achieved": dom[3], "peak": peak, "unit": "TFLOP/s", "frac": round(exec_frac, 4),
                               "algorithmic": dom[2], "algorithmic_frac": round(dom[2] / peak, 4),
                               "step_mfma_frac": round(roof["exec_flops_total"] / roof_dt / 1e12 / peak, 4),
                               "measured_over": "the timed region" if roof_in_region else f"a pass of {roof_steps} steps right after the timed region "
                                                "(under a process group the per-launch events cost the step 5 %: the timed region runs without them)",
                               "traffic": traffic, "traffic_source": (src + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of an earlier run of this command, "
                                                                      "not measured in this run)") if src else None,
                               "kernel": roof["kernel"], "launches_per_step": roof["launches"] // roof_steps,
                               "avg_launch_us": round(roof["avg_us"], 2), "share_of_step": round(roof["time_s"] / roof_dt, 3),
                               "by_kernel": roof["by_kernel"],
                               "by_kernel_columns": ["launches", "ms", "algorithmic TFLOP/s (direct-conv FLOPs)", "executed TFLOP/s"],
                               "note": "achieved / frac = multiply-adds the kernel actually issues on the matrix pipe (Winograd F(2x2): 16/36, "
                                       "F(4x4) position-GEMMs: 36/144 of the direct convolution) / HIP-event launch time; algorithmic = direct-conv FLOPs / the same time",
                               "concurrency": "weight-gradient kernels run on a second HIP stream during backward: launch durations (live events and rocprofv3 "
                                              "alike) include time shared with them; `standalone` repeats the measurement single-stream after the timed region"}
            if "x3" in roof["kernel"]:
                # split-operand kernels: the fp32 multiply-adds above are issued as six bf16 MFMAs each (x = h + m + l, six of nine cross
                # products) - the occupancy of the pipe they actually run on is 6 x that rate against the dense bf16 peak
                out["roofline"]["bf16_matrix_pipe"] = {"achieved": round(6.0 * dom[3], 1), "peak": PEAK_TFLOPS["bf16"], "unit": "TFLOP/s",
                                                       "frac": round(6.0 * dom[3] / PEAK_TFLOPS["bf16"], 4),
                                                       "note": "fp32-accurate arithmetic on the bf16 matrix cores: `achieved` / `frac` above count the fp32 "
                                                               "multiply-adds against the fp32 MFMA peak (the arithmetic's own roofline), this block counts "
                                                               "the bf16 MFMAs issued (6 per fp32 product) against the dense bf16 peak"}
            if hbm is not None:
                out["roofline"]["hbm"] = hbm
            if alone is not None:      # [launches, total ms, algorithmic, executed] of the same kernel in three single-stream steps after the timed region
                out["roofline"]["standalone"] = {"achieved": alone[3], "frac": round(alone[3] / peak, 4), "algorithmic": alone[2],
                                                 "avg_launch_us": round(1e3 * alone[1] / alone[0], 2)}
            if args.dtype != "f32" and roof.get("bytes_per_s"):
                # the bf16-operand kernels are HBM-bound by design (16x the fp32 matrix rate): their roofline is the memory one.
                # achieved = ALGORITHMIC bytes (fp32 activations read once + written once + the bf16 weights) / HIP-event launch time
                r = out["roofline"]
                r["mfma"] = {"achieved": r["achieved"], "peak": r["peak"], "unit": r["unit"], "frac": r["frac"]}
                r.update({"bound": "hbm", "achieved": round(roof["bytes_per_s"] / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                          "frac": round(roof["bytes_per_s"] / 1e9 / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_launch": int(roof["bytes_per_launch"]),
                          "note": "achieved = algorithmic HBM bytes per launch (fp32 activations read + written once, bf16 weights once) / HIP-event "
                                  "launch time against the 8 TB/s HBM3E peak (6.3 TB/s measured-achievable); `mfma` holds the matrix-pipe view"})
                r["by_kernel_columns"] = r["by_kernel_columns"] + ["algorithmic GB/s"]
                if alone is not None and len(alone) > 4:
                    r["standalone"] = {"achieved": alone[4], "frac": round(alone[4] / HBM_PEAK_GBS, 4), "avg_launch_us": round(1e3 * alone[1] / alone[0], 2)}
        if world == 1 and not args.no_cpu_baseline and args.model == "runet":
            out["cpu_baseline"] = cpu_baseline(args.batch, args.size, 1234)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if use_dist:
        dist.destroy_process_group()


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    if args.selftest_launch:
        return selftest_rank()
    run_rank(args)


if __name__ == "__main__":
    main()
