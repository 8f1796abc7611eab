"""Per-layer micro-benchmark of the bf16-operand convolution kernels (fwd / dgrad / wgrad) at the RobustUNet config-2 shapes (16 x 256 x 256):
time, algorithmic TFLOP/s and the HBM-bound time (fp32 activations read + written once, at 6.3 TB/s)."""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

ops = importlib.import_module("eusipco-2026-robust-unet_amd.ops")
dev = torch.device("cuda:0")
N = int(os.environ.get("BENCH_N", 16))
S = int(os.environ.get("BENCH_S", 256))
LAYERS = [  # cin, cout, spatial divisor, k, dil
    (64, 64, 1, 3, 1), (128, 64, 1, 3, 1), (64, 128, 2, 3, 1), (128, 128, 2, 3, 1), (256, 128, 2, 3, 1),
    (128, 256, 4, 3, 1), (256, 256, 4, 3, 1), (512, 256, 4, 3, 1), (256, 512, 8, 3, 1), (512, 512, 8, 3, 1),
    (1024, 512, 8, 3, 1), (512, 256, 16, 3, 2), (1024, 1024, 16, 3, 1), (128, 64, 1, 1, 1), (64, 32, 1, 1, 1), (1024, 512, 8, 1, 1),
]


def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0, "hbm": 0.0}
with ops.precision("bf16"):
    for cin, cout, div, k, dil in LAYERS:
        h = S // div
        x = torch.randn((N, h, h, cin), device=dev)
        w = torch.randn((k, k, cin, cout), device=dev) * 0.05
        dy = torch.randn((N, h, h, cout), device=dev)
        flop = 2.0 * N * h * h * cin * cout * k * k
        hbm = 4.0 * N * h * h * (cin + cout) / 6.3e12 * 1e3
        tf = timeit(lambda: ops.conv_fwd(x, w, dil=dil))
        td = timeit(lambda: ops.conv_dgrad(dy, w, dil=dil))
        tw = timeit(lambda: ops.conv_wgrad(x, dy, k, k, dil=dil, on_side=False))
        tot["fwd"] += tf; tot["dgrad"] += td; tot["wgrad"] += tw; tot["hbm"] += hbm
        print(f"{cin:5d}->{cout:5d} @{h:4d} k{k} d{dil}: fwd {tf:7.3f} ms {flop/tf/1e9:7.1f} TF | dgrad {td:7.3f} ms {flop/td/1e9:7.1f} TF"
              f" | wgrad {tw:7.3f} ms {flop/tw/1e9:7.1f} TF | HBM-bound {hbm:6.3f} ms", flush=True)
print("sum ms:", {k: round(v, 3) for k, v in tot.items()})
