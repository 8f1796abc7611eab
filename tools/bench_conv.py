
"""Per-layer micro-benchmark of the implicit-GEMM conv kernels (fwd / dgrad / wgrad) at the
RobustUNet config-2 shapes (16 x 256 x 256).  Prints TFLOP/s against the 157.3 TF fp32-MFMA peak."""
import importlib
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

ops = importlib.import_module("eusipco-2026-robust-unet_amd.ops")
dev = torch.device("cuda:0")
N = int(os.environ.get("BENCH_N", 16))
S = int(os.environ.get("BENCH_S", 256))
LAYERS = [  # cin, cout, spatial divisor, k, dil
    (64, 64, 1, 3, 1), (128, 64, 1, 3, 1), (64, 128, 2, 3, 1), (128, 128, 2, 3, 1), (256, 128, 2, 3, 1),
    (128, 256, 4, 3, 1), (256, 256, 4, 3, 1), (512, 256, 4, 3, 1), (256, 512, 8, 3, 1), (512, 512, 8, 3, 1),
    (1024, 512, 8, 3, 1), (512, 256, 16, 3, 2), (1024, 1024, 16, 3, 1), (128, 64, 1, 1, 1), (1024, 512, 8, 1, 1),
]


def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
for cin, cout, div, k, dil in LAYERS:
    h = S // div
    x = torch.randn((N, h, h, cin), device=dev)
    w = torch.randn((k, k, cin, cout), device=dev) * 0.05
    dy = torch.randn((N, h, h, cout), device=dev)
    flop = 2.0 * N * h * h * cin * cout * k * k
    tf = timeit(lambda: ops.conv_fwd(x, w, dil=dil))
    td = timeit(lambda: ops.conv_dgrad(dy, w, dil=dil))
    tw = timeit(lambda: ops.conv_wgrad(x, dy, k, k, dil=dil))
    tot["fwd"] += tf; tot["dgrad"] += td; tot["wgrad"] += tw
    wino = ""
    if k == 3 and dil == 1 and ops.wino_ok(h, h, cin, cout):
        U = ops.wino_weights(w); Ud = ops.wino_weights(w, dgrad=True)
        twf = timeit(lambda: ops.wino_conv(x, U)); twd = timeit(lambda: ops.wino_conv(dy, Ud))
        tot["wino_fwd"] = tot.get("wino_fwd", 0.0) + twf; tot["wino_dgrad"] = tot.get("wino_dgrad", 0.0) + twd
        wino = f" | wino fwd {twf:7.3f} ms {flop/twf/1e9:6.1f} TF dgrad {twd:7.3f} ms {flop/twd/1e9:6.1f} TF"
    if k == 3 and dil == 1 and ops.wino4_ok(h, h, cin, cout) and cin >= 128:
        U4 = ops.wino4_weights(w); U4d = ops.wino4_weights(w, dgrad=True)
        t4f = timeit(lambda: ops.wino4_conv(x, U4)); t4d = timeit(lambda: ops.wino4_conv(dy, U4d)); t4w = timeit(lambda: ops.wino4_wgrad(x, dy))
        t4u = timeit(lambda: ops.wino4_weights(w))
        for kk, vv in (("w4_fwd", t4f), ("w4_dgrad", t4d), ("w4_wgrad", t4w), ("w4_weights", t4u)):
            tot[kk] = tot.get(kk, 0.0) + vv
        wino += f" | F4 fwd {t4f:6.3f} dgrad {t4d:6.3f} wgrad {t4w:6.3f} U {t4u:6.3f} ms"
    print(f"{cin:5d}->{cout:5d} @{h:4d} k{k} d{dil}: fwd {tf:7.3f} ms {flop/tf/1e9:6.1f} TF | dgrad {td:7.3f} ms {flop/td/1e9:6.1f} TF"
          f" | wgrad {tw:7.3f} ms {flop/tw/1e9:6.1f} TF" + wino, flush=True)
print("sum ms:", tot)
