
"""Run one conv shape a few times (for rocprofv3 --pmc): conv_one.py cin cout spatial k mode[fwd|dgrad|wgrad] [iters]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
ops = importlib.import_module("eusipco-2026-robust-unet_amd.ops")
cin, cout, h, k = (int(v) for v in sys.argv[1:5])
mode = sys.argv[5]
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 3
dev = torch.device("cuda:0")
N = int(os.environ.get("BENCH_N", 16))
x = torch.randn((N, h, h, cin), device=dev)
w = torch.randn((k, k, cin, cout), device=dev) * 0.05
dy = torch.randn((N, h, h, cout), device=dev)
with ops.precision(os.environ.get("RUNET_PREC", "f32")):       # RUNET_PREC=bf16: the bf16-operand kernels
    for _ in range(iters):
        if mode == "fwd":
            ops.conv_fwd(x, w)
        elif mode == "dgrad":
            ops.conv_dgrad(dy, w)
        else:
            ops.conv_wgrad(x, dy, k, k, on_side=False)
torch.cuda.synchronize()
