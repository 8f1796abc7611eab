"""Timing of the fused F(2x2) x3 kernel on the step's four shapes, for the ablation knobs (RUNET_WINO_X3_ABL)."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
ops = importlib.import_module("eusipco-2026-robust-unet_amd.ops")
dev = torch.device("cuda:0")
for cin, cout, h in ((64, 64, 256), (128, 64, 256), (128, 128, 128), (64, 128, 128)):
    x = torch.randn((16, h, h, cin), device=dev)
    w = torch.randn((3, 3, cin, cout), device=dev) * 0.05
    U = ops.wino_weights(w)
    for _ in range(3):
        ops.wino_conv(x, U)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.wino_conv(x, U)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 10
    print(f"{cin:4d}->{cout:4d} @{h}: {t * 1e3:7.1f} us  {2.0 * 16 * h * h * 9 * cin * cout / t / 1e9:6.1f} TF (direct-conv FLOPs)", flush=True)
