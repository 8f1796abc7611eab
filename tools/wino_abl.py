import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("eusipco-2026-robust-unet_amd.ops")
dev = torch.device("cuda:0")
for cin, cout, h in ((256, 256, 64), (64, 64, 256)):
    x = torch.randn((16, h, h, cin), device=dev); w = torch.randn((3, 3, cin, cout), device=dev) * 0.05
    U = ops.wino_weights(w)
    for _ in range(3): ops.wino_conv(x, U)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.wino_conv(x, U)
    e1.record(); torch.cuda.synchronize()
    print(os.environ.get("RUNET_WINO_ABL", "0"), cin, cout, h, "ms", e0.elapsed_time(e1) / 10)
