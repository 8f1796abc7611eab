"""How far the host runs ahead of the GPU in the train step, and whether the caching allocator goes to the driver inside the steady state:
per step, host time to enqueue (perf_counter around step()), device allocations / frees (torch.cuda.memory_stats), and the GPU step time.
usage: python tools/host_lead.py [batch size steps [f32|bf16|fp16 [runet|deeplab|unet]]]"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pkg = importlib.import_module("eusipco-2026-robust-unet_amd")
trainer = importlib.import_module("eusipco-2026-robust-unet_amd.trainer")
n, size, steps = (int(a) for a in (sys.argv[1:4] if len(sys.argv) >= 4 else (16, 256, 12)))
prec = sys.argv[4] if len(sys.argv) > 4 else "f32"
which = sys.argv[5] if len(sys.argv) > 5 else "runet"
dev = torch.device("cuda:0")
m = {"runet": lambda: pkg.RobustUNet(3, 1, 64), "deeplab": lambda: pkg.DeepLabV3Plus(n_classes=1), "unet": lambda: pkg.UNet(3, 2)}[which]().to(dev).train()
if prec != "f32":
    m.set_precision(prec)
step = trainer.TrainStep(m, lr=1e-3, weight_decay=1e-4, loss_scale=1024.0 if prec == "fp16" else None)
x, y = pkg.synthetic_batch(n, size, seed=1)
if which == "unet":
    y = (y[:, 0] > 0.5).long()
x, y = x.to(dev), y.to(dev)
for _ in range(5):
    step(x, y)
torch.cuda.synchronize()
keys = ("num_device_alloc", "num_device_free", "num_alloc_retries", "num_sync_all_streams")
s0 = torch.cuda.memory_stats()
t_host, allocs = [], []
ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
ev[0].record()
for i in range(steps):
    t0 = time.perf_counter()
    step(x, y)
    t_host.append(time.perf_counter() - t0)
    allocs.append(torch.cuda.memory_stats().get("num_device_alloc", 0))
    ev[i + 1].record()
torch.cuda.synchronize()
s1 = torch.cuda.memory_stats()
print("host ms per step:", " ".join(f"{1e3 * t:.2f}" for t in t_host))
print("device allocations per step:", " ".join(str(b - a) for a, b in zip([s0.get("num_device_alloc", 0)] + allocs, allocs)))
print("gpu  ms per step:", " ".join(f"{ev[i].elapsed_time(ev[i + 1]):.2f}" for i in range(steps)))
print({k: (s0.get(k), s1.get(k)) for k in keys})
print("reserved GB", torch.cuda.memory_reserved() / 1e9, "allocated peak GB", torch.cuda.max_memory_allocated() / 1e9)
