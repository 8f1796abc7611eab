
"""Convergence sanity of the full path (base 64, side streams, Winograd F(2x2)/F(4x4)): fit 8 fixed synthetic tiles for 150 steps."""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("eusipco-2026-robust-unet_amd")
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = pkg.RobustUNet(3, 1, 64).to(dev).train()
step = pkg.TrainStep(model, lr=1e-3, weight_decay=1e-4)
x, y = pkg.synthetic_batch(8, 128, seed=7)
x, y = x.to(dev), y.to(dev)
ev = pkg.ModelEvaluator(dev)
for i in range(151):
    loss = step(x, y)
    if i % 25 == 0:
        model.eval()
        with torch.no_grad():
            iou = sum(m["iou"] for m in ev.batch_metrics(model(x), y)) / 8
        model.train()
        print(f"step {i:4d}  loss {float(loss.detach()):.4f}  eval IoU on the training tiles {iou:.3f}", flush=True)
assert torch.isfinite(loss) and float(loss.detach()) < 0.3, float(loss.detach())
