"""Convergence sanity of the full path (base 64, side streams): fit 8 fixed synthetic tiles for 150 steps, in fp32 (Winograd F(2x2) / F(4x4)),
with bf16 operands, and with fp16 operands + loss scaling; the three loss / IoU curves side by side.
usage: python tools/train_sanity.py [f32 bf16 fp16]"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("eusipco-2026-robust-unet_amd")
dev = torch.device("cuda:0")
modes = sys.argv[1:] or ["f32", "bf16", "fp16"]
x, y = pkg.synthetic_batch(8, 128, seed=7)
x, y = x.to(dev), y.to(dev)
ev = pkg.ModelEvaluator(dev)
curves = {}
for mode in modes:
    torch.manual_seed(0)
    model = pkg.RobustUNet(3, 1, 64).to(dev).train().set_precision(mode)
    step = pkg.TrainStep(model, lr=1e-3, weight_decay=1e-4, loss_scale=1024.0 if mode == "fp16" else None)
    rows = []
    for i in range(151):
        loss = step(x, y)
        if i % 25 == 0:
            model.eval()
            with torch.no_grad():
                iou = sum(m["iou"] for m in ev.batch_metrics(model(x), y)) / 8
            model.train()
            rows.append((i, float(loss.detach()), iou))
    curves[mode] = rows
    if mode == "fp16":
        print("fp16 loss scale / skipped steps:", step.adjust_loss_scale())
    assert torch.isfinite(loss) and float(loss.detach()) < 0.3, (mode, float(loss.detach()))
print("step  " + "  ".join(f"{m:>8s} loss   IoU " for m in modes))
for k in range(len(curves[modes[0]])):
    print(f"{curves[modes[0]][k][0]:4d}  " + "  ".join(f"{curves[m][k][1]:13.4f} {curves[m][k][2]:5.3f}" for m in modes))
