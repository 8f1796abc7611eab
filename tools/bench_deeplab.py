
"""DeepLabV3+ baseline (BASELINE.json config 4) train-step throughput on one MI355X: 256x256, batch 16, fp32, BCE + Adam.
Not a bench line of the contract (bench.py measures the Robust U-Net metric); the figure is quoted in DESIGN.md."""
import importlib
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("eusipco-2026-robust-unet_amd")


def main():
    n, size, steps, warm = 16, 256, 30, 5
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = pkg.DeepLabV3Plus().to(dev).train()
    opt = pkg.FusedAdam(model.parameters(), lr=1e-4, weight_decay=1e-4)
    x, y = pkg.synthetic_batch(n, size, seed=1234)
    x, y = x.to(dev), y.to(dev)

    def step():
        opt.zero_grad()
        loss = pkg.bce_loss(model(x), y)
        loss.backward()
        opt.step()
        return loss

    for _ in range(warm):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"model": "DeepLabV3+ baseline", "images_per_s": round(n * steps / dt, 1), "ms_per_step": round(1e3 * dt / steps, 3),
                      "batch": n, "size": size, "final_loss": round(float(loss.item()), 5)}))


if __name__ == "__main__":
    main()
