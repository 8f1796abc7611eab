#!/usr/bin/env python3
"""Gradient-parity report of the full train step against the reference goldens (tests/golden/model_*.npz): per parameter tensor,
max |err| / scale and the fraction of elements above 5e-3 of the scale.  Run with RUNET_NO_WINOGRAD4=1 / RUNET_NO_WINOGRAD=1 to
compare the convolution algorithms' rounding behaviour."""
import importlib
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("eusipco-2026-robust-unet_amd")
oracle = importlib.import_module("oracle.robust_unet_ref")
GOLD = os.path.join(ROOT, "tests", "golden")


def main(tag):
    meta = json.load(open(os.path.join(GOLD, f"model_{tag}.json")))
    gold = dict(np.load(os.path.join(GOLD, f"model_{tag}.npz")))
    dev = torch.device("cuda:0")
    base, n, size, seed = meta["base"], meta["n"], meta["size"], meta["seed"]
    model = pkg.RobustUNet(3, 1, base)
    model.load_state_dict(oracle.init_state(3, 1, base, seed=seed, perturb_bn=True))
    model = model.to(dev).train()
    model.set_dropout_masks(oracle.dropout_masks(n, base, seed=seed))
    x, y = pkg.synthetic_batch(n, size, seed=seed)
    prob, logit = model(x.to(dev), return_logits=True)
    pkg.bce_loss(prob, y.to(dev)).backward()
    print(tag, "prob max err", float(np.abs(prob.detach().cpu().numpy() - gold["prob"]).max()),
          "logit max err", float(np.abs(logit.detach().cpu().numpy() - gold["logit"]).max()), "logit scale", float(np.abs(gold["logit"]).max()))
    rows = []
    for k, p in model.named_parameters():
        if f"grad/{k}" not in gold:
            continue
        gref = gold[f"grad/{k}"]
        scale = float(np.abs(gref).max())
        if scale < 1e-6:
            continue
        err = np.abs(p.grad.cpu().numpy() - gref)
        rows.append((float(err.max() / scale), float((err > 5e-3 * scale).mean()), k, err.size))
    rows.sort(reverse=True)
    for r in rows[:12]:
        print(f"  {r[2]:40s} max err/scale {r[0]:.2e}  frac>5e-3 {r[1]:.3f}  n={r[3]}")
    print("  median max err/scale", float(np.median([r[0] for r in rows])))


if __name__ == "__main__":
    for tag in ("b16_n2_s64", "b64_n2_s64"):
        main(tag)
