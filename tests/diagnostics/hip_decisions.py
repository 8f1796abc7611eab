#!/usr/bin/env python3
"""GPU diagnostic: which discrete decisions (ReLU masks, spatial-attention channel maxima) of the HIP train step differ from the fp32
oracle's on the same inputs, how close to a tie each one is, and the gradient agreement per parameter tensor - against the plain
oracle and against the oracle evaluated under the HIP step's own ReLU decisions (tests/decisions.py).

Usage (GPU box): python tests/diagnostics/hip_decisions.py [base n size seed ...]      (default: 64 2 64 5 11 12 13 14)
"""
import importlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(HERE))
import decisions as D  # noqa: E402

pkg = importlib.import_module(D.PKG_NAME)
oracle = importlib.import_module("oracle.robust_unet_ref")


def compare(base, n, size, seed):
    dev = torch.device("cuda:0")
    st = oracle.init_state(3, 1, base, seed=seed, perturb_bn=True)
    masks = oracle.dropout_masks(n, base, seed=seed)
    x, y = pkg.synthetic_batch(n, size, seed=seed)
    model = pkg.RobustUNet(3, 1, base)
    model.load_state_dict(st)
    model = model.to(dev).train()
    model.set_dropout_masks(masks)
    ctxs, prob, logit = D.hip_step(model, x, y, dev)
    g_plain, named, rp, rl = D.oracle_step(oracle, st, masks, x, y)
    flips = D.differing_decisions(ctxs, named, masks)
    print(f"base {base} n {n} size {size} seed {seed}: logit max err {float((logit - rl).abs().max()):.2e} (scale {float(rl.abs().max()):.1f}); "
          f"p==1 flips {int(((prob == 1) != (rp == 1)).sum())}; differing decisions: {len(flips)}")
    for f in flips[:20]:
        print("   %-13s %-15s at %s: oracle value / margin %.3e (tensor scale %.2e)" % f)
    g_forced, _, _, _ = D.oracle_step(oracle, st, masks, x, y, forced=D.forced_from_hip(ctxs, masks))
    for label, g in (("plain oracle", g_plain), ("oracle under the HIP step's ReLU decisions", g_forced)):
        rows = D.grad_errors(model, g)
        print("   gradient max err / scale vs %s: median %.2e; worst: %s" % (label, float(np.median([r[0] for r in rows])),
                                                                            ", ".join("%s %.1e" % (r[1], r[0]) for r in rows[:4])))


if __name__ == "__main__":
    torch.set_num_threads(16)
    a = [int(v) for v in sys.argv[1:]]
    base, n, size = a[:3] if len(a) >= 3 else (64, 2, 64)
    for seed in a[3:] or [5, 11, 12, 13, 14]:
        compare(base, n, size, seed)
