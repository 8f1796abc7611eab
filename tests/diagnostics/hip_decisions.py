#!/usr/bin/env python3
"""GPU diagnostic: which discrete decisions (ReLU masks, pool winners, attention maxima) of the HIP train step differ from the fp32
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
    blocks = importlib.import_module(D.PKG_NAME + ".blocks")
    real_cs, sums = blocks.chan_sum, []

    def cs_spy(t, out):          # every bias gradient of the step: fp64 re-summation of the SAME device tensor the kernel summed
        sums.append((tuple(t.shape), t.detach().double().sum((0, 1, 2)).cpu(), t.detach().double().abs().sum((0, 1, 2)).cpu(), out))
        return real_cs(t, out)

    blocks.chan_sum = cs_spy
    try:
        dec, prob, logit = D.hip_step(model, x, y, dev)
    finally:
        blocks.chan_sum = real_cs
    for shape, s64, sabs, out in sums:
        if shape[3] >= 32 and shape[1] >= 64:
            print("   chan_sum over %s: kernel vs fp64 of its own input: max |err| / sum|terms| = %.2e" % (shape, float(((out.cpu().double() - s64).abs() / sabs).max())))
    g_plain, named, rp, rl = D.oracle_step(oracle, st, masks, x, y)
    flips = D.differing_decisions(dec, named, masks)
    print(f"base {base} n {n} size {size} seed {seed}: logit max err {float((logit - rl).abs().max()):.2e} (scale {float(rl.abs().max()):.1f}); "
          f"p==1 flips {int(((prob == 1) != (rp == 1)).sum())}; differing decisions: {len(flips)}")
    for f in flips[:20]:
        print("   %-13s %-15s at %s: margin %.3e (tensor scale %.2e)" % f)
    g_forced, named_f, _, _ = D.oracle_step(oracle, st, masks, x, y, forced=D.forced_from_hip(dec, masks))
    for k, (ref64, sabs) in named_f["up_bias"].items():
        hip = dict(model.named_parameters())[k].grad.cpu().double()
        print("   %s: |hip - fp64 resum of oracle terms| / sum|terms| = %.2e;  |oracle fp32 - same| / sum|terms| = %.2e;  |sum| / sum|terms| = %.2e" % (
            k, float(((hip - ref64).abs() / sabs).max()), float(((g_forced[k].double() - ref64).abs() / sabs).max()), float((ref64.abs() / sabs).max())))
    for label, g, ub in (("plain oracle", g_plain, named["up_bias"]), ("oracle under the HIP step's decisions", g_forced, named_f["up_bias"])):
        rows = D.grad_errors(model, g, ub)
        print("   gradient max err / scale vs %s: median %.2e; worst: %s" % (label, float(np.median([r[0] for r in rows])),
                                                                            ", ".join("%s %.1e" % (r[1], r[0]) for r in rows[:4])))


if __name__ == "__main__":
    torch.set_num_threads(16)
    a = [int(v) for v in sys.argv[1:]]
    base, n, size = a[:3] if len(a) >= 3 else (64, 2, 64)
    for seed in a[3:] or [5, 11, 12, 13, 14]:
        compare(base, n, size, seed)
