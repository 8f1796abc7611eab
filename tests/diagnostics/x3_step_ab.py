"""A/B of one HIP train step with the F(4x4) position GEMMs on the f32 MFMA and on the split-operand bf16 path (ops.USE_X3), against each
other and against the decision-forced oracle: per-tensor gradient errors (top rows), decisions that differ between the two HIP runs.
usage: python tests/diagnostics/x3_step_ab.py [n size seed]"""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

import decisions as D

pkg = importlib.import_module(D.PKG_NAME)
ops = importlib.import_module(D.PKG_NAME + ".ops")
oracle = importlib.import_module("oracle.robust_unet_ref")
n, size, seed = (int(a) for a in (sys.argv[1:4] if len(sys.argv) >= 4 else (1, 1024, 45)))
torch.set_num_threads(16)
dev = torch.device("cuda:0")
x, y = pkg.synthetic_batch(n, size, seed=seed)
st = oracle.init_state(3, 1, 64, seed=seed, perturb_bn=True)
masks = oracle.dropout_masks(n, 64, seed=seed)
runs = {}
for use in (False, True):
    ops.USE_X3 = use
    model = pkg.RobustUNet(3, 1, 64)
    model.load_state_dict(st)
    model = model.to(dev).train()
    model.set_dropout_masks(masks)
    dec, prob, logit = D.hip_step(model, x, y, dev)
    runs[use] = (dec, prob, logit, {k: p.grad.detach().cpu().clone() for k, p in model.named_parameters()}, model)
g0, g1 = runs[False][3], runs[True][3]
rows = sorted(((float((g1[k] - g0[k]).abs().max()) / max(float(g0[k].abs().max()), 1e-30), k) for k in g0), reverse=True)
print("x3 vs f32-MFMA HIP gradients, max |d| / scale, top 20:")
for r in rows[:20]:
    print(f"   {r[0]:.3e} {r[1]}")
print("prob max diff", float((runs[True][1] - runs[False][1]).abs().max()))
d0, d1 = runs[False][0], runs[True][0]
for k in D.RB:
    a = int(((d0["act"][k]["a1"] > 0) != (d1["act"][k]["a1"] > 0)).sum())
    o = int(((d0["act"][k]["out"] > 0) != (d1["act"][k]["out"] > 0)).sum())
    sa = int((d0["sa"][k] != d1["sa"][k]).sum())
    ca = int((d0["ca"][k] != d1["ca"][k]).sum())
    print(f"   {k}: relu(bn1) flips {a}, relu(out) flips {o}, sa max flips {sa}, ca max flips {ca}")
for k in D.RB + ("bottleneck.1",):
    for what in ("a1", "out"):
        if what in d0["act"][k]:
            a, b = d0["act"][k][what], d1["act"][k][what]
            print(f"   {k}.{what}: x3 vs f32 max |d| / scale {float((a - b).abs().max() / a.abs().max()):.3e}  rms {float((a - b).pow(2).mean().sqrt() / a.pow(2).mean().sqrt()):.3e}")
for lvl in (1, 2, 3, 4):
    print(f"   pool{lvl} flips {int((d0['pool'][lvl] != d1['pool'][lvl]).sum())}")
for use in (False, True):
    dec, prob, logit, g, model = runs[use]
    _, named, rp, rl = D.oracle_step(oracle, st, masks, x, y) if use is False else (None, named, rp, rl)
    flips = D.differing_decisions(dec, named, masks)
    worst = max([(m / s, b, w_) for b, w_, p, m, s in flips] or [(0, "", "")])
    gref, named_f, _, _ = D.oracle_step(oracle, st, masks, x, y, forced=D.forced_from_hip(dec, masks))
    rr = D.grad_errors(model, gref, named_f["up_bias"])
    kinds = {}
    for b, w_, p, m, s in flips:
        kinds[(b, w_)] = kinds.get((b, w_), 0) + 1
    print(f"x3={int(use)}: {len(flips)} decisions differ from the oracle (largest margin / scale {worst}); by kind {kinds}")
    print("   gradient errors under forced decisions, top 15:", [(f"{e:.2e}", k) for e, k in rr[:15]])
