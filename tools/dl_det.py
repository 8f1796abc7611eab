"""Run-to-run determinism of the DeepLabV3+ train step at the benchmarked size (16 x 256^2): three steps from the same state, every gradient
tensor compared bitwise; prints the tensors that differ.  (Found the LDS float atomics of head3x3_bwd_kernel in round 3.)"""
import importlib, sys, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("eusipco-2026-robust-unet_amd")
dl = importlib.import_module("oracle.deeplab_ref")
DEV="cuda:0"
st = dl.init_state(seed=21, perturb_bn=True)
x, y = pkg.synthetic_batch(16, 256, seed=21)
xd, yd = x.to(DEV), y.to(DEV)
res=[]
for _ in range(3):
    m = pkg.DeepLabV3Plus(n_classes=1); m.load_state_dict(st); m = m.to(DEV).train()
    p = m(xd); l = pkg.bce_loss(p, yd); l.backward(); torch.cuda.synchronize()
    res.append({k: q.grad.detach().clone() for k, q in m.named_parameters()})
for k in res[0]:
    d1 = float((res[0][k]-res[1][k]).abs().max()); d2 = float((res[0][k]-res[2][k]).abs().max())
    if d1 or d2: print(k, d1, d2, float(res[0][k].abs().max()))
