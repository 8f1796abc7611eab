"""F(4x4) convolutions with the position GEMMs on the f32 MFMA (gemm.hip) and on the split-operand bf16 path (gemm_split.hip), both against a
float64 CPU convolution: forward, data gradient, weight gradient (fresh V and the forward's kept V), at chosen magnitudes of dy.
usage: python tools/diag_x3.py [n h w cin cout dil dy_scale]..."""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

ops = importlib.import_module("eusipco-2026-robust-unet_amd.ops")
dev = "cuda:0"
CASES = [(1, 64, 64, 1024, 1024, 1, 1.0), (1, 64, 64, 1024, 1024, 1, 1e-9), (1, 128, 128, 256, 512, 1, 1e-9), (1, 64, 64, 1024, 256, 4, 1e-9),
         (1, 64, 64, 1024, 256, 2, 1.0), (2, 32, 32, 1024, 1024, 1, 1.0)]


def rel(got, ref):
    d = (got.double().cpu() - ref).abs()
    return float(d.max() / ref.abs().max()), float((d.pow(2).mean() / ref.pow(2).mean()).sqrt())


torch.set_num_threads(16)
for n, h, w, cin, cout, dil, sc in CASES:
    g = torch.Generator().manual_seed(h + cin + cout + dil)
    x = torch.randn(n, cin, h, w, generator=g).relu()
    wt = torch.randn(cout, cin, 3, 3, generator=g) / (9 * cin) ** 0.5
    dy = torch.randn(n, cout, h, w, generator=g) * sc
    xd_, wd_ = x.double().requires_grad_(True), wt.double().requires_grad_(True)
    y = F.conv2d(xd_, wd_, None, 1, dil, dil)
    y.backward(dy.double())
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
    dyd = dy.permute(0, 2, 3, 1).contiguous().to(dev)
    wd = wt.permute(2, 3, 1, 0).contiguous().to(dev)
    for use in (False, True):
        ops.USE_X3 = use
        U, Ud = ops.wino4_weights(wd), ops.wino4_weights(wd, dgrad=True)
        keep = {}
        f = ops.wino4_conv(xd, U, None, keep_v=keep, dil=dil).permute(0, 3, 1, 2)
        f2 = ops.wino4_conv(xd, U, None, dil=dil).permute(0, 3, 1, 2)
        dg = ops.wino4_conv(dyd, Ud, dil=dil).permute(0, 3, 1, 2)
        wg = ops.wino4_wgrad(xd, dyd, dil=dil).permute(3, 2, 0, 1)
        wg2 = ops.wino4_wgrad(xd, dyd, v=keep["V"], dil=dil).permute(3, 2, 0, 1)
        torch.cuda.synchronize()
        print(f"{(n, h, w, cin, cout, dil, sc)} x3={int(use)}: fwd {rel(f, y.detach())} fwd(composite) {rel(f2, y.detach())} dgrad {rel(dg, xd_.grad)} "
              f"wgrad {rel(wg, wd_.grad)} wgrad(kept V) {rel(wg2, wd_.grad)}", flush=True)
