"""Plain 2-class U-Net of the reference's older trainer on the gfx950 kernels.

Drop-in for `UNet` in /root/reference/train_water_segmentation.py:209-288 (the model `predict_coastline.py:351` loads): same
constructor, attribute tree and state_dict (enc1..4 / bottleneck / dec4..1 = Sequential(Conv2d(3x3, bias), BatchNorm2d, ReLU, Conv2d,
BatchNorm2d, ReLU), upconv4..1 = ConvTranspose2d(k2, s2), final = Conv2d(64, n_classes, 1), pool), forward(x [N, 3, H, W]) ->
logits [N, n_classes, H, W]; trained with nn.CrossEntropyLoss on int64 masks (:304) - `ops.cross_entropy` is the fused equivalent.

Built from the Robust U-Net path's kernels: 3x3 convolutions (Winograd / implicit GEMM / bf16), BatchNorm +
ReLU, 2x2 max-pool, k2-s2 transposed convolution.  As there, the whole network is ONE autograd node with an explicit backward;
`torch.cat([upsampled, skip])` is never materialised: the encoder block writes its output straight into the right half of the
decoder's input buffer and the transposed convolution into the left half.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import blocks as B
from . import ops
from ._lib import check, lib
from .model import BatchNorm2d, Conv2d, ConvTranspose2d, _Act, _logical, _require_cuda

CH = (64, 128, 256, 512)


def _conv_block(cin, cout):
    return nn.Sequential(Conv2d(cin, cout, 3, padding=1), BatchNorm2d(cout), _Act(), Conv2d(cout, cout, 3, padding=1), BatchNorm2d(cout), _Act())


class _Pool(nn.Module):
    def __init__(self):
        super().__init__()
        self.kernel_size, self.stride = 2, 2

    def forward(self, x):
        from .model import MaxPool2d
        return MaxPool2d(2)(x)


class UNet(nn.Module):
    def __init__(self, n_channels=3, n_classes=2):
        super().__init__()
        if not 1 <= n_classes <= 4:
            # the reference takes any n_classes; this head is a 64 -> 4 padded 1x1 GEMM (channel quads) and ops.cross_entropy takes 2..8
            # classes, so 1..4 run here (the reference trains 2); INTEGRATION.md states the bound
            raise ValueError("the fused head handles 1..4 classes (the reference uses 2)")
        self.n_channels, self.n_classes = n_channels, n_classes
        self.enc1 = _conv_block(n_channels, 64)
        self.enc2 = _conv_block(64, 128)
        self.enc3 = _conv_block(128, 256)
        self.enc4 = _conv_block(256, 512)
        self.bottleneck = _conv_block(512, 1024)
        self.upconv4 = ConvTranspose2d(1024, 512, 2, stride=2)
        self.dec4 = _conv_block(1024, 512)
        self.upconv3 = ConvTranspose2d(512, 256, 2, stride=2)
        self.dec3 = _conv_block(512, 256)
        self.upconv2 = ConvTranspose2d(256, 128, 2, stride=2)
        self.dec2 = _conv_block(256, 128)
        self.upconv1 = ConvTranspose2d(128, 64, 2, stride=2)
        self.dec1 = _conv_block(128, 64)
        self.final = Conv2d(64, n_classes, 1)
        self.pool = _Pool()
        self.precision = "f32"

    def __setattr__(self, name, value):
        # ddp.GradAllReducer(sync_bn=True) / set_sync_bn(True) install a cross-rank BatchNorm hook on the model; this model's blocks use
        # per-rank statistics only - refuse loudly instead of silently training a different function than the caller asked for
        if name == "sync_bn_hook" and value is not None:
            raise NotImplementedError("the plain UNet has no SyncBatchNorm path (per-rank BatchNorm statistics only): "
                                      "construct GradAllReducer(sync_bn=False)")
        super().__setattr__(name, value)

    def set_precision(self, mode):
        if mode not in ops.PRECISIONS:
            raise ValueError(f"precision must be one of {ops.PRECISIONS}")
        self.precision = mode
        return self

    def forward(self, x):
        _require_cuda(x)
        if x.shape[2] % 16 or x.shape[3] % 16:
            raise ValueError("H and W must be multiples of 16 (four 2x2 poolings)")
        params = [p for _, p in self.named_parameters()]
        if torch.is_grad_enabled() and any(p.requires_grad for p in params):
            return _UNetFn.apply(x, self, *params)
        with ops.precision(self.precision):
            return unet_forward(self, x, save=False)[0]


# ------------------------------------------------------------------------------------------------------------ blocks
def _block_forward(x, seq, training, sm, out=None, save=True):
    c1, bn1, c2, bn2 = seq[0], seq[1], seq[3], seq[4]
    w1, w2 = ops.hwio(c1.weight), ops.hwio(c2.weight)
    t1 = ops.conv_fwd(x, w1, c1.bias)
    s1, h1, mean1, invstd1, _ = B.bn_coeff(t1, bn1.state(), training, sm)
    a1 = B.bn_apply(t1, s1, h1, None, relu=True)
    t2 = ops.conv_fwd(a1, w2, c2.bias)
    s2, h2, mean2, invstd2, _ = B.bn_coeff(t2, bn2.state(), training, sm)
    a2 = B.bn_apply(t2, s2, h2, None, relu=True, out=out)
    ctx = None
    if save:
        ctx = dict(x=x, w1=w1, w2=w2, t1=t1, a1=a1, t2=t2, s1=s1, h1=h1, mean1=mean1, invstd1=invstd1, s2=s2, h2=h2, mean2=mean2, invstd2=invstd2,
                   training=training, cin_w=w1.shape[2])
    return a2, ctx


def _block_backward(c, dout, G, pre, need_dx=True):
    dev = dout.device
    cout = c["w2"].shape[3]

    def vec(n):
        return torch.empty(n, device=dev, dtype=torch.float32)

    sums2 = vec(2 * cout)
    dt2 = B.bn_backward(dout, c["t2"], c["mean2"], c["invstd2"], c["s2"], sums2, relu_shift=c["h2"], training=c["training"])
    G[pre + ".4.weight"], G[pre + ".4.bias"] = sums2[:cout], sums2[cout:]
    G[pre + ".3.weight"] = ops.conv_wgrad(c["a1"], dt2, 3, 3)
    G[pre + ".3.bias"] = B.chan_sum(dt2, vec(cout))
    da1 = ops.conv_dgrad(dt2, c["w2"])
    del dt2
    sums1 = vec(2 * cout)
    dt1 = B.bn_backward(da1, c["t1"], c["mean1"], c["invstd1"], c["s1"], sums1, relu_shift=c["h1"], out=da1, training=c["training"])
    G[pre + ".1.weight"], G[pre + ".1.bias"] = sums1[:cout], sums1[cout:]
    G[pre + ".0.weight"] = ops.conv_wgrad(c["x"], dt1, 3, 3, cin_w=c["cin_w"], on_side=need_dx)
    G[pre + ".0.bias"] = B.chan_sum(dt1, vec(cout))
    return ops.conv_dgrad(dt1, c["w1"]) if need_dx else None


def _head_weights(net):
    """final (64 -> n_classes, 1x1) zero-padded to 4 output channels: the implicit-GEMM kernels work on channel quads."""
    w = ops.hwio(net.final.weight)                      # [1, 1, 64, classes]
    w4 = torch.zeros((1, 1, w.shape[2], 4), device=w.device, dtype=torch.float32)
    w4[..., :net.n_classes].copy_(w)
    b4 = torch.zeros(4, device=w.device, dtype=torch.float32)
    b4[:net.n_classes].copy_(net.final.bias.detach())
    return w4, b4


def unet_forward(net: UNet, x, save=True):
    tr = net.training
    dev = x.device
    sm = B.Small(dev)
    n = x.shape[0]
    C = {}
    ops.branches_pay(n, x.shape[2], x.shape[3])
    if save:
        ops.prefetch_derived()
    cur = B.to_nhwc_pad(x, (net.n_channels + 3) // 4 * 4)
    cats = {}
    for lvl, ch in enumerate(CH, 1):
        _, h, w, _ = cur.shape
        cat = ops.empty_nhwc(n, h, w, 2 * ch, cur)
        skip = cat[..., ch:]
        _, C[f"enc{lvl}"] = _block_forward(cur, getattr(net, f"enc{lvl}"), tr, sm, out=skip, save=save)
        cats[lvl] = cat
        cur, C[f"pool{lvl}"] = B.maxpool_forward(skip)
    y, C["bottleneck"] = _block_forward(cur, net.bottleneck, tr, sm, save=save)
    for lvl in (4, 3, 2, 1):
        up = getattr(net, f"upconv{lvl}")
        ch = CH[lvl - 1]
        wup = ops.hwio_t(up.weight)
        ops.convt_fwd(y, wup, up.bias, out=cats[lvl][..., :ch])
        if save:
            C[f"up{lvl}"] = (y, wup)
        y, C[f"dec{lvl}"] = _block_forward(cats[lvl], getattr(net, f"dec{lvl}"), tr, sm, save=save)
    w4, b4 = _head_weights(net)
    z4 = ops.conv_fwd(y, w4, b4)
    _, h, w, _ = z4.shape
    logits = torch.empty((n, net.n_classes, h, w), device=dev, dtype=torch.float32)
    check(lib.runet_nhwc_to_nchw(z4.data_ptr(), 4, logits.data_ptr(), n, net.n_classes, h * w, ops.stream()))
    if save:
        C["head"] = (y, w4)
    return logits, (C if save else None)


def unet_backward(net: UNet, C, dlogits):
    """-> {parameter name: gradient in the parameter's PHYSICAL layout}"""
    G = {}
    dev = dlogits.device
    y, w4 = C["head"]
    k = net.n_classes
    dz4 = B.to_nhwc_pad(dlogits.contiguous(), 4)
    G["final.weight"] = ops.conv_wgrad(y, dz4, 1, 1, on_side=False)[..., :k].contiguous()
    G["final.bias"] = B.chan_sum(dz4, torch.empty(4, device=dev, dtype=torch.float32))[:k]
    dy = ops.conv_dgrad(dz4, w4)
    dskip = {}
    for lvl in (1, 2, 3, 4):
        ch = CH[lvl - 1]
        dcat = _block_backward(C[f"dec{lvl}"], dy, G, f"dec{lvl}")
        dup, dskip[lvl] = dcat[..., :ch], dcat[..., ch:]
        yin, wup = C[f"up{lvl}"]
        G[f"upconv{lvl}.weight"] = ops.convt_wgrad(yin, dup)
        G[f"upconv{lvl}.bias"] = B.chan_sum(dup, torch.empty(ch, device=dev, dtype=torch.float32))
        dy = ops.convt_dgrad(dup, wup)
    dcur = _block_backward(C["bottleneck"], dy, G, "bottleneck")
    for lvl in (4, 3, 2, 1):
        B.maxpool_backward(dcur, C[f"pool{lvl}"], dx=dskip[lvl])          # adds the pooled path's gradient to the skip's
        dcur = _block_backward(C[f"enc{lvl}"], dskip[lvl], G, f"enc{lvl}", need_dx=lvl > 1)
    return G


class _UNetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, net, *params):
        with ops.precision(net.precision):
            logits, C = unet_forward(net, x, save=True)
        ctx.C, ctx.net = C, net
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        if ctx.C is None:
            raise RuntimeError("UNet backward called twice (activations were released after the first pass)")
        net = ctx.net
        with ops.precision(net.precision), ops.wgrad_side_stream():
            G = unet_backward(net, ctx.C, dlogits)
        ctx.C = None
        out = []
        for name, p in net.named_parameters():
            g = G[name]
            if name.startswith("upconv") and name.endswith("weight"):
                g = g.permute(2, 3, 0, 1)                  # physical [2, 2, cin, cout] -> logical [cin, cout, 2, 2]
            elif g.dim() == 4:
                g = g.permute(3, 2, 0, 1)                  # physical HWIO -> logical OIHW
            out.append(g)
        ops.deliver_grads(net, [p for _, p in net.named_parameters()], out)      # fixed addresses, assigned here (not returned to autograd)
        return (None, None) + (None,) * len(out)
