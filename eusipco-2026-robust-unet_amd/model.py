"""Drop-in `RobustUNet` nn.Module (and its blocks) backed by the gfx950 kernels.

Same constructor signatures, attribute tree, parameter registration order and state_dict keys /
shapes / dtypes as /root/reference/Main_Final.py:82-321, so `load_state_dict` interchanges with
the reference in both directions.  Differences that are invisible through that surface:

* convolution weights are *stored* HWIO (logical OIHW tensors with permuted strides), which is what
  the implicit-GEMM kernels read and what the weight-gradient kernel writes - no repacking per step;
* activations inside the network are NHWC; the NCHW boundary is converted once at the input
  (C=3) and is free at the output (C=1);
* `RobustUNet.forward` is ONE autograd node whose backward is an explicit kernel sequence
  (blocks.py); skip-connection gradients are accumulated by the kernels, not by autograd.

Leaf modules (Conv2d, BatchNorm2d, ConvTranspose2d, Dropout2d) only hold parameters/buffers.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from . import blocks as B
from . import ops


# ----------------------------------------------------------------------------- parameter holders
class _Holder(nn.Module):
    def forward(self, *a, **k):
        raise NotImplementedError(f"{type(self).__name__} is a parameter holder; call the enclosing block "
                                  "(ResidualBlock / DilatedBlock / AttentionGate / RobustUNet)")


class Conv2d(_Holder):
    def __init__(self, in_channels, out_channels, kernel_size, padding=0, dilation=1, bias=True, stride=1):
        super().__init__()
        k = kernel_size
        self.in_channels, self.out_channels, self.kernel_size = in_channels, out_channels, (k, k)
        self.padding, self.dilation, self.stride = (padding, padding), (dilation, dilation), (stride, stride)
        self.weight = nn.Parameter(torch.empty(k, k, in_channels, out_channels).permute(3, 2, 0, 1))   # HWIO memory
        if bias:
            self.bias = nn.Parameter(torch.empty(out_channels))
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()

    def reset_parameters(self):  # torch's nn.Conv2d defaults
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            fan_in = self.in_channels * self.kernel_size[0] * self.kernel_size[1]
            bound = 1.0 / math.sqrt(fan_in)
            nn.init.uniform_(self.bias, -bound, bound)

    def extra_repr(self):
        return f"{self.in_channels}, {self.out_channels}, kernel_size={self.kernel_size}, dilation={self.dilation}, bias={self.bias is not None}"


class ConvTranspose2d(_Holder):
    def __init__(self, in_channels, out_channels, kernel_size=2, stride=2):
        super().__init__()
        assert kernel_size == 2 and stride == 2, "the hot path only has the k2-s2 transposed convolution"
        self.in_channels, self.out_channels = in_channels, out_channels
        self.weight = nn.Parameter(torch.empty(2, 2, in_channels, out_channels).permute(2, 3, 0, 1))   # [cin,cout,2,2] logical
        self.bias = nn.Parameter(torch.empty(out_channels))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        bound = 1.0 / math.sqrt(out_channels * 4)
        nn.init.uniform_(self.bias, -bound, bound)


class BatchNorm2d(_Holder):
    def __init__(self, num_features, eps=1e-5, momentum=0.1):
        super().__init__()
        self.num_features, self.eps, self.momentum = num_features, eps, momentum
        self.weight = nn.Parameter(torch.ones(num_features))
        self.bias = nn.Parameter(torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))

    def state(self):
        return B.BNState(self.weight, self.bias, self.running_mean, self.running_var, self.num_batches_tracked)


class Dropout2d(_Holder):
    """Per-(sample, channel) Bernoulli mask scaled by 1/(1-p); `mask` may be injected for parity runs."""

    def __init__(self, p=0.5):
        super().__init__()
        self.p = p
        self.mask = None

    def draw(self, n, c, device):
        if self.mask is not None:
            m = self.mask.to(device=device, dtype=torch.float32)
            assert tuple(m.shape) == (n, c), f"injected dropout mask has shape {tuple(m.shape)}, need {(n, c)}"
            return m.contiguous()
        if self.p <= 0.0:
            return None
        return torch.empty((n, c), device=device, dtype=torch.float32).bernoulli_(1.0 - self.p).div_(1.0 - self.p)


class _Act(nn.Module):
    def forward(self, x):
        raise NotImplementedError("activation is fused into the enclosing block's kernels")


def _nhwc(x):
    """NCHW tensor -> NHWC view of channels_last memory (copies only if the memory is not channels_last)."""
    if x.dim() != 4:
        raise ValueError("expected a 4-D NCHW tensor")
    if x.dtype != torch.float32:
        raise TypeError("the kernels compute in fp32")
    return x.contiguous(memory_format=torch.channels_last).permute(0, 2, 3, 1)


def _nchw(y):
    return y.permute(0, 3, 1, 2)


def _require_cuda(x):
    if not x.is_cuda:
        raise RuntimeError("Robust U-Net kernels run on an MI355X (HIP) device only; there is no CPU path. "
                           "Move the model and the inputs to 'cuda'.")


# ----------------------------------------------------------------------------- attention modules
class ChannelAttention(nn.Module):
    """Parameter layout and call surface of the reference module (Main_Final.py:82-101)."""

    def __init__(self, in_channels, ratio=16):
        super().__init__()
        self.avg_pool, self.max_pool = _Act(), _Act()
        self.fc = nn.Sequential(Conv2d(in_channels, in_channels // ratio, 1, bias=False), _Act(),
                                Conv2d(in_channels // ratio, in_channels, 1, bias=False))
        self.sigmoid = _Act()

    def forward(self, x):
        """Standalone call (inside ResidualBlock the same kernels run fused into the block's tail)."""
        _require_cuda(x)
        return _CAFn.apply(x, self.fc[0].weight, self.fc[2].weight)


class _CAFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w0, w2):
        y, ctx.c = B.ca_forward(_nhwc(x), ops.hwio(w0), ops.hwio(w2))
        return _nchw(y)

    @staticmethod
    def backward(ctx, dy):
        sink = B.DictSink(dy.device)
        dx = B.ca_backward(ctx.c, _nhwc(dy), sink)
        ctx.c = None
        return _nchw(dx), _logical("fc.0.weight", sink.g["fc.0.weight"]), _logical("fc.2.weight", sink.g["fc.2.weight"])


class SpatialAttention(nn.Module):
    def __init__(self, kernel_size=7):
        super().__init__()
        assert kernel_size == 7
        self.conv1 = Conv2d(2, 1, kernel_size, padding=kernel_size // 2, bias=False)
        self.sigmoid = _Act()

    def forward(self, x):
        """Standalone call (inside ResidualBlock the same kernels run fused into the block's tail)."""
        _require_cuda(x)
        return _SAFn.apply(x, self.conv1.weight)


class _SAFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w):
        y, ctx.c = B.sa_forward(_nhwc(x), ops.hwio(w))
        return _nchw(y)

    @staticmethod
    def backward(ctx, dy):
        sink = B.DictSink(dy.device)
        dx = B.sa_backward(ctx.c, _nhwc(dy), sink)
        ctx.c = None
        return _nchw(dx), _logical("conv1.weight", sink.g["conv1.weight"])


class AttentionGate(nn.Module):
    def __init__(self, F_g, F_l, F_int):
        super().__init__()
        self.W_g = nn.Sequential(Conv2d(F_g, F_int, 1, bias=True), BatchNorm2d(F_int))
        self.W_x = nn.Sequential(Conv2d(F_l, F_int, 1, bias=True), BatchNorm2d(F_int))
        self.psi = nn.Sequential(Conv2d(F_int, 1, 1, bias=True), BatchNorm2d(1), _Act())
        self.relu = _Act()

    def _params(self):
        return [self.W_g[0].weight, self.W_g[0].bias, self.W_g[1].weight, self.W_g[1].bias, self.W_x[0].weight, self.W_x[0].bias,
                self.W_x[1].weight, self.W_x[1].bias, self.psi[0].weight, self.psi[0].bias, self.psi[1].weight, self.psi[1].bias]

    def handles(self, up: "ConvTranspose2d | None" = None):
        return B.UpGateParams(ops.hwio_t(up.weight) if up is not None else None, up.bias if up is not None else None,
                              ops.hwio(self.W_g[0].weight), self.W_g[0].bias, self.W_g[1].state(),
                              ops.hwio(self.W_x[0].weight), self.W_x[0].bias, self.W_x[1].state(),
                              ops.hwio(self.psi[0].weight), self.psi[0].bias, self.psi[1].state())

    def forward(self, g, x):
        _require_cuda(x)
        return _GateFn.apply(g, x, self, *self._params())


class _GateFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, g, x, mod, *params):
        gn, xn = _nhwc(g), _nhwc(x)
        p = mod.handles()
        out = ops.empty_nhwc(*xn.shape, xn)
        sm = B.Small(x.device)
        gc = B.gate_forward(gn, xn, p, mod.training, out, sm)
        ctx.gc, ctx.io, ctx.p = gc, (gn, xn), p
        return _nchw(out)

    @staticmethod
    def backward(ctx, dout):
        gn, xn = ctx.io
        dup = torch.zeros(gn.shape, device=gn.device, dtype=torch.float32)
        sink = B.DictSink(gn.device)
        dskip = B.gate_backward(ctx.gc, gn, xn, ctx.p, _nhwc(dout), dup, sink)
        ctx.gc = None
        names = ["W_g.0.weight", "W_g.0.bias", "W_g.1.weight", "W_g.1.bias", "W_x.0.weight", "W_x.0.bias", "W_x.1.weight",
                 "W_x.1.bias", "psi.0.weight", "psi.0.bias", "psi.1.weight", "psi.1.bias"]
        return (_nchw(dup), _nchw(dskip), None) + tuple(_logical(k, sink.g[k]) for k in names)


def _logical(name, t):
    """physical gradient -> tensor shaped/strided like the parameter it belongs to."""
    if t.dim() == 4:
        return t.permute(2, 3, 0, 1) if name.startswith("up") and name.endswith("weight") else t.permute(3, 2, 0, 1)
    return t


# ----------------------------------------------------------------------------- ResidualBlock
_RB_NAMES = ["conv1.weight", "bn1.weight", "bn1.bias", "conv2.weight", "bn2.weight", "bn2.bias", "ca.fc.0.weight", "ca.fc.2.weight",
             "sa.conv1.weight"]
_RB_SC = ["shortcut.0.weight", "shortcut.1.weight", "shortcut.1.bias"]


class ResidualBlock(nn.Module):
    def __init__(self, in_channels, out_channels, dropout_rate=0.1):
        super().__init__()
        self.conv1 = Conv2d(in_channels, out_channels, 3, padding=1, bias=False)
        self.bn1 = BatchNorm2d(out_channels)
        self.conv2 = Conv2d(out_channels, out_channels, 3, padding=1, bias=False)
        self.bn2 = BatchNorm2d(out_channels)
        self.dropout = Dropout2d(dropout_rate)
        self.relu = _Act()
        self.ca = ChannelAttention(out_channels)
        self.sa = SpatialAttention()
        if in_channels != out_channels:
            self.shortcut = nn.Sequential(Conv2d(in_channels, out_channels, 1, bias=False), BatchNorm2d(out_channels))
        else:
            self.shortcut = nn.Identity()
        self.in_channels, self.out_channels = in_channels, out_channels

    def param_names(self):
        return _RB_NAMES + (_RB_SC if self.in_channels != self.out_channels else [])

    def _params(self):
        ps = [self.conv1.weight, self.bn1.weight, self.bn1.bias, self.conv2.weight, self.bn2.weight, self.bn2.bias,
              self.ca.fc[0].weight, self.ca.fc[2].weight, self.sa.conv1.weight]
        if self.in_channels != self.out_channels:
            ps += [self.shortcut[0].weight, self.shortcut[1].weight, self.shortcut[1].bias]
        return ps

    def handles(self):
        sc = self.in_channels != self.out_channels
        return B.RBParams(ops.hwio(self.conv1.weight), self.bn1.state(), ops.hwio(self.conv2.weight), self.bn2.state(),
                          ops.hwio(self.ca.fc[0].weight), ops.hwio(self.ca.fc[2].weight), ops.hwio(self.sa.conv1.weight),
                          ops.hwio(self.shortcut[0].weight) if sc else None, self.shortcut[1].state() if sc else None)

    def forward(self, x):
        _require_cuda(x)
        return _RBFn.apply(x, self, *self._params())


def _pad_channels(xn, mult=4):
    """NHWC view whose channel count is not a multiple of 4 (RGB stem) -> zero-padded dense copy."""
    n, h, w, c = xn.shape
    if c % mult == 0:
        return xn
    return B.to_nhwc_pad(xn.permute(0, 3, 1, 2), (c + mult - 1) // mult * mult)


class _RBFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mod, *params):
        xn = _pad_channels(_nhwc(x))
        n, c = x.shape[0], mod.out_channels
        mask = mod.dropout.draw(n, c, x.device) if mod.training else None
        out, c_ = B.rb_forward(xn, mod.handles(), mod.training, mask, save=any(ctx.needs_input_grad))
        ctx.c, ctx.names, ctx.need_dx, ctx.cin = c_, mod.param_names(), ctx.needs_input_grad[0], x.shape[1]
        return _nchw(out)

    @staticmethod
    def backward(ctx, dout):
        sink = B.DictSink(dout.device)
        dx = B.rb_backward(ctx.c, _nhwc(dout), sink, need_dx=ctx.need_dx)
        ctx.c = None
        if dx is not None:
            dx = _nchw(dx[..., :ctx.cin]) if dx.shape[3] != ctx.cin else _nchw(dx)
        return (dx, None) + tuple(_logical(k, sink.g[k]) for k in ctx.names)


# ----------------------------------------------------------------------------- DilatedBlock
class DilatedBlock(nn.Module):
    def __init__(self, in_channels, out_channels):
        super().__init__()
        q = out_channels // 4
        self.conv1 = Conv2d(in_channels, q, 1)
        self.conv2 = Conv2d(in_channels, q, 3, padding=1, dilation=1)
        self.conv3 = Conv2d(in_channels, q, 3, padding=2, dilation=2)
        self.conv4 = Conv2d(in_channels, q, 3, padding=4, dilation=4)
        self.bn = BatchNorm2d(out_channels)
        self.relu = _Act()

    def param_names(self):
        return [f"conv{i}.{k}" for i in (1, 2, 3, 4) for k in ("weight", "bias")] + ["bn.weight", "bn.bias"]

    def _params(self):
        cs = (self.conv1, self.conv2, self.conv3, self.conv4)
        return [t for c in cs for t in (c.weight, c.bias)] + [self.bn.weight, self.bn.bias]

    def handles(self):
        cs = (self.conv1, self.conv2, self.conv3, self.conv4)
        return B.DilParams([ops.hwio(c.weight) for c in cs], [c.bias for c in cs], self.bn.state())

    def forward(self, x):
        _require_cuda(x)
        return _DilFn.apply(x, self, *self._params())


class _DilFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mod, *params):
        out, c_ = B.dilated_forward(_nhwc(x), mod.handles(), mod.training)
        ctx.c, ctx.names, ctx.need_dx = c_, mod.param_names(), ctx.needs_input_grad[0]
        return _nchw(out)

    @staticmethod
    def backward(ctx, dout):
        sink = B.DictSink(dout.device)
        dx = B.dilated_backward(ctx.c, _nhwc(dout), sink, need_dx=ctx.need_dx)
        ctx.c = None
        return (_nchw(dx) if dx is not None else None, None) + tuple(_logical(k, sink.g[k]) for k in ctx.names)


class MaxPool2d(nn.Module):
    def __init__(self, kernel_size=2):
        super().__init__()
        assert kernel_size == 2
        self.kernel_size = kernel_size

    def forward(self, x):
        _require_cuda(x)
        return _PoolFn.apply(x)


class _PoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        y, idx = B.maxpool_forward(_nhwc(x))
        ctx.idx = idx
        return _nchw(y)

    @staticmethod
    def backward(ctx, dy):
        return _nchw(B.maxpool_backward(_nhwc(dy), ctx.idx))


# ----------------------------------------------------------------------------- RobustUNet
class RobustUNet(nn.Module):
    """forward(x: float32 [N, n_channels, H, W], H and W multiples of 16) -> sigmoid probabilities [N, n_classes=1, H, W]."""

    LEVEL_DROPOUT = {"inc": 0.1, "down1": 0.1, "down2": 0.2, "down3": 0.2, "bottleneck": 0.3, "dec4": 0.2, "dec3": 0.2,
                     "dec2": 0.1, "dec1": 0.1}

    def __init__(self, n_channels=3, n_classes=1, base_channels=64):
        super().__init__()
        if n_classes != 1:
            raise ValueError("the fused output head implements the reference's n_classes=1 sigmoid head")
        if base_channels % 16 != 0:
            raise ValueError("base_channels must be a multiple of 16 (ChannelAttention uses C // 16)")
        b = base_channels
        self.n_channels, self.n_classes, self.base_channels = n_channels, n_classes, b
        self.inc = ResidualBlock(n_channels, b, dropout_rate=0.1)
        self.down1 = nn.Sequential(MaxPool2d(2), ResidualBlock(b, b * 2, dropout_rate=0.1))
        self.down2 = nn.Sequential(MaxPool2d(2), ResidualBlock(b * 2, b * 4, dropout_rate=0.2))
        self.down3 = nn.Sequential(MaxPool2d(2), ResidualBlock(b * 4, b * 8, dropout_rate=0.2))
        self.bottleneck = nn.Sequential(MaxPool2d(2), DilatedBlock(b * 8, b * 16), ResidualBlock(b * 16, b * 16, dropout_rate=0.3))
        self.att4 = AttentionGate(b * 8, b * 8, b * 4)
        self.att3 = AttentionGate(b * 4, b * 4, b * 2)
        self.att2 = AttentionGate(b * 2, b * 2, b)
        self.att1 = AttentionGate(b, b, b // 2)
        self.up4 = ConvTranspose2d(b * 16, b * 8, 2, stride=2)
        self.dec4 = ResidualBlock(b * 16, b * 8, dropout_rate=0.2)
        self.up3 = ConvTranspose2d(b * 8, b * 4, 2, stride=2)
        self.dec3 = ResidualBlock(b * 8, b * 4, dropout_rate=0.2)
        self.up2 = ConvTranspose2d(b * 4, b * 2, 2, stride=2)
        self.dec2 = ResidualBlock(b * 4, b * 2, dropout_rate=0.1)
        self.up1 = ConvTranspose2d(b * 2, b, 2, stride=2)
        self.dec1 = ResidualBlock(b * 2, b, dropout_rate=0.1)
        self.outc = nn.Sequential(Conv2d(b, n_classes, 1), _Act())
        self.sync_bn_hook = None     # set by ddp.GradAllReducer(sync_bn=True)
        self.precision = "f32"       # operand type of the convolutions' multiply-adds: set_precision("bf16") for BASELINE configs 3 / 5
        self._arena = None
        self._named = None
        self._initialize_weights()

    def _initialize_weights(self):
        for m in self.modules():
            if isinstance(m, Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    # ---- plumbing -------------------------------------------------------------------------
    def named_params(self):
        """list(self.named_parameters()), cached: the module walk costs ~0.5 ms and runs three times per step (the module tree is
        fixed after construction; Module._apply - .to(), .cuda(), .float() - drops the cache)."""
        if self._named is None:
            self._named = list(self.named_parameters())
        return self._named

    def _apply(self, fn, recurse=True):
        self._named = None
        return super()._apply(fn, recurse)

    def _rbs(self):
        return {"inc": self.inc, "down1.1": self.down1[1], "down2.1": self.down2[1], "down3.1": self.down3[1],
                "bottleneck.2": self.bottleneck[2], "dec4": self.dec4, "dec3": self.dec3, "dec2": self.dec2, "dec1": self.dec1}

    def grad_arena(self):
        """The flat gradient buffer (created on first use, rebuilt if the parameters moved device)."""
        dev = self.outc[0].weight.device
        if self._arena is None or self._arena.device != dev:
            self._arena = GradArena(self)
        return self._arena

    def set_precision(self, mode):
        """'f32' (the reference's arithmetic) or 'bf16': convolution operands rounded to bf16, fp32 accumulation; parameters (fp32
        masters), activations in HBM, BatchNorm, attention, loss and optimizer stay fp32 (ops.precision)."""
        if mode not in ops.PRECISIONS:
            raise ValueError(f"precision must be one of {ops.PRECISIONS}")
        self.precision = mode
        return self

    def set_dropout_masks(self, masks):
        """masks: {block prefix: [N, C] keep-mask already divided by 1-p} or None to restore random draws."""
        for k, rb in self._rbs().items():
            rb.dropout.mask = None if masks is None else masks[k]

    def forward(self, x, return_logits=False):
        _require_cuda(x)
        if x.shape[2] % 16 or x.shape[3] % 16:
            raise ValueError("H and W must be multiples of 16 (four 2x2 poolings)")
        named = self.named_params()
        names = [k for k, _ in named]
        params = [p for _, p in named]
        if torch.is_grad_enabled() and any(p.requires_grad for p in params):
            prob, logit = _NetFn.apply(x, self, names, return_logits, *params)
        else:
            prob, logit, _ = net_forward(self, x, save=False, want_logit=return_logits)
        return (prob, logit) if return_logits else prob


def _draw_masks(net, rbs, n, dev):
    """Dropout2d keep-masks (already divided by 1 - p) of all nine blocks: ONE Bernoulli draw over the concatenated channels with a
    per-channel probability row, sliced per block - 2 launches instead of 18 tiny ones in front of the first convolution.  Injected masks
    (tests, set_dropout_masks) keep the per-block path."""
    if any(rb.dropout.mask is not None for rb in rbs.values()):
        return {k: rb.dropout.draw(n, rb.out_channels, dev) for k, rb in rbs.items()}
    key = (str(dev), n, tuple((k, rb.out_channels, rb.dropout.p) for k, rb in rbs.items()))
    plan = getattr(net, "_mask_plan", None)
    if plan is None or plan[0] != key:
        # block-major flat layout: block k owns n * C_k consecutive entries, so every slice is a contiguous [n, C_k] mask
        keep = torch.cat([torch.full((n * rb.out_channels,), 1.0 - rb.dropout.p) for rb in rbs.values()])
        inv = torch.where(keep > 0, 1.0 / keep.clamp_min(1e-12), torch.zeros_like(keep))
        offs, o = {}, 0
        for k, rb in rbs.items():
            offs[k] = (o, o + n * rb.out_channels)
            o += n * rb.out_channels
        plan = net._mask_plan = (key, keep.to(dev), inv.to(dev), offs)
    _, keep, inv, offs = plan
    m = torch.bernoulli(keep).mul_(inv)
    return {k: (m[a:b].view(n, rb.out_channels) if rb.dropout.p > 0.0 else None) for (k, rb), (a, b) in zip(rbs.items(), offs.values())}


def net_forward(net: RobustUNet, x, save, want_logit=False):
    with ops.precision(net.precision):
        return _net_forward(net, x, save, want_logit)


def _net_forward(net: RobustUNet, x, save, want_logit=False):
    tr = net.training
    hook = net.sync_bn_hook if tr else None
    n = x.shape[0]
    dev = x.device
    ops.branches_pay(n, x.shape[2], x.shape[3])
    if save:
        ops.prefetch_derived()          # stale Winograd filters / packed weights: refilled on the side stream while the stem runs
    x0 = B.to_nhwc_pad(x, (net.n_channels + 3) // 4 * 4)
    rbs = net._rbs()
    masks = _draw_masks(net, rbs, n, dev) if tr else {k: None for k in rbs}
    C = {}
    x1, C["inc"] = B.rb_forward(x0, rbs["inc"].handles(), tr, masks["inc"], save, hook)
    skips = [x1]
    cur = x1
    for lvl in (1, 2, 3):
        pooled, C[f"pool{lvl}"] = B.maxpool_forward(cur)
        cur, C[f"down{lvl}.1"] = B.rb_forward(pooled, rbs[f"down{lvl}.1"].handles(), tr, masks[f"down{lvl}.1"], save, hook)
        skips.append(cur)
    pooled, C["pool4"] = B.maxpool_forward(cur)
    xd, C["bottleneck.1"] = B.dilated_forward(pooled, net.bottleneck[1].handles(), tr, save, hook)
    y, C["bottleneck.2"] = B.rb_forward(xd, rbs["bottleneck.2"].handles(), tr, masks["bottleneck.2"], save, hook)
    for lvl in (4, 3, 2, 1):
        att, up = getattr(net, f"att{lvl}"), getattr(net, f"up{lvl}")
        cat, C[f"upgate{lvl}"] = B.upgate_forward(y, skips[lvl - 1], att.handles(up), tr, save, hook)
        y, C[f"dec{lvl}"] = B.rb_forward(cat, rbs[f"dec{lvl}"].handles(), tr, masks[f"dec{lvl}"], save, hook)
    w, b = ops.hwio(net.outc[0].weight), net.outc[0].bias
    prob, logit = B.outc_forward(y, w, b, want_logit)
    if save:
        C["head"] = (y, w, prob)
        C["precision"] = net.precision
    return prob, logit, (C if save else None)


# backward completes the blocks in this order; the gradient arena is laid out the same way, so that
# "everything up to block k" is one contiguous range (RCCL buckets are slices of it, no packing copies)
BACKWARD_ORDER = ("outc.0", "dec1", "att1", "up1", "dec2", "att2", "up2", "dec3", "att3", "up3", "dec4", "att4", "up4",
                  "bottleneck.2", "bottleneck.1", "down3.1", "down2.1", "down1.1", "inc")


class GradArena:
    """One flat fp32 buffer holding every parameter gradient, in backward-completion order.

    `p.grad` of each parameter is a view into it (same logical shape / physical strides as the
    parameter).  Each tensor starts on a 16-byte boundary, except a `.bias` that directly follows
    the `.weight` of the same module, which is packed right behind it (the BN / psi / outc kernels
    write the pair as one vector)."""

    def __init__(self, net):
        named = list(net.named_parameters())
        self.device = named[0][1].device
        self.off, self.numel, self.block_end = {}, {}, {}
        cur, seen = 0, set()
        for blk in BACKWARD_ORDER:
            prev = None
            for name, p in named:
                if not name.startswith(blk + "."):
                    continue
                packed = prev is not None and name.endswith(".bias") and prev == name[:-5] + ".weight"
                if not packed:
                    cur = (cur + 3) // 4 * 4
                self.off[name], self.numel[name] = cur, p.numel()
                cur += p.numel()
                prev = name
                seen.add(name)
            self.block_end[blk] = cur
        assert len(seen) == len(named), "BACKWARD_ORDER does not cover every parameter"
        self.total = (cur + 3) // 4 * 4
        self.flat = torch.zeros(self.total, device=self.device, dtype=torch.float32)
        self.on_block_done = None        # ddp.GradAllReducer hook: f(end_offset_in_floats)

    def buf(self, prefix, items):
        off0 = self.off[prefix + items[0][0]]
        cur = off0
        for name, sh in items:
            n = B._numel(sh)
            assert self.off[prefix + name] == cur and self.numel[prefix + name] == n, f"arena layout mismatch at {prefix + name}"
            cur += n
        return self.flat[off0:cur]

    def grad_view(self, name, p):
        return torch.as_strided(self.flat, p.shape, p.stride(), self.off[name])

    def done(self, blk):
        if self.on_block_done is not None:
            self.on_block_done(self.block_end[blk])


def net_backward(C, dprob, sink, done=lambda blk: None):
    """Explicit backward pass; parameter gradients are written through `sink` (GradArena or DictSink)."""
    with ops.precision(C.get("precision", "f32")), ops.wgrad_side_stream():
        _net_backward(C, dprob, sink, done)


def _net_backward(C, dprob, sink, done):
    y, w, prob = C["head"]
    dy = B.outc_backward(dprob, prob, y, w, sink, pre="outc.0.")
    done("outc.0")
    dskip = {}
    for lvl in (1, 2, 3, 4):
        dcat = B.rb_backward(C[f"dec{lvl}"], dy, sink, pre=f"dec{lvl}.")
        done(f"dec{lvl}")
        dy, dskip[lvl] = B.upgate_backward(C[f"upgate{lvl}"], dcat, sink, f"att{lvl}.", f"up{lvl}.")
        done(f"up{lvl}")
    dxd = B.rb_backward(C["bottleneck.2"], dy, sink, pre="bottleneck.2.")
    done("bottleneck.2")
    dpool = B.dilated_backward(C["bottleneck.1"], dxd, sink, pre="bottleneck.1.")
    done("bottleneck.1")
    dcur = B.maxpool_backward(dpool, C["pool4"], dx=dskip[4])
    for lvl in (3, 2, 1):
        dpool = B.rb_backward(C[f"down{lvl}.1"], dcur, sink, pre=f"down{lvl}.1.")
        done(f"down{lvl}.1")
        dcur = B.maxpool_backward(dpool, C[f"pool{lvl}"], dx=dskip[lvl])
    B.rb_backward(C["inc"], dcur, sink, pre="inc.", need_dx=False)
    done("inc")


class _NetFn(torch.autograd.Function):
    """The whole network as one autograd node.  Its backward writes the parameter gradients straight into the
    model's GradArena and assigns `p.grad` itself (returning None to autograd for the parameters): no
    per-tensor accumulate kernels, gradients stay contiguous for RCCL and the fused optimizer.  If some
    `p.grad` is already populated (gradient accumulation), the new gradients are added to it instead."""

    @staticmethod
    def forward(ctx, x, net, names, want_logit, *params):
        prob, logit, C = net_forward(net, x, save=True, want_logit=want_logit)
        ctx.C, ctx.net = C, net
        if logit is None:
            logit = prob.new_empty(0)
        ctx.mark_non_differentiable(logit)
        return prob, logit

    @staticmethod
    def backward(ctx, dprob, _dlogit):
        if ctx.C is None:
            raise RuntimeError("RobustUNet backward called twice (activations were released after the first pass)")
        net = ctx.net
        named = net.named_params()
        arena = net.grad_arena()
        if all(p.grad is None for _, p in named):
            net_backward(ctx.C, dprob.contiguous(), arena, arena.done)
            for k, p in named:
                if p.requires_grad:
                    p.grad = arena.grad_view(k, p)
        else:
            sink = B.DictSink(dprob.device)
            net_backward(ctx.C, dprob.contiguous(), sink)
            for k, p in named:
                if not p.requires_grad:
                    continue
                g = _logical(k, sink.g[k])
                if p.grad is None:
                    p.grad = g
                else:
                    p.grad.add_(g)
        ctx.C = None
        return (None, None, None, None) + (None,) * len(named)
