"""`torch.library` registration of the hot path's leaf operators (namespace `runet`), as SURVEY.md section 8(b) lists them:
PyTorch-ROCm custom ops over the C ABI (include/runet_hip.h).  Each op has a fake (meta) implementation, so FakeTensor tracing /
`torch.compile` see shapes without running a kernel, and the differentiable ones carry `register_autograd` formulas that call the
hand-written data-gradient / weight-gradient kernels.

    torch.ops.runet.conv2d_nhwc(x, w_hwio, bias, dilation)      3x3 / 1x1 'same' convolution, NHWC activations, HWIO weights
    torch.ops.runet.conv2d_nhwc_dgrad / conv2d_nhwc_wgrad        its two gradients (ops in their own right)
    torch.ops.runet.convt2x2s2_nhwc(x, w_hwio, bias)             ConvTranspose2d(k2, s2)  (+ _dgrad / _wgrad)
    torch.ops.runet.maxpool2_nhwc(x) -> (y, idx)                 MaxPool2d(2) with the argmax byte  (+ maxpool2_nhwc_bwd)
    torch.ops.runet.bce_loss(prob, target)                       nn.BCELoss() mean, ATen clamp semantics (Main_Final.py:551)
    torch.ops.runet.cross_entropy(logits, target)                nn.CrossEntropyLoss() mean (train_water_segmentation.py:304)
    torch.ops.runet.seg_counts(pred, target, threshold)          per-image tp / predicted / target / agreement counts (Main_Final.py:519-547)
    torch.ops.runet.bilinear_resize(x, ho, wo)                   F.interpolate(bilinear, align_corners=False) (Main_Final.py:577-578)

The nn.Module surface (model.RobustUNet: ONE autograd node per network, explicit kernel sequences in blocks.py) does not go through
the dispatcher per leaf op - a dispatcher round trip costs ~20 us and a train step has ~650 launches; these registrations are the
operator-level surface for callers that compose the kernels themselves.  All ops run on the HIP device only (no CPU kernels are
registered: a CPU tensor raises NotImplementedError from the dispatcher).
"""
from __future__ import annotations

import torch
from torch.library import custom_op, register_autograd

from . import ops

_DEV = "cuda"


def _chk_nhwc(x):
    if x.dim() != 4 or x.dtype != torch.float32:
        raise ValueError("expected a float32 [N, H, W, C] tensor")


# ------------------------------------------------------------------------------------------------ convolution
@custom_op("runet::conv2d_nhwc", mutates_args=(), device_types=_DEV)
def conv2d_nhwc(x: torch.Tensor, w_hwio: torch.Tensor, bias: torch.Tensor | None, dilation: int = 1) -> torch.Tensor:
    _chk_nhwc(x)
    return ops.conv_fwd(x.contiguous(), w_hwio.contiguous(), bias, dil=int(dilation))


@conv2d_nhwc.register_fake
def _(x, w_hwio, bias, dilation=1):
    n, h, w, _ = x.shape
    return x.new_empty((n, h, w, w_hwio.shape[3]))


@custom_op("runet::conv2d_nhwc_dgrad", mutates_args=(), device_types=_DEV)
def conv2d_nhwc_dgrad(dy: torch.Tensor, w_hwio: torch.Tensor, dilation: int = 1) -> torch.Tensor:
    _chk_nhwc(dy)
    return ops.conv_dgrad(dy.contiguous(), w_hwio.contiguous(), dil=int(dilation))


@conv2d_nhwc_dgrad.register_fake
def _(dy, w_hwio, dilation=1):
    n, h, w, _ = dy.shape
    return dy.new_empty((n, h, w, w_hwio.shape[2]))


@custom_op("runet::conv2d_nhwc_wgrad", mutates_args=(), device_types=_DEV)
def conv2d_nhwc_wgrad(x: torch.Tensor, dy: torch.Tensor, kh: int, kw: int, dilation: int = 1) -> torch.Tensor:
    _chk_nhwc(x)
    return ops.conv_wgrad(x.contiguous(), dy.contiguous(), int(kh), int(kw), dil=int(dilation), on_side=False)


@conv2d_nhwc_wgrad.register_fake
def _(x, dy, kh, kw, dilation=1):
    return x.new_empty((kh, kw, x.shape[3], dy.shape[3]))


def _conv_setup(ctx, inputs, output):
    x, w, bias, dil = inputs
    ctx.save_for_backward(x, w)
    ctx.dil, ctx.has_bias = int(dil), bias is not None


def _conv_bwd(ctx, dy):
    x, w = ctx.saved_tensors
    dy = dy.contiguous()
    dx = torch.ops.runet.conv2d_nhwc_dgrad(dy, w, ctx.dil) if ctx.needs_input_grad[0] else None
    dw = torch.ops.runet.conv2d_nhwc_wgrad(x, dy, w.shape[0], w.shape[1], ctx.dil) if ctx.needs_input_grad[1] else None
    db = dy.sum(dim=(0, 1, 2)) if ctx.has_bias and ctx.needs_input_grad[2] else None
    return dx, dw, db, None


register_autograd("runet::conv2d_nhwc", _conv_bwd, setup_context=_conv_setup)


# ------------------------------------------------------------------------------------------------ transposed convolution (k2, s2)
@custom_op("runet::convt2x2s2_nhwc", mutates_args=(), device_types=_DEV)
def convt2x2s2_nhwc(x: torch.Tensor, w_hwio: torch.Tensor, bias: torch.Tensor | None) -> torch.Tensor:
    _chk_nhwc(x)
    return ops.convt_fwd(x.contiguous(), w_hwio.contiguous(), bias)


@convt2x2s2_nhwc.register_fake
def _(x, w_hwio, bias):
    n, h, w, _ = x.shape
    return x.new_empty((n, 2 * h, 2 * w, w_hwio.shape[3]))


@custom_op("runet::convt2x2s2_nhwc_dgrad", mutates_args=(), device_types=_DEV)
def convt2x2s2_nhwc_dgrad(dy: torch.Tensor, w_hwio: torch.Tensor) -> torch.Tensor:
    _chk_nhwc(dy)
    return ops.convt_dgrad(dy.contiguous(), w_hwio.contiguous())


@convt2x2s2_nhwc_dgrad.register_fake
def _(dy, w_hwio):
    n, h2, w2, _ = dy.shape
    return dy.new_empty((n, h2 // 2, w2 // 2, w_hwio.shape[2]))


@custom_op("runet::convt2x2s2_nhwc_wgrad", mutates_args=(), device_types=_DEV)
def convt2x2s2_nhwc_wgrad(x: torch.Tensor, dy: torch.Tensor) -> torch.Tensor:
    _chk_nhwc(x)
    return ops._convt_wgrad(x.contiguous(), dy.contiguous())


@convt2x2s2_nhwc_wgrad.register_fake
def _(x, dy):
    return x.new_empty((2, 2, x.shape[3], dy.shape[3]))


def _convt_setup(ctx, inputs, output):
    x, w, bias = inputs
    ctx.save_for_backward(x, w)
    ctx.has_bias = bias is not None


def _convt_bwd(ctx, dy):
    x, w = ctx.saved_tensors
    dy = dy.contiguous()
    dx = torch.ops.runet.convt2x2s2_nhwc_dgrad(dy, w) if ctx.needs_input_grad[0] else None
    dw = torch.ops.runet.convt2x2s2_nhwc_wgrad(x, dy) if ctx.needs_input_grad[1] else None
    db = dy.sum(dim=(0, 1, 2)) if ctx.has_bias and ctx.needs_input_grad[2] else None
    return dx, dw, db


register_autograd("runet::convt2x2s2_nhwc", _convt_bwd, setup_context=_convt_setup)


# ------------------------------------------------------------------------------------------------ max pooling
@custom_op("runet::maxpool2_nhwc", mutates_args=(), device_types=_DEV)
def maxpool2_nhwc(x: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
    _chk_nhwc(x)
    from . import blocks
    return blocks.maxpool_forward(x.contiguous())


@maxpool2_nhwc.register_fake
def _(x):
    n, h, w, c = x.shape
    return x.new_empty((n, h // 2, w // 2, c)), x.new_empty((n, h // 2, w // 2, c), dtype=torch.uint8)


@custom_op("runet::maxpool2_nhwc_bwd", mutates_args=(), device_types=_DEV)
def maxpool2_nhwc_bwd(dy: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    from . import blocks
    return blocks.maxpool_backward(dy.contiguous(), idx)


@maxpool2_nhwc_bwd.register_fake
def _(dy, idx):
    n, h, w, c = dy.shape
    return dy.new_empty((n, 2 * h, 2 * w, c))


def _pool_setup(ctx, inputs, output):
    ctx.save_for_backward(output[1])
    ctx.mark_non_differentiable(output[1])


def _pool_bwd(ctx, dy, _didx):
    (idx,) = ctx.saved_tensors
    return torch.ops.runet.maxpool2_nhwc_bwd(dy.contiguous(), idx)


register_autograd("runet::maxpool2_nhwc", _pool_bwd, setup_context=_pool_setup)


# ------------------------------------------------------------------------------------------------ losses / metrics / resize
@custom_op("runet::bce_loss", mutates_args=(), device_types=_DEV)
def bce_loss(prob: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    return ops.bce_loss(prob.detach(), target)


@bce_loss.register_fake
def _(prob, target):
    return prob.new_empty(())


@custom_op("runet::bce_loss_bwd", mutates_args=(), device_types=_DEV)
def bce_loss_bwd(prob: torch.Tensor, target: torch.Tensor, gout: torch.Tensor) -> torch.Tensor:
    prob, target = prob.contiguous(), target.contiguous().to(torch.float32)
    dprob = torch.empty_like(prob)
    ops.check(ops.lib.runet_bce_bwd(prob.data_ptr(), target.data_ptr(), gout.contiguous().data_ptr(), dprob.data_ptr(), prob.numel(), ops.stream()))
    return dprob


@bce_loss_bwd.register_fake
def _(prob, target, gout):
    return torch.empty_like(prob)


def _bce_setup(ctx, inputs, output):
    ctx.save_for_backward(*inputs)


def _bce_bwd(ctx, gout):
    prob, target = ctx.saved_tensors
    return torch.ops.runet.bce_loss_bwd(prob, target, gout), None


register_autograd("runet::bce_loss", _bce_bwd, setup_context=_bce_setup)


@custom_op("runet::cross_entropy", mutates_args=(), device_types=_DEV)
def cross_entropy(logits: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    return ops.cross_entropy(logits.detach(), target)


@cross_entropy.register_fake
def _(logits, target):
    return logits.new_empty(())


@custom_op("runet::cross_entropy_bwd", mutates_args=(), device_types=_DEV)
def cross_entropy_bwd(logits: torch.Tensor, target: torch.Tensor, gout: torch.Tensor) -> torch.Tensor:
    logits, target = logits.contiguous(), target.contiguous()
    n, c, h, w = logits.shape
    dz = torch.empty_like(logits)
    ops.check(ops.lib.runet_ce_bwd(logits.data_ptr(), target.data_ptr(), gout.contiguous().to(torch.float32).data_ptr(), dz.data_ptr(), n, c, h * w,
                                   ops.stream()))
    return dz


@cross_entropy_bwd.register_fake
def _(logits, target, gout):
    return torch.empty_like(logits)


def _ce_bwd(ctx, gout):
    logits, target = ctx.saved_tensors
    return torch.ops.runet.cross_entropy_bwd(logits, target, gout), None


register_autograd("runet::cross_entropy", _ce_bwd, setup_context=_bce_setup)


@custom_op("runet::seg_counts", mutates_args=(), device_types=_DEV)
def seg_counts(pred: torch.Tensor, target: torch.Tensor, threshold: float = 0.5) -> torch.Tensor:
    return ops.seg_counts(pred, target, float(threshold))


@seg_counts.register_fake
def _(pred, target, threshold=0.5):
    return pred.new_empty((pred.shape[0], 4), dtype=torch.int64)


@custom_op("runet::bilinear_resize", mutates_args=(), device_types=_DEV)
def bilinear_resize(x: torch.Tensor, ho: int, wo: int) -> torch.Tensor:
    return ops.bilinear_resize(x.detach(), (int(ho), int(wo)))


@bilinear_resize.register_fake
def _(x, ho, wo):
    return x.new_empty((x.shape[0], x.shape[1], ho, wo))


@custom_op("runet::bilinear_resize_bwd", mutates_args=(), device_types=_DEV)
def bilinear_resize_bwd(dy: torch.Tensor, h: int, w: int) -> torch.Tensor:
    dy = dy.contiguous()
    n, c, ho, wo = dy.shape
    dx = torch.empty((n, c, h, w), device=dy.device, dtype=torch.float32)
    ops.check(ops.lib.runet_bilinear_bwd(dy.data_ptr(), dx.data_ptr(), n * c, h, w, ho, wo, ops.stream()))
    return dx


@bilinear_resize_bwd.register_fake
def _(dy, h, w):
    return dy.new_empty((dy.shape[0], dy.shape[1], h, w))


def _bil_setup(ctx, inputs, output):
    ctx.hw = (inputs[0].shape[2], inputs[0].shape[3])


def _bil_bwd(ctx, dy):
    return torch.ops.runet.bilinear_resize_bwd(dy, ctx.hw[0], ctx.hw[1]), None, None


register_autograd("runet::bilinear_resize", _bil_bwd, setup_context=_bil_setup)

OPS = ("conv2d_nhwc", "conv2d_nhwc_dgrad", "conv2d_nhwc_wgrad", "convt2x2s2_nhwc", "convt2x2s2_nhwc_dgrad", "convt2x2s2_nhwc_wgrad",
       "maxpool2_nhwc", "maxpool2_nhwc_bwd", "bce_loss", "bce_loss_bwd", "cross_entropy", "cross_entropy_bwd", "seg_counts",
       "bilinear_resize", "bilinear_resize_bwd")
