"""CPU oracle for the DeepLabV3+ baseline (TEST INFRASTRUCTURE ONLY -- never imported by the product path).

Functional restatement, on torch CPU fp32 ops, of the reference's `DeepLabV3Plus` / `ASPP`
(/root/reference/Main_Final.py:325-433): a state-dict keyed forward so that the same dictionary drives the oracle, the
reference (tests/golden/make_golden.py loads it into the reference class) and the HIP modules.  Pinned by
tests/golden/deeplab_n2_s64.npz, which that script produced from the reference itself.
"""
from __future__ import annotations

import importlib
import math
from collections import OrderedDict

import torch
import torch.nn.functional as F

_rng = importlib.import_module("eusipco-2026-robust-unet_amd.portable_rng")

# (key prefix, kind, geometry)  kind: conv (cout, cin, k) | bn (c) | convT (cin, cout, k)
LAYERS = (
    ("conv1.0", "conv", (64, 3, 7)), ("conv1.1", "bn", 64),
    ("conv2.1", "conv", (128, 64, 3)), ("conv2.2", "bn", 128),
    ("conv3.0", "conv", (256, 128, 3)), ("conv3.1", "bn", 256),
    ("conv4.0", "conv", (512, 256, 3)), ("conv4.1", "bn", 512),
    ("aspp.conv1", "conv", (256, 512, 1)), ("aspp.conv2", "conv", (256, 512, 3)), ("aspp.conv3", "conv", (256, 512, 3)),
    ("aspp.conv4", "conv", (256, 512, 3)), ("aspp.conv5", "conv", (256, 512, 1)), ("aspp.conv_out", "conv", (256, 1280, 1)),
    ("aspp.bn", "bn", 256),
    ("decoder.0", "convT", (256, 128, 4)), ("decoder.1", "bn", 128),
    ("decoder.3", "convT", (128, 64, 4)), ("decoder.4", "bn", 64),
    ("decoder.6", "convT", (64, 32, 4)), ("decoder.7", "bn", 32),
    ("decoder.9", "convT", (32, 16, 4)), ("decoder.10", "bn", 16),
    ("decoder.12", "conv", (1, 16, 3)),
)


def init_state(seed=0, perturb_bn=True):
    """torch-default initialisation *distributions* (the reference's DeepLabV3Plus has no custom init: Conv2d/ConvTranspose2d
    kaiming_uniform(a=sqrt 5) = U(+-1/sqrt(fan_in)), bias U(+-1/sqrt(fan_in))), drawn from the portable generator."""
    st = OrderedDict()
    for name, kind, g in LAYERS:
        if kind == "bn":
            s = lambda k: _rng.name_seed(f"deeplab.{name}.{k}", seed)
            st[f"{name}.weight"] = torch.from_numpy(_rng.normal_f32((g,), s("weight"), 0.1, 1.0)) if perturb_bn else torch.ones(g)
            st[f"{name}.bias"] = torch.from_numpy(_rng.normal_f32((g,), s("bias"), 0.1, 0.0)) if perturb_bn else torch.zeros(g)
            st[f"{name}.running_mean"] = torch.zeros(g)
            st[f"{name}.running_var"] = torch.ones(g)
            st[f"{name}.num_batches_tracked"] = torch.zeros((), dtype=torch.int64)
            continue
        if kind == "conv":
            cout, cin, k = g
            shape, fan_in = (cout, cin, k, k), cin * k * k
        else:
            cin, cout, k = g
            shape, fan_in = (cin, cout, k, k), cout * k * k     # torch computes fan_in from dim 1 of the transposed weight
        bound = 1.0 / math.sqrt(fan_in)
        st[f"{name}.weight"] = torch.from_numpy(_rng.uniform_f32(shape, _rng.name_seed(f"deeplab.{name}.weight", seed), -bound, bound))
        st[f"{name}.bias"] = torch.from_numpy(_rng.uniform_f32((shape[0] if kind == "conv" else shape[1],),
                                                               _rng.name_seed(f"deeplab.{name}.bias", seed), -bound, bound))
    return st


def param_names():
    return [k for k in init_state(0, False) if not k.split(".")[-1].startswith(("running", "num_batches"))]


def _bn_relu(P, name, x, training):
    y = F.batch_norm(x, P[f"{name}.running_mean"], P[f"{name}.running_var"], P[f"{name}.weight"], P[f"{name}.bias"], training, 0.1, 1e-5)
    if training:
        P[f"{name}.num_batches_tracked"] += 1
    return F.relu(y)


def _conv(P, name, x, stride=1, padding=0, dilation=1):
    return F.conv2d(x, P[f"{name}.weight"], P[f"{name}.bias"], stride, padding, dilation)


def aspp(P, x, training=True):
    """ASPP (Main_Final.py:325-357) on its own: keys "aspp.*" of the state."""
    size = x.shape[2:]
    branches = [_conv(P, "aspp.conv1", x)]
    for i, d in ((2, 6), (3, 12), (4, 18)):
        branches.append(_conv(P, f"aspp.conv{i}", x, 1, d, d))
    pooled = _conv(P, "aspp.conv5", F.adaptive_avg_pool2d(x, 1))
    branches.append(F.interpolate(pooled, size=size, mode="bilinear", align_corners=False))
    return _bn_relu(P, "aspp.bn", _conv(P, "aspp.conv_out", torch.cat(branches, 1)), training)


def forward(P, x, training=True, taps=None, return_logit=False):
    """P: state dict (tensors; parameters may require grad).  -> sigmoid probabilities [N,1,H,W] (return_logit: (prob, logit), both attached)"""
    def tap(k, v):
        if taps is not None:
            taps[k] = v.detach().clone()
        return v
    x = tap("conv1", _bn_relu(P, "conv1.1", _conv(P, "conv1.0", x, 2, 3), training))
    x = F.max_pool2d(x, 3, 2, 1)
    x = tap("conv2", _bn_relu(P, "conv2.2", _conv(P, "conv2.1", x, 1, 1), training))
    x = tap("conv3", _bn_relu(P, "conv3.1", _conv(P, "conv3.0", x, 2, 1), training))
    x = tap("conv4", _bn_relu(P, "conv4.1", _conv(P, "conv4.0", x, 2, 1), training))
    x = tap("aspp", aspp(P, x, training))
    for i in (0, 3, 6, 9):
        x = F.conv_transpose2d(x, P[f"decoder.{i}.weight"], P[f"decoder.{i}.bias"], stride=2, padding=1)
        x = tap(f"decoder.{i}", _bn_relu(P, f"decoder.{i + 1}", x, training))
    logit = tap("logit", _conv(P, "decoder.12", x, 1, 1))
    return (torch.sigmoid(logit), logit) if return_logit else torch.sigmoid(logit)
