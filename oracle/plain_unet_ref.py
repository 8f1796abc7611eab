"""ORACLE - test infrastructure only, never the product path.

CPU restatement (stock torch/ATen fp32 ops, functional style over a flat name->tensor state) of the reference's plain 2-class U-Net
and its loss (paths relative to /root/reference):

  forward      train_water_segmentation.py:262-288   (UNet.forward; conv_block :252-260 = Conv3x3+bias, BatchNorm2d, ReLU, twice)
  state_spec   train_water_segmentation.py:222-250   (module tree / registration order)
  ce_mean      train_water_segmentation.py:304       (nn.CrossEntropyLoss(), mean over pixels)

Pinning: tests/golden/make_golden.py imports the reference file itself (with empty stub modules for cv2 / osgeo / torchvision, which
the build container lacks and the model code never touches) and stores golden vectors in tests/golden/unet_*.npz;
tests/test_oracle_golden.py checks this file against them.
"""
from __future__ import annotations

import importlib
import math
from collections import OrderedDict

import torch
import torch.nn.functional as F

_rng = importlib.import_module("eusipco-2026-robust-unet_amd.portable_rng")
BN_EPS, BN_MOMENTUM = 1e-5, 0.1
CH = (64, 128, 256, 512)


def _block_spec(pre, cin, cout):
    return [(f"{pre}.0", (cout, cin, 3, 3), "conv"), (f"{pre}.1", cout, "bn"), (f"{pre}.3", (cout, cout, 3, 3), "conv"), (f"{pre}.4", cout, "bn")]


def module_spec(n_channels=3, n_classes=2):
    s = _block_spec("enc1", n_channels, 64) + _block_spec("enc2", 64, 128) + _block_spec("enc3", 128, 256) + _block_spec("enc4", 256, 512)
    s += _block_spec("bottleneck", 512, 1024)
    for lvl, (cin, cout) in zip((4, 3, 2, 1), ((1024, 512), (512, 256), (256, 128), (128, 64))):
        s += [(f"upconv{lvl}", (cin, cout, 2, 2), "convT")] + _block_spec(f"dec{lvl}", cin, cout)
    s += [("final", (n_classes, 64, 1, 1), "conv")]
    return s


def init_state(n_channels=3, n_classes=2, seed=0, perturb_bn=True):
    """torch's default initialisation DISTRIBUTIONS (the reference class defines no initialiser) from the portable generator:
    conv / convT weights and biases U(-1/sqrt(fan_in), +), BatchNorm gamma = 1 / beta = 0 (jittered when perturb_bn)."""
    st = OrderedDict()
    for name, shape, kind in module_spec(n_channels, n_classes):
        if kind == "bn":
            c = shape
            st[f"{name}.weight"] = torch.from_numpy(_rng.normal_f32((c,), _rng.name_seed(name + ".weight", seed), 0.1, 1.0)) if perturb_bn else torch.ones(c)
            st[f"{name}.bias"] = torch.from_numpy(_rng.normal_f32((c,), _rng.name_seed(name + ".bias", seed), 0.1, 0.0)) if perturb_bn else torch.zeros(c)
            st[f"{name}.running_mean"], st[f"{name}.running_var"] = torch.zeros(c), torch.ones(c)
            st[f"{name}.num_batches_tracked"] = torch.zeros((), dtype=torch.int64)
        else:
            fan_in = shape[1] * shape[2] * shape[3]
            bound = 1.0 / math.sqrt(fan_in)
            st[f"{name}.weight"] = torch.from_numpy(_rng.uniform_f32(shape, _rng.name_seed(name + ".weight", seed), -bound, bound))
            cout = shape[1] if kind == "convT" else shape[0]
            st[f"{name}.bias"] = torch.from_numpy(_rng.uniform_f32((cout,), _rng.name_seed(name + ".bias", seed), -bound, bound))
    return st


def param_names(n_channels=3, n_classes=2):
    return [k for k in init_state(n_channels, n_classes) if not k.endswith(("running_mean", "running_var", "num_batches_tracked"))]


def _bn(P, name, x, training):
    y = F.batch_norm(x, P[f"{name}.running_mean"], P[f"{name}.running_var"], P[f"{name}.weight"], P[f"{name}.bias"], training, BN_MOMENTUM, BN_EPS)
    if training:
        P[f"{name}.num_batches_tracked"] += 1
    return y


def conv_block(P, pre, x, training):
    x = F.relu(_bn(P, f"{pre}.1", F.conv2d(x, P[f"{pre}.0.weight"], P[f"{pre}.0.bias"], padding=1), training))
    return F.relu(_bn(P, f"{pre}.4", F.conv2d(x, P[f"{pre}.3.weight"], P[f"{pre}.3.bias"], padding=1), training))


def forward(P, x, training=True):
    """x [N, 3, H, W] -> logits [N, classes, H, W]"""
    e1 = conv_block(P, "enc1", x, training)
    e2 = conv_block(P, "enc2", F.max_pool2d(e1, 2), training)
    e3 = conv_block(P, "enc3", F.max_pool2d(e2, 2), training)
    e4 = conv_block(P, "enc4", F.max_pool2d(e3, 2), training)
    y = conv_block(P, "bottleneck", F.max_pool2d(e4, 2), training)
    for lvl, skip in ((4, e4), (3, e3), (2, e2), (1, e1)):
        y = F.conv_transpose2d(y, P[f"upconv{lvl}.weight"], P[f"upconv{lvl}.bias"], stride=2)
        y = conv_block(P, f"dec{lvl}", torch.cat([y, skip], dim=1), training)
    return F.conv2d(y, P["final.weight"], P["final.bias"])


def ce_mean(logits, target):
    return F.cross_entropy(logits, target)
