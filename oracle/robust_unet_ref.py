"""ORACLE — test infrastructure only, never the product path.

CPU restatement (stock torch/ATen fp32 ops, functional style over a flat name->tensor
state) of the reference's Robust U-Net training hot path.  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this file; the
product package must never route through it.

Each function cites the reference lines it restates (paths relative to /root/reference):

  channel_attention   Main_Final.py:82-101
  spatial_attention   Main_Final.py:104-117
  attention_gate      Main_Final.py:120-148
  residual_block      Main_Final.py:151-196
  dilated_block       Main_Final.py:199-223
  forward             Main_Final.py:290-321   (RobustUNet.forward)
  param_spec/init     Main_Final.py:229-288   (RobustUNet.__init__/_initialize_weights)
  bce_mean            Main_Final.py:551,580   (nn.BCELoss(), log clamped at -100)
  adam_step           Main_Final.py:552,582   (torch.optim.Adam lr, weight_decay -> L2-coupled)
  seg_metrics         Main_Final.py:519-547   (ModelEvaluator.calculate_metrics)
  labelme_mask        Main_Final.py:62-78     (CoastalDataset.create_mask_from_labelme)

Pinning: the reference's own tests pin nothing (it has none); this restatement is pinned
by golden vectors generated in the build container from the reference itself
(`tests/golden/make_golden.py` imports /root/reference/Main_Final.py and dumps
`tests/golden/*.npz`); `tests/test_oracle_golden.py` checks this file against them.
"""
from __future__ import annotations

import importlib
import json
import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

_rng = importlib.import_module("eusipco-2026-robust-unet_amd.portable_rng")

BN_EPS = 1e-5
BN_MOMENTUM = 0.1
WATER_LABELS = ("water", "sea", "海水", "水体")


# ----------------------------------------------------------------------------- spec
def _rb_spec(prefix, cin, cout):
    s = [
        (f"{prefix}.conv1.weight", (cout, cin, 3, 3), "conv"),
        (f"{prefix}.bn1", cout, "bn"),
        (f"{prefix}.conv2.weight", (cout, cout, 3, 3), "conv"),
        (f"{prefix}.bn2", cout, "bn"),
        (f"{prefix}.ca.fc.0.weight", (cout // 16, cout, 1, 1), "conv"),
        (f"{prefix}.ca.fc.2.weight", (cout, cout // 16, 1, 1), "conv"),
        (f"{prefix}.sa.conv1.weight", (1, 2, 7, 7), "conv"),
    ]
    if cin != cout:
        s += [(f"{prefix}.shortcut.0.weight", (cout, cin, 1, 1), "conv"), (f"{prefix}.shortcut.1", cout, "bn")]
    return s


def _conv_b(prefix, cout, cin, k):
    return [(f"{prefix}.weight", (cout, cin, k, k), "conv"), (f"{prefix}.bias", (cout,), ("cbias", cin * k * k))]


def _att_spec(prefix, fg, fl, fint):
    return (_conv_b(f"{prefix}.W_g.0", fint, fg, 1) + [(f"{prefix}.W_g.1", fint, "bn")]
            + _conv_b(f"{prefix}.W_x.0", fint, fl, 1) + [(f"{prefix}.W_x.1", fint, "bn")]
            + _conv_b(f"{prefix}.psi.0", 1, fint, 1) + [(f"{prefix}.psi.1", 1, "bn")])


def _up_spec(prefix, cin, cout):
    return [(f"{prefix}.weight", (cin, cout, 2, 2), ("convT", cout * 4)),
            (f"{prefix}.bias", (cout,), ("cbias", cout * 4))]


def module_spec(n_channels=3, n_classes=1, base=64):
    """Modules in the reference's registration order (Main_Final.py:232-277)."""
    b = base
    s = []
    s += _rb_spec("inc", n_channels, b)
    s += _rb_spec("down1.1", b, 2 * b)
    s += _rb_spec("down2.1", 2 * b, 4 * b)
    s += _rb_spec("down3.1", 4 * b, 8 * b)
    for i, k in zip((1, 2, 3, 4), (1, 3, 3, 3)):
        s += _conv_b(f"bottleneck.1.conv{i}", 4 * b, 8 * b, k)
    s += [("bottleneck.1.bn", 16 * b, "bn")]
    s += _rb_spec("bottleneck.2", 16 * b, 16 * b)
    s += _att_spec("att4", 8 * b, 8 * b, 4 * b)
    s += _att_spec("att3", 4 * b, 4 * b, 2 * b)
    s += _att_spec("att2", 2 * b, 2 * b, b)
    s += _att_spec("att1", b, b, b // 2)
    s += _up_spec("up4", 16 * b, 8 * b) + _rb_spec("dec4", 16 * b, 8 * b)
    s += _up_spec("up3", 8 * b, 4 * b) + _rb_spec("dec3", 8 * b, 4 * b)
    s += _up_spec("up2", 4 * b, 2 * b) + _rb_spec("dec2", 4 * b, 2 * b)
    s += _up_spec("up1", 2 * b, b) + _rb_spec("dec1", 2 * b, b)
    s += _conv_b("outc.0", n_classes, b, 1)
    return s


def state_spec(n_channels=3, n_classes=1, base=64):
    """Flat (key, shape, dtype, kind) list in state_dict order (params and buffers)."""
    out = []
    for name, shape, kind in module_spec(n_channels, n_classes, base):
        if kind == "bn":
            c = shape
            out += [(f"{name}.weight", (c,), torch.float32, "bn_w"), (f"{name}.bias", (c,), torch.float32, "bn_b"),
                    (f"{name}.running_mean", (c,), torch.float32, "buf_mean"),
                    (f"{name}.running_var", (c,), torch.float32, "buf_var"),
                    (f"{name}.num_batches_tracked", (), torch.int64, "buf_nbt")]
        else:
            out.append((name, tuple(shape), torch.float32, kind))
    return out


def param_names(n_channels=3, n_classes=1, base=64):
    return [k for k, _, _, kind in state_spec(n_channels, n_classes, base) if not str(kind).startswith("buf")]


def init_state(n_channels=3, n_classes=1, base=64, seed=0, perturb_bn=True):
    """Portable re-statement of the reference initialisation *distributions*
    (Main_Final.py:282-288 + torch defaults for biases / ConvTranspose2d), drawn from the
    portable generator so any process can rebuild the identical state.
    perturb_bn=True additionally jitters BN gamma/beta (test coverage; gamma=1/beta=0 hides bugs)."""
    st = OrderedDict()
    for key, shape, dtype, kind in state_spec(n_channels, n_classes, base):
        s = _rng.name_seed(key, seed)
        if kind == "conv":  # kaiming_normal_(fan_out, relu): std = sqrt(2 / (Cout*kh*kw))
            std = math.sqrt(2.0 / (shape[0] * shape[2] * shape[3]))
            t = torch.from_numpy(_rng.normal_f32(shape, s, std))
        elif isinstance(kind, tuple) and kind[0] in ("cbias", "convT"):  # U(-1/sqrt(fan_in), +)
            bound = 1.0 / math.sqrt(kind[1])
            t = torch.from_numpy(_rng.uniform_f32(shape, s, -bound, bound))
        elif kind == "bn_w":
            t = torch.from_numpy(_rng.normal_f32(shape, s, 0.1, 1.0)) if perturb_bn else torch.ones(shape)
        elif kind == "bn_b":
            t = torch.from_numpy(_rng.normal_f32(shape, s, 0.1, 0.0)) if perturb_bn else torch.zeros(shape)
        elif kind == "buf_mean":
            t = torch.zeros(shape)
        elif kind == "buf_var":
            t = torch.ones(shape)
        elif kind == "buf_nbt":
            t = torch.zeros((), dtype=torch.int64)
        else:
            raise ValueError(kind)
        st[key] = t
    return st


DROPOUT_SITES = (("inc", 0.1), ("down1.1", 0.1), ("down2.1", 0.2), ("down3.1", 0.2), ("bottleneck.2", 0.3),
                 ("dec4", 0.2), ("dec3", 0.2), ("dec2", 0.1), ("dec1", 0.1))


def dropout_masks(n, base=64, seed=0):
    """Per-(n, c) Dropout2d keep-masks already scaled by 1/(1-p), keyed by block prefix."""
    chans = {"inc": base, "down1.1": 2 * base, "down2.1": 4 * base, "down3.1": 8 * base, "bottleneck.2": 16 * base,
             "dec4": 8 * base, "dec3": 4 * base, "dec2": 2 * base, "dec1": base}
    out = {}
    for name, p in DROPOUT_SITES:
        keep = _rng.bernoulli_keep((n, chans[name]), _rng.name_seed("dropout." + name, seed), p)
        out[name] = torch.from_numpy(keep / np.float32(1.0 - p))
    return out


# ----------------------------------------------------------------------------- blocks
def _bn(P, name, x, training):
    y = F.batch_norm(x, P[f"{name}.running_mean"], P[f"{name}.running_var"], P[f"{name}.weight"], P[f"{name}.bias"],
                     training, BN_MOMENTUM, BN_EPS)
    if training:
        P[f"{name}.num_batches_tracked"] += 1
    return y


def channel_attention(P, pre, x):
    w0, w2 = P[f"{pre}.fc.0.weight"], P[f"{pre}.fc.2.weight"]
    mlp = lambda v: F.conv2d(F.relu(F.conv2d(v, w0)), w2)
    return x * torch.sigmoid(mlp(F.adaptive_avg_pool2d(x, 1)) + mlp(F.adaptive_max_pool2d(x, 1)))


def spatial_attention(P, pre, x):
    m = torch.cat([x.mean(dim=1, keepdim=True), x.max(dim=1, keepdim=True)[0]], dim=1)
    return x * torch.sigmoid(F.conv2d(m, P[f"{pre}.conv1.weight"], padding=3))


def residual_block(P, pre, x, training, mask=None, taps=None):
    """mask: [N, C] dropout keep-mask already scaled by 1/(1-p) (None = no dropout)."""
    if f"{pre}.shortcut.0.weight" in P:
        res = _bn(P, f"{pre}.shortcut.1", F.conv2d(x, P[f"{pre}.shortcut.0.weight"]), training)
    else:
        res = x
    out = F.relu(_bn(P, f"{pre}.bn1", F.conv2d(x, P[f"{pre}.conv1.weight"], padding=1), training))
    if training and mask is not None:
        out = out * mask[:, :, None, None]
    out = _bn(P, f"{pre}.bn2", F.conv2d(out, P[f"{pre}.conv2.weight"], padding=1), training)
    out = channel_attention(P, f"{pre}.ca", out)
    out = spatial_attention(P, f"{pre}.sa", out)
    return F.relu(out + res)


def dilated_block(P, pre, x, training):
    ys = [F.conv2d(x, P[f"{pre}.conv1.weight"], P[f"{pre}.conv1.bias"])]
    for i, d in ((2, 1), (3, 2), (4, 4)):
        ys.append(F.conv2d(x, P[f"{pre}.conv{i}.weight"], P[f"{pre}.conv{i}.bias"], padding=d, dilation=d))
    return F.relu(_bn(P, f"{pre}.bn", torch.cat(ys, dim=1), training))


def attention_gate(P, pre, g, x, training):
    g1 = _bn(P, f"{pre}.W_g.1", F.conv2d(g, P[f"{pre}.W_g.0.weight"], P[f"{pre}.W_g.0.bias"]), training)
    x1 = _bn(P, f"{pre}.W_x.1", F.conv2d(x, P[f"{pre}.W_x.0.weight"], P[f"{pre}.W_x.0.bias"]), training)
    s = F.conv2d(F.relu(g1 + x1), P[f"{pre}.psi.0.weight"], P[f"{pre}.psi.0.bias"])
    return x * torch.sigmoid(_bn(P, f"{pre}.psi.1", s, training))


def forward(P, x, training=True, masks=None, taps=None):
    """x: float32 [N, C, H, W] (H, W multiples of 16).  Returns (prob, pre_sigmoid).
    taps (optional dict) collects the per-stage activations named in SURVEY.md section 8(c)."""
    masks = masks or {}
    rb = lambda pre, v: residual_block(P, pre, v, training, masks.get(pre))
    x1 = rb("inc", x)
    x2 = rb("down1.1", F.max_pool2d(x1, 2))
    x3 = rb("down2.1", F.max_pool2d(x2, 2))
    x4 = rb("down3.1", F.max_pool2d(x3, 2))
    xd = dilated_block(P, "bottleneck.1", F.max_pool2d(x4, 2), training)
    x5 = rb("bottleneck.2", xd)
    if taps is not None:
        taps.update(x1=x1, x2=x2, x3=x3, x4=x4, xd=xd, x5=x5)
    y = x5
    for lvl, skip in ((4, x4), (3, x3), (2, x2), (1, x1)):
        up = F.conv_transpose2d(y, P[f"up{lvl}.weight"], P[f"up{lvl}.bias"], stride=2)
        att = attention_gate(P, f"att{lvl}", up, skip, training)
        y = rb(f"dec{lvl}", torch.cat([att, up], dim=1))
        if taps is not None:
            taps[f"up{lvl}"], taps[f"att{lvl}"], taps[f"dec{lvl}"] = up, att, y
    logit = F.conv2d(y, P["outc.0.weight"], P["outc.0.bias"])
    return torch.sigmoid(logit), logit


class _BCEMean(torch.autograd.Function):
    """nn.BCELoss(reduction='mean') as ATen defines it: forward clamps both logs at -100;
    backward is (p - y) / max(p * (1 - p), 1e-12) / numel (no 0*inf NaNs at saturated p)."""

    @staticmethod
    def forward(ctx, prob, target):
        ctx.save_for_backward(prob, target)
        lp = torch.clamp(torch.log(prob), min=-100.0)
        l1p = torch.clamp(torch.log(1.0 - prob), min=-100.0)
        return -(target * lp + (1.0 - target) * l1p).mean()

    @staticmethod
    def backward(ctx, gout):
        prob, target = ctx.saved_tensors
        g = (prob - target) / torch.clamp(prob * (1.0 - prob), min=1e-12) / prob.numel()
        return g * gout, None


def bce_mean(prob, target):
    return _BCEMean.apply(prob, target)


def adam_step(params, grads, exp_avg, exp_avg_sq, step, lr=1e-4, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=1e-4):
    """torch.optim.Adam semantics (L2-coupled weight decay, bias-corrected), in place; `step` is 1-based."""
    bc1, bc2 = 1.0 - beta1 ** step, 1.0 - beta2 ** step
    with torch.no_grad():
        for p, g, m, v in zip(params, grads, exp_avg, exp_avg_sq):
            g = g + weight_decay * p
            m.mul_(beta1).add_(g, alpha=1.0 - beta1)
            v.mul_(beta2).addcmul_(g, g, value=1.0 - beta2)
            denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
            p.addcdiv_(m, denom, value=-lr / bc1)


def seg_metrics(pred, target, threshold=0.5):
    """Per-image metrics in float64 on the host: strict `>`; +1e-8 denominators; empty union -> IoU 0."""
    pb = (np.asarray(pred) > threshold).reshape(-1)
    tb = np.asarray(target).reshape(-1).astype(bool)
    inter = float(np.logical_and(pb, tb).sum())
    union = float(np.logical_or(pb, tb).sum())
    tp, fp, fn = inter, float(pb.sum()) - inter, float(tb.sum()) - inter
    prec, rec = tp / (tp + fp + 1e-8), tp / (tp + fn + 1e-8)
    return {"accuracy": float((pb == tb).mean()), "iou": inter / (union + 1e-8), "precision": prec, "recall": rec,
            "f1_score": 2 * prec * rec / (prec + rec + 1e-8)}


def labelme_mask(label_path, image_size):
    """image_size = (W, H) as PIL reports it.  Any failure -> all-zero mask (reference's broad except)."""
    from PIL import Image, ImageDraw
    try:
        with open(label_path, "r", encoding="utf-8") as f:
            data = json.load(f)
        canvas = Image.new("L", image_size, 0)
        draw = ImageDraw.Draw(canvas)
        for shape in data.get("shapes", []):
            if shape["label"].lower() in WATER_LABELS:
                pts = [(int(p[0]), int(p[1])) for p in shape["points"]]
                if len(pts) >= 3:
                    draw.polygon(pts, fill=1)
        return np.array(canvas, dtype=np.uint8)
    except Exception:
        return np.zeros((image_size[1], image_size[0]), dtype=np.uint8)


# ----------------------------------------------------------------------------- train-step helper
class OracleNet(torch.nn.Module):
    """Thin nn.Module shell over the functional oracle (CPU baseline timing, gloo DDP tests)."""

    def __init__(self, n_channels=3, n_classes=1, base_channels=64, seed=0, perturb_bn=False):
        super().__init__()
        st = init_state(n_channels, n_classes, base_channels, seed, perturb_bn)
        self.keys = list(st.keys())
        self.base = base_channels
        self._pnames = param_names(n_channels, n_classes, base_channels)
        pset = set(self._pnames)
        self.plist = torch.nn.ParameterList([torch.nn.Parameter(st[k]) for k in self._pnames])
        self._bufs = {k: v for k, v in st.items() if k not in pset}

    def state(self):
        P = dict(self._bufs)
        P.update({k: p for k, p in zip(self._pnames, self.plist)})
        return P

    def forward(self, x, masks=None):
        return forward(self.state(), x, self.training, masks)[0]
