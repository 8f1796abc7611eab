"""Decision-aware gradient parity for the two "next" models (plain 2-class U-Net, DeepLabV3+): the same two-part check tests/decisions.py
runs for the Robust U-Net, for networks whose only discrete decisions are ReLU masks, max-pool winners and - with a sigmoid + BCE head -
the fp32 saturation of the sigmoid.  Test infrastructure; the oracle is the checker.

  1. every decision on which the HIP step and the oracle differ must be a near-tie in the oracle (|ReLU input| or the gap to the pooling
     window's maximum within NEAR_TIE of the tensor's scale; a saturation flip within 4 ulp of 1.0);
  2. with the oracle evaluated under the HIP step's OWN decisions every gradient tensor must agree tightly.

The oracle modules call `F.relu` / `F.max_pool2d` through their module-level `F`; `run_oracle` swaps in a recorder that logs every decision
in call order and, given `forced` (the HIP step's decisions in the same order), takes them from there.
"""
import torch
import torch.nn.functional as F

from decisions import NEAR_TIE, SAT_SCALE, _ForcedProb, _saturated


class SeqRecorder:
    def __init__(self, forced=None):
        self.log = []            # (kind, decision, values) in call order
        self.forced = forced

    def __getattr__(self, name):
        return getattr(F, name)

    def _f(self):
        k = len(self.log)
        return None if self.forced is None or k >= len(self.forced) else self.forced[k]

    def relu(self, x, inplace=False):
        f = self._f()
        self.log.append(("relu", (x > 0).detach().clone(), x.detach().clone()))
        return F.relu(x) if f is None else x * f.to(x.dtype)

    def max_pool2d(self, x, kernel_size, stride=None, padding=0):
        f = self._f()
        y, idx = F.max_pool2d(x, kernel_size, stride, padding, return_indices=True)
        self.log.append(("pool", idx.detach().clone(), x.detach().clone()))
        if f is not None:        # f: flat index h * W + w into the input plane, [n, c, ho, wo]
            n, c, h, w = x.shape
            y = torch.gather(x.reshape(n, c, h * w), 2, f.reshape(n, c, -1)).reshape(y.shape)
        return y


def run_oracle(mod, step, forced=None, forced_prob=None):
    """step(rec) -> (loss, prob or None, logit or None) evaluated with `mod.F` replaced by a recorder.  Backward is run here.
    -> (recorder log, prob detached or None)"""
    rec = SeqRecorder(forced)
    real = mod.F
    mod.F = rec
    try:
        loss_fn, prob, logit = step(rec)
        if forced_prob is not None and prob is not None:
            where = _saturated(forced_prob) != _saturated(prob.detach())
            if bool(where.any()):
                prob = _ForcedProb.apply(logit, prob, forced_prob, where)
        loss_fn(prob).backward()
    finally:
        mod.F = real
    return rec.log, (prob.detach().clone() if prob is not None else None)


def pool_flat_2x2(code, w_in):
    """HIP 2x2 winner byte (dy * 2 + dx) [n, c, ho, wo] -> ATen flat index into the input plane"""
    n, c, ho, wo = code.shape
    oh = torch.arange(ho).view(1, 1, -1, 1)
    ow = torch.arange(wo).view(1, 1, 1, -1)
    return (2 * oh + code // 2) * w_in + 2 * ow + code % 2


def pool_flat_3s2(code, w_in):
    """HIP MaxPool2d(3, 2, 1) winner byte (ky * 3 + kx) -> ATen flat index"""
    n, c, ho, wo = code.shape
    oh = torch.arange(ho).view(1, 1, -1, 1)
    ow = torch.arange(wo).view(1, 1, 1, -1)
    return (2 * oh - 1 + code // 3) * w_in + 2 * ow - 1 + code % 3


def differing(hip, log, hip_prob=None, ref_prob=None):
    """hip: the HIP step's decisions in the oracle's call order (bool masks / flat pool indices).  -> [(index, kind, position, margin, scale)]"""
    assert len(hip) == len(log), (len(hip), len(log))
    flips = []
    for i, (h, (kind, ref, vals)) in enumerate(zip(hip, log)):
        scale = float(vals.abs().max())
        if kind == "relu":
            for pos in (h != ref).nonzero():
                pos = tuple(int(p) for p in pos)
                flips.append((i, kind, pos, abs(float(vals[pos])), scale))
        else:
            n, c, hh, ww = vals.shape
            flat = vals.reshape(n, c, hh * ww)
            for pos in (h != ref).nonzero():
                n_, c_, y_, x_ = (int(p) for p in pos)
                flips.append((i, kind, (n_, c_, y_, x_), float(flat[n_, c_, ref[n_, c_, y_, x_]] - flat[n_, c_, h[n_, c_, y_, x_]]), scale))
    if hip_prob is not None and ref_prob is not None:
        for pos in (_saturated(hip_prob) != _saturated(ref_prob)).nonzero():
            pos = tuple(int(p) for p in pos)
            flips.append((len(log), "sigmoid saturation", pos, abs(float(hip_prob[pos]) - float(ref_prob[pos])), SAT_SCALE))
    return flips


def assert_near_ties(flips):
    for i, kind, pos, margin, scale in flips:
        assert margin <= NEAR_TIE * scale, f"decision {i} ({kind}) at {pos}: the HIP step and the oracle differ where the oracle sees no tie (margin {margin:.3e}, scale {scale:.2e})"


def grad_errors(named_hip, named_ref, skip=()):
    """-> sorted [(max |err| / scale, name)]; scale = the reference tensor's largest magnitude, at least 1e-3 of the largest gradient anywhere
    (bias sums behind a BatchNorm cancel to rounding level)."""
    gmax = max(float(g.abs().max()) for g in named_ref.values())
    rows = []
    for k, g in named_ref.items():
        if k in skip:
            continue
        sc = max(float(g.abs().max()), 1e-3 * gmax)
        rows.append((float((named_hip[k] - g).abs().max()) / sc, k))
    rows.sort(reverse=True)
    return rows
