"""Forward / backward of the Robust U-Net building blocks as explicit kernel sequences.

Every function here takes NHWC fp32 device tensors (see ops.ld) and calls the C ABI directly;
the backward functions consume the context returned by the matching forward.  No autograd,
no torch compute ops: torch only allocates device memory.  model.py wraps these in
autograd.Functions so that `loss.backward()` on the drop-in nn.Module works.

Reference semantics: /root/reference/Main_Final.py ResidualBlock :151-196, DilatedBlock :199-223,
AttentionGate :120-148 + ConvTranspose2d + torch.cat :301-303, outc :274-277, MaxPool2d :235.
"""
from __future__ import annotations

import os

import torch

from . import ops
from ._lib import check, lib

BN_EPS = 1e-5
BN_MOMENTUM = 0.1


class Small:
    """Carves small per-channel scratch vectors out of pooled device buffers (fewer allocator calls)."""

    CHUNK = 1 << 16

    def __init__(self, device):
        self.device = device
        self.buf = None
        self.off = 0

    def _take(self, n, dtype):
        n4 = (n + 3) // 4 * 4
        if n4 > self.CHUNK:
            return torch.empty(n, device=self.device, dtype=dtype)
        if self.buf is None or self.off + n4 > self.CHUNK:
            self.buf = torch.empty(self.CHUNK, device=self.device, dtype=torch.float32)
            self.off = 0
        t = self.buf[self.off:self.off + n]
        self.off += n4
        return t if dtype == torch.float32 else t.view(dtype)

    def f(self, n):
        return self._take(n, torch.float32)

    def i(self, n):
        return self._take(n, torch.int32)


_scratch = {}


def scratch(nfloats, device):
    """Reduction workspace (partials) per device AND stream (forward branches / weight gradients run beside the main chain); grows monotonically."""
    return ops.grow(_scratch, (device.index, ops.stream()), nfloats, device, 1 << 21)


def _ws(n, hw, c, device):
    return scratch(lib.runet_reduce_workspace_floats(n, hw, c), device)


_zero_vec = {}


def zeros(n, device):
    """Read-only vector of zeros (eval-mode BatchNorm backward: the batch-statistics terms of dx vanish)."""
    buf = _zero_vec.get(device.index)
    if buf is None or buf.numel() < n:
        buf = torch.zeros(max(int(n), 4096), device=device, dtype=torch.float32)
        _zero_vec[device.index] = buf
    return buf


class DictSink:
    """Default gradient sink: fresh device tensors, collected in a dict keyed by parameter name."""

    def __init__(self, device):
        self.device = device
        self.g = {}

    def buf(self, prefix, items):
        """items: [(param name, physical shape)] laid out back to back -> flat float32 view over all of them."""
        total = sum(_numel(sh) for _, sh in items)
        flat = torch.empty(total, device=self.device, dtype=torch.float32)
        off = 0
        for name, sh in items:
            n = _numel(sh)
            self.g[prefix + name] = flat[off:off + n].view(sh)
            off += n
        return flat


def _numel(shape):
    n = 1
    for d in shape:
        n *= d
    return n


class BNState:
    """Physical handles of one BatchNorm2d (parameters + buffers)."""
    __slots__ = ("weight", "bias", "running_mean", "running_var", "nbt")

    def __init__(self, weight, bias, running_mean, running_var, nbt):
        self.weight, self.bias, self.running_mean, self.running_var, self.nbt = weight, bias, running_mean, running_var, nbt


FUSED_BN_STATS = os.environ.get("RUNET_BN_STATS_3", "0") != "1"      # RUNET_BN_STATS_3=1: statistics / combine / finalize as three launches


def bn_coeff(x, bn: BNState, training, sm: Small, want_minmax=False, stats_hook=None, fused=None):
    """Batch (training) or running (eval) statistics -> (scale, shift, save_mean, save_invstd[, per-(n,c) stats]).
    fused: the dict the producing convolution filled (ops.conv_fwd(stats=...)): its epilogue already holds (count, mean, M2) partials of x."""
    n, h, w, c = x.shape
    hw = h * w
    st = ops.stream()
    scale, shift, mean, invstd = sm.f(c), sm.f(c), sm.f(c), sm.f(c)
    nc = None
    if training and not want_minmax and stats_hook is None and FUSED_BN_STATS and fused and "part" in fused:
        check(lib.runet_bn_stats_finalize(fused["part"].data_ptr(), fused["nparts"], c, bn.weight.data_ptr(), bn.bias.data_ptr(),
                                          bn.running_mean.data_ptr(), bn.running_var.data_ptr(), bn.nbt.data_ptr(), BN_MOMENTUM, BN_EPS,
                                          scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), invstd.data_ptr(), st))
        return scale, shift, mean, invstd, None
    if training and not want_minmax and stats_hook is None and FUSED_BN_STATS:
        # no per-image statistics wanted: partials -> batch statistics -> coefficients in two launches
        check(lib.runet_bn_stats(x.data_ptr(), ops.ld(x), n, hw, c, _ws(n, hw, c, x.device).data_ptr(), bn.weight.data_ptr(), bn.bias.data_ptr(),
                                 bn.running_mean.data_ptr(), bn.running_var.data_ptr(), bn.nbt.data_ptr(), BN_MOMENTUM, BN_EPS,
                                 scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), invstd.data_ptr(), st))
        return scale, shift, mean, invstd, None
    if training or want_minmax:
        mean_nc, m2_nc = sm.f(n * c), sm.f(n * c)
        if want_minmax:
            max_nc, min_nc, imax, imin = sm.f(n * c), sm.f(n * c), sm.i(n * c), sm.i(n * c)
            nc = (mean_nc, m2_nc, max_nc, min_nc, imax, imin)
        ws = _ws(n, hw, c, x.device)
        check(lib.runet_chan_stats(x.data_ptr(), ops.ld(x), n, hw, c, ws.data_ptr(), mean_nc.data_ptr(), m2_nc.data_ptr(),
                                   nc[2].data_ptr() if nc else None, nc[3].data_ptr() if nc else None,
                                   nc[4].data_ptr() if nc else None, nc[5].data_ptr() if nc else None, int(want_minmax), st))
        mean_all, m2_all, n_eff = mean_nc, m2_nc, n
        if stats_hook is not None and training:   # SyncBN: every rank's per-image (mean, M2) rows, Chan-combined by bn_finalize
            mean_all, m2_all, n_eff = stats_hook.gather_stats(mean_nc, m2_nc, n, c)
    if training:
        check(lib.runet_bn_finalize(mean_all.data_ptr(), m2_all.data_ptr(), n_eff, c, hw, bn.weight.data_ptr(), bn.bias.data_ptr(),
                                    bn.running_mean.data_ptr(), bn.running_var.data_ptr(), bn.nbt.data_ptr(), BN_MOMENTUM, BN_EPS, 1,
                                    scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), invstd.data_ptr(), st))
    else:
        check(lib.runet_bn_finalize(None, None, n, c, hw, bn.weight.data_ptr(), bn.bias.data_ptr(), bn.running_mean.data_ptr(),
                                    bn.running_var.data_ptr(), None, BN_MOMENTUM, BN_EPS, 0, scale.data_ptr(), shift.data_ptr(),
                                    mean.data_ptr(), invstd.data_ptr(), st))
    return scale, shift, mean, invstd, nc


def bn_coeff_pair(xa, bna: BNState, xb, bnb: BNState, training, sm: Small, stats_hook=None):
    """Two BatchNorms whose inputs are ready at the same point of the pass.  Under SyncBN their per-image statistics travel in ONE
    all-gather (the messages are latency-bound: a ResidualBlock's shortcut with its bn1, an attention gate's W_g with its W_x); otherwise
    two plain bn_coeff calls.  -> ((scale, shift, mean, invstd) of a, the same of b)"""
    if stats_hook is None or not training:
        return bn_coeff(xa, bna, training, sm)[:4], bn_coeff(xb, bnb, training, sm)[:4]
    st = ops.stream()
    local = []
    for x in (xa, xb):
        n, h, w, c = x.shape
        mean_nc, m2_nc = sm.f(n * c), sm.f(n * c)
        check(lib.runet_chan_stats(x.data_ptr(), ops.ld(x), n, h * w, c, _ws(n, h * w, c, x.device).data_ptr(), mean_nc.data_ptr(), m2_nc.data_ptr(),
                                   None, None, None, None, 0, st))
        local.append((mean_nc, m2_nc, n, c))
    res = []
    for x, bn, (mean_all, m2_all, n_eff) in zip((xa, xb), (bna, bnb), stats_hook.gather_stats_many(local)):
        n, h, w, c = x.shape
        scale, shift, mean, invstd = sm.f(c), sm.f(c), sm.f(c), sm.f(c)
        check(lib.runet_bn_finalize(mean_all.data_ptr(), m2_all.data_ptr(), n_eff, c, h * w, bn.weight.data_ptr(), bn.bias.data_ptr(),
                                    bn.running_mean.data_ptr(), bn.running_var.data_ptr(), bn.nbt.data_ptr(), BN_MOMENTUM, BN_EPS, 1,
                                    scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), invstd.data_ptr(), st))
        res.append((scale, shift, mean, invstd))
    return res[0], res[1]


def bn_apply(x, scale, shift, mask=None, relu=False, out=None):
    n, h, w, c = x.shape
    if out is None:
        out = ops.empty_nhwc(n, h, w, c, x)
    check(lib.runet_bn_apply(x.data_ptr(), ops.ld(x), out.data_ptr(), ops.ld(out), n * h * w, h * w, c, scale.data_ptr(), shift.data_ptr(),
                             mask.data_ptr() if mask is not None else None, int(relu), ops.stream()))
    return out


def bn_bwd_reduce(dy, x, mean, invstd, scale, sums, act=None, mask=None, relu_shift=None):
    """First half of bn_backward: the LOCAL (dgamma | dbeta) sums into `sums` [2c]."""
    n, h, w, c = x.shape
    hw = h * w
    if relu_shift is not None:
        act = None
    actp, lda = (act.data_ptr(), ops.ld(act)) if act is not None else (None, 0)
    rsc, rsh = (scale.data_ptr(), relu_shift.data_ptr()) if relu_shift is not None else (None, None)
    check(lib.runet_bn_bwd_reduce(dy.data_ptr(), ops.ld(dy), x.data_ptr(), ops.ld(x), actp, lda, n, hw, c, mean.data_ptr(), invstd.data_ptr(),
                                  mask.data_ptr() if mask is not None else None, _ws(n, hw, c, x.device).data_ptr(), sums.data_ptr(), rsc, rsh,
                                  ops.stream()))


def bn_bwd_apply(dy, x, mean, invstd, scale, use, m_total, act=None, mask=None, out=None, relu_shift=None):
    """Second half: dx from the sums `use` that enter it (local, all-reduced with the global element count m_total, or zeros in eval mode)."""
    n, h, w, c = x.shape
    hw = h * w
    if relu_shift is not None:
        act = None
    actp, lda = (act.data_ptr(), ops.ld(act)) if act is not None else (None, 0)
    if out is None:
        out = ops.empty_nhwc(n, h, w, c, x)
    check(lib.runet_bn_bwd_apply(dy.data_ptr(), ops.ld(dy), x.data_ptr(), ops.ld(x), actp, lda, out.data_ptr(), ops.ld(out), n * hw, hw, c,
                                 mean.data_ptr(), invstd.data_ptr(), scale.data_ptr(), use.data_ptr(), mask.data_ptr() if mask is not None else None,
                                 m_total, relu_shift.data_ptr() if relu_shift is not None else None, ops.stream()))
    return out


def bn_backward(dy, x, mean, invstd, scale, sums, act=None, mask=None, out=None, sync=None, relu_shift=None, training=True):
    """sums: [2c] destination for (dgamma | dbeta), always the LOCAL sums.  act/mask: fused relu(+dropout) backward from the saved
    activation; relu_shift (the forward's shift vector, with `scale` the forward's scale): the same from x alone, act is not read.
    sync (SyncBN): dx uses the all-reduced sums and the global element count.  training=False (the forward normalised with the
    RUNNING statistics, `mean` / `invstd` are those): BatchNorm is a per-channel affine map, dx = dy * scale with no batch-statistics
    terms, while dgamma / dbeta keep their form (sums over dy * xhat and dy).  -> dx"""
    n, h, w, c = x.shape
    bn_bwd_reduce(dy, x, mean, invstd, scale, sums, act, mask, relu_shift)
    if not training:
        use, m_total = zeros(2 * c, x.device), 0
    else:
        use, m_total = (sums, 0) if sync is None else sync.reduce_sums(sums, n * h * w)
    return bn_bwd_apply(dy, x, mean, invstd, scale, use, m_total, act, mask, out, relu_shift)


CHAN_SUM_ON_SIDE = os.environ.get("RUNET_CHAN_SUM_MAIN", "0") != "1"
# the shortcut BatchNorm's backward sums taken in rb_bwd2 (runet_rb_bwd2_bn) instead of by bn_bwd_reduce(dv, r): one tensor read and one launch
# less per block, but measured SLOWER in the step (A/B 545.0 vs 540.5 img/s: the reduction kernel gets twice the LDS and registers, the pass it
# replaces ran at HBM speed) - opt-in (RUNET_FUSED_SHORTCUT_BN_SUMS=1), kept equal by tests/test_gpu_blocks.py
FUSED_SHORTCUT_BN_SUMS = os.environ.get("RUNET_FUSED_SHORTCUT_BN_SUMS", "0") == "1"
FUSED_GATE_BN_SUMS = os.environ.get("RUNET_NO_FUSED_GATE_BN_SUMS", "0") != "1"      # the attention gates' BatchNorm-backward sums taken in ag_bwd2


def chan_sum(x, out):
    """Per-channel sum over all pixels (bias gradients).  Nothing in the backward chain waits for a bias gradient, so inside the backward
    pass (ops.wgrad_side_stream active) the two launches go to the weight-gradient stream like the weight gradients themselves."""
    n, h, w, c = x.shape

    def run():
        ws = _ws(n, h * w, c, x.device)
        check(lib.runet_chan_sum(x.data_ptr(), ops.ld(x), n * h * w, c, ws.data_ptr(), out.data_ptr(), 0, ops.stream()))
        return out
    if CHAN_SUM_ON_SIDE and ops.side_stream() is not None:
        return ops._on_side(run, (x, out))
    return run()


# =============================================================================== ResidualBlock
class RBParams:
    __slots__ = ("w1", "bn1", "w2", "bn2", "w0p", "w2p", "wsa", "ws", "bns", "cin_w")

    def __init__(self, w1, bn1, w2, bn2, w0p, w2p, wsa, ws=None, bns=None):
        self.w1, self.bn1, self.w2, self.bn2, self.w0p, self.w2p, self.wsa, self.ws, self.bns = w1, bn1, w2, bn2, w0p, w2p, wsa, ws, bns
        self.cin_w = w1.shape[2]


def rb_forward(x, p: RBParams, training, mask=None, save=True, stats_hook=None):
    """x: [N,H,W,Cx] with Cx >= cin_w (stem: RGB zero-padded to 4).  -> (out [N,H,W,C], ctx or None)"""
    n, h, w, _ = x.shape
    c = p.w1.shape[3]
    cr = p.w0p.shape[3]
    hw, P = h * w, n * h * w
    st = ops.stream()
    sm = Small(x.device)
    kv1, kv2 = ({}, {}) if save else (None, None)      # F(4x4) layers: the transformed inputs are kept for the weight gradients
    t1, br, fstem = None, None, (None, None)
    if p.ws is not None:
        if ops._stem_case(x.shape[3], p.cin_w, c, 3) and p.ws.shape[2] == p.cin_w:
            fstem = ({}, {}) if (training and stats_hook is None) else (None, None)
            t1, r = ops.stem_conv(x, p.w1, p.ws, stats3=fstem[0], stats1=fstem[1])       # RGB stem: conv1 and the shortcut convolution in one launch
            if stats_hook is None:
                ss, hs, mean_s, invstd_s, _ = bn_coeff(r, p.bns, training, sm, fused=fstem[1])
        elif stats_hook is not None:
            r = ops.conv_fwd(x, p.ws)                  # SyncBN: no branch; the shortcut's statistics share bn1's message below
        else:
            # the shortcut branch (1x1 convolution + its BatchNorm statistics) runs beside conv1 .. the attention maps, joins at rb_out
            sm.f(4)                                    # the arena's buffer is allocated on the main stream
            r = ops.main_pool(ops.empty_nhwc(n, h, w, c, x))          # ... and so is the branch's result (side_branch.join)
            br = ops.side_branch(stats_hook is None)
            with br:
                fs = {} if training else None
                ops.conv_fwd(x, p.ws, out=r, stats=fs)
                ss, hs, mean_s, invstd_s, _ = bn_coeff(r, p.bns, training, sm, fused=fs)
    else:
        r, ss, hs, mean_s, invstd_s = x, None, None, None, None
    f1 = {} if (training and stats_hook is None) else None
    if t1 is None:
        t1 = ops.conv_fwd(x, p.w1, keep_v=kv1, stats=f1)
    elif fstem[0]:
        f1 = fstem[0]                              # the stem kernel took bn1's statistics in its epilogue
    if p.ws is not None and stats_hook is not None:
        (ss, hs, mean_s, invstd_s), (s1, h1, mean1, invstd1) = bn_coeff_pair(r, p.bns, t1, p.bn1, training, sm, stats_hook)
    else:
        s1, h1, mean1, invstd1, _ = bn_coeff(t1, p.bn1, training, sm, stats_hook=stats_hook, fused=f1)
    use_mask = mask if training else None
    if ops.fuses_act_input(t1, p.w2):
        # conv2 on the F(4x4) path: BatchNorm + ReLU + Dropout2d ride in its input transform's loads, a1 is never written (rb_a1 recomputes it
        # for whoever wants to look at it); the backward pass needs t1 and the kept V only
        a1 = None
        t2 = ops.conv_fwd(t1, p.w2, keep_v=kv2, pre=(s1, h1, use_mask))
    else:
        a1 = bn_apply(t1, s1, h1, use_mask, relu=True)
        t2 = ops.conv_fwd(a1, p.w2, keep_v=kv2)
    if not save:
        del t1
    s2, h2, mean2, invstd2, nc = bn_coeff(t2, p.bn2, training, sm, want_minmax=True, stats_hook=stats_hook)
    mean_nc, _, max_nc, min_nc, imax, imin = nc
    A, B, ca, avg, mx, tval = (sm.f(n * c) for _ in range(6))
    idx = sm.i(n * c)
    check(lib.runet_ca_coeff(mean_nc.data_ptr(), max_nc.data_ptr(), min_nc.data_ptr(), imax.data_ptr(), imin.data_ptr(), s2.data_ptr(),
                             h2.data_ptr(), p.w0p.data_ptr(), p.w2p.data_ptr(), n, c, cr, A.data_ptr(), B.data_ptr(), ca.data_ptr(),
                             avg.data_ptr(), mx.data_ptr(), idx.data_ptr(), tval.data_ptr(), st))
    smap = torch.empty((P, 2), device=x.device, dtype=torch.float32)
    amax = torch.empty(P, device=x.device, dtype=torch.int32)
    check(lib.runet_sa_reduce(t2.data_ptr(), ops.ld(t2), A.data_ptr(), B.data_ptr(), P, hw, c, smap.data_ptr(), amax.data_ptr(), st))
    sa = torch.empty(P, device=x.device, dtype=torch.float32)
    check(lib.runet_sa_conv7(smap.data_ptr(), p.wsa.data_ptr(), sa.data_ptr(), n, h, w, st))
    out = ops.empty_nhwc(n, h, w, c, x)
    if br is not None:
        br.join(r)
    check(lib.runet_rb_out(t2.data_ptr(), ops.ld(t2), A.data_ptr(), B.data_ptr(), sa.data_ptr(), r.data_ptr(), ops.ld(r),
                           ss.data_ptr() if ss is not None else None, hs.data_ptr() if hs is not None else None, out.data_ptr(),
                           ops.ld(out), P, hw, c, st))
    if not save:
        return out, None
    ctx = dict(x=x, r=r, t1=t1, a1=a1, t2=t2, out=out, mask=use_mask, p=p, training=training, sync=stats_hook if training else None, s1=s1, h1=h1, mean1=mean1, invstd1=invstd1, s2=s2, h2=h2,
               mean2=mean2, invstd2=invstd2, ss=ss, mean_s=mean_s, invstd_s=invstd_s, A=A, B=B, ca=ca, avg=avg, mx=mx, idx=idx,
               tval=tval, mean_nc=mean_nc, smap=smap, amax=amax, sa=sa, v1=kv1.get("V"), v2=kv2.get("V"))
    return out, ctx


def rb_a1(ctx):
    """The activation between conv1 and conv2, relu(bn1(conv1 x)) * dropout mask: saved, or - where conv2's input transform produced it on the
    fly (rb_forward) - evaluated by the same expression (bn_apply_kernel and the fused transform share bn_pre: identical bits)."""
    if ctx["a1"] is not None:
        return ctx["a1"]
    return bn_apply(ctx["t1"], ctx["s1"], ctx["h1"], ctx["mask"], relu=True)


def rb_backward(ctx, dout, sink, pre="", need_dx=True):
    """Parameter gradients go to `sink` (names prefixed with `pre`).  -> dx or None"""
    p: RBParams = ctx["p"]
    x, r, t1, a1, t2, out = ctx["x"], ctx["r"], ctx["t1"], ctx["a1"], ctx["t2"], ctx["out"]
    n, h, w, c = out.shape
    cr = p.w0p.shape[3]
    hw, P = h * w, n * h * w
    st = ops.stream()
    dev = x.device
    sm = Small(dev)
    A, B, sa, smap, amax = ctx["A"], ctx["B"], ctx["sa"], ctx["smap"], ctx["amax"]
    dv = ops.empty_nhwc(n, h, w, c, x)
    dq = torch.empty(P, device=dev, dtype=torch.float32)
    check(lib.runet_rb_bwd1(dout.data_ptr(), ops.ld(dout), out.data_ptr(), ops.ld(out), t2.data_ptr(), ops.ld(t2), A.data_ptr(), B.data_ptr(),
                            sa.data_ptr(), dv.data_ptr(), ops.ld(dv), dq.data_ptr(), P, hw, c, st))
    dsm = torch.empty((P, 2), device=dev, dtype=torch.float32)
    dwsa = sink.buf(pre, [("sa.conv1.weight", (7, 7, 2, 1))])
    ws = scratch(lib.runet_sa_conv7_bwd_workspace_floats(n, h, w), dev)
    check(lib.runet_sa_conv7_bwd(smap.data_ptr(), dq.data_ptr(), p.wsa.data_ptr(), dsm.data_ptr(), dwsa.data_ptr(), ws.data_ptr(), n, h, w, st))
    sdu, sdut = sm.f(n * c), sm.f(n * c)
    sync, tr = ctx["sync"], ctx["training"]
    sums_s_fused = None
    if p.ws is not None and FUSED_SHORTCUT_BN_SUMS and not (sync is not None and tr):
        # dv is the shortcut BatchNorm's incoming gradient: its backward reduction rides in this pass (one read of r instead of a pass over dv and r)
        sums_s_fused = sink.buf(pre, [("shortcut.1.weight", (c,)), ("shortcut.1.bias", (c,))])
        ws = scratch(lib.runet_rb_bwd2_bn_workspace_floats(n, hw, c), dev)
        check(lib.runet_rb_bwd2_bn(dv.data_ptr(), ops.ld(dv), t2.data_ptr(), ops.ld(t2), sa.data_ptr(), dsm.data_ptr(), amax.data_ptr(), r.data_ptr(),
                                   ops.ld(r), ctx["mean_s"].data_ptr(), ctx["invstd_s"].data_ptr(), n, hw, c, ws.data_ptr(), ws.numel(),
                                   sdu.data_ptr(), sdut.data_ptr(), sums_s_fused.data_ptr(), st))
    else:
        ws = _ws(n, hw, c, dev)
        check(lib.runet_rb_bwd2(dv.data_ptr(), ops.ld(dv), t2.data_ptr(), ops.ld(t2), sa.data_ptr(), dsm.data_ptr(), amax.data_ptr(), n, hw, c,
                                ws.data_ptr(), sdu.data_ptr(), sdut.data_ptr(), st))
    davg, dmx = sm.f(n * c), sm.f(n * c)
    sums2 = sink.buf(pre, [("bn2.weight", (c,)), ("bn2.bias", (c,))])
    dw0p = sink.buf(pre, [("ca.fc.0.weight", (1, 1, c, cr))])
    dw2p = sink.buf(pre, [("ca.fc.2.weight", (1, 1, cr, c))])
    ws = scratch(lib.runet_ca_bwd_workspace_floats(n, c, cr), dev)
    check(lib.runet_ca_bwd(sdu.data_ptr(), sdut.data_ptr(), ctx["s2"].data_ptr(), ctx["h2"].data_ptr(), ctx["ca"].data_ptr(),
                           ctx["avg"].data_ptr(), ctx["mx"].data_ptr(), p.w0p.data_ptr(), p.w2p.data_ptr(), ctx["mean_nc"].data_ptr(),
                           ctx["tval"].data_ptr(), ctx["mean2"].data_ptr(), ctx["invstd2"].data_ptr(), n, c, cr, ws.data_ptr(),
                           davg.data_ptr(), dmx.data_ptr(), sums2.data_ptr(), dw0p.data_ptr(), dw2p.data_ptr(), st))
    dt2 = ops.empty_nhwc(n, h, w, c, x)
    use_s = None
    if not tr:
        use2, m_total = zeros(2 * c, dev), 0
    elif sync is None:
        use2, m_total = sums2, 0
    elif p.ws is not None:                   # SyncBN: the shortcut BatchNorm's sums (dv is final since rb_bwd1) share bn2's all-reduce
        sums_s = sink.buf(pre, [("shortcut.1.weight", (c,)), ("shortcut.1.bias", (c,))])
        bn_bwd_reduce(dv, r, ctx["mean_s"], ctx["invstd_s"], ctx["ss"], sums_s)
        (use2, m_total), (use_s, m_s) = sync.reduce_sums_many([sums2, sums_s], [P, P])
    else:
        use2, m_total = sync.reduce_sums(sums2, P)
    check(lib.runet_rb_bwd3(dv.data_ptr(), ops.ld(dv), t2.data_ptr(), ops.ld(t2), sa.data_ptr(), dsm.data_ptr(), amax.data_ptr(),
                            ctx["ca"].data_ptr(), davg.data_ptr(), dmx.data_ptr(), ctx["idx"].data_ptr(), ctx["mean2"].data_ptr(),
                            ctx["invstd2"].data_ptr(), ctx["s2"].data_ptr(), use2.data_ptr(), dt2.data_ptr(), ops.ld(dt2), P, hw, c, m_total, st))
    # the data gradient first: on the adjoint F(4x4) path it leaves Z = A dy A^T behind, which the weight gradient (side stream) reuses
    kz = {}
    da1 = ops.conv_dgrad(dt2, p.w2, keep_z=kz)
    if a1 is None:
        assert ctx.get("v2") is not None, "conv2 took its activation on the fly: its weight gradient needs the kept V"
        a1 = t1                                        # shape only: the weight gradient reads V
    ops.conv_wgrad(a1, dt2, 3, 3, out=sink.buf(pre, [("conv2.weight", (3, 3, c, c))]), v=ctx.get("v2"), z=kz.get("Z"))
    ctx["v2"] = None
    del dt2, kz
    sums1 = sink.buf(pre, [("bn1.weight", (c,)), ("bn1.bias", (c,))])
    kz1, dx1 = {}, None
    fused_dt1 = need_dx and p.cin_w == x.shape[3] and ops.fuses_bn_bwd_input(da1, p.w1)
    if fused_dt1:
        # conv1 on the adjoint F(4x4) path: the BatchNorm-backward dx rides in the loads of Z = A dt1 A^T, which both of conv1's gradients read
        # - dt1 itself is never written (two passes over the tensor and one launch less)
        bn_bwd_reduce(da1, t1, ctx["mean1"], ctx["invstd1"], ctx["s1"], sums1, None, ctx["mask"], ctx["h1"])
        if not tr:
            use1, m1 = zeros(2 * c, dev), 0
        else:
            use1, m1 = (sums1, 0) if sync is None else sync.reduce_sums(sums1, P)
        bn1 = dict(x=t1, mean=ctx["mean1"], invstd=ctx["invstd1"], scale=ctx["s1"], shift=ctx["h1"], sums=use1, m_total=m1, mask=ctx["mask"])
        if p.ws is not None:
            dx1 = ops.conv_dgrad(da1, p.w1, keep_z=kz1, bn=bn1)
        else:
            dx1 = ops.conv_dgrad(da1, p.w1, out=dv, accumulate=True, keep_z=kz1, bn=bn1)
        dt1 = da1                                      # shape only: the weight gradient reads Z
    else:
        dt1 = bn_backward(da1, t1, ctx["mean1"], ctx["invstd1"], ctx["s1"], sums1, mask=ctx["mask"], out=da1, sync=sync, relu_shift=ctx["h1"], training=tr)
        # the first block of the network (need_dx False) ends the backward chain: nothing is left on the main stream to overlap with, so its
        # last weight gradient runs there, next to the conv2 weight gradient still on the side stream
        if need_dx and p.ws is not None:
            dx1 = ops.conv_dgrad(dt1, p.w1, keep_z=kz1)
    ops.conv_wgrad(x, dt1, 3, 3, cin_w=p.cin_w, out=sink.buf(pre, [("conv1.weight", (3, 3, p.cin_w, c))]), v=ctx.get("v1"), on_side=need_dx,
                   z=kz1.get("Z"))
    ctx["v1"] = None
    dx = None
    if p.ws is not None:
        if use_s is not None:
            dr = bn_bwd_apply(dv, r, ctx["mean_s"], ctx["invstd_s"], ctx["ss"], use_s, m_s, out=dv)
        elif sums_s_fused is not None:
            dr = bn_bwd_apply(dv, r, ctx["mean_s"], ctx["invstd_s"], ctx["ss"], sums_s_fused if tr else zeros(2 * c, dev), 0, out=dv)
        else:
            sums_s = sink.buf(pre, [("shortcut.1.weight", (c,)), ("shortcut.1.bias", (c,))])
            dr = bn_backward(dv, r, ctx["mean_s"], ctx["invstd_s"], ctx["ss"], sums_s, out=dv, sync=sync, training=tr)
        ops.conv_wgrad(x, dr, 1, 1, cin_w=p.cin_w, out=sink.buf(pre, [("shortcut.0.weight", (1, 1, p.cin_w, c))]))
        if need_dx:
            dx = dx1
            ops.conv_dgrad(dr, p.ws, out=dx, accumulate=True)
    elif need_dx:
        dx = dv
        if not fused_dt1:
            ops.conv_dgrad(dt1, p.w1, out=dx, accumulate=True)
    return dx


# =============================================================================== standalone attention modules
# ChannelAttention (Main_Final.py:82-101) and SpatialAttention (:104-117) called on their own.  Inside ResidualBlock both are fused into
# the block's tail (rb_forward); on their own they are the same kernels with an identity BatchNorm in front (scale 1, shift 0) and
# neutral stand-ins for the other attention (per-pixel factor 1 / per-channel factor 1), so nothing new is computed differently.
def _consts(n, c, P, dev):
    one_c, zero_c = torch.ones(max(2 * c, n * c), device=dev), zeros(max(2 * c, n * c), dev)
    return one_c, zero_c


def ca_forward(x, w0p, w2p, save=True):
    """y = x * sigmoid(mlp(avgpool x) + mlp(maxpool x)).  x NHWC.  -> (y, ctx)"""
    n, h, w, c = x.shape
    cr = w0p.shape[3]
    dev, st = x.device, ops.stream()
    sm = Small(dev)
    mean_nc, m2_nc, max_nc, min_nc = sm.f(n * c), sm.f(n * c), sm.f(n * c), sm.f(n * c)
    imax, imin, idx = sm.i(n * c), sm.i(n * c), sm.i(n * c)
    check(lib.runet_chan_stats(x.data_ptr(), ops.ld(x), n, h * w, c, _ws(n, h * w, c, dev).data_ptr(), mean_nc.data_ptr(), m2_nc.data_ptr(),
                               max_nc.data_ptr(), min_nc.data_ptr(), imax.data_ptr(), imin.data_ptr(), 1, st))
    one, zero = _consts(n, c, 0, dev)
    A, B, ca, avg, mx, tval = (sm.f(n * c) for _ in range(6))
    check(lib.runet_ca_coeff(mean_nc.data_ptr(), max_nc.data_ptr(), min_nc.data_ptr(), imax.data_ptr(), imin.data_ptr(), one.data_ptr(),
                             zero.data_ptr(), w0p.data_ptr(), w2p.data_ptr(), n, c, cr, A.data_ptr(), B.data_ptr(), ca.data_ptr(), avg.data_ptr(),
                             mx.data_ptr(), idx.data_ptr(), tval.data_ptr(), st))
    y = bn_apply(x, one, zero, ca, relu=False)          # (x * 1 + 0) * ca[n, c]
    return y, (dict(x=x, w0p=w0p, w2p=w2p, ca=ca, avg=avg, mx=mx, idx=idx, tval=tval, mean_nc=mean_nc) if save else None)


def ca_backward(ctx, dy, sink, pre=""):
    x, w0p, w2p = ctx["x"], ctx["w0p"], ctx["w2p"]
    n, h, w, c = x.shape
    cr = w0p.shape[3]
    hw, P = h * w, n * h * w
    dev, st = x.device, ops.stream()
    sm = Small(dev)
    one, zero = _consts(n, c, P, dev)
    ones_p = torch.ones(P, device=dev)
    zeros_p2 = torch.zeros((P, 2), device=dev)
    none_p = torch.full((P,), -1, device=dev, dtype=torch.int32)
    sdu, sdut, davg, dmx = sm.f(n * c), sm.f(n * c), sm.f(n * c), sm.f(n * c)
    check(lib.runet_rb_bwd2(dy.data_ptr(), ops.ld(dy), x.data_ptr(), ops.ld(x), ones_p.data_ptr(), zeros_p2.data_ptr(), none_p.data_ptr(), n, hw, c,
                            _ws(n, hw, c, dev).data_ptr(), sdu.data_ptr(), sdut.data_ptr(), st))
    dw0p = sink.buf(pre, [("fc.0.weight", (1, 1, c, cr))])
    dw2p = sink.buf(pre, [("fc.2.weight", (1, 1, cr, c))])
    sums = sm.f(2 * c)
    ws = scratch(lib.runet_ca_bwd_workspace_floats(n, c, cr), dev)
    check(lib.runet_ca_bwd(sdu.data_ptr(), sdut.data_ptr(), one.data_ptr(), zero.data_ptr(), ctx["ca"].data_ptr(), ctx["avg"].data_ptr(),
                           ctx["mx"].data_ptr(), w0p.data_ptr(), w2p.data_ptr(), ctx["mean_nc"].data_ptr(), ctx["tval"].data_ptr(), zero.data_ptr(),
                           one.data_ptr(), n, c, cr, ws.data_ptr(), davg.data_ptr(), dmx.data_ptr(), sums.data_ptr(), dw0p.data_ptr(), dw2p.data_ptr(), st))
    dx = ops.empty_nhwc(n, h, w, c, x)
    check(lib.runet_rb_bwd3(dy.data_ptr(), ops.ld(dy), x.data_ptr(), ops.ld(x), ones_p.data_ptr(), zeros_p2.data_ptr(), none_p.data_ptr(),
                            ctx["ca"].data_ptr(), davg.data_ptr(), dmx.data_ptr(), ctx["idx"].data_ptr(), zero.data_ptr(), one.data_ptr(), one.data_ptr(),
                            zero.data_ptr(), dx.data_ptr(), ops.ld(dx), P, hw, c, 0, st))
    return dx


def sa_forward(x, wsa, save=True):
    """y = x * sigmoid(conv7x7([mean_c x, max_c x])).  x NHWC.  -> (y, ctx)"""
    n, h, w, c = x.shape
    P = n * h * w
    dev, st = x.device, ops.stream()
    one, zero = _consts(n, c, P, dev)
    smap = torch.empty((P, 2), device=dev, dtype=torch.float32)
    amax = torch.empty(P, device=dev, dtype=torch.int32)
    check(lib.runet_sa_reduce(x.data_ptr(), ops.ld(x), one.data_ptr(), zero.data_ptr(), P, h * w, c, smap.data_ptr(), amax.data_ptr(), st))
    sa = torch.empty(P, device=dev, dtype=torch.float32)
    check(lib.runet_sa_conv7(smap.data_ptr(), wsa.data_ptr(), sa.data_ptr(), n, h, w, st))
    y = ops.empty_nhwc(n, h, w, c, x)
    check(lib.runet_mul_pixel(x.data_ptr(), ops.ld(x), sa.data_ptr(), y.data_ptr(), ops.ld(y), P, c, st))
    return y, (dict(x=x, wsa=wsa, smap=smap, amax=amax, sa=sa) if save else None)


def sa_backward(ctx, dy, sink, pre=""):
    x, wsa, smap, amax, sa = ctx["x"], ctx["wsa"], ctx["smap"], ctx["amax"], ctx["sa"]
    n, h, w, c = x.shape
    hw, P = h * w, n * h * w
    dev, st = x.device, ops.stream()
    sm = Small(dev)
    one, zero = _consts(n, c, P, dev)
    dv = ops.empty_nhwc(n, h, w, c, x)
    dq = torch.empty(P, device=dev, dtype=torch.float32)
    check(lib.runet_rb_bwd1(dy.data_ptr(), ops.ld(dy), None, 0, x.data_ptr(), ops.ld(x), one.data_ptr(), zero.data_ptr(), sa.data_ptr(),
                            dv.data_ptr(), ops.ld(dv), dq.data_ptr(), P, hw, c, st))
    dsm = torch.empty((P, 2), device=dev, dtype=torch.float32)
    dwsa = sink.buf(pre, [("conv1.weight", (7, 7, 2, 1))])
    ws = scratch(lib.runet_sa_conv7_bwd_workspace_floats(n, h, w), dev)
    check(lib.runet_sa_conv7_bwd(smap.data_ptr(), dq.data_ptr(), wsa.data_ptr(), dsm.data_ptr(), dwsa.data_ptr(), ws.data_ptr(), n, h, w, st))
    none_nc = torch.full((n * c,), -1, device=dev, dtype=torch.int32)
    dx = ops.empty_nhwc(n, h, w, c, x)
    check(lib.runet_rb_bwd3(dv.data_ptr(), ops.ld(dv), x.data_ptr(), ops.ld(x), sa.data_ptr(), dsm.data_ptr(), amax.data_ptr(), one.data_ptr(),
                            zero.data_ptr(), zero.data_ptr(), none_nc.data_ptr(), zero.data_ptr(), one.data_ptr(), one.data_ptr(), zero.data_ptr(),
                            dx.data_ptr(), ops.ld(dx), P, hw, c, 0, st))
    return dx


# =============================================================================== DilatedBlock
class DilParams:
    __slots__ = ("w", "b", "bn")

    def __init__(self, ws, bs, bn):
        self.w, self.b, self.bn = ws, bs, bn


DIL = (1, 1, 2, 4)


def dilated_forward(x, p: DilParams, training, save=True, stats_hook=None):
    n, h, w, _ = x.shape
    q = p.w[0].shape[3]
    sm = Small(x.device)
    cat = ops.empty_nhwc(n, h, w, 4 * q, x)
    for i in range(4):
        ops.conv_fwd(x, p.w[i], p.b[i], out=cat[..., i * q:(i + 1) * q], dil=DIL[i])
    s, hsh, mean, invstd, _ = bn_coeff(cat, p.bn, training, sm, stats_hook=stats_hook)
    out = bn_apply(cat, s, hsh, None, relu=True)
    if not save:
        return out, None
    return out, dict(x=x, cat=cat, out=out, p=p, s=s, h=hsh, mean=mean, invstd=invstd, training=training, sync=stats_hook if training else None)


def dilated_backward(ctx, dout, sink, pre="", need_dx=True):
    p: DilParams = ctx["p"]
    x, cat, out = ctx["x"], ctx["cat"], ctx["out"]
    q = p.w[0].shape[3]
    cin = x.shape[3]
    c = 4 * q
    # parameter order: conv1.weight, conv1.bias, ..., conv4.bias, bn.weight, bn.bias
    wb = [sink.buf(pre, [(f"conv{i + 1}.weight", (1 if i == 0 else 3, 1 if i == 0 else 3, cin, q)), (f"conv{i + 1}.bias", (q,))])
          for i in range(4)]
    sums = sink.buf(pre, [("bn.weight", (c,)), ("bn.bias", (c,))])
    dcat = bn_backward(dout, cat, ctx["mean"], ctx["invstd"], ctx["s"], sums, mask=None, sync=ctx["sync"], relu_shift=ctx["h"], training=ctx["training"])
    dx = None
    for i in range(4):
        sl = dcat[..., i * q:(i + 1) * q]
        k = 1 if i == 0 else 3
        nw = k * k * cin * q
        kz = {}
        if need_dx:
            dx = ops.conv_dgrad(sl, p.w[i], out=dx, dil=DIL[i], accumulate=i > 0, keep_z=kz)
        ops.conv_wgrad(x, sl, k, k, dil=DIL[i], out=wb[i][:nw], z=kz.get("Z"))
        chan_sum(sl, wb[i][nw:])
    return dx


# =============================================================================== up-conv + attention gate + concat
class UpGateParams:
    __slots__ = ("wup", "bup", "wg", "bg", "bng", "wx", "bx", "bnx", "wpsi", "bpsi", "bnp")

    def __init__(self, wup, bup, wg, bg, bng, wx, bx, bnx, wpsi, bpsi, bnp):
        (self.wup, self.bup, self.wg, self.bg, self.bng, self.wx, self.bx, self.bnx, self.wpsi, self.bpsi, self.bnp) = (
            wup, bup, wg, bg, bng, wx, bx, bnx, wpsi, bpsi, bnp)


def gate_x_branch(skip, p: UpGateParams, training, sm, stats_hook=None):
    """W_x(skip) + its BatchNorm statistics on the side stream: independent of the gate signal, so upgate_forward starts it in front of
    the transposed convolution.  -> (branch handle, x1, (scale, shift, mean, invstd))"""
    sm.f(4)
    n, h, w, _ = skip.shape
    x1 = ops.main_pool(ops.empty_nhwc(n, h, w, p.wx.shape[3], skip))      # allocated on the main stream (side_branch.join)
    br = ops.side_branch(stats_hook is None)
    with br:
        fx = {} if (training and stats_hook is None) else None
        ops.conv_fwd(skip, p.wx, p.bx, out=x1, stats=fx)
        # SyncBN: the statistics wait for gate_forward, where they share W_g's message
        cx = bn_coeff(x1, p.bnx, training, sm, fused=fx)[:4] if stats_hook is None else None
    return br, x1, cx


def gate_forward(up, skip, p: UpGateParams, training, att_out, sm, stats_hook=None, xb=None):
    """AttentionGate(g=up, x=skip) -> writes skip*psi into att_out; returns ctx pieces.  xb: gate_x_branch() started earlier."""
    n, h, w, c = skip.shape
    f = p.wg.shape[3]
    P = n * h * w
    st = ops.stream()
    if xb is None:
        xb = gate_x_branch(skip, p, training, sm, stats_hook)
    fg = {} if (training and stats_hook is None) else None
    g1 = ops.conv_fwd(up, p.wg, p.bg, stats=fg)
    br, x1, cx = xb
    if cx is None:
        br.join(x1)
        (sg, hg, mean_g, invstd_g), cx = bn_coeff_pair(g1, p.bng, x1, p.bnx, training, sm, stats_hook)
    else:
        sg, hg, mean_g, invstd_g, _ = bn_coeff(g1, p.bng, training, sm, fused=fg)
        br.join(x1)
    sx, hx, mean_x, invstd_x = cx
    s = torch.empty((n, h, w, 1), device=skip.device, dtype=torch.float32)
    check(lib.runet_ag_psi(g1.data_ptr(), ops.ld(g1), x1.data_ptr(), ops.ld(x1), sg.data_ptr(), hg.data_ptr(), sx.data_ptr(), hx.data_ptr(),
                           p.wpsi.data_ptr(), p.bpsi.data_ptr(), s.data_ptr(), P, f, st))
    sp, hp, mean_p, invstd_p, _ = bn_coeff(s, p.bnp, training, sm, stats_hook=stats_hook)
    check(lib.runet_ag_out(skip.data_ptr(), ops.ld(skip), s.data_ptr(), sp.data_ptr(), hp.data_ptr(), att_out.data_ptr(), ops.ld(att_out), P, c, st))
    return dict(training=training, sync=stats_hook if training else None, g1=g1, x1=x1, s=s, sg=sg, hg=hg, mean_g=mean_g, invstd_g=invstd_g, sx=sx, hx=hx, mean_x=mean_x, invstd_x=invstd_x,
                sp=sp, hp=hp, mean_p=mean_p, invstd_p=invstd_p)


def gate_backward(gc, up, skip, p: UpGateParams, datt, dup, sink, pre=""):
    """datt: grad of the gated skip (view); dup: grad buffer of `up`, accumulated into.  -> dskip"""
    n, h, w, c = skip.shape
    f = p.wg.shape[3]
    cg = up.shape[3]
    P = n * h * w
    st = ops.stream()
    dev = skip.device
    # parameter order: W_g.0.weight, W_g.0.bias, W_g.1.weight, W_g.1.bias, W_x.0.*, W_x.1.*, psi.0.weight, psi.0.bias, psi.1.*
    wg_b = sink.buf(pre, [("W_g.0.weight", (1, 1, cg, f)), ("W_g.0.bias", (f,))])
    sums_g = sink.buf(pre, [("W_g.1.weight", (f,)), ("W_g.1.bias", (f,))])
    wx_b = sink.buf(pre, [("W_x.0.weight", (1, 1, c, f)), ("W_x.0.bias", (f,))])
    sums_x = sink.buf(pre, [("W_x.1.weight", (f,)), ("W_x.1.bias", (f,))])
    dwpsi_db = sink.buf(pre, [("psi.0.weight", (1, 1, f, 1)), ("psi.0.bias", (1,))])
    sums_p = sink.buf(pre, [("psi.1.weight", (1,)), ("psi.1.bias", (1,))])
    dskip = ops.empty_nhwc(n, h, w, c, skip)
    dsbn = torch.empty((n, h, w, 1), device=dev, dtype=torch.float32)
    check(lib.runet_ag_bwd1(datt.data_ptr(), ops.ld(datt), skip.data_ptr(), ops.ld(skip), gc["s"].data_ptr(), gc["sp"].data_ptr(),
                            gc["hp"].data_ptr(), dskip.data_ptr(), ops.ld(dskip), dsbn.data_ptr(), P, c, st))
    sync, tr = gc["sync"], gc["training"]
    ds = bn_backward(dsbn, gc["s"], gc["mean_p"], gc["invstd_p"], gc["sp"], sums_p, out=dsbn, sync=sync, training=tr)
    dpre = ops.empty_nhwc(n, h, w, f, skip)
    fused_sums = FUSED_GATE_BN_SUMS and not (sync is not None and tr)
    if fused_sums:
        # the gate's two BatchNorm-backward reductions ride in the kernel that produces their incoming gradient
        ws = scratch(lib.runet_ag_bwd2_bn_workspace_floats(P, f), dev)
        check(lib.runet_ag_bwd2_bn(ds.data_ptr(), gc["g1"].data_ptr(), ops.ld(gc["g1"]), gc["x1"].data_ptr(), ops.ld(gc["x1"]), gc["sg"].data_ptr(),
                                   gc["hg"].data_ptr(), gc["sx"].data_ptr(), gc["hx"].data_ptr(), p.wpsi.data_ptr(), gc["mean_g"].data_ptr(),
                                   gc["invstd_g"].data_ptr(), gc["mean_x"].data_ptr(), gc["invstd_x"].data_ptr(), dpre.data_ptr(), ops.ld(dpre),
                                   ws.data_ptr(), ws.numel(), dwpsi_db.data_ptr(), sums_g.data_ptr(), sums_x.data_ptr(), P, f, st))
        use_g, use_x = (sums_g, sums_x) if tr else (zeros(2 * f, dev), zeros(2 * f, dev))
        dg1 = bn_bwd_apply(dpre, gc["g1"], gc["mean_g"], gc["invstd_g"], gc["sg"], use_g, 0)
        ops.conv_wgrad(up, dg1, 1, 1, out=wg_b[:cg * f])
        chan_sum(dg1, wg_b[cg * f:])
        ops.conv_dgrad(dg1, p.wg, out=dup, accumulate=True)
        del dg1
        dx1 = bn_bwd_apply(dpre, gc["x1"], gc["mean_x"], gc["invstd_x"], gc["sx"], use_x, 0, out=dpre)
        ops.conv_wgrad(skip, dx1, 1, 1, out=wx_b[:c * f])
        chan_sum(dx1, wx_b[c * f:])
        ops.conv_dgrad(dx1, p.wx, out=dskip, accumulate=True)
        return dskip
    ws = _ws(n, h * w, f, dev)
    check(lib.runet_ag_bwd2(ds.data_ptr(), gc["g1"].data_ptr(), ops.ld(gc["g1"]), gc["x1"].data_ptr(), ops.ld(gc["x1"]), gc["sg"].data_ptr(),
                            gc["hg"].data_ptr(), gc["sx"].data_ptr(), gc["hx"].data_ptr(), p.wpsi.data_ptr(), dpre.data_ptr(), ops.ld(dpre),
                            ws.data_ptr(), dwpsi_db.data_ptr(), P, f, st))
    if sync is not None and tr:              # SyncBN: both BatchNorms' sums are ready here - one all-reduce
        bn_bwd_reduce(dpre, gc["g1"], gc["mean_g"], gc["invstd_g"], gc["sg"], sums_g)
        bn_bwd_reduce(dpre, gc["x1"], gc["mean_x"], gc["invstd_x"], gc["sx"], sums_x)
        (use_g, m_g), (use_x, m_x) = sync.reduce_sums_many([sums_g, sums_x], [P, P])
        dg1 = bn_bwd_apply(dpre, gc["g1"], gc["mean_g"], gc["invstd_g"], gc["sg"], use_g, m_g)
    else:
        dg1 = bn_backward(dpre, gc["g1"], gc["mean_g"], gc["invstd_g"], gc["sg"], sums_g, training=tr)
    ops.conv_wgrad(up, dg1, 1, 1, out=wg_b[:cg * f])
    chan_sum(dg1, wg_b[cg * f:])
    ops.conv_dgrad(dg1, p.wg, out=dup, accumulate=True)
    del dg1
    if sync is not None and tr:
        dx1 = bn_bwd_apply(dpre, gc["x1"], gc["mean_x"], gc["invstd_x"], gc["sx"], use_x, m_x, out=dpre)
    else:
        dx1 = bn_backward(dpre, gc["x1"], gc["mean_x"], gc["invstd_x"], gc["sx"], sums_x, out=dpre, training=tr)
    ops.conv_wgrad(skip, dx1, 1, 1, out=wx_b[:c * f])
    chan_sum(dx1, wx_b[c * f:])
    ops.conv_dgrad(dx1, p.wx, out=dskip, accumulate=True)
    return dskip


def upgate_forward(y, skip, p: UpGateParams, training, save=True, stats_hook=None):
    """cat([AttentionGate(up, skip), up]) with up = ConvTranspose2d(y), written into one buffer."""
    n, h, w, c = skip.shape
    sm = Small(skip.device)
    cat = ops.empty_nhwc(n, h, w, 2 * c, skip)
    up = cat[..., c:]
    xb = gate_x_branch(skip, p, training, sm, stats_hook)
    ops.convt_fwd(y, p.wup, p.bup, out=up)
    gc = gate_forward(up, skip, p, training, cat[..., :c], sm, stats_hook, xb=xb)
    if not save:
        return cat, None
    gc.update(y=y, skip=skip, cat=cat, p=p)
    return cat, gc


def upgate_backward(ctx, dcat, sink, pre_att, pre_up):
    """dcat [N,H,W,2C] is consumed (its right half accumulates the gate's gradient).  -> (dy, dskip)"""
    p: UpGateParams = ctx["p"]
    y, skip, cat = ctx["y"], ctx["skip"], ctx["cat"]
    c = skip.shape[3]
    cin = y.shape[3]
    up, dup, datt = cat[..., c:], dcat[..., c:], dcat[..., :c]
    dskip = gate_backward(ctx, up, skip, p, datt, dup, sink, pre=pre_att)
    up_b = sink.buf(pre_up, [("weight", (2, 2, cin, c)), ("bias", (c,))])
    ops.convt_wgrad(y, dup, out=up_b[:4 * cin * c])
    chan_sum(dup, up_b[4 * cin * c:])
    dy = ops.convt_dgrad(dup, p.wup)
    return dy, dskip


# =============================================================================== pool / stem / head
def maxpool_forward(x):
    n, h, w, c = x.shape
    y = ops.empty_nhwc(n, h // 2, w // 2, c, x)
    idx = torch.empty((n, h // 2, w // 2, c), device=x.device, dtype=torch.uint8)
    check(lib.runet_maxpool2_fwd(x.data_ptr(), ops.ld(x), y.data_ptr(), ops.ld(y), idx.data_ptr(), n, h, w, c, ops.stream()))
    return y, idx


def maxpool_backward(dy, idx, dx=None):
    """dx given: accumulate into it (skip-connection gradient already there)."""
    n, ho, wo, c = dy.shape
    acc = dx is not None
    if dx is None:
        dx = ops.empty_nhwc(n, 2 * ho, 2 * wo, c, dy)
    check(lib.runet_maxpool2_bwd(dy.data_ptr(), ops.ld(dy), idx.data_ptr(), dx.data_ptr(), ops.ld(dx), n, 2 * ho, 2 * wo, c, int(acc), ops.stream()))
    return dx


def to_nhwc_pad(x_nchw, c_pad):
    n, c, h, w = x_nchw.shape
    y = torch.empty((n, h, w, c_pad), device=x_nchw.device, dtype=torch.float32)
    sn, sc, sh, sw = x_nchw.stride()
    check(lib.runet_to_nhwc_pad(x_nchw.data_ptr(), sn, sc, sh, sw, y.data_ptr(), n, c, h, w, c_pad, ops.stream()))
    return y


def outc_forward(x, w, b, want_logit=False):
    n, h, wd, c = x.shape
    prob = torch.empty((n, 1, h, wd), device=x.device, dtype=torch.float32)
    logit = torch.empty((n, 1, h, wd), device=x.device, dtype=torch.float32) if want_logit else None
    check(lib.runet_outc_fwd(x.data_ptr(), ops.ld(x), w.data_ptr(), b.data_ptr(), logit.data_ptr() if want_logit else None, prob.data_ptr(),
                             n * h * wd, c, ops.stream()))
    return prob, logit


def outc_backward(dprob, prob, x, w, sink, pre=""):
    n, h, wd, c = x.shape
    dx = ops.empty_nhwc(n, h, wd, c, x)
    dw_db = sink.buf(pre, [("weight", (1, 1, c, 1)), ("bias", (1,))])
    ws = _ws(n, h * wd, c, x.device)
    check(lib.runet_outc_bwd(dprob.data_ptr(), prob.data_ptr(), x.data_ptr(), ops.ld(x), w.data_ptr(), dx.data_ptr(), ops.ld(dx), ws.data_ptr(),
                             dw_db.data_ptr(), n * h * wd, c, ops.stream()))
    return dx
