"""Thin Python wrappers over the C ABI (include/runet_hip.h).

Activations are torch tensors used purely as device-memory handles: logical shape
[N, H, W, C] (NHWC), last-dim stride 1, pixel stride `ld = t.stride(2)` (a tensor may be a
channel slice of a wider concat buffer).  All launches go to torch's current HIP stream.
"""
from __future__ import annotations

import ctypes
import os
import threading
import weakref

import torch

from ._lib import check, lib

CONV_FWD, CONV_DGRAD, CONVT_FWD, CONVT_DGRAD, CONV_DGRAD_T, CONVT_DGRAD_T = 0, 1, 2, 3, 4, 5

_ws = {}
# Set by trainer.TrainStep once a step has been captured into a hipGraph: the graph holds the ADDRESSES of the scratch buffers, so a
# later, larger eager step (ragged batch, another model) must not free them when it grows a pool - superseded buffers are retired here.
PIN_SCRATCH = False
_retired = []


def grow(pool, key, nfloats, device, floor):
    """Scratch buffer `key` of `pool` with at least `nfloats` floats (grows geometrically; see PIN_SCRATCH)."""
    buf = pool.get(key)
    if buf is None or buf.numel() < nfloats:
        if buf is not None and PIN_SCRATCH:
            _retired.append(buf)
        buf = torch.empty(max(int(nfloats), floor, 2 * buf.numel() if buf is not None else 0), device=device, dtype=torch.float32)
        if getattr(_tls, "override", None) is not None:
            # allocated from torch's CURRENT stream's pool while the launches are redirected to the weight-gradient stream (_on_side): tell
            # the allocator, or a buffer dropped at the next growth could be handed out again while that stream still works on it
            buf.record_stream(_side["active"])
        pool[key] = buf
    return buf


_tls = threading.local()     # .override: raw handle of the weight-gradient stream while THIS thread redirects a launch to it (_on_side);
#                              per thread: the autograd engine runs backward passes on its own threads


def stream():
    """Raw handle of the stream the next kernel goes to.  torch.cuda.current_stream() costs ~8 us (device-index resolution, availability check); the
    ops ask for the stream ~500 times per step, which is 4 ms of a host-bound small step - the raw getter costs 0.3 us."""
    o = getattr(_tls, "override", None)
    if o is not None:
        return o
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


# ---- operand precision of the matrix-core kernels.  "f32": the reference's arithmetic (fp32 MFMA, Winograd forms included).
# "bf16" (BASELINE.json configs 3 / 5): every convolution with >= 8 input channels rounds its two operands to bf16 on the way into LDS and
# accumulates in fp32 (csrc/conv_bf16.hip); weights stay fp32 masters, activations / BatchNorm / attention / loss / Adam stay fp32.
_PRECISION = "f32"
PRECISIONS = ("f32", "bf16", "fp16")      # "fp16": the same kernels with IEEE half operands (needs a loss scale: trainer.TrainStep(loss_scale=...))
_LOWP = {"bf16": (torch.bfloat16, "bf16"), "fp16": (torch.float16, "fp16")}


class precision:
    """with ops.precision("bf16"): ...   (model.RobustUNet.set_precision wraps forward and backward in it)"""

    def __init__(self, mode):
        if mode not in PRECISIONS:
            raise ValueError(f"precision must be one of {PRECISIONS}")
        self.mode = mode

    def __enter__(self):
        global _PRECISION
        self.prev, _PRECISION = _PRECISION, self.mode
        return self

    def __exit__(self, *exc):
        global _PRECISION
        _PRECISION = self.prev
        return False


# ---- derived-weight cache: Winograd-domain filters (fp32 path) and packed bf16 weights are functions of the weights alone, and the
# weights change once per optimizer step, so they are produced once per step instead of once per use (forward + data gradient of the same
# layer, 36 + 24 transform launches per step).  FusedAdam updates parameters through raw pointers (no torch version bump), so it bumps
# WEIGHT_EPOCH; torch-side writes (load_state_dict, optimizers from torch.optim) bump the tensor's _version.
WEIGHT_EPOCH = 0
_derived = {}


def bump_weight_epoch():
    global WEIGHT_EPOCH
    WEIGHT_EPOCH += 1
    if len(_derived) > 4096:
        _derived.clear()


def _cached(w, kind, make, desc=None):
    """make(out) -> the derived tensor (out = None: allocate; else refill that tensor in place).  make is kept in the cache entry (for
    prefetch_derived) and must capture the weight's raw pointer, not the tensor: the entry holds the parameter by weak reference only.
    desc = (RUNET_DERIVE_* kind, weight pointer, cin, cout, mode): the same refill as an entry of the multi-tensor launch (_derive_multi)."""
    base = w._base if w._base is not None else w
    key = (w.data_ptr(), kind)
    tag = (base._version, WEIGHT_EPOCH, tuple(w.shape))
    hit = _derived.get(key)
    # the weak reference pins the entry to THIS parameter tensor: another model's weight allocated later at the same address misses
    if hit is not None and hit[0] == tag and hit[2]() is base and hit[5] == base.data_ptr():
        ev = hit[4]
        if ev is not None:                         # refilled by prefetch_derived on the side stream: order this stream behind it, once
            sid = stream()
            if sid not in ev[1]:
                torch.cuda.current_stream().wait_event(ev[0])
                ev[1].add(sid)
        return hit[1]
    val = make(None)
    _derived[key] = [tag, val, weakref.ref(base), make, None, base.data_ptr(), desc]
    return val


PREFETCH_DERIVED = os.environ.get("RUNET_PREFETCH_DERIVED", "1") != "0"
# All stale split-operand weights of a step in one launch (csrc/derive_multi.hip) instead of one launch per tensor (67 per Robust U-Net step).
DERIVE_MULTI = os.environ.get("RUNET_NO_DERIVE_MULTI", "0") != "1"
DERIVE_WINO4, DERIVE_WINO2, DERIVE_PACK = 0, 1, 2
_derive_tables = {}


def _derive_multi(entries):
    """One launch on ops.stream() refilling, in place, every cache entry of `entries` (all with a descriptor).  The device table depends on
    addresses only, so it is built and uploaded the first time a set of entries is seen - never during a stream capture (-> False then)."""
    key = (entries[0][1].device.index,) + tuple((ent[6], ent[1].data_ptr()) for ent in entries)      # addresses repeat across devices
    tab = _derive_tables.get(key)
    if tab is None:
        if torch.cuda.is_current_stream_capturing():
            return False
        nb = lib.runet_derive_desc_bytes()
        host = ctypes.create_string_buffer(nb * len(entries))
        first = 0
        for i, ent in enumerate(entries):
            kind, wp, cin, cout, mode = ent[6]
            blocks = lib.runet_derive_desc(ctypes.addressof(host), i, kind, wp, ent[1].data_ptr(), cin, cout, mode, first)
            if blocks <= 0:
                raise RuntimeError(f"runet_derive_desc refused entry {ent[6]}")
            first += blocks
        table = torch.frombuffer(bytearray(host.raw), dtype=torch.uint8).to(entries[0][1].device)
        if len(_derive_tables) >= 64:
            if PIN_SCRATCH:                      # a captured step may hold the address of a table
                _retired.extend(t[0] for t in _derive_tables.values())
            _derive_tables.clear()
        tab = _derive_tables[key] = (table, len(entries), first)
    check(lib.runet_derive_multi(tab[0].data_ptr(), tab[1], tab[2], stream()))
    return True


def _refill(group):
    """group: [(entry, base)] stale entries -> refilled in place on ops.stream(): those with a descriptor in one launch, the rest one by one."""
    multi = [e for e, _ in group if e[6] is not None] if DERIVE_MULTI else []
    done = len(multi) >= 2 and _derive_multi(multi)
    for ent, base in group:
        if not (done and ent[6] is not None):
            ent[3](ent[1])
        ent[0] = (base._version, WEIGHT_EPOCH, ent[0][2])


def prefetch_derived(allow_side=True):
    """Called at the start of a training forward pass: refill every derived weight (Winograd-domain filters, packed split-operand / bf16 /
    fp16 weights) that the optimizer step has made stale.  GPU-bound step sizes: on the idle side stream, in the order the previous step
    first used them, forward kinds first.  The main stream waits for ONE event (recorded behind the forward kinds) the first time it touches
    a refilled entry, and for a second one behind the data-gradient kinds; no per-entry events (those cost more than the launches they moved:
    an earlier version with one event per entry was 0.7 % slower).  Otherwise (small steps, no side stream, a hipGraph capture): the entries
    of the multi-tensor launch on the current stream, the rest lazily at their first use as before.  Refilled IN PLACE: every reader of the
    old values was enqueued before the optimizer launch, which the stream used here is ordered behind.  -> number of entries refilled"""
    if not (PREFETCH_DERIVED and _derived):
        return 0
    on_side = allow_side and FWD_BRANCHES and _branch_now and USE_WGRAD_STREAM and not torch.cuda.is_current_stream_capturing()
    if not on_side and (not DERIVE_MULTI or _side.get("active") is not None):
        return 0
    todo = []
    for key, ent in list(_derived.items()):
        base = ent[2]()
        if base is None or base.data_ptr() != ent[5]:
            # parameter gone, or its storage was replaced under the surviving Parameter object (`p.data = ...`, a .to() round trip): the
            # refill closure holds the OLD raw pointer - drop the entry, the next use rebuilds it from the live storage
            del _derived[key]
        elif (ent[0][0], ent[0][1]) != (base._version, WEIGHT_EPOCH):
            todo.append((key[1].endswith("d") or key[1].endswith("t") or key[1] == "T", ent, base))
    if not todo:
        return 0
    if not on_side:
        group = [(e, b) for _, e, b in todo if e[6] is not None]
        if len(group) < 2 or not _derive_multi([e for e, _ in group]):
            return 0
        for ent, base in group:
            ent[0] = (base._version, WEIGHT_EPOCH, ent[0][2])
            ent[4] = None
        return len(group)
    br = side_branch()
    if br.s is None:
        return 0
    side_id = br.s.cuda_stream
    with br:
        for want_bwd in (False, True):             # forward kinds first: they are needed within the first half millisecond
            group = [(e, b) for is_bwd, e, b in todo if is_bwd == want_bwd]
            if not group:
                continue
            _refill(group)
            ev = [torch.cuda.Event(), {side_id}]
            ev[0].record(br.s)
            for ent, _ in group:
                ent[4] = ev
    br.s = None                                    # no join: consumers wait on the events
    return len(todo)


# Measured: reading the data gradients' weights from a cached per-tap transposed copy (modes *_DGRAD_T, n-contiguous staging) instead of the
# k-contiguous staging of the forward weight changes nothing in the step (421.9 vs 422.2 img/s: those launches are bound by their activation
# traffic, not by the weight tile) and costs 22 transpose launches (0.33 ms of kernel time) per step, so it is off by default.
USE_TRANSPOSED_DGRAD = os.environ.get("RUNET_TRANSPOSED_DGRAD", "0") == "1"


def transposed_weights(w_hwio):
    """[taps][cin][cout] -> [taps][cout][cin] (cached per optimizer step) for the *_DGRAD_T modes of runet_conv_igemm."""
    kh, kw, cin, cout = w_hwio.shape
    wp, dev = w_hwio.data_ptr(), w_hwio.device          # the closure is kept in the cache: it must not hold the parameter alive

    def make(out):
        wt = out if out is not None else torch.empty((kh, kw, cout, cin), device=dev, dtype=torch.float32)
        check(lib.runet_transpose_taps(wp, wt.data_ptr(), kh * kw, cin, cout, stream()))
        return wt
    return _cached(w_hwio, "T", make)


def _bf16_case(cin, cin_w):
    """A reduced-precision (bf16 / fp16 operand) kernel serves this convolution."""
    return _PRECISION != "f32" and cin_w == cin and cin >= 8


def _lp(name):
    """C-ABI entry point of the active reduced-precision type, e.g. _lp("runet_conv_igemm_{}") -> lib.runet_conv_igemm_bf16."""
    return getattr(lib, name.format(_LOWP[_PRECISION][1]))


def bf16_weights(w_hwio, transpose=False):
    """HWIO fp32 weight [kh, kw, cin, cout] -> packed bf16 [taps][K/8][N][8] (forward: K = cin; transpose: K = cout, data gradients)."""
    kh, kw, cin, cout = w_hwio.shape

    prec = _PRECISION                                   # bound now: a refill ahead of use runs outside the precision context
    tdtype, tname = _LOWP[prec]
    pack_elems, pack = getattr(lib, f"runet_{tname}_pack_elems"), getattr(lib, f"runet_{tname}_pack_weights")
    wp, dev = w_hwio.data_ptr(), w_hwio.device

    def make(out):
        k, n = (cout, cin) if transpose else (cin, cout)
        buf = out if out is not None else torch.empty(pack_elems(kh * kw, k, n), device=dev, dtype=tdtype)
        check(pack(wp, buf.data_ptr(), kh * kw, cin, cout, int(transpose), stream()))
        return buf
    return _cached(w_hwio, prec + ("t" if transpose else ""), make)


def _igemm_bf16(mode, x, w_hwio, bias, out, n, h, wd, cin, cout, kh, kw, dil, accumulate, transpose):
    wp = bf16_weights(w_hwio, transpose)
    prof = _PROFILE is not None
    if prof:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    check(_lp("runet_conv_igemm_{}")(x.data_ptr(), ld(x), wp.data_ptr(), bias.data_ptr() if bias is not None else None, out.data_ptr(), ld(out),
                                    n, h, wd, cin, cout, kh, kw, dil, mode, int(accumulate), stream()))
    if prof:
        e1.record()
        taps = 4 if kh == 2 else kh * kw
        fl = 2.0 * n * h * wd * taps * cin * cout
        opix = 4 * n * h * wd if mode == CONVT_FWD else n * h * wd          # pixels written (mode CONVT_DGRAD reads 4x the pixels instead)
        ipix = 4 * n * h * wd if mode == CONVT_DGRAD else n * h * wd
        nbytes = 4.0 * (ipix * cin + opix * cout) + 2.0 * taps * cin * cout
        name = ("conv3x3_{}_kernel" if (kh == 3 and dil == 1 and mode in (CONV_FWD, CONV_DGRAD)) else "igemm_{}_kernel").format(_PRECISION)
        _PROFILE.append((name, fl, fl, e0, e1, nbytes))
    return out


def ld(t):
    """Pixel stride of an NHWC view; validates the layout contract."""
    n, h, w, c = t.shape
    s = t.stride()
    if not (s[3] == 1 and s[1] == w * s[2] and (n == 1 or s[0] == h * w * s[2]) and s[2] >= c):
        raise ValueError(f"not an NHWC (slice) view: shape {tuple(t.shape)} strides {s}")
    return s[2]


def empty_nhwc(n, h, w, c, like):
    return torch.empty((n, h, w, c), device=like.device, dtype=torch.float32)


def workspace(nfloats, device):
    key = (device.index, "wgrad", stream())      # one scratch per stream (weight gradients may run on a side stream)
    return grow(_ws, key, nfloats, device, 1 << 22)


def hwio(w):
    """Physical [kh, kw, cin, cout] view of a conv weight stored the way model.py stores it
    (logical OIHW tensor whose memory is HWIO); falls back to a copy for foreign layouts."""
    v = w.permute(2, 3, 1, 0)
    return v if v.is_contiguous() else v.contiguous()


def hwio_t(w):
    """ConvTranspose2d weight (logical [cin, cout, 2, 2]) -> physical [2, 2, cin, cout]."""
    v = w.permute(2, 3, 0, 1)
    return v if v.is_contiguous() else v.contiguous()


# ---- optional live timing of the implicit-GEMM launches (bench.py roofline): HIP events on the launch stream
_PROFILE = None


def start_conv_profile():
    global _PROFILE
    _PROFILE = []
    return _PROFILE


def stop_conv_profile(prof):
    """-> dict(kernel=dominant instantiation, avg_us, launches, time_s, exec_flops_total,
               by_kernel={name: [launches, ms, algorithmic TFLOP/s, executed TFLOP/s]})
    algorithmic = the direct convolution's FLOPs; executed = the multiply-adds the kernel issues on the
    matrix pipe (Winograd F(2x2): 16/36 of them, F(4x4) position-GEMMs: their own GEMM FLOPs)."""
    global _PROFILE
    _PROFILE = None
    torch.cuda.synchronize()
    agg = {}
    for ent in prof:
        name, flops, xflops, e0, e1 = ent[:5]
        a = agg.setdefault(name, [0, 0.0, 0.0, 0.0, 0.0])
        a[0] += 1
        a[1] += e0.elapsed_time(e1) * 1e-3
        a[2] += flops
        a[3] += xflops
        a[4] += ent[5] if len(ent) > 5 else 0.0          # algorithmic HBM bytes (bf16 kernels: activations read + written once, weights once)
    dom = max(agg, key=lambda k: agg[k][1])
    n, t, f, xf, nb = agg[dom]
    by = {k: [v[0], round(v[1] * 1e3, 3), round(v[2] / v[1] / 1e12, 2), round(v[3] / v[1] / 1e12, 2)] + ([round(v[4] / v[1] / 1e9, 1)] if v[4] else [])
          for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])}
    return {"kernel": dom, "avg_us": t / n * 1e6, "launches": n, "time_s": t, "exec_flops_total": sum(v[3] for v in agg.values()), "by_kernel": by,
            "bytes_per_s": nb / t if nb else None, "bytes_per_launch": nb / n if nb else None}


def _igemm(mode, x, ldx, w, bias, y, ldy, n, h, wd, cin, cin_w, cout, kh, kw, dil, accumulate):
    if _PROFILE is None:
        check(lib.runet_conv_igemm(x, ldx, w, bias, y, ldy, n, h, wd, cin, cin_w, cout, kh, kw, dil, mode, accumulate, stream()))
        return
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    check(lib.runet_conv_igemm(x, ldx, w, bias, y, ldy, n, h, wd, cin, cin_w, cout, kh, kw, dil, mode, accumulate, stream()))
    e1.record()
    taps = 4 if kh == 2 else kh * kw
    name = lib.runet_conv_igemm_kernel_name(n, h, wd, min(cin, cin_w), cout, kh, mode).decode()
    fl = 2.0 * n * h * wd * taps * min(cin, cin_w) * cout
    _PROFILE.append((name, fl, fl, e0, e1))


USE_WINOGRAD = os.environ.get("RUNET_NO_WINOGRAD", "0") != "1"


def _wino_case(h, w, kh, dil, k, n, cin_w, n_img=1, ldx=0, ldy=0):
    """Fused F(2x2) kernel usable: shape supported and the tensors within its 32-bit buffer offsets (else: implicit GEMM, 64-bit)."""
    return (USE_WINOGRAD and kh == 3 and dil == 1 and cin_w == k
            and bool(lib.runet_wino_fits(n_img, h, w, max(ldx, k), max(ldy, n), k, n)))


USE_WINOGRAD4 = USE_WINOGRAD and os.environ.get("RUNET_NO_WINOGRAD4", "0") != "1"


USE_WINOGRAD4_DILATED = USE_WINOGRAD4 and os.environ.get("RUNET_NO_WINOGRAD4_DILATED", "0") != "1"


# Channels the wider / narrower side needs and the largest image (pixels) for the unfused F(4x4) path.  128 / 128 since the data gradient
# shares Z = A dy A^T with the weight gradient (adjoint form): before that, sending the 128 -> 128 @ 128^2 layers here cut 1.1 ms of kernel time
# but added 4 GB of Winograd-domain traffic and did not move the step (533.0 vs 532.7 img/s); with it 538.2 -> 547.1 (three A/B pairs).
WINO4_MIN_WIDE = int(os.environ.get("RUNET_WINO4_MIN_WIDE", "128"))
WINO4_MIN_NARROW = int(os.environ.get("RUNET_WINO4_MIN_NARROW", "128"))
WINO4_MAX_PIXELS = int(os.environ.get("RUNET_WINO4_MAX_PIXELS", str(128 * 128)))


def _wino4_case(h, w, kh, dil, k, n, cin_w):
    """Deep layers (>= 256 channels on one side, <= 128x128 pixels): unfused F(4x4,3x3) beats the fused F(2x2) kernel (tools/bench_conv.py).
    Dilated 3x3 convolutions (the bottleneck's DilatedBlock, dilation 2 and 4) take the same path as dil*dil dilation-1 convolutions over
    the sub-images of every image (csrc/conv_winograd4.hip pix()): 4x fewer multiplies than the implicit GEMM they used before."""
    if dil != 1 and not (USE_WINOGRAD4_DILATED and h % dil == 0 and w % dil == 0):
        return False
    return (USE_WINOGRAD4 and kh == 3 and cin_w == k and max(k, n) >= WINO4_MIN_WIDE and min(k, n) >= WINO4_MIN_NARROW and h * w <= WINO4_MAX_PIXELS
            and bool(lib.runet_wino4_supported(h // dil, w // dil, k, n)))


USE_STEM_KERNELS = os.environ.get("RUNET_NO_STEM_KERNELS", "0") != "1"


def _stem_case(cin, cin_w, cout, kh, dil=1):
    """The RGB stem (1..3 real channels in a 4-channel NHWC tensor): csrc/conv_stem.hip instead of the general kernels."""
    return USE_STEM_KERNELS and cin == 4 and kh in (1, 3) and dil == 1 and bool(lib.runet_stem_supported(cin_w, cout))


def stem_conv(x, w3, w1=None, stats3=None, stats1=None):
    """x [N,H,W,4] (RGB + zero) -> (conv3x3(x, w3), conv1x1(x, w1) or None) in one launch; w3 [3,3,cin_w,C], w1 [1,1,cin_w,C].
    stats3 / stats1: dicts that receive the BatchNorm statistics partials of the two outputs (see conv_fwd's `stats`)."""
    n, h, w, _ = x.shape
    _, _, cin_w, cout = w3.shape
    y3 = empty_nhwc(n, h, w, cout, x)
    y1 = empty_nhwc(n, h, w, cout, x) if w1 is not None else None
    args = (x.data_ptr(), ld(x), w3.data_ptr(), w1.data_ptr() if w1 is not None else None, y3.data_ptr(), ld(y3),
            y1.data_ptr() if y1 is not None else None, ld(y1) if y1 is not None else 0, n, h, w, cin_w, cout)
    if EPILOGUE_STATS and stats3 is not None and (w1 is None or stats1 is not None):
        parts = lib.runet_stem_conv_stats_parts(n, h, w)
        s3 = _stats_buf(stats3, parts, cout, x.device)
        s1 = _stats_buf(stats1, parts, cout, x.device) if w1 is not None else None
        check(lib.runet_stem_conv_stats(*args, s3, s1, stream()))
    else:
        check(lib.runet_stem_conv(*args, stream()))
    return y3, y1


# ---- 1x1 convolutions and the k2-s2 transposed convolution on the BF16 matrix cores (csrc/conv_x3.hip: the split-operand scheme of the
# F(4x4) position GEMMs behind the SIMPLE geometry of the implicit GEMM): fp32-accurate, the weight pre-split once per optimizer step.
# RUNET_NO_CONV_X3=1 (or RUNET_NO_X3=1): the f32-MFMA implicit GEMM everywhere.
# Where it pays (tools/conv_launches.py, 16 x 256^2 step, single stream, x3 vs f32-MFMA igemm in us): the contraction (taps x channels) must be
# deep enough for the matrix time to matter - K >= 256: 1x1 forward 512->256 @ 64^2 194 -> 118, 1024->512 @ 32^2 157 -> 104, transposed forward
# 1024->512 @ 16^2 180 -> 104, transposed data gradient 256->512 @ 32^2 159 -> 104, `+=` data gradient 512->1024 @ 32^2 183 -> 123; K = 128 only
# with >= 128 output channels (128->256 @ 64^2 59 -> 52; 128->64 @ 128^2 61 -> 73 loses); K = 64 launches are bound by streaming their
# activations and the smaller f32 tiles keep more loads in flight (64->32 @ 256^2 95 -> 175): those stay on the implicit GEMM.
USE_CONV_X3 = os.environ.get("RUNET_NO_CONV_X3", "0") != "1"
CONV_X3_MIN_K = int(os.environ.get("RUNET_CONV_X3_MIN_K", "128"))        # smallest contraction (taps x channels); from 2x this on: any width
CONV_X3_WIDE_N = int(os.environ.get("RUNET_CONV_X3_WIDE_N", "128"))      # output channels needed while the contraction is below 2x MIN_K
# The transposed forward with a 128-deep contraction takes the split-operand kernel also with fewer than CONV_X3_WIDE_N output channels (up1:
# 128 -> 64 @ 128^2: its four taps make it matrix-bound on the f32 implicit GEMM, 267 us at 1.5 TB/s; 204 us here).  RUNET_NO_CONVT_X3_NARROW=1: off.
CONVT_X3_NARROW = os.environ.get("RUNET_NO_CONVT_X3_NARROW", "0") != "1"
_X3_KIND = {CONV_FWD: "cx3f", CONV_DGRAD: "cx3d", CONVT_FWD: "cx3uf", CONVT_DGRAD: "cx3ud"}      # "...d": refilled with the backward kinds


def _conv_x3_case(x, cin, cin_w, cout, taps=1):
    """taps: taps inside the contraction (4 for the transposed data gradient, else 1)."""
    k = cin * taps
    return (USE_X3 and USE_CONV_X3 and _PRECISION == "f32" and cin_w == cin and cin % 16 == 0 and cout % 4 == 0 and k >= CONV_X3_MIN_K
            and (k >= 2 * CONV_X3_MIN_K or cout >= CONV_X3_WIDE_N) and ld(x) % 4 == 0 and x.data_ptr() % 16 == 0)


def conv_x3_weights(w_hwio, mode):
    """The module's forward weight (1x1 [1,1,Ci,Co] or transposed [2,2,Ci,Co]) -> split planes for runet_conv_x3 in `mode`."""
    _, _, ci, co = w_hwio.shape
    k, n = (co, ci) if mode in (CONV_DGRAD, CONVT_DGRAD) else (ci, co)
    wp, dev = w_hwio.data_ptr(), w_hwio.device

    def make(out):
        buf = out if out is not None else torch.empty(lib.runet_conv_x3_pack_elems(k, n, mode), device=dev, dtype=torch.bfloat16)
        check(lib.runet_conv_x3_pack(wp, buf.data_ptr(), k, n, mode, stream()))
        return buf
    return _cached(w_hwio, _X3_KIND[mode], make, desc=(DERIVE_PACK, wp, k, n, mode))


# BatchNorm statistics taken in the producing convolution's epilogue (runet_conv_x3_stats / runet_wino_conv_x3_stats) instead of by a pass of
# runet_bn_stats over the tensor; RUNET_NO_EPILOGUE_STATS=1: the separate pass everywhere.
EPILOGUE_STATS = os.environ.get("RUNET_NO_EPILOGUE_STATS", "0") != "1"


def _stats_buf(stats, nparts, c, dev):
    """stats: the caller's dict - receives the partials buffer [nparts][c][3] and its row count (blocks.bn_coeff(fused=...) consumes them)."""
    part = torch.empty(nparts * c * 3, device=dev, dtype=torch.float32)
    stats["part"], stats["nparts"] = part, nparts
    return part.data_ptr()


def _conv_x3(mode, x, w_hwio, bias, out, n, h, w, cin, cout, accumulate, stats=None):
    """h, w: the image the 1x1 convolution runs over; the low-resolution side for the transposed modes."""
    wp = conv_x3_weights(w_hwio, mode)
    prof = _PROFILE is not None
    if prof:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    if stats is not None and EPILOGUE_STATS and mode == CONV_FWD:
        sp = _stats_buf(stats, lib.runet_conv_x3_stats_parts(n, h, w), cout, x.device)
        check(lib.runet_conv_x3_stats(x.data_ptr(), ld(x), wp.data_ptr(), bias.data_ptr() if bias is not None else None, out.data_ptr(), ld(out), n, h, w,
                                      cin, cout, mode, int(accumulate), sp, stream()))
    else:
        check(lib.runet_conv_x3(x.data_ptr(), ld(x), wp.data_ptr(), bias.data_ptr() if bias is not None else None, out.data_ptr(), ld(out), n, h, w,
                                cin, cout, mode, int(accumulate), stream()))
    if prof:
        e1.record()
        fl = 2.0 * n * h * w * (4 if mode in (CONVT_FWD, CONVT_DGRAD) else 1) * cin * cout
        _PROFILE.append((lib.runet_conv_x3_kernel_name(n, h, w, cout, mode).decode(), fl, fl, e0, e1))
    return out


def conv_fwd(x, w_hwio, bias=None, out=None, dil=1, accumulate=False, keep_v=None, stats=None, pre=None):
    """stats: dict; when the kernel that serves this convolution can take the BatchNorm statistics of its output in its epilogue it fills
    stats["part"] / stats["nparts"] (blocks.bn_coeff(fused=stats) then skips its pass over the tensor), otherwise leaves it empty.
    pre = (scale, shift, factor_nc or None), only where fuses_act_input(x, w_hwio): the convolution of max(x * scale + shift, 0) * factor."""
    n, h, w, cin = x.shape
    kh, kw, cin_w, cout = w_hwio.shape
    if pre is not None:
        assert fuses_act_input(x, w_hwio) and dil == 1
        return wino4_conv(x, wino4_weights(w_hwio), bias, out=out, accumulate=accumulate, keep_v=keep_v, dil=dil, pre=pre, stats=stats)
    if _bf16_case(cin, cin_w):
        if out is None:
            out = empty_nhwc(n, h, w, cout, x)
        return _igemm_bf16(CONV_FWD, x, w_hwio, bias, out, n, h, w, cin, cout, kh, kw, dil, accumulate, False)
    if _wino4_case(h, w, kh, dil, cin, cout, cin_w):
        return wino4_conv(x, wino4_weights(w_hwio), bias, out=out, accumulate=accumulate, keep_v=keep_v, dil=dil, stats=stats)
    if _wino_case(h, w, kh, dil, cin, cout, cin_w, n, ld(x), ld(out) if out is not None else cout):
        return wino_conv(x, wino_weights(w_hwio), bias, out=out, accumulate=accumulate, stats=stats)
    if out is None:
        out = empty_nhwc(n, h, w, cout, x)
    if kh == 1 and _conv_x3_case(x, cin, cin_w, cout):
        return _conv_x3(CONV_FWD, x, w_hwio, bias, out, n, h, w, cin, cout, accumulate, stats=stats)
    _igemm(CONV_FWD, x.data_ptr(), ld(x), w_hwio.data_ptr(), bias.data_ptr() if bias is not None else None,
           out.data_ptr(), ld(out), n, h, w, cin, cin_w, cout, kh, kw, dil, int(accumulate))
    return out


# the fused F(2x2) kernel with its position products on the BF16 matrix cores (csrc/conv_winograd_x3.hip); RUNET_NO_WINO_X3=1: the f32-MFMA kernel
USE_WINO_X3 = os.environ.get("RUNET_NO_WINO_X3", "0") != "1"


def wino_weights(w_hwio, dgrad=False):
    """HWIO 3x3 weight -> Winograd-domain U[16][K][N] (forward: K=cin, N=cout; dgrad: rotated filter, K=cout, N=cin); under USE_X3 the
    split-plane form Up[16][3][K/8][N][8] bf16 that runet_wino_conv_x3 reads."""
    _, _, cin, cout = w_hwio.shape
    k, n = (cout, cin) if dgrad else (cin, cout)
    wp, dev = w_hwio.data_ptr(), w_hwio.device
    if USE_X3 and USE_WINO_X3 and k % 16 == 0:
        def make_x3(out):
            Up = out if out is not None else torch.empty(lib.runet_wino_x3_pack_elems(k, n), device=dev, dtype=torch.bfloat16)
            check(lib.runet_wino_weights_x3(wp, Up.data_ptr(), cin, cout, int(dgrad), stream()))
            Up.kn = (k, n)
            return Up
        return _cached(w_hwio, "wino2xd" if dgrad else "wino2x", make_x3, desc=(DERIVE_WINO2, wp, cin, cout, int(dgrad)))

    def make(out):
        U = out if out is not None else torch.empty((16, k, n), device=dev, dtype=torch.float32)
        check(lib.runet_wino_weights(wp, U.data_ptr(), cin, cout, int(dgrad), stream()))
        return U
    return _cached(w_hwio, "wino2d" if dgrad else "wino2", make)


def wino_conv(x, U, bias=None, out=None, accumulate=False, stats=None):
    """3x3 'same' convolution (or its data gradient, with dgrad weights) through the fused Winograd kernel."""
    n, h, w, k = x.shape
    x3 = U.dtype == torch.bfloat16                      # split-plane filter (wino_weights under USE_X3)
    nn_ = U.kn[1] if x3 else U.shape[2]
    if out is None:
        out = empty_nhwc(n, h, w, nn_, x)
    if _PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    if x3 and stats is not None and EPILOGUE_STATS:
        sp = _stats_buf(stats, lib.runet_wino_conv_x3_stats_parts(n, h, w), nn_, x.device)
        check(lib.runet_wino_conv_x3_stats(x.data_ptr(), ld(x), U.data_ptr(), bias.data_ptr() if bias is not None else None, out.data_ptr(), ld(out),
                                           n, h, w, k, nn_, int(accumulate), sp, stream()))
    else:
        fn = lib.runet_wino_conv_x3 if x3 else lib.runet_wino_conv
        check(fn(x.data_ptr(), ld(x), U.data_ptr(), bias.data_ptr() if bias is not None else None, out.data_ptr(), ld(out),
                 n, h, w, k, nn_, int(accumulate), stream()))
    if _PROFILE is not None:
        e1.record()
        fl = 2.0 * n * h * w * 9 * k * nn_
        _PROFILE.append(("wino_conv_x3_kernel" if x3 else "wino_conv_kernel", fl, fl * 16.0 / 36.0, e0, e1))
    return out


def wino_ok(h, w, cin, cout):
    return bool(lib.runet_wino_supported(h, w, cin, cout))


# BatchNorm + ReLU (+ Dropout2d) folded into the loads of the F(4x4) input transforms (csrc/conv_winograd4.hip W4Pre): the activation in front
# of a ResidualBlock's conv2 and the BatchNorm-backward dx in front of conv1's gradients are never written.  RUNET_NO_FUSED_BN_INPUT=1: two steps.
FUSE_BN_INPUT = os.environ.get("RUNET_NO_FUSED_BN_INPUT", "0") != "1"


def fuses_act_input(t, w_hwio):
    """conv_fwd(t, w, pre=(scale, shift, factor)) is available: the 3x3 convolution takes the unfused F(4x4) path in fp32."""
    n, h, w, cin = t.shape
    kh, kw, cin_w, cout = w_hwio.shape
    return FUSE_BN_INPUT and not _bf16_case(cin, cin_w) and _wino4_case(h, w, kh, 1, cin, cout, cin_w)


def fuses_bn_bwd_input(dy, w_hwio):
    """conv_dgrad(dy, w, keep_z=..., bn=...) is available: the data gradient takes the adjoint F(4x4) path (Z shared with the weight gradient)."""
    n, h, w, cout = dy.shape
    kh, kw, cin, cout_w = w_hwio.shape
    return (FUSE_BN_INPUT and cout_w == cout and not _bf16_case(cout, cout) and _wino4_case(h, w, kh, 1, cout, cin, cout) and USE_W4_ADJOINT
            and _x3_case(cout) and cin >= 16)


def conv_dgrad(dy, w_hwio, out=None, dil=1, accumulate=False, keep_z=None, bn=None):
    """keep_z: dict that receives {"Z": ...} when the layer takes the adjoint F(4x4) path - hand it to conv_wgrad(..., z=...) of the same layer.
    bn = dict(x, mean, invstd, scale, shift, sums, m_total, mask), only where fuses_bn_bwd_input(dy, w_hwio): the data gradient of
    blocks.bn_bwd_apply(dy, x, ..., relu_shift=shift) without that tensor being written; the weight gradient must then take keep_z's Z."""
    n, h, w, cout = dy.shape
    kh, kw, cin, cout_w = w_hwio.shape
    assert cout_w == cout
    if bn is not None:
        assert fuses_bn_bwd_input(dy, w_hwio) and dil == 1 and keep_z is not None
        return wino4_dgrad_adj(dy, w_hwio, out=out, accumulate=accumulate, dil=dil, keep_z=keep_z, bn=bn)
    if _bf16_case(cout, cout):
        if out is None:
            out = empty_nhwc(n, h, w, cin, dy)
        return _igemm_bf16(CONV_DGRAD, dy, w_hwio, None, out, n, h, w, cout, cin, kh, kw, dil, accumulate, True)
    if _wino4_case(h, w, kh, dil, cout, cin, cout):
        if USE_W4_ADJOINT and _x3_case(cout):
            return wino4_dgrad_adj(dy, w_hwio, out=out, accumulate=accumulate, dil=dil, keep_z=keep_z)
        return wino4_conv(dy, wino4_weights(w_hwio, dgrad=True), None, out=out, accumulate=accumulate, dil=dil)
    if _wino_case(h, w, kh, dil, cout, cin, cout, n, ld(dy), ld(out) if out is not None else cin):
        return wino_conv(dy, wino_weights(w_hwio, dgrad=True), None, out=out, accumulate=accumulate)
    if out is None:
        out = empty_nhwc(n, h, w, cin, dy)
    if kh == 1 and _conv_x3_case(dy, cout, cout, cin):
        return _conv_x3(CONV_DGRAD, dy, w_hwio, None, out, n, h, w, cout, cin, accumulate)
    if USE_TRANSPOSED_DGRAD:
        _igemm(CONV_DGRAD_T, dy.data_ptr(), ld(dy), transposed_weights(w_hwio).data_ptr(), None, out.data_ptr(), ld(out), n, h, w,
               cout, cout, cin, kh, kw, dil, int(accumulate))
    else:
        _igemm(CONV_DGRAD, dy.data_ptr(), ld(dy), w_hwio.data_ptr(), None, out.data_ptr(), ld(out), n, h, w,
               cout, cout, cin, kh, kw, dil, int(accumulate))
    return out


# ---- weight gradients on a side stream: nothing in the backward chain waits for them (only the optimizer / the gradient all-reduce do),
# and they are matrix-core work while much of the chain (BatchNorm / attention backward, Winograd transforms) is HBM-bound.
USE_WGRAD_STREAM = os.environ.get("RUNET_NO_WGRAD_STREAM", "0") != "1"
GRAPH_SIDE = os.environ.get("RUNET_GRAPH_SIDE", "0") == "1"      # measurement knob: keep the fork / join inside a hipGraph capture
SIDE_RECORD_STREAM = os.environ.get("RUNET_SIDE_RECORD_STREAM", "0") == "1"      # measurement knob: the round-2 lifetime rule (Tensor.record_stream)
_side = {}


class wgrad_side_stream:
    """with ops.wgrad_side_stream(): ...backward...  -> conv_wgrad / convt_wgrad launch on a second HIP stream that waits for their inputs;
    leaving the block makes the current stream wait for it.  ops.side_stream() is the active stream (the all-reduce must wait for it too)."""

    def __enter__(self):
        # not under hipGraph capture: a captured fork/join replays correctly but slowly (23 ms instead of 12 ms at 2x256x256 with 8 hardware queues)
        if USE_WGRAD_STREAM and torch.cuda.is_available() and (GRAPH_SIDE or not torch.cuda.is_current_stream_capturing()):
            dev = torch.cuda.current_device()
            if ("s", dev) not in _side:
                _side[("s", dev)] = torch.cuda.Stream(device=dev)
            _side["active"] = _side[("s", dev)]
            _side["active_raw"] = _side["active"].cuda_stream
            _side["torch_switch"] = torch.cuda.is_current_stream_capturing()      # captured fork / join (GRAPH_SIDE): through torch's stream objects
        return self

    def __exit__(self, *exc):
        s = _side.pop("active", None)
        if s is not None:
            torch.cuda.current_stream().wait_stream(s)
        # Tensors the weight-gradient stream read or wrote were kept alive up to here (_on_side) instead of being handed to
        # Tensor.record_stream: they are released in host order BEHIND the wait above, so the caching allocator (which reuses a block for the
        # stream it was allocated on, i.e. the current one) can take them back at once and the same way every step.  With record_stream the
        # reuse of a block depended on whether the GPU had already passed the recording event when the host - up to a step ahead - asked
        # again: the pool never reached a steady state (tools/host_lead.py: ~2 hipMalloc per step, +0.6 GB reserved per step without end,
        # host time per step 25 ms instead of 7).
        _side.pop("keep", None)
        return False


def side_stream():
    return _side.get("active")


# ---- forward-pass branches on the same side stream: the pass is one dependent chain, but a ResidualBlock's shortcut (1x1 convolution +
# BatchNorm statistics) does not depend on its main branch, nor the attention gate's W_x(skip) on the transposed convolution that produces
# the gate signal.  The side stream is idle in the forward pass; the branch joins where its result is consumed.
FWD_BRANCHES = os.environ.get("RUNET_FWD_BRANCHES", "1") != "0"
# Branching costs host time (stream switches, event waits: +1.5-2.5 ms per step measured at 2 x 256^2 and 2 x 64^2, where the step is bound
# by the host issuing launches) and buys GPU time: the models switch it on per forward pass from the input size (branches_pay).
BRANCH_MIN_PIXELS = int(os.environ.get("RUNET_BRANCH_MIN_PIXELS", str(1 << 18)))
_branch_now = True


def branches_pay(n, h, w):
    """Called by a model at the start of its forward pass: forward branches / derived-weight prefetch only for GPU-bound step sizes."""
    global _branch_now
    _branch_now = n * h * w >= BRANCH_MIN_PIXELS
    return _branch_now


class side_branch:
    """br = ops.side_branch(); with br: ...independent work...; later br.join(tensors allocated inside the branch)."""

    def __init__(self, enabled=True):
        self.s = None
        if (enabled and FWD_BRANCHES and _branch_now and USE_WGRAD_STREAM and torch.cuda.is_available() and _side.get("active") is None
                and not torch.cuda.is_current_stream_capturing()):
            dev = torch.cuda.current_device()
            if ("s", dev) not in _side:
                _side[("s", dev)] = torch.cuda.Stream(device=dev)
            self.s = _side[("s", dev)]

    def __enter__(self):
        if self.s is not None:
            self.s.wait_stream(torch.cuda.current_stream())
            self.ctx = torch.cuda.stream(self.s)
            self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.s is not None:
            self.ctx.__exit__(*exc)
        return False

    def join(self, *tensors):
        """The current stream waits for the branch.  `tensors`: results the branch wrote that the current stream goes on to use - allocate
        them BEFORE entering the branch (from the current stream's pool, like everything else the pass keeps); a tensor that was allocated
        inside the branch (the branch stream's pool) is marked with record_stream instead, which is correct but lets the allocator's reuse
        depend on GPU progress (see wgrad_side_stream.__exit__)."""
        if self.s is not None:
            cur = torch.cuda.current_stream()
            cur.wait_stream(self.s)
            for t in tensors:
                if t is not None and (SIDE_RECORD_STREAM or getattr(t, "_runet_main_pool", False) is False):
                    t.record_stream(cur)
            self.s = None


def main_pool(t):
    """Mark a tensor allocated on the current stream that a forward branch will fill: side_branch.join() then needs no record_stream for it."""
    t._runet_main_pool = True
    return t


def _on_side(fn, tensors):
    """Run fn's launches on the weight-gradient stream, behind everything enqueued on the current stream so far.  The kernels take their
    stream from ops.stream(), so the redirection is an override of that handle plus one event record / wait in C - torch's current stream
    is not switched (wait_stream + the stream context manager cost ~25 us of host time per call, 50 calls per step).  Only the live
    conv profile (HIP events recorded on torch's current stream) still goes through the torch-level switch."""
    s = _side.get("active")
    if s is None:
        return fn()
    if _PROFILE is not None or _side.get("torch_switch"):
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            r = fn()
    else:
        raw = _side["active_raw"]
        check(lib.runet_stream_wait(raw, stream()))
        _tls.override = raw
        try:
            r = fn()
        finally:
            _tls.override = None
    if SIDE_RECORD_STREAM:
        for t in tensors:
            if t is not None:
                t.record_stream(s)
        return r
    keep = _side.setdefault("keep", [])
    for t in tensors:
        if t is not None:
            keep.append(t)                       # released by wgrad_side_stream.__exit__, behind the join (no record_stream: see there)
    return r


def conv_wgrad(x, dy, kh, kw, cin_w=None, dil=1, out=None, v=None, on_side=True, z=None):
    if on_side and _side.get("active") is not None:
        if out is None:
            out = torch.empty((kh, kw, x.shape[3] if cin_w is None else cin_w, dy.shape[3]), device=x.device, dtype=torch.float32)
        return _on_side(lambda: _conv_wgrad(x, dy, kh, kw, cin_w, dil, out, v, z), (x, dy, out, v, z))
    return _conv_wgrad(x, dy, kh, kw, cin_w, dil, out, v, z)


def _conv_wgrad(x, dy, kh, kw, cin_w=None, dil=1, out=None, v=None, z=None):
    n, h, w, cin = x.shape
    cout = dy.shape[3]
    cin_w = cin if cin_w is None else cin_w
    if out is None:
        out = torch.empty((kh, kw, cin_w, cout), device=x.device, dtype=torch.float32)
    if _bf16_case(cin, cin_w):
        return _wgrad_bf16(x, dy, out, n, h, w, cin, cout, kh, kw, dil, 0)
    if _wino4_case(h, w, kh, dil, cin, cout, cin_w) and cout >= 16:
        return wino4_wgrad(x, dy, out=out, v=v, dil=dil, z=z)
    if _stem_case(cin, cin_w, cout, kh, dil):
        ws = workspace(lib.runet_stem_wgrad_workspace_floats(n, h, w, cin_w, cout, kh), x.device)
        check(lib.runet_stem_wgrad(x.data_ptr(), ld(x), dy.data_ptr(), ld(dy), out.data_ptr(), ws.data_ptr(), ws.numel(), n, h, w, cin_w, cout, kh,
                                   stream()))
        return out
    prof = _PROFILE is not None
    if prof:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    if USE_WINOGRAD and kh == 3 and dil == 1 and cin_w == cin and h % 2 == 0 and w % 2 == 0 and cin >= 16:
        name = "wino_wgrad_direct_kernel(+reduce)"
        nws = lib.runet_wino_wgrad_workspace_floats(n, h, w, cin, cout)
        ws = workspace(nws, x.device)
        check(lib.runet_wino_wgrad(x.data_ptr(), ld(x), dy.data_ptr(), ld(dy), out.data_ptr(), ws.data_ptr(), ws.numel(), n, h, w, cin, cout, stream()))
    else:
        name = "wgrad_tile_kernel/wgrad_kernel(+reduce)"
        nws = lib.runet_conv_wgrad_workspace_floats(n, h, w, cin_w, cout, kh, kw)
        ws = workspace(nws, x.device)
        check(lib.runet_conv_wgrad(x.data_ptr(), ld(x), dy.data_ptr(), ld(dy), out.data_ptr(), ws.data_ptr(), ws.numel(),
                                   n, h, w, cin, cin_w, cout, kh, kw, dil, 0, stream()))
    if prof:
        e1.record()
        fl = 2.0 * n * h * w * kh * kw * cin_w * cout
        _PROFILE.append((name, fl, fl * 16.0 / 36.0 if name.startswith("wino") else fl, e0, e1))
    return out


def _wgrad_bf16(x, dy, out, n, h, w, cin, cout, kh, kw, dil, transposed):
    prof = _PROFILE is not None
    if prof:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    nws = _lp("runet_conv_wgrad_{}_workspace_floats")(n, h, w, cin, cout, kh, kw, dil, transposed)
    ws = workspace(nws, x.device)
    check(_lp("runet_conv_wgrad_{}")(x.data_ptr(), ld(x), dy.data_ptr(), ld(dy), out.data_ptr(), ws.data_ptr(), ws.numel(), n, h, w, cin, cout,
                                    kh, kw, dil, transposed, stream()))
    if prof:
        e1.record()
        fl = 2.0 * n * h * w * kh * kw * cin * cout
        nbytes = 4.0 * n * h * w * (cin + (4 if transposed else 1) * cout) + 4.0 * kh * kw * cin * cout
        _PROFILE.append((f"wgrad_{_PRECISION}_kernel", fl, fl, e0, e1, nbytes))      # the span includes the slab reduce
    return out


def convt_fwd(x, w_hwio, bias=None, out=None):
    n, h, w, cin = x.shape
    _, _, cin_w, cout = w_hwio.shape
    if out is None:
        out = empty_nhwc(n, 2 * h, 2 * w, cout, x)
    if _bf16_case(cin, cin_w):
        return _igemm_bf16(CONVT_FWD, x, w_hwio, bias, out, n, h, w, cin, cout, 2, 2, 1, False, False)
    if _conv_x3_case(x, cin, cin_w, cout) or (CONVT_X3_NARROW and cout >= 32 and _conv_x3_case(x, cin, cin_w, max(cout, CONV_X3_WIDE_N))):
        return _conv_x3(CONVT_FWD, x, w_hwio, bias, out, n, h, w, cin, cout, False)
    _igemm(CONVT_FWD, x.data_ptr(), ld(x), w_hwio.data_ptr(), bias.data_ptr() if bias is not None else None,
           out.data_ptr(), ld(out), n, h, w, cin, cin_w, cout, 2, 2, 1, 0)
    return out


def convt_dgrad(dy, w_hwio, out=None, accumulate=False):
    n, h2, w2, cout = dy.shape
    _, _, cin, cout_w = w_hwio.shape
    h, w = h2 // 2, w2 // 2
    if out is None:
        out = empty_nhwc(n, h, w, cin, dy)
    if _bf16_case(cout, cout):
        return _igemm_bf16(CONVT_DGRAD, dy, w_hwio, None, out, n, h, w, cout, cin, 2, 2, 1, accumulate, True)
    if _conv_x3_case(dy, cout, cout, cin, taps=4):
        return _conv_x3(CONVT_DGRAD, dy, w_hwio, None, out, n, h, w, cout, cin, accumulate)
    if USE_TRANSPOSED_DGRAD:
        _igemm(CONVT_DGRAD_T, dy.data_ptr(), ld(dy), transposed_weights(w_hwio).data_ptr(), None, out.data_ptr(), ld(out), n, h, w,
               cout, cout, cin, 2, 2, 1, int(accumulate))
    else:
        _igemm(CONVT_DGRAD, dy.data_ptr(), ld(dy), w_hwio.data_ptr(), None, out.data_ptr(), ld(out), n, h, w,
               cout, cout, cin, 2, 2, 1, int(accumulate))
    return out


def convt_wgrad(x, dy, out=None):
    if _side.get("active") is not None:
        if out is None:
            out = torch.empty((2, 2, x.shape[3], dy.shape[3]), device=x.device, dtype=torch.float32)
        return _on_side(lambda: _convt_wgrad(x, dy, out), (x, dy, out))
    return _convt_wgrad(x, dy, out)


def _convt_wgrad(x, dy, out=None):
    n, h, w, cin = x.shape
    cout = dy.shape[3]
    if out is None:
        out = torch.empty((2, 2, cin, cout), device=x.device, dtype=torch.float32)
    if _bf16_case(cin, cin):
        return _wgrad_bf16(x, dy, out, n, h, w, cin, cout, 2, 2, 1, 1)
    nws = lib.runet_conv_wgrad_workspace_floats(n, h, w, cin, cout, 2, 2)
    ws = workspace(nws, x.device)
    check(lib.runet_conv_wgrad(x.data_ptr(), ld(x), dy.data_ptr(), ld(dy), out.data_ptr(), ws.data_ptr(), ws.numel(),
                               n, h, w, cin, cin, cout, 2, 2, 1, 1, stream()))
    return out


# ------------------------------------------------------------------------------- loss / metrics
class _BCELoss(torch.autograd.Function):
    """nn.BCELoss() (mean) on probabilities, ATen semantics (/root/reference/Main_Final.py:551,580)."""

    @staticmethod
    def forward(ctx, prob, target):
        prob, target = prob.contiguous(), target.contiguous()
        n = prob.numel()
        loss = torch.empty((), device=prob.device, dtype=torch.float32)
        part = torch.empty(1024, device=prob.device, dtype=torch.float64)
        check(lib.runet_bce_fwd(prob.data_ptr(), target.data_ptr(), n, part.data_ptr(), loss.data_ptr(), stream()))
        ctx.save_for_backward(prob, target)
        return loss

    @staticmethod
    def backward(ctx, gout):
        prob, target = ctx.saved_tensors
        dprob = torch.empty_like(prob)
        gout = gout.contiguous()
        check(lib.runet_bce_bwd(prob.data_ptr(), target.data_ptr(), gout.data_ptr(), dprob.data_ptr(), prob.numel(), stream()))
        return dprob, None


def bce_loss(prob, target):
    if not prob.is_cuda:
        raise RuntimeError("bce_loss runs on the HIP device only")
    if prob.shape != target.shape:
        raise ValueError(f"shape mismatch {tuple(prob.shape)} vs {tuple(target.shape)}")
    return _BCELoss.apply(prob, target.to(torch.float32))


class _CrossEntropy(torch.autograd.Function):
    """nn.CrossEntropyLoss() (mean over all pixels) on [N, C, H, W] logits and int64 [N, H, W] targets
    (/root/reference/train_water_segmentation.py:304)."""

    @staticmethod
    def forward(ctx, logits, target):
        logits, target = logits.contiguous(), target.contiguous()
        n, c, h, w = logits.shape
        loss = torch.empty((), device=logits.device, dtype=torch.float32)
        part = torch.empty(1024, device=logits.device, dtype=torch.float64)
        check(lib.runet_ce_fwd(logits.data_ptr(), target.data_ptr(), n, c, h * w, part.data_ptr(), loss.data_ptr(), stream()))
        ctx.save_for_backward(logits, target)
        return loss

    @staticmethod
    def backward(ctx, gout):
        logits, target = ctx.saved_tensors
        n, c, h, w = logits.shape
        dz = torch.empty_like(logits)
        gout = gout.contiguous().to(torch.float32)
        check(lib.runet_ce_bwd(logits.data_ptr(), target.data_ptr(), gout.data_ptr(), dz.data_ptr(), n, c, h * w, stream()))
        return dz, None


def cross_entropy(logits, target):
    if not logits.is_cuda or logits.dtype != torch.float32 or logits.dim() != 4:
        raise RuntimeError("cross_entropy takes float32 [N, C, H, W] logits on the HIP device")
    if target.dtype != torch.int64 or tuple(target.shape) != (logits.shape[0], logits.shape[2], logits.shape[3]):
        raise ValueError(f"target must be int64 [N, H, W], got {target.dtype} {tuple(target.shape)}")
    if not 2 <= logits.shape[1] <= 8:
        raise ValueError("2..8 classes")
    return _CrossEntropy.apply(logits, target)


class _Bilinear(torch.autograd.Function):
    """F.interpolate(x, size, mode='bilinear', align_corners=False) on [N, C, H, W] device tensors (the reference's harness resizes
    the model output to the mask size when they differ: /root/reference/Main_Final.py:577-578,596-597,648-649)."""

    @staticmethod
    def forward(ctx, x, size):
        x = x.contiguous()
        n, c, h, w = x.shape
        ho, wo = int(size[0]), int(size[1])
        y = torch.empty((n, c, ho, wo), device=x.device, dtype=torch.float32)
        check(lib.runet_bilinear_fwd(x.data_ptr(), y.data_ptr(), n * c, h, w, ho, wo, stream()))
        ctx.shape = (n, c, h, w, ho, wo)
        return y

    @staticmethod
    def backward(ctx, dy):
        n, c, h, w, ho, wo = ctx.shape
        dy = dy.contiguous()
        dx = torch.empty((n, c, h, w), device=dy.device, dtype=torch.float32)
        check(lib.runet_bilinear_bwd(dy.data_ptr(), dx.data_ptr(), n * c, h, w, ho, wo, stream()))
        return dx, None


def bilinear_resize(x, size):
    if not x.is_cuda or x.dtype != torch.float32 or x.dim() != 4:
        raise RuntimeError("bilinear_resize takes a float32 [N, C, H, W] HIP-device tensor")
    return _Bilinear.apply(x, tuple(size))


def match_size(outputs, masks):
    """The reference harness's size guard: outputs resized to the mask's H x W when the shapes differ."""
    if outputs.shape != masks.shape:
        outputs = bilinear_resize(outputs, masks.shape[-2:])
    return outputs


def seg_counts(pred, target, threshold=0.5):
    """pred, target: [N, ...] -> int64 [N, 4] = (tp, predicted positives, target positives, agreeing pixels)."""
    pred, target = pred.contiguous(), target.contiguous().to(torch.float32)
    n = pred.shape[0]
    per = pred.numel() // n
    counts = torch.empty((n, 4), device=pred.device, dtype=torch.int64)
    check(lib.runet_seg_counts(pred.data_ptr(), target.data_ptr(), counts.data_ptr(), n, per, float(threshold), stream()))
    return counts


# ------------------------------------------------------------------------------- general geometry (DeepLabV3+ baseline)
def conv_out_hw(h, w, k, stride, pad, dil):
    return (h + 2 * pad - dil * (k - 1) - 1) // stride + 1, (w + 2 * pad - dil * (k - 1) - 1) // stride + 1


def conv_general_fwd(x, w_hwio, bias, stride, pad, dil=1, out=None):
    n, h, w, cin = x.shape
    kh, kw, cin_w, cout = w_hwio.shape
    ho, wo = conv_out_hw(h, w, kh, stride, pad, dil)
    if out is None:
        out = empty_nhwc(n, ho, wo, cout, x)
    check(lib.runet_conv2d_general(x.data_ptr(), ld(x), w_hwio.data_ptr(), bias.data_ptr() if bias is not None else None, out.data_ptr(),
                                   ld(out), n, h, w, cin, cin_w, cout, kh, kw, stride, pad, dil, CONV_FWD, 0, stream()))
    return out


def conv_general_dgrad(dy, w_hwio, hin, win, stride, pad, dil=1, out=None, accumulate=False):
    n, ho, wo, cout = dy.shape
    kh, kw, cin, _ = w_hwio.shape
    if out is None:
        out = empty_nhwc(n, hin, win, cin, dy)
    check(lib.runet_conv2d_general(dy.data_ptr(), ld(dy), w_hwio.data_ptr(), None, out.data_ptr(), ld(out), n, hin, win, cout, cout, cin,
                                   kh, kw, stride, pad, dil, CONV_DGRAD, int(accumulate), stream()))
    return out


def conv_general_wgrad(x, dy, kh, kw, stride, pad, dil=1, cin_w=None, out=None):
    n, h, w, cin = x.shape
    cout = dy.shape[3]
    cin_w = cin if cin_w is None else cin_w
    if out is None:
        out = torch.empty((kh, kw, cin_w, cout), device=x.device, dtype=torch.float32)
    ws = workspace(lib.runet_conv_wgrad_general_workspace_floats(dy.shape[0] * dy.shape[1] * dy.shape[2], cin_w, cout, kh, kw), x.device)
    check(lib.runet_conv_wgrad_general(x.data_ptr(), ld(x), dy.data_ptr(), ld(dy), out.data_ptr(), ws.data_ptr(), ws.numel(), n, h, w, cin, cin_w,
                                       cout, kh, kw, stride, pad, dil, 0, stream()))
    return out


def convt4_fwd(x, w_hwio, bias, out=None):
    n, h, w, cin = x.shape
    cout = w_hwio.shape[3]
    if out is None:
        out = empty_nhwc(n, 2 * h, 2 * w, cout, x)
    check(lib.runet_convt4_igemm(x.data_ptr(), ld(x), w_hwio.data_ptr(), bias.data_ptr() if bias is not None else None, out.data_ptr(), ld(out),
                                 n, h, w, cin, cout, CONVT_FWD, 0, stream()))
    return out


def convt4_dgrad(dy, w_hwio, out=None):
    n, h2, w2, cout = dy.shape
    cin = w_hwio.shape[2]
    if out is None:
        out = empty_nhwc(n, h2 // 2, w2 // 2, cin, dy)
    check(lib.runet_convt4_igemm(dy.data_ptr(), ld(dy), w_hwio.data_ptr(), None, out.data_ptr(), ld(out), n, h2 // 2, w2 // 2, cout, cin,
                                 CONVT_DGRAD, 0, stream()))
    return out


def convt4_wgrad(x, dy, out=None):
    n, h, w, cin = x.shape
    cout = dy.shape[3]
    if out is None:
        out = torch.empty((4, 4, cin, cout), device=x.device, dtype=torch.float32)
    ws = workspace(lib.runet_conv_wgrad_general_workspace_floats(n * h * w, cin, cout, 4, 4), x.device)
    check(lib.runet_conv_wgrad_general(x.data_ptr(), ld(x), dy.data_ptr(), ld(dy), out.data_ptr(), ws.data_ptr(), ws.numel(), n, h, w, cin, cin,
                                       cout, 4, 4, 2, 1, 1, 1, stream()))
    return out


# ------------------------------------------------------------------------------- Winograd F(4x4,3x3), unfused (deep layers)
_ws4 = {}


def _workspace4(nfloats, device):
    key = (device.index, stream())
    return grow(_ws4, key, nfloats, device, 1)


def wino4_ok(h, w, k, n):
    return bool(lib.runet_wino4_supported(h, w, k, n))


# fp32 position-GEMMs on the BF16 matrix cores: exact three-way operand split, six bf16 MFMAs per product (csrc/gemm_split.hip) - fp32-accurate
# (errors against float64 equal to the f32-MFMA kernels', tools/bench_gemm_x3.py) at 6/16 of the matrix time.  RUNET_NO_X3=1: f32 MFMA.
USE_X3 = os.environ.get("RUNET_NO_X3", "0") != "1"


def _x3_case(k):
    return USE_X3 and k % 16 == 0


# The F(4x4) data gradient as the ADJOINT of the forward algorithm (csrc/conv_winograd4.hip wino4_output_adj_kernel): dx = overlap-add of
# B [U^T .* (A dy A^T)] B^T.  Z = A dy A^T is the transform the weight gradient of the same layer needs, so one pass over dy serves both, and the
# filter needs no rotated second transform.  RUNET_NO_W4_ADJOINT=1: the data gradient as a convolution with the rotated filter (round 2's form).
USE_W4_ADJOINT = os.environ.get("RUNET_NO_W4_ADJOINT", "0") != "1"


def wino4_weights(w_hwio, dgrad=False, adjoint=False):
    """HWIO 3x3 weight -> U[36][K][N] (forward: K=cin, N=cout; dgrad: rotated filter, K=cout, N=cin); under USE_X3 the split-plane
    packing of it (runet_gemm_x3_pack: [36][3][K/8][N][8] bf16) that runet_wino4_conv_x3 / runet_gemm_x3_batched read."""
    _, _, cin, cout = w_hwio.shape
    dgrad = dgrad or adjoint
    k, n = (cout, cin) if dgrad else (cin, cout)
    wp, dev = w_hwio.data_ptr(), w_hwio.device
    if _x3_case(k):
        def make_x3(out):
            Up = out if out is not None else torch.empty(lib.runet_gemm_x3_pack_elems(36, k, n), device=dev, dtype=torch.bfloat16)
            check(lib.runet_wino4_weights_x3(wp, Up.data_ptr(), cin, cout, 2 if adjoint else int(dgrad), stream()))      # transform + split in one pass
            Up.kn = (k, n)
            return Up
        return _cached(w_hwio, "wino4xad" if adjoint else ("wino4xd" if dgrad else "wino4x"), make_x3,
                       desc=(DERIVE_WINO4, wp, cin, cout, 2 if adjoint else int(dgrad)))
    assert not adjoint, "the adjoint data gradient exists for the split-operand path only"

    def make(out):
        U = out if out is not None else torch.empty((36, k, n), device=dev, dtype=torch.float32)
        check(lib.runet_wino4_weights(wp, U.data_ptr(), cin, cout, int(dgrad), stream()))
        return U
    return _cached(w_hwio, "wino4d" if dgrad else "wino4", make)


def wino4_conv(x, U, bias=None, out=None, accumulate=False, keep_v=None, dil=1, pre=None, stats=None):
    """keep_v: dict that receives {"V": transformed input [36*T*K]} - the weight gradient of the same convolution reuses it
    (conv_wgrad(..., v=...)) instead of transforming x again.  pre, stats: see conv_fwd."""
    n, h, w, k = x.shape
    x3 = U.dtype == torch.bfloat16                      # split-plane packing (wino4_weights under USE_X3)
    nn_ = U.kn[1] if x3 else U.shape[2]
    if out is None:
        out = empty_nhwc(n, h, w, nn_, x)
    bp = bias.data_ptr() if bias is not None else None
    t = n * (h // 4) * (w // 4)
    sparts = lib.runet_wino4_output_stats_parts(n, h, w, nn_, dil) if (stats is not None and EPILOGUE_STATS) else 0
    if _PROFILE is None and keep_v is None and pre is None and sparts == 0:
        ws = _workspace4(lib.runet_wino4_workspace_floats(n, h, w, k, nn_), x.device)
        fn = lib.runet_wino4_conv_x3 if x3 else lib.runet_wino4_conv
        check(fn(x.data_ptr(), ld(x), U.data_ptr(), bp, out.data_ptr(), ld(out), n, h, w, k, nn_, dil, int(accumulate), ws.data_ptr(),
                 ws.numel(), stream()))
        return out
    # the same three kernels through their own entry points (V kept for the backward pass / HIP events around the position-GEMM alone)
    if keep_v is not None:
        vt = torch.empty(36 * t * k, device=x.device, dtype=torch.float32)
        keep_v["V"] = vt
        V, M = vt.data_ptr(), _workspace4(36 * t * nn_, x.device).data_ptr()
    else:
        ws = _workspace4(36 * t * (k + nn_), x.device)
        V, M = ws.data_ptr(), ws.data_ptr() + 4 * 36 * t * k
    if pre is None:
        check(lib.runet_wino4_input(x.data_ptr(), ld(x), k, n, h, w, dil, 0, V, stream()))
    else:
        sc, sh, fac = pre
        check(lib.runet_wino4_input_act(x.data_ptr(), ld(x), k, n, h, w, dil, sc.data_ptr(), sh.data_ptr(), fac.data_ptr() if fac is not None else None,
                                        V, stream()))
    if _PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    if x3:
        check(lib.runet_gemm_x3_batched(V, k, t * k, U.data_ptr(), M, nn_, t * nn_, 36, t, k, nn_, stream()))
    else:
        check(lib.runet_gemm_batched(V, k, t * k, U.data_ptr(), k * nn_, M, nn_, t * nn_, 36, t, k, nn_, stream()))
    if _PROFILE is not None:
        e1.record()
        fl = 2.0 * 36 * t * k * nn_          # the position-GEMMs' own FLOPs; the convolution they implement is 4x that in direct-conv FLOPs
        kname = lib.runet_gemm_x3_kernel_name(36, t, k, nn_) if x3 else lib.runet_gemm_batched_kernel_name(36, t, k, nn_)
        _PROFILE.append((kname.decode(), 4.0 * fl, fl, e0, e1))
    if sparts > 0:
        sp = _stats_buf(stats, sparts, nn_, x.device)
        check(lib.runet_wino4_output_stats(M, nn_, n, h, w, bp, out.data_ptr(), ld(out), int(accumulate), sp, stream()))
    else:
        check(lib.runet_wino4_output(M, nn_, n, h, w, dil, bp, out.data_ptr(), ld(out), int(accumulate), stream()))
    return out


def wino4_dgrad_adj(dy, w_hwio, out=None, accumulate=False, dil=1, keep_z=None, bn=None):
    """Data gradient of conv3x3(x, w) by the adjoint form.  keep_z: dict that receives {"Z": A dy A^T [36*T*cout]} for the weight gradient of
    the same layer (conv_wgrad(..., z=...))."""
    n, h, w, cout = dy.shape
    cin = w_hwio.shape[2]
    Ua = wino4_weights(w_hwio, adjoint=True)
    if out is None:
        out = empty_nhwc(n, h, w, cin, dy)
    t = n * (h // 4) * (w // 4)
    if keep_z is not None:
        zt = torch.empty(36 * t * cout, device=dy.device, dtype=torch.float32)
        keep_z["Z"] = zt
        Z, M = zt.data_ptr(), _workspace4(36 * t * cin, dy.device).data_ptr()
    else:
        ws = _workspace4(36 * t * (cout + cin), dy.device)
        Z, M = ws.data_ptr(), ws.data_ptr() + 4 * 36 * t * cout
    if bn is None:
        check(lib.runet_wino4_input(dy.data_ptr(), ld(dy), cout, n, h, w, dil, 1, Z, stream()))
    else:
        bx, mk = bn["x"], bn.get("mask")
        check(lib.runet_wino4_input_bn_bwd(dy.data_ptr(), ld(dy), bx.data_ptr(), ld(bx), cout, n, h, w, dil, bn["mean"].data_ptr(), bn["invstd"].data_ptr(),
                                           bn["scale"].data_ptr(), bn["shift"].data_ptr(), bn["sums"].data_ptr(), mk.data_ptr() if mk is not None else None,
                                           int(bn["m_total"]), Z, stream()))
    if _PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    check(lib.runet_gemm_x3_batched(Z, cout, t * cout, Ua.data_ptr(), M, cin, t * cin, 36, t, cout, cin, stream()))
    if _PROFILE is not None:
        e1.record()
        fl = 2.0 * 36 * t * cout * cin
        _PROFILE.append((lib.runet_gemm_x3_kernel_name(36, t, cout, cin).decode(), 4.0 * fl, fl, e0, e1))
    check(lib.runet_wino4_output_adj(M, cin, n, h, w, dil, out.data_ptr(), ld(out), int(accumulate), stream()))
    return out


def wino4_wgrad(x, dy, out=None, v=None, dil=1, z=None):
    """v: the forward pass's transformed input (wino4_conv(keep_v=...)) - skips the B^T d B pass over x;  z: A dy A^T from the adjoint data
    gradient of the same layer (wino4_dgrad_adj(keep_z=...)) - skips the pass over dy."""
    n, h, w, cin = x.shape
    cout = dy.shape[3]
    if out is None:
        out = torch.empty((3, 3, cin, cout), device=x.device, dtype=torch.float32)
    ws = _workspace4(lib.runet_wino4_wgrad_workspace_floats(n, h, w, cin, cout), x.device)
    x3 = USE_X3 and (n * (h // 4) * (w // 4)) % 16 == 0 and cin > 64 and cout > 64
    if _PROFILE is None and v is None and z is None and not x3:
        check(lib.runet_wino4_wgrad(x.data_ptr(), ld(x), dy.data_ptr(), ld(dy), out.data_ptr(), ws.data_ptr(), ws.numel(), n, h, w, cin, cout, dil, stream()))
        return out
    t = n * (h // 4) * (w // 4)
    V = ws.data_ptr()
    Z = V + 4 * 36 * t * cin
    dU = Z + 4 * 36 * t * cout
    rps = lib.runet_wino4_wgrad_rows_per_split(n, h, w, cin, cout)
    if v is None:
        check(lib.runet_wino4_input(x.data_ptr(), ld(x), cin, n, h, w, dil, 0, V, stream()))
    else:
        assert v.numel() == 36 * t * cin
        V = v.data_ptr()
    if z is None:
        check(lib.runet_wino4_input(dy.data_ptr(), ld(dy), cout, n, h, w, dil, 1, Z, stream()))
    else:
        assert z.numel() == 36 * t * cout
        Z = z.data_ptr()
    if _PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    if x3:
        check(lib.runet_gemm_x3_tn_batched(V, cin, t * cin, Z, cout, t * cout, dU, 36, t, cin, cout, rps, stream()))
    else:
        check(lib.runet_gemm_tn_batched(V, cin, t * cin, Z, cout, t * cout, dU, 36, t, cin, cout, rps, stream()))
    if _PROFILE is not None:
        e1.record()
        fl = 2.0 * 36 * t * cin * cout
        _PROFILE.append(("gemm_tn_x3_kernel" if x3 else ("gemm_tn_kernel<true>" if t % 16 == 0 else "gemm_tn_kernel<false>"), 4.0 * fl, fl, e0, e1))
    check(lib.runet_wino4_wgrad_output(dU, -(-t // rps), cin, cout, out.data_ptr(), stream()))
    return out


# ---- gradients at fixed addresses for the models without a flat arena (DeepLabV3+, plain U-Net).  Their backward passes produce fresh
# tensors; handed to autograd these became new `p.grad` tensors every step, so FusedAdam's device pointer table was rebuilt and uploaded
# (a blocking host-to-device copy) every step - the host could never run ahead of the GPU (tools/host_lead.py: host time == GPU time) - and a
# captured step was impossible.  deliver_grads copies them (one multi-tensor launch) into per-parameter buffers that live as long as the model
# and assigns those as p.grad itself.
def deliver_grads(net, params, grads):
    """params / grads: matching lists (grads in the parameters' logical shapes).  Sets p.grad (adds to an existing one) -> None"""
    bufs = getattr(net, "_grad_bufs", None)
    if bufs is None or len(bufs) != len(params) or any(b.shape != p.shape or b.device != p.device for b, p in zip(bufs, params)):
        bufs = net._grad_bufs = [torch.empty_like(p, memory_format=torch.preserve_format) for p in params]
    fresh = [p.grad is None for p in params]
    if all(fresh):
        torch._foreach_copy_(bufs, list(grads))
        for p, b in zip(params, bufs):
            if p.requires_grad:
                p.grad = b
        return
    for p, g, f in zip(params, grads, fresh):      # gradient accumulation: the rare path
        if not p.requires_grad:
            continue
        if f:
            p.grad = g.clone()
        else:
            p.grad.add_(g)
