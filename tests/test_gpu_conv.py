"""GPU parity: implicit-GEMM convolutions (fwd / dgrad / wgrad / transposed) through the C ABI
vs stock torch CPU fp32 ops on the same seeded inputs.  Tolerance: fp32 MFMA accumulates in a
different order than oneDNN, so |err| <= 2e-4 * (1 + |ref|) on O(1) data with K <= 9216."""
import importlib

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

prng = importlib.import_module("eusipco-2026-robust-unet_amd.portable_rng")


def _ops():
    return importlib.import_module("eusipco-2026-robust-unet_amd.ops")


def rnd(shape, seed, std=1.0):
    return torch.from_numpy(prng.normal_f32(shape, seed, std))


def close(a, b, tol=2e-4):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    err = (a - b).abs()
    lim = tol * (1.0 + b.abs())
    assert bool((err <= lim).all()), f"max err {err.max().item():.3e} (ref max {b.abs().max().item():.3e})"


CASES = [
    # n, h, w, cin, cout, k, dil
    (2, 8, 8, 16, 32, 3, 1),
    (2, 16, 16, 32, 64, 3, 1),
    (1, 12, 20, 64, 128, 3, 1),      # non-square, tile tails
    (2, 16, 16, 32, 16, 3, 2),       # dilation 2, narrow cout
    (2, 16, 16, 32, 48, 3, 4),       # dilation 4, cout not a multiple of 32
    (3, 4, 4, 128, 256, 3, 1),       # tiny spatial (config-1 bottleneck)
    (2, 16, 16, 64, 32, 1, 1),       # 1x1
    (2, 8, 8, 256, 8, 1, 1),         # 1x1, cout 8
    (1, 64, 64, 64, 64, 3, 1),       # 256x64 tile path (P = 4096 < threshold) and more
    (2, 256, 256, 16, 64, 3, 1),     # large M: 256-row tiles
    (2, 16, 16, 256, 128, 1, 1),     # wide 1x1: weight gradient through the TN GEMM (16-row tiles: FAST loader), SIMPLE igemm loader
    (1, 6, 10, 128, 384, 1, 1),      # wide 1x1, 60 pixels (not a multiple of 16: the general TN loader), cout not a multiple of 128
    (2, 16, 16, 256, 128, 3, 2),     # DilatedBlock shapes: dilation through the F(4x4) sub-image path when Winograd is on
    (1, 16, 32, 128, 256, 3, 4),
]


@pytest.mark.parametrize("wino", [False, True])
@pytest.mark.parametrize("n,h,w,cin,cout,k,dil", CASES)
def test_conv_fwd_dgrad_wgrad(n, h, w, cin, cout, k, dil, wino, monkeypatch):
    ops = _ops()
    monkeypatch.setattr(ops, "USE_WINOGRAD", wino)      # direct implicit GEMM and (where the shape allows) Winograd
    dev = torch.device("cuda:0")
    x = rnd((n, cin, h, w), 1)
    wt = rnd((cout, cin, k, k), 2, std=(2.0 / (cin * k * k)) ** 0.5)
    b = rnd((cout,), 3)
    gy = rnd((n, cout, h, w), 4)
    xr = x.clone().requires_grad_(True)
    wr = wt.clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, b, padding=dil * (k // 2), dilation=dil)
    yr.backward(gy)

    xg = x.permute(0, 2, 3, 1).contiguous().to(dev)
    wg = wt.permute(2, 3, 1, 0).contiguous().to(dev)      # HWIO
    gyg = gy.permute(0, 2, 3, 1).contiguous().to(dev)
    y = ops.conv_fwd(xg, wg, b.to(dev), dil=dil)
    close(y.permute(0, 3, 1, 2), yr)
    dx = ops.conv_dgrad(gyg, wg, dil=dil)
    close(dx.permute(0, 3, 1, 2), xr.grad)
    dw = ops.conv_wgrad(xg, gyg, k, k, dil=dil)
    close(dw.permute(3, 2, 0, 1), wr.grad, tol=3e-4 * max(1.0, (n * h * w / 256.0) ** 0.5))
    # accumulate + channel-slice destination (concat buffer)
    buf = torch.full((n, h, w, cout + 16), 7.0, device=dev)
    ops.conv_fwd(xg, wg, None, out=buf[..., 16:], dil=dil, accumulate=True)
    close(buf[..., 16:].permute(0, 3, 1, 2), yr - b.view(1, -1, 1, 1) + 7.0)
    assert float(buf[..., :16].min()) == 7.0 and float(buf[..., :16].max()) == 7.0


def test_conv_stem_rgb_padded_to_4():
    """Cin = 3 stem: x zero-padded to 4 channels, weight keeps its 3 rows (cin_w = 3)."""
    ops = _ops()
    dev = torch.device("cuda:0")
    n, h, w, cout = 2, 32, 32, 64
    x = rnd((n, 3, h, w), 5)
    wt = rnd((cout, 3, 3, 3), 6, std=0.2)
    gy = rnd((n, cout, h, w), 7)
    wr = wt.clone().requires_grad_(True)
    yr = F.conv2d(x, wr, None, padding=1)
    yr.backward(gy)
    xp = torch.zeros((n, h, w, 4))
    xp[..., :3] = x.permute(0, 2, 3, 1)
    xg = xp.to(dev)
    wg = wt.permute(2, 3, 1, 0).contiguous().to(dev)
    close(ops.conv_fwd(xg, wg).permute(0, 3, 1, 2), yr)
    dw = ops.conv_wgrad(xg, gy.permute(0, 2, 3, 1).contiguous().to(dev), 3, 3, cin_w=3)
    assert tuple(dw.shape) == (3, 3, 3, cout)
    close(dw.permute(3, 2, 0, 1), wr.grad, tol=5e-4)


@pytest.mark.parametrize("n,h,w,cin_w,cout", [(2, 32, 32, 3, 64), (1, 24, 40, 3, 32), (3, 8, 72, 1, 64), (2, 16, 16, 2, 32), (2, 256, 256, 3, 64)])
def test_stem_kernels(n, h, w, cin_w, cout):
    """csrc/conv_stem.hip: conv3x3 + the 1x1 shortcut of the RGB stem in one launch, and both weight gradients (pixels as the MFMA
    contraction), against torch CPU conv2d; tiles that overhang the image, 1..3 real channels, garbage in the padding channel."""
    ops = _ops()
    dev = torch.device("cuda:0")
    x = rnd((n, cin_w, h, w), 11)
    w3 = rnd((cout, cin_w, 3, 3), 12, std=0.3)
    w1 = rnd((cout, cin_w, 1, 1), 13, std=0.5)
    g3, g1 = rnd((n, cout, h, w), 14), rnd((n, cout, h, w), 15)
    w3r, w1r = w3.clone().requires_grad_(True), w1.clone().requires_grad_(True)
    y3r, y1r = F.conv2d(x, w3r, None, padding=1), F.conv2d(x, w1r, None)
    (y3r * g3).sum().backward()
    (y1r * g1).sum().backward()
    xp = torch.full((n, h, w, 4), 123.0)              # channels >= cin_w must not matter
    xp[..., :cin_w] = x.permute(0, 2, 3, 1)
    xg = xp.to(dev)
    w3g, w1g = w3.permute(2, 3, 1, 0).contiguous().to(dev), w1.permute(2, 3, 1, 0).contiguous().to(dev)
    assert ops._stem_case(4, cin_w, cout, 3)
    y3, y1 = ops.stem_conv(xg, w3g, w1g)
    close(y3.permute(0, 3, 1, 2), y3r)
    close(y1.permute(0, 3, 1, 2), y1r)
    y3b, none = ops.stem_conv(xg, w3g)
    assert none is None and torch.equal(y3b, y3)
    tol = 3e-4 * max(1.0, (n * h * w / 256.0) ** 0.5)
    dw3 = ops.conv_wgrad(xg, g3.permute(0, 2, 3, 1).contiguous().to(dev), 3, 3, cin_w=cin_w)
    assert tuple(dw3.shape) == (3, 3, cin_w, cout)
    close(dw3.permute(3, 2, 0, 1), w3r.grad, tol=tol)
    dw1 = ops.conv_wgrad(xg, g1.permute(0, 2, 3, 1).contiguous().to(dev), 1, 1, cin_w=cin_w)
    close(dw1.permute(3, 2, 0, 1), w1r.grad, tol=tol)
    again = ops.conv_wgrad(xg, g3.permute(0, 2, 3, 1).contiguous().to(dev), 3, 3, cin_w=cin_w)
    assert torch.equal(again, dw3), "the stem weight gradient must be bitwise reproducible"


@pytest.mark.parametrize("n,h,w,cin,cout", [(2, 4, 4, 64, 32), (1, 16, 16, 128, 64), (2, 8, 8, 32, 16)])
def test_conv_transpose_k2s2(n, h, w, cin, cout):
    ops = _ops()
    dev = torch.device("cuda:0")
    x = rnd((n, cin, h, w), 8)
    wt = rnd((cin, cout, 2, 2), 9, std=(1.0 / cin) ** 0.5)
    b = rnd((cout,), 10)
    gy = rnd((n, cout, 2 * h, 2 * w), 11)
    xr, wr = x.clone().requires_grad_(True), wt.clone().requires_grad_(True)
    yr = F.conv_transpose2d(xr, wr, b, stride=2)
    yr.backward(gy)
    xg = x.permute(0, 2, 3, 1).contiguous().to(dev)
    wg = wt.permute(2, 3, 0, 1).contiguous().to(dev)      # [2,2,cin,cout]
    gyg = gy.permute(0, 2, 3, 1).contiguous().to(dev)
    buf = torch.zeros((n, 2 * h, 2 * w, 2 * cout), device=dev)
    ops.convt_fwd(xg, wg, b.to(dev), out=buf[..., cout:])          # right half of a concat buffer
    close(buf[..., cout:].permute(0, 3, 1, 2), yr)
    assert float(buf[..., :cout].abs().max()) == 0.0
    close(ops.convt_dgrad(gyg, wg).permute(0, 3, 1, 2), xr.grad)
    close(ops.convt_wgrad(xg, gyg).permute(2, 3, 0, 1), wr.grad, tol=5e-4)


def test_conv_rejects_bad_shapes():
    ops = _ops()
    dev = torch.device("cuda:0")
    x = torch.zeros((1, 4, 4, 6), device=dev)           # cin not a multiple of 4
    w = torch.zeros((3, 3, 6, 16), device=dev)
    with pytest.raises(RuntimeError):
        ops.conv_fwd(x, w)


@pytest.mark.parametrize("n,h,w,cin,cout", [(2, 8, 8, 16, 32), (1, 12, 20, 64, 128), (3, 4, 4, 128, 256), (2, 16, 16, 32, 48),
                                            (2, 64, 64, 64, 64), (1, 32, 32, 256, 64), (5, 6, 10, 48, 32)])
def test_winograd_fwd_and_dgrad(n, h, w, cin, cout):
    """Fused Winograd F(2x2,3x3) kernel vs torch CPU conv (forward with bias, data gradient with accumulate into a slice)."""
    ops = _ops()
    dev = torch.device("cuda:0")
    x = rnd((n, cin, h, w), 21)
    wt = rnd((cout, cin, 3, 3), 22, std=(2.0 / (cin * 9)) ** 0.5)
    b = rnd((cout,), 23)
    gy = rnd((n, cout, h, w), 24)
    xr = x.clone().requires_grad_(True)
    yr = F.conv2d(xr, wt, b, padding=1)
    yr.backward(gy)
    xg = x.permute(0, 2, 3, 1).contiguous().to(dev)
    wg = wt.permute(2, 3, 1, 0).contiguous().to(dev)
    gyg = gy.permute(0, 2, 3, 1).contiguous().to(dev)
    assert ops.wino_ok(h, w, cin, cout)
    y = ops.wino_conv(xg, ops.wino_weights(wg), b.to(dev))
    close(y.permute(0, 3, 1, 2), yr, tol=3e-4)
    buf = torch.full((n, h, w, cin + 8), 3.0, device=dev)
    ops.wino_conv(gyg, ops.wino_weights(wg, dgrad=True), None, out=buf[..., 8:], accumulate=True)
    close(buf[..., 8:].permute(0, 3, 1, 2), xr.grad + 3.0, tol=3e-4)
    assert float(buf[..., :8].min()) == 3.0 and float(buf[..., :8].max()) == 3.0


@pytest.mark.parametrize("n,h,w,cin,cout,dil", [(2, 8, 8, 32, 48, 1), (3, 12, 20, 64, 32, 1), (2, 16, 16, 256, 128, 1), (1, 4, 4, 128, 256, 1),
                                                (16, 16, 16, 128, 128, 1), (2, 16, 16, 64, 32, 2), (2, 16, 16, 32, 64, 4), (3, 8, 24, 16, 16, 2),
                                                (1, 32, 16, 128, 64, 4), (2, 16, 16, 512, 256, 2)])
def test_winograd_f4_matches_torch(n, h, w, cin, cout, dil):
    """Unfused F(4x4,3x3): forward (+bias, +accumulate), data gradient and weight gradient against torch CPU conv2d.  dil > 1: the
    dilated convolution as dil*dil dilation-1 convolutions over the sub-images (the DilatedBlock's conv3 / conv4, Main_Final.py:207-208)."""
    import importlib
    import torch.nn.functional as F
    ops = importlib.import_module("eusipco-2026-robust-unet_amd.ops")
    g = torch.Generator().manual_seed(n * 100 + h + cin)
    x = torch.randn(n, cin, h, w, generator=g, requires_grad=True)
    wt = (torch.randn(cout, cin, 3, 3, generator=g) / (9 * cin) ** 0.5).requires_grad_(True)
    b = torch.randn(cout, generator=g)
    y = F.conv2d(x, wt, b, 1, dil, dil)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    dev = "cuda:0"
    xd = x.detach().permute(0, 2, 3, 1).contiguous().to(dev)
    dyd = dy.permute(0, 2, 3, 1).contiguous().to(dev)
    wd = wt.detach().permute(2, 3, 1, 0).contiguous().to(dev)
    assert ops.wino4_ok(h // dil, w // dil, cin, cout)

    def close(got, ref, tol, msg):
        ref = ref.detach().numpy()
        np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=tol, atol=tol * float(np.abs(ref).max()), err_msg=msg)

    U, Ud = ops.wino4_weights(wd), ops.wino4_weights(wd, dgrad=True)
    got = ops.wino4_conv(xd, U, b.to(dev), dil=dil)
    close(got.permute(0, 3, 1, 2), y, 2e-4, "fwd")
    base = torch.randn(n, h, w, cout, generator=g).to(dev)
    acc = base.clone()
    ops.wino4_conv(xd, U, None, out=acc, accumulate=True, dil=dil)
    close((acc - base).permute(0, 3, 1, 2), y - b.view(1, -1, 1, 1), 5e-4, "fwd accumulate")
    close(ops.wino4_conv(dyd, Ud, dil=dil).permute(0, 3, 1, 2), x.grad, 2e-4, "dgrad")
    if ops._x3_case(cout):
        # the data gradient as the adjoint of the forward algorithm (gather-form output transform), its accumulate form, and the weight
        # gradient on the Z = A dy A^T it leaves behind
        kz = {}
        close(ops.wino4_dgrad_adj(dyd, wd, dil=dil, keep_z=kz).permute(0, 3, 1, 2), x.grad, 2e-4, "dgrad (adjoint form)")
        accd = torch.full((n, h, w, cin), 0.5, device=dev)
        ops.wino4_dgrad_adj(dyd, wd, out=accd, accumulate=True, dil=dil)
        close((accd - 0.5).permute(0, 3, 1, 2), x.grad, 5e-4, "dgrad (adjoint form, accumulate)")
        close(ops.wino4_wgrad(xd, dyd, dil=dil, z=kz["Z"]).permute(3, 2, 0, 1), wt.grad, 5e-4, "wgrad from the adjoint data gradient's Z")
        assert torch.equal(ops.conv_dgrad(dyd, wd, dil=dil), ops.wino4_dgrad_adj(dyd, wd, dil=dil)) or not ops._wino4_case(h, w, 3, dil, cout, cin, cout)
    close(ops.wino4_wgrad(xd, dyd, dil=dil).permute(3, 2, 0, 1), wt.grad, 5e-4, "wgrad")
    keep = {}
    ops.wino4_conv(xd, U, b.to(dev), keep_v=keep, dil=dil)             # the forward's transformed input reused by the weight gradient
    close(ops.wino4_wgrad(xd, dyd, v=keep["V"], dil=dil).permute(3, 2, 0, 1), wt.grad, 5e-4, "wgrad from kept V")
    # channel-slice views (concat buffers): pixel stride > channels
    big = torch.zeros(n, h, w, cin + 32, device=dev)
    big[..., 16:16 + cin] = xd
    outb = torch.zeros(n, h, w, cout + 8, device=dev)
    ops.wino4_conv(big[..., 16:16 + cin], U, b.to(dev), out=outb[..., 4:4 + cout], dil=dil)
    close(outb[..., 4:4 + cout].permute(0, 3, 1, 2), y, 2e-4, "fwd strided")
    assert float(outb[..., :4].abs().max()) == 0.0 and float(outb[..., 4 + cout:].abs().max()) == 0.0


@pytest.mark.parametrize("k,dil", [(1, 1), (3, 2)])
def test_data_gradient_on_pre_transposed_weights_equals_the_default_path(k, dil):
    """Modes RUNET_CONV_DGRAD_T / RUNET_CONVT_DGRAD_T (weights transposed per tap by runet_transpose_taps, n-contiguous staging): same
    products in the same order as the k-contiguous staging of the forward weight -> bit-identical results."""
    import importlib
    ops = importlib.import_module("eusipco-2026-robust-unet_amd.ops")
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(k * 10 + dil)
    dy = torch.randn(2, 16, 16, 48, generator=g).to(dev)
    w = (torch.randn(k, k, 32, 48, generator=g) * 0.1).to(dev)
    wt = (torch.randn(2, 2, 32, 48, generator=g) * 0.1).to(dev)
    ref, reft = ops.conv_dgrad(dy, w, dil=dil), ops.convt_dgrad(dy, wt)
    ops.USE_TRANSPOSED_DGRAD = True
    try:
        got, gott = ops.conv_dgrad(dy, w, dil=dil), ops.convt_dgrad(dy, wt)
    finally:
        ops.USE_TRANSPOSED_DGRAD = False
    assert torch.equal(got, ref) and torch.equal(gott, reft)


def test_opt_in_direct_gemm_paths_in_a_child_process():
    """RUNET_GEMM_TN_DIRECT (register-direct TN GEMM for the F(4x4) and all 1x1 weight gradients; off by default because it slows the step
    down from the side stream, see csrc/gemm.hip) is read once per process: the F(4x4) and generic convolution tests run again in a
    child with it set."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, RUNET_GEMM_TN_DIRECT="1")
    here = os.path.abspath(__file__)
    r = subprocess.run([sys.executable, "-m", "pytest", here, "-q", "-x", "-k", "winograd_f4 or fwd_dgrad_wgrad", "-p", "no:cacheprovider"],
                       env=env, capture_output=True, text=True, timeout=600, cwd=os.path.dirname(os.path.dirname(here)))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.parametrize("rows,k,n,scale", [(256, 64, 64, 1.0), (200, 48, 36, 1.0), (1024, 512, 256, 1e-9), (130, 1024, 132, 1.0), (4096, 16, 128, 1e4)])
def test_split_operand_gemms_against_float64(rows, k, n, scale):
    """csrc/gemm_split.hip: the fp32 position GEMMs on the BF16 matrix cores (x = h + m + l, six bf16 products) are fp32-accurate - error
    against a float64 product of the same operands within 4e-6 of the result's scale (the f32-MFMA kernels: the same), also for tiny and
    large operands (the split keeps fp32's exponent range), ragged rows / columns, and the TN form with split-K slabs."""
    L = importlib.import_module("eusipco-2026-robust-unet_amd._lib")
    lib, check = L.lib, L.check
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    B = 3
    g = torch.Generator().manual_seed(rows + k + n)
    a = (torch.randn(B, rows, k, generator=g) * scale).to(dev)
    b = torch.randn(B, k, n, generator=g).to(dev)
    bp = torch.empty(lib.runet_gemm_x3_pack_elems(B, k, n), device=dev, dtype=torch.bfloat16)
    check(lib.runet_gemm_x3_pack(b.data_ptr(), k * n, bp.data_ptr(), B, k, n, st))
    c = torch.full((B, rows, n + 4), 7.0, device=dev)          # row stride > n: the columns beyond n must stay untouched
    check(lib.runet_gemm_x3_batched(a.data_ptr(), k, rows * k, bp.data_ptr(), c.data_ptr(), n + 4, rows * (n + 4), B, rows, k, n, st))
    ref = torch.bmm(a.double().cpu(), b.double().cpu())
    err = float((c[..., :n].double().cpu() - ref).abs().max() / ref.abs().max())
    assert err <= 4e-6, err
    assert float(c[..., n:].min()) == 7.0 and float(c[..., n:].max()) == 7.0
    if rows % 16 == 0:
        z = torch.randn(B, rows, n, generator=g).to(dev)
        rps = max(16, (rows // 3 + 15) // 16 * 16)
        splits = -(-rows // rps)
        cu = torch.empty(splits, B, k, n, device=dev)
        check(lib.runet_gemm_x3_tn_batched(a.data_ptr(), k, rows * k, z.data_ptr(), n, rows * n, cu.data_ptr(), B, rows, k, n, rps, st))
        ref2 = torch.bmm(a.double().cpu().transpose(1, 2), z.double().cpu())
        err2 = float((cu.double().sum(0).cpu() - ref2).abs().max() / ref2.abs().max())
        assert err2 <= 4e-6, err2


@pytest.mark.parametrize("n,h,w,ci,co", [(2, 16, 16, 64, 32), (1, 6, 10, 128, 384), (3, 8, 8, 16, 4), (2, 32, 32, 256, 128), (1, 4, 4, 1024, 512),
                                         (2, 12, 20, 48, 36)])
def test_conv_x3_matches_torch(n, h, w, ci, co, monkeypatch):
    """csrc/conv_x3.hip: 1x1 convolution and k2-s2 transposed convolution, forward and data gradient, on the split-operand path against
    float64 torch CPU ops (2e-5 of the result's scale: fp32-accurate), with bias, accumulate, channel-slice sources and destinations."""
    ops = _ops()
    monkeypatch.setattr(ops, "CONV_X3_MIN_K", 16)
    monkeypatch.setattr(ops, "CONV_X3_WIDE_N", 4)
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(n * 1000 + h + ci + co)
    x = torch.randn(n, ci, h, w, generator=g, dtype=torch.float64, requires_grad=True)
    w1 = (torch.randn(co, ci, 1, 1, generator=g, dtype=torch.float64) / ci ** 0.5).requires_grad_(True)
    wt = (torch.randn(ci, co, 2, 2, generator=g, dtype=torch.float64) / ci ** 0.5).requires_grad_(True)
    b = torch.randn(co, generator=g, dtype=torch.float64)
    y1 = F.conv2d(x, w1, b)
    dy1 = torch.randn(y1.shape, generator=g, dtype=torch.float64)
    gx1, = torch.autograd.grad(y1, x, dy1)
    yt = F.conv_transpose2d(x, wt, b, stride=2)
    dyt = torch.randn(yt.shape, generator=g, dtype=torch.float64)
    gxt, = torch.autograd.grad(yt, x, dyt)

    def nhwc(t):
        return t.detach().float().permute(0, 2, 3, 1).contiguous().to(dev)

    def close(got, ref, msg, tol=2e-5):
        ref = ref.detach()
        err = float((got.permute(0, 3, 1, 2).double().cpu() - ref).abs().max() / ref.abs().max())
        assert err <= tol, f"{msg}: {err:.3e}"

    xd, w1d, wtd, bd = nhwc(x), w1.detach().float().permute(2, 3, 1, 0).contiguous().to(dev), wt.detach().float().permute(2, 3, 0, 1).contiguous().to(dev), b.float().to(dev)
    seen = []
    real = ops._conv_x3
    monkeypatch.setattr(ops, "_conv_x3", lambda mode, *a, **kw: (seen.append(mode), real(mode, *a, **kw))[1])
    close(ops.conv_fwd(xd, w1d, bd), y1, "1x1 forward")
    # channel-slice source and destination (concat buffers), accumulate
    big = torch.zeros(n, h, w, ci + 32, device=dev)
    big[..., 16:16 + ci] = xd
    outb = torch.full((n, h, w, co + 8), 2.0, device=dev)
    ops.conv_fwd(big[..., 16:16 + ci], w1d, None, out=outb[..., 4:4 + co], accumulate=True)
    close(outb[..., 4:4 + co], y1 - b.view(1, -1, 1, 1) + 2.0, "1x1 forward, slices, accumulate")
    assert float(outb[..., :4].min()) == 2.0 and float(outb[..., 4 + co:].max()) == 2.0
    close(ops.conv_dgrad(nhwc(dy1), w1d), gx1, "1x1 data gradient")
    acc = torch.full((n, h, w, ci), -1.0, device=dev)
    ops.conv_dgrad(nhwc(dy1), w1d, out=acc, accumulate=True)
    close(acc, gx1 - 1.0, "1x1 data gradient, accumulate")
    cat = torch.zeros(n, 2 * h, 2 * w, 2 * co, device=dev)
    ops.convt_fwd(xd, wtd, bd, out=cat[..., co:])
    close(cat[..., co:], yt, "transposed forward into the right half of a concat buffer")
    assert float(cat[..., :co].abs().max()) == 0.0
    close(ops.convt_dgrad(nhwc(dyt), wtd), gxt, "transposed data gradient")
    dcat = torch.zeros(n, 2 * h, 2 * w, 2 * co, device=dev)
    dcat[..., co:] = nhwc(dyt)
    close(ops.convt_dgrad(dcat[..., co:], wtd), gxt, "transposed data gradient from a slice")
    # every call whose contraction is a multiple of 16 channels took the split-operand kernel (data gradients contract over `co`)
    want = [ops.CONV_FWD, ops.CONVT_FWD] + ([ops.CONV_DGRAD, ops.CONVT_DGRAD] if co % 16 == 0 else [])
    assert sorted(set(seen)) == sorted(want), seen


@pytest.mark.parametrize("n,h,w,ci,co,k", [(2, 16, 16, 128, 128, 1), (3, 10, 12, 256, 64, 1), (1, 6, 10, 128, 384, 1), (2, 32, 32, 64, 64, 3),
                                           (2, 20, 28, 32, 96, 3), (1, 8, 16, 16, 64, 3), (16, 64, 64, 128, 64, 3),
                                           (2, 16, 16, 128, 128, 3), (1, 8, 12, 128, 512, 3), (3, 12, 20, 256, 128, 3)])      # the last three: F(4x4) output transform
def test_epilogue_statistics_equal_the_separate_pass(n, h, w, ci, co, k, monkeypatch):
    """BatchNorm statistics taken in the producing convolution's epilogue (runet_conv_x3_stats, runet_wino_conv_x3_stats,
    runet_wino4_output_stats + runet_bn_stats_finalize) against the separate pass over the stored tensor (runet_bn_stats): scale / shift / saved mean / inverse
    standard deviation and the running statistics agree to 2e-6 relative (both are Chan combinations of exact per-block moments; the
    partition of the pixels differs), also with ragged row / tile counts and a bias."""
    ops = _ops()
    blocks = importlib.import_module("eusipco-2026-robust-unet_amd.blocks")
    monkeypatch.setattr(ops, "CONV_X3_MIN_K", 16)
    monkeypatch.setattr(ops, "CONV_X3_WIDE_N", 4)
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(n + h + ci + co)
    x = (torch.randn(n, h, w, ci, generator=g) + 0.3).to(dev)
    wt = (torch.randn(k, k, ci, co, generator=g) / (k * k * ci) ** 0.5).to(dev)
    b = torch.randn(co, generator=g).to(dev)

    def bn_state():
        return blocks.BNState(torch.full((co,), 1.5, device=dev), torch.full((co,), -0.25, device=dev), torch.zeros(co, device=dev),
                              torch.ones(co, device=dev), torch.zeros((), device=dev, dtype=torch.int64))
    st = {}
    y = ops.conv_fwd(x, wt, b, stats=st)
    assert "part" in st, "this shape should have taken a kernel with epilogue statistics"
    bn_a, bn_b = bn_state(), bn_state()
    got = blocks.bn_coeff(y, bn_a, True, blocks.Small(dev), fused=st)[:4]
    ref = blocks.bn_coeff(y, bn_b, True, blocks.Small(dev))[:4]
    for a, r, name in zip(got, ref, ("scale", "shift", "mean", "invstd")):
        err = float((a - r).abs().max() / r.abs().max())
        assert err <= 2e-6, (name, err)
    assert float((bn_a.running_mean - bn_b.running_mean).abs().max()) <= 2e-6 * float(bn_b.running_mean.abs().max() + 1e-3)
    assert float((bn_a.running_var - bn_b.running_var).abs().max() / bn_b.running_var.abs().max()) <= 2e-6
    assert int(bn_a.nbt) == 1 and int(bn_b.nbt) == 1
    # the convolution's own result is unchanged by asking for statistics
    assert torch.equal(y, ops.conv_fwd(x, wt, b))


@pytest.mark.parametrize("n,h,w,ci,co", [(2, 16, 16, 128, 64), (1, 16, 32, 256, 128), (2, 32, 48, 64, 64), (3, 8, 16, 192, 96)])
def test_transposed_conv_weight_gradient_on_the_split_operand_gemm(n, h, w, ci, co):
    """ConvTranspose2d(k2, s2) weight gradient (Main_Final.py:261-270) through runet_conv_wgrad's split-operand TN route (image width a multiple
    of 16, both channel counts >= 64) against float64 torch: 2e-5 of the gradient's scale, also from the right half of a concat buffer."""
    ops = _ops()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(n + h + w + ci)
    x = torch.randn(n, ci, h, w, generator=g, dtype=torch.float64)
    wt = (torch.randn(ci, co, 2, 2, generator=g, dtype=torch.float64) / ci ** 0.5).requires_grad_(True)
    y = F.conv_transpose2d(x, wt, None, stride=2)
    dy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    y.backward(dy)
    ref = wt.grad.permute(2, 3, 0, 1)                       # [2, 2, ci, co]
    xd = x.float().permute(0, 2, 3, 1).contiguous().to(dev)
    dcat = torch.zeros(n, 2 * h, 2 * w, 2 * co, device=dev)
    dcat[..., co:] = dy.float().permute(0, 2, 3, 1).to(dev)
    got = ops.convt_wgrad(xd, dcat[..., co:])
    err = float((got.double().cpu() - ref).abs().max() / ref.abs().max())
    assert err <= 2e-5, err


@pytest.mark.parametrize("n,h,w,co", [(2, 24, 40, 64), (1, 20, 32, 64), (3, 8, 100, 32)])
def test_stem_epilogue_statistics_equal_the_separate_pass(n, h, w, co):
    """the RGB stem's two outputs (conv1 and the 1x1 shortcut of the first ResidualBlock, one launch): BatchNorm statistics from the kernel's
    epilogue (runet_stem_conv_stats) against the pass over the stored tensors, ragged tile rows / columns included; outputs unchanged"""
    ops = _ops()
    blocks = importlib.import_module("eusipco-2026-robust-unet_amd.blocks")
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(n + h + w)
    x = torch.zeros(n, h, w, 4)
    x[..., :3] = torch.rand(n, h, w, 3, generator=g)
    x = x.to(dev)
    w3 = (torch.randn(3, 3, 3, co, generator=g) / 27 ** 0.5).to(dev)
    w1 = (torch.randn(1, 1, 3, co, generator=g) / 3 ** 0.5).to(dev)

    def bn_state():
        return blocks.BNState(torch.full((co,), 1.5, device=dev), torch.full((co,), -0.25, device=dev), torch.zeros(co, device=dev),
                              torch.ones(co, device=dev), torch.zeros((), device=dev, dtype=torch.int64))
    s3, s1 = {}, {}
    y3, y1 = ops.stem_conv(x, w3, w1, stats3=s3, stats1=s1)
    z3, z1 = ops.stem_conv(x, w3, w1)
    assert torch.equal(y3, z3) and torch.equal(y1, z1)
    assert "part" in s3 and "part" in s1
    for y, st in ((y3, s3), (y1, s1)):
        bn_a, bn_b = bn_state(), bn_state()
        got = blocks.bn_coeff(y, bn_a, True, blocks.Small(dev), fused=st)[:4]
        ref = blocks.bn_coeff(y, bn_b, True, blocks.Small(dev))[:4]
        for a, r, name in zip(got, ref, ("scale", "shift", "mean", "invstd")):
            err = float((a - r).abs().max() / r.abs().max())
            assert err <= 2e-6, (name, err)
        assert float((bn_a.running_var - bn_b.running_var).abs().max() / bn_b.running_var.abs().max()) <= 2e-6
