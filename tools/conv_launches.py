"""Per-launch table of every matrix-core launch of one train step: entry point, shape, time, TFLOP/s and GB/s (minimum bytes).

Wraps the C-ABI entry points on the ctypes handle (the product code is untouched), runs the step single-stream so that a launch's
time is its own, and prints the launches in issue order plus a summary per (entry point, mode).  Answers "which layer is the slow one".
    python tools/conv_launches.py [--batch 16] [--size 256] [--dtype f32]
"""
import argparse
import importlib
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

MODES = {0: "fwd", 1: "dgrad", 2: "convT fwd", 3: "convT dgrad", 4: "dgrad(T)", 5: "convT dgrad(T)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--two-streams", action="store_true", help="keep the weight gradients on the side stream (times then include sharing)")
    args = ap.parse_args()
    pkg = importlib.import_module("eusipco-2026-robust-unet_amd")
    ops = importlib.import_module("eusipco-2026-robust-unet_amd.ops")
    trainer = importlib.import_module("eusipco-2026-robust-unet_amd.trainer")
    lib = ops.lib
    ops.USE_WGRAD_STREAM = bool(args.two_streams)
    rows = []
    recording = [False]

    def wrap(name, describe):
        fn = getattr(lib, name)

        def inner(*a):
            if not recording[0]:
                return fn(*a)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s = torch.cuda.current_stream()
            e0.record(s)
            rc = fn(*a)
            e1.record(s)
            rows.append((name, describe(a), e0, e1))
            return rc
        setattr(lib, name, inner)

    def d_igemm(a):      # x ldx w bias y ldy n h w cin cin_w cout kh kw dil mode acc stream
        n, h, w, cin, cin_w, cout, kh, kw, dil, mode, acc = a[6:17]
        taps = 4 if kh == 2 else kh * kw
        fl = 2.0 * n * h * w * taps * min(cin, cin_w) * cout
        opix = 4 * n * h * w if mode == 2 else n * h * w
        ipix = 4 * n * h * w if mode in (3, 5) else n * h * w
        by = 4.0 * (ipix * cin + opix * cout * (2 if acc else 1) + taps * cin * cout)
        return f"{MODES.get(mode, mode):>14s} {n}x{h}x{w} {cin:4d}->{cout:4d} k{kh} d{dil} ld{a[1]}/{a[5]}{' +=' if acc else ''}", fl, by

    def d_wgrad(a):      # x ldx dy ldy dw ws wsf n h w cin cin_w cout kh kw dil transposed stream
        n, h, w, cin, cin_w, cout, kh, kw, dil, tr = a[7:17]
        taps = 4 if tr else kh * kw
        fl = 2.0 * n * h * w * taps * min(cin, cin_w) * cout
        by = 4.0 * (n * h * w * (cin + (4 if tr else 1) * cout) + taps * cin * cout)
        return f"{'wgrad' + (' convT' if tr else ''):>14s} {n}x{h}x{w} {cin:4d}->{cout:4d} k{kh} d{dil}", fl, by

    def d_wino(a):       # x ldx U bias y ldy n h w k n acc stream
        n, h, w, k, nn, acc = a[6:12]
        return f"{'F(2x2)':>14s} {n}x{h}x{w} {k:4d}->{nn:4d}{' +=' if acc else ''}", 2.0 * n * h * w * 9 * k * nn, 4.0 * n * h * w * (k + nn * (2 if acc else 1)) + 64.0 * k * nn

    def d_wino_wgrad(a):  # x ldx dy ldy dw ws wsf n h w cin cout stream
        n, h, w, cin, cout = a[7:12]
        return f"{'F(2x2) wgrad':>14s} {n}x{h}x{w} {cin:4d}->{cout:4d}", 2.0 * n * h * w * 9 * cin * cout, 4.0 * n * h * w * (cin + cout)

    def d_wino4(a):      # x ldx U bias y ldy n h w k n dil acc ws wsf stream
        n, h, w, k, nn, dil, acc = a[6:13]
        return f"{'F(4x4) (3 k.)':>14s} {n}x{h}x{w} {k:4d}->{nn:4d} d{dil}{' +=' if acc else ''}", 2.0 * n * h * w * 9 * k * nn, 4.0 * n * h * w * (k + nn) * 3.25 + 144.0 * k * nn

    def d_wino4_wgrad(a):  # x ldx dy ldy dw ws wsf n h w cin cout dil stream
        n, h, w, cin, cout, dil = a[7:13]
        return f"{'F(4x4) wgrad':>14s} {n}x{h}x{w} {cin:4d}->{cout:4d} d{dil}", 2.0 * n * h * w * 9 * cin * cout, 4.0 * n * h * w * (cin + cout) * 2.25

    def d_stem(a):       # x ldx w3 w1 y3 ldy3 y1 ldy1 n h w cin_w cout stream
        n, h, w, cin_w, cout = a[8:13]
        both = a[3] is not None
        return (f"{'stem 3x3' + ('+1x1' if both else ''):>14s} {n}x{h}x{w} {cin_w:4d}->{cout:4d}", 2.0 * n * h * w * (10 if both else 9) * cin_w * cout,
                4.0 * n * h * w * (4 + cout * (2 if both else 1)))

    def d_stem_wgrad(a):  # x ldx dy ldy dw ws wsf n h w cin_w cout ksize stream
        n, h, w, cin_w, cout, ks = a[7:13]
        return f"{'stem wgrad':>14s} {n}x{h}x{w} {cin_w:4d}->{cout:4d} k{ks}", 2.0 * n * h * w * ks * ks * cin_w * cout, 4.0 * n * h * w * (4 + cout)

    def d_gemm_tn(a):    # a lda sa b ldb sb c batch rows k n rps stream
        batch, rows, k, n = a[7:11]
        return f"{'gemm TN (F4 wg)':>14s} {batch}x[{rows}x{k}]^T[{rows}x{n}]", 4.0 * 2.0 * batch * rows * k * n, 4.0 * batch * rows * (k + n)

    def d_gemm_nn(a):    # a lda sa b sb c ldc sc batch rows k n stream
        batch, rows, k, n = a[8:12]
        return f"{'gemm NN (F4)':>14s} {batch}x[{rows}x{k}][{k}x{n}]", 4.0 * 2.0 * batch * rows * k * n, 4.0 * batch * (rows * (k + n) + k * n)

    def d_conv_x3(a):    # x ldx wpacked bias y ldy n h w cin cout mode acc stream
        n, h, w, cin, cout, mode, acc = a[6:13]
        taps = 4 if mode in (2, 3) else 1
        fl = 2.0 * n * h * w * taps * cin * cout
        opix = 4 * n * h * w if mode == 2 else n * h * w
        ipix = 4 * n * h * w if mode == 3 else n * h * w
        by = 4.0 * (ipix * cin + opix * cout * (2 if acc else 1)) + 6.0 * taps * cin * cout
        return f"{MODES.get(mode, mode):>14s} {n}x{h}x{w} {cin:4d}->{cout:4d} x3 ld{a[1]}/{a[5]}{' +=' if acc else ''}", fl, by

    def d_gemm_x3_tn(a):  # a lda sa b ldb sb c batch rows k n rps stream
        batch, rows, k, n = a[7:11]
        return f"{'gemm TN x3':>14s} {batch}x[{rows}x{k}]^T[{rows}x{n}]", 4.0 * 2.0 * batch * rows * k * n, 4.0 * batch * rows * (k + n)

    def d_gemm_x3_nn(a):  # a lda sa packed c ldc sc batch rows k n stream
        batch, rows, k, n = a[7:11]
        return f"{'gemm NN x3':>14s} {batch}x[{rows}x{k}][{k}x{n}]", 4.0 * 2.0 * batch * rows * k * n, 4.0 * batch * (rows * (k + n)) + 6.0 * batch * k * n

    wrap("runet_conv_x3", d_conv_x3)
    wrap("runet_wino4_conv_x3", d_wino4)
    wrap("runet_gemm_x3_tn_batched", d_gemm_x3_tn)
    wrap("runet_gemm_x3_batched", d_gemm_x3_nn)
    wrap("runet_gemm_tn_batched", d_gemm_tn)
    wrap("runet_gemm_batched", d_gemm_nn)
    wrap("runet_stem_conv", d_stem)
    wrap("runet_stem_wgrad", d_stem_wgrad)
    wrap("runet_conv_igemm", d_igemm)
    wrap("runet_conv_wgrad", d_wgrad)
    wrap("runet_wino_conv", d_wino)
    wrap("runet_wino_conv_x3", d_wino)
    wrap("runet_wino_wgrad", d_wino_wgrad)
    wrap("runet_wino4_conv", d_wino4)
    wrap("runet_wino4_wgrad", d_wino4_wgrad)
    for t in ("bf16", "fp16"):
        wrap(f"runet_conv_igemm_{t}", lambda a: d_igemm(a[:10] + (a[9],) + a[10:]))
        wrap(f"runet_conv_wgrad_{t}", lambda a: d_wgrad(a[:11] + (a[10],) + a[11:]))

    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    m = pkg.RobustUNet(3, 1, 64).to(dev).train().set_precision(args.dtype)
    step = trainer.TrainStep(m, loss_scale=1024.0 if args.dtype == "fp16" else None)
    x, y = pkg.synthetic_batch(args.batch, args.size, seed=1)
    x, y = x.to(dev), y.to(dev)
    for _ in range(3):
        step(x, y)
    torch.cuda.synchronize()
    reps = 3
    recording[0] = True
    for _ in range(reps):
        step(x, y)
    torch.cuda.synchronize()
    recording[0] = False
    per = len(rows) // reps
    summ = defaultdict(lambda: [0, 0.0, 0.0, 0.0])
    print(f"{'#':>3s} {'entry point':24s} {'launch':58s} {'us':>8s} {'TFLOP/s':>8s} {'GB/s':>7s}")
    total = 0.0
    for i in range(per):
        name, (desc, fl, by), _, _ = rows[i]
        us = min(rows[i + r * per][2].elapsed_time(rows[i + r * per][3]) for r in range(reps)) * 1e3
        total += us
        print(f"{i:3d} {name:24s} {desc:58s} {us:8.1f} {fl / us / 1e6:8.1f} {by / us / 1e3:7.0f}")
        key = (name, desc.split()[0] + (" " + desc.split()[1] if desc.split()[0] in ("convT", "F(2x2)", "F(4x4)") and not desc.split()[1][0].isdigit() else ""))
        s = summ[key]
        s[0] += 1; s[1] += us; s[2] += fl; s[3] += by
    print(f"\n{per} matrix-core launches per step, {total / 1e3:.2f} ms single-stream")
    for (name, kind), (cnt, us, fl, by) in sorted(summ.items(), key=lambda kv: -kv[1][1]):
        print(f"  {name:24s} {kind:16s} {cnt:3d} launches {us / 1e3:7.3f} ms  {fl / us / 1e6:7.1f} TFLOP/s  {by / us / 1e3:6.0f} GB/s")


if __name__ == "__main__":
    main()
