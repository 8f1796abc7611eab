"""Is runet_conv_igemm's tile choice the fastest one?  Times every tile variant (runet_igemm_force_variant) on the 1x1 convolutions of the
16 x 256^2 train step - forward and data gradient, with the step's row strides (concat halves) and `+=` epilogues - and prints the
automatic choice next to the best.    python tools/igemm_variants.py [--batch 16] [--size 256]
"""
import argparse
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

NAMES = ["128x32", "256x64", "128x64", "128x128", "64x64"]
# (spatial divisor, cin, cout, ldx, ldy, accumulate, mode: 0 fwd / 1 dgrad (cin = channels READ, cout = channels WRITTEN))
SHAPES = [
    (1, 128, 64, 128, 64, 0, 0), (1, 64, 32, 64, 32, 0, 0), (1, 64, 32, 128, 32, 0, 0),
    (2, 256, 128, 256, 128, 0, 0), (2, 128, 64, 128, 64, 0, 0), (2, 128, 64, 256, 64, 0, 0), (2, 64, 128, 64, 128, 0, 0),
    (4, 512, 256, 512, 256, 0, 0), (4, 128, 256, 128, 256, 0, 0), (4, 256, 128, 256, 128, 0, 0), (4, 256, 128, 512, 128, 0, 0),
    (8, 1024, 512, 1024, 512, 0, 0), (8, 512, 256, 512, 256, 0, 0), (8, 256, 512, 256, 512, 0, 0), (8, 512, 256, 1024, 256, 0, 0),
    (16, 512, 256, 512, 1024, 0, 0),
    (1, 64, 128, 64, 128, 1, 1), (1, 32, 64, 32, 64, 1, 1), (1, 32, 64, 32, 128, 1, 1),
    (2, 128, 256, 128, 256, 1, 1), (2, 64, 128, 64, 128, 1, 1), (2, 64, 128, 64, 256, 1, 1), (2, 128, 64, 128, 64, 1, 1),
    (4, 256, 512, 256, 512, 1, 1), (4, 128, 256, 128, 256, 1, 1), (4, 128, 256, 128, 512, 1, 1), (4, 256, 128, 256, 128, 1, 1),
    (8, 512, 1024, 512, 1024, 1, 1), (8, 256, 512, 256, 512, 1, 1), (8, 256, 512, 256, 1024, 1, 1), (8, 512, 256, 512, 256, 1, 1),
    (16, 256, 512, 1024, 512, 0, 1),
    # k2-s2 transposed convolution: forward (mode 2: x [n, h, h, cin] -> the left half of the decoder's concat buffer [n, 2h, 2h, 2 cout]) and
    # data gradient (mode 3: dy = that half -> dx [n, h, h, cin], here cin = channels READ, cout = channels WRITTEN)
    (16, 1024, 512, 1024, 1024, 0, 2), (8, 512, 256, 512, 512, 0, 2), (4, 256, 128, 256, 256, 0, 2), (2, 128, 64, 128, 128, 0, 2),
    (16, 512, 1024, 1024, 1024, 0, 3), (8, 256, 512, 512, 512, 0, 3), (4, 128, 256, 256, 256, 0, 3), (2, 64, 128, 128, 128, 0, 3),
]


def lowp(args):
    """The same sweep for runet_conv_igemm_bf16 / _fp16, through the ops layer (packed weights come from its cache)."""
    ops = importlib.import_module("eusipco-2026-robust-unet_amd.ops")
    lib, check = ops.lib, ops.check
    dev = torch.device("cuda:0")
    names = ["128x32", "128x64", "128x128"]
    gain = 0.0
    print(f"{'shape':52s} " + " ".join(f"{n:>8s}" for n in names) + "   auto -> best (us)")
    with ops.precision(args.dtype):
        for div, cin, cout, ldx, ldy, acc, mode in SHAPES:
            h, n = args.size // div, args.batch
            big = 2 * h
            src = torch.randn((n, big if mode == 3 else h, big if mode == 3 else h, ldx), device=dev)[..., :cin]
            dst = torch.zeros((n, big if mode == 2 else h, big if mode == 2 else h, ldy), device=dev)[..., :cout]
            k = 2 if mode >= 2 else 1
            w = torch.randn((k, k, cin, cout) if mode in (0, 2) else (k, k, cout, cin), device=dev) * 0.05
            run = {0: lambda: ops.conv_fwd(src, w, out=dst, accumulate=bool(acc)), 1: lambda: ops.conv_dgrad(src, w, out=dst, accumulate=bool(acc)),
                   2: lambda: ops.convt_fwd(src, w, out=dst), 3: lambda: ops.convt_dgrad(src, w, out=dst)}[mode]

            def timeit(iters=10):
                run()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(iters):
                    run()
                e1.record()
                torch.cuda.synchronize()
                return 1e3 * e0.elapsed_time(e1) / iters
            ts = []
            for v in range(3):
                check(lib.runet_igemm_lowp_force_variant(v))
                ts.append(timeit())
            check(lib.runet_igemm_lowp_force_variant(-1))
            auto = timeit()
            best = min(range(3), key=lambda v: ts[v])
            gain += max(0.0, auto - ts[best])
            desc = f"{('fwd', 'dgrad', 'convT fwd', 'convT dgrad')[mode]:11s} {n}x{h}x{h} {cin:4d}->{cout:4d} ld{ldx}/{ldy}{' +=' if acc else ''}"
            print(f"{desc:52s} " + " ".join(f"{t:8.1f}" for t in ts) + f"   {auto:7.1f} -> {ts[best]:7.1f} {names[best]}", flush=True)
    print(f"sum of (auto - best) over these launches: {gain:.0f} us per step")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--dtype", default="f32", help="bf16 / fp16: the reduced-precision kernel's three variants (128 pixels x 32 / 64 / 128 channels)")
    args = ap.parse_args()
    if args.dtype != "f32":
        return lowp(args)
    ops = importlib.import_module("eusipco-2026-robust-unet_amd.ops")
    lib, check = ops.lib, ops.check
    dev = torch.device("cuda:0")
    gain = 0.0
    print(f"{'shape':52s} " + " ".join(f"{n:>8s}" for n in NAMES) + "   auto -> best (us)")
    for div, cin, cout, ldx, ldy, acc, mode in SHAPES:
        h = args.size // div
        n = args.batch
        k = 2 if mode >= 2 else 1
        x = torch.randn((n, 2 * h if mode == 3 else h, 2 * h if mode == 3 else h, ldx), device=dev)
        y = torch.zeros((n, 2 * h if mode == 2 else h, 2 * h if mode == 2 else h, ldy), device=dev)
        # fwd: w [k, k, cin, cout]; the data gradients read dy with `cin` channels and the weight [k, k, cout_written, cin_read]
        w = torch.randn((k, k, cin, cout) if mode in (0, 2) else (k, k, cout, cin), device=dev) * 0.05
        st = torch.cuda.current_stream().cuda_stream

        def run():
            check(lib.runet_conv_igemm(x.data_ptr(), ldx, w.data_ptr(), None, y.data_ptr(), ldy, n, h, h, cin, cin, cout, k, k, 1, mode, acc, st))

        def timeit(iters=10):
            run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                run()
            e1.record()
            torch.cuda.synchronize()
            return 1e3 * e0.elapsed_time(e1) / iters
        ts = []
        for v in range(5):
            check(lib.runet_igemm_force_variant(v))
            ts.append(timeit())
        check(lib.runet_igemm_force_variant(-1))
        auto = timeit()
        best = min(range(5), key=lambda v: ts[v])
        gain += max(0.0, auto - ts[best])
        desc = f"{('fwd', 'dgrad', 'convT fwd', 'convT dgrad')[mode]:11s} {n}x{h}x{h} {cin:4d}->{cout:4d} ld{ldx}/{ldy}{' +=' if acc else ''}"
        print(f"{desc:52s} " + " ".join(f"{t:8.1f}" for t in ts) + f"   {auto:7.1f} -> {ts[best]:7.1f} {NAMES[best]}", flush=True)
    print(f"sum of (auto - best) over these launches: {gain:.0f} us per step")


if __name__ == "__main__":
    main()
