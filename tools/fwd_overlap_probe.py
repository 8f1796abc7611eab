"""How much forward time could two half-batches on two streams save?  Upper-bound probe: the no-grad forward of the whole batch on one
stream against the two halves running concurrently on two streams (each half with its own batch statistics - numerically not the same
step, but the same kernels at half size, free to overlap HBM-bound and matrix-bound phases).
    RUNET_FWD_BRANCHES=0 python tools/fwd_overlap_probe.py [--batch 16] [--size 256]"""
import argparse
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--dtype", default="f32")
    args = ap.parse_args()
    pkg = importlib.import_module("eusipco-2026-robust-unet_amd")
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    m = pkg.RobustUNet(3, 1, 64).to(dev).train().set_precision(args.dtype)
    x, _ = pkg.synthetic_batch(args.batch, args.size, seed=1)
    x = x.to(dev)
    h = args.batch // 2
    xa, xb = x[:h].contiguous(), x[h:].contiguous()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

    def full():
        with torch.no_grad():
            m(x)

    def halves():
        with torch.no_grad():
            cur = torch.cuda.current_stream()
            s1.wait_stream(cur); s2.wait_stream(cur)
            with torch.cuda.stream(s1):
                m(xa)
            with torch.cuda.stream(s2):
                m(xb)
            cur.wait_stream(s1); cur.wait_stream(s2)

    def halves_serial():
        with torch.no_grad():
            m(xa); m(xb)

    for name, fn in (("whole batch, one stream", full), ("two halves, one stream", halves_serial), ("two halves, two streams", halves)):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        print(f"{name:28s} {e0.elapsed_time(e1) / 20:.3f} ms per forward")


if __name__ == "__main__":
    main()
