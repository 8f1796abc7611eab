"""cProfile of the host side of a train step at a small (host-bound) size: where do the ~15 us per launch go?
    python tools/host_profile.py [--batch 2] [--size 256] [--steps 30]"""
import argparse
import cProfile
import importlib
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--steps", type=int, default=30)
    args = ap.parse_args()
    pkg = importlib.import_module("eusipco-2026-robust-unet_amd")
    trainer = importlib.import_module("eusipco-2026-robust-unet_amd.trainer")
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    m = pkg.RobustUNet(3, 1, 64).to(dev).train()
    step = trainer.TrainStep(m)
    x, y = pkg.synthetic_batch(args.batch, args.size, seed=1)
    x, y = x.to(dev), y.to(dev)
    for _ in range(5):
        step(x, y)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(args.steps):
        step(x, y)
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("tottime")
    print(f"per step: total host time {sum(v[2] for v in st.stats.values()) / args.steps * 1e3:.2f} ms (under the profiler)")
    st.print_stats(35)


if __name__ == "__main__":
    main()
