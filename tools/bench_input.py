"""Does the input pipeline keep up with the train step?  (the reference decodes, rasterises and resizes inline with
num_workers=0 and a blocking .to(device) per batch.)  Writes a synthetic Labelme dataset (PNG tiles + polygon JSON) to a temporary directory,
then measures (a) CoastalDataset + DataLoader images/s for several worker counts (host only), (b) the same through DevicePrefetcher onto the
GPU, (c) a training epoch with the prefetcher against the step's own rate on resident data.
usage: python tools/bench_input.py [n_images=256] [size=256]"""
import importlib
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from PIL import Image

pkg = importlib.import_module("eusipco-2026-robust-unet_amd")
data = importlib.import_module("eusipco-2026-robust-unet_amd.data")


def make_dataset(root, n, size):
    os.makedirs(os.path.join(root, "images")); os.makedirs(os.path.join(root, "labels"))
    rng = np.random.default_rng(0)
    for i in range(n):
        Image.fromarray(rng.integers(0, 255, (size, size, 3), dtype=np.uint8)).save(os.path.join(root, "images", f"t{i:04d}.png"))
        with open(os.path.join(root, "labels", f"t{i:04d}.json"), "w") as f:
            json.dump({"shapes": data.synthetic_shapes(size, i), "imageHeight": size, "imageWidth": size}, f)


def rate(loader, n_img, to_dev=None):
    t0 = time.perf_counter()
    for batch in (data.DevicePrefetcher(loader, to_dev) if to_dev is not None else loader):
        pass
    if to_dev is not None:
        torch.cuda.synchronize()
    return n_img / (time.perf_counter() - t0)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    size = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    with tempfile.TemporaryDirectory() as root:
        make_dataset(root, n, size)
        for workers in (0, 4, 8, 16):
            loaders = data.prepare_dataset(os.path.join(root, "images"), os.path.join(root, "labels"), batch_size=16, image_size=(size, size),
                                           num_workers=workers, pin_memory=True)
            tr = loaders[0]
            n_tr = len(tr.dataset)
            rate(tr, n_tr)                                   # first epoch: worker start-up
            host = rate(tr, n_tr)
            line = f"workers {workers:2d}: host pipeline {host:8.1f} img/s"
            if torch.cuda.is_available():
                dev = torch.device("cuda:0")
                line += f", through DevicePrefetcher {rate(tr, n_tr, dev):8.1f} img/s"
            print(line, flush=True)
        if torch.cuda.is_available():
            dev = torch.device("cuda:0")
            model = pkg.RobustUNet(3, 1, 64).to(dev).train()
            step = pkg.TrainStep(model)
            tr = data.prepare_dataset(os.path.join(root, "images"), os.path.join(root, "labels"), batch_size=16, image_size=(size, size),
                                      num_workers=16, pin_memory=True)[0]
            x, y = pkg.synthetic_batch(16, size)
            x, y = x.to(dev), y.to(dev)
            for _ in range(5):
                step(x, y)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                step(x, y)
            torch.cuda.synchronize()
            resident = 160 / (time.perf_counter() - t0)
            for ep in range(2):
                t0 = time.perf_counter()
                k = 0
                for images, masks in data.DevicePrefetcher(tr, dev):
                    step(images, masks)
                    k += images.shape[0]
                torch.cuda.synchronize()
                fed = k / (time.perf_counter() - t0)
            print(f"train step on resident data {resident:.1f} img/s; epoch fed by 16 workers + pinned memory + DevicePrefetcher {fed:.1f} img/s")


if __name__ == "__main__":
    main()
