"""Labelme -> binary-mask dataloader, drop-in for the reference's `CoastalDataset` /
`prepare_dataset` (/root/reference/Main_Final.py:28-78, 671-711).

Same constructor signature, same `(image, mask)` item contract (image float32 [3,S,S]
ImageNet-normalised, mask float32 [1,S,S] in {0,1}), same fallbacks (unreadable image ->
grey 512x512, unreadable/invalid JSON -> all-zero mask), same sorted 80/20 split.  The
reference builds its transform from torchvision; torchvision is not a dependency here, so
the three transforms it uses (Resize on a PIL image, ToTensor, Normalize) are restated on
PIL/NumPy below with the same semantics (PIL bilinear resize, uint8 HWC -> float CHW /255,
(x-mean)/std).

Also holds the synthetic tile generator used by tests and bench.py:
images ~ N(0,1) from the portable generator, masks from 1-3 random convex polygons pushed
through the same rasteriser as the Labelme path.
"""
from __future__ import annotations

import json
import math
import os

import numpy as np
import torch
from PIL import Image, ImageDraw
from torch.utils.data import DataLoader, Dataset

from . import portable_rng as prng

WATER_LABELS = ("water", "sea", "海水", "水体")
IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


# --------------------------------------------------------------------------- transforms
class Resize:
    def __init__(self, size):
        self.size = (size, size) if isinstance(size, int) else tuple(size)  # (H, W)

    def __call__(self, img):
        return img.resize((self.size[1], self.size[0]), Image.BILINEAR)


class ToTensor:
    def __call__(self, img):
        a = np.asarray(img, dtype=np.uint8)
        if a.ndim == 2:
            a = a[:, :, None]
        return torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1))).float().div_(255.0)


class Normalize:
    def __init__(self, mean, std):
        self.mean = torch.tensor(mean, dtype=torch.float32).view(-1, 1, 1)
        self.std = torch.tensor(std, dtype=torch.float32).view(-1, 1, 1)

    def __call__(self, t):
        return (t - self.mean) / self.std


class Compose:
    def __init__(self, transforms):
        self.transforms = list(transforms)

    def __call__(self, x):
        for t in self.transforms:
            x = t(x)
        return x


# --------------------------------------------------------------------------- rasteriser
def rasterize_shapes(shapes, image_size):
    """Union of water polygons -> uint8 [H, W]; image_size is PIL's (W, H).
    Label match is case-insensitive; vertices truncated with int(); <3 vertices skipped."""
    canvas = Image.new("L", image_size, 0)
    draw = ImageDraw.Draw(canvas)
    for shape in shapes:
        if shape["label"].lower() in WATER_LABELS:
            pts = [(int(p[0]), int(p[1])) for p in shape["points"]]
            if len(pts) >= 3:
                draw.polygon(pts, fill=1)
    return np.array(canvas, dtype=np.uint8)


class CoastalDataset(Dataset):
    """(image float32 [3, S, S], mask float32 [1, S, S]) per item, as /root/reference/Main_Final.py:28-78.  return_path=True gives the
    variant of /root/reference/Extended_Baseline_Comparison.py:43-70, whose items carry the image path as a third element."""

    def __init__(self, image_paths, label_paths, transform=None, image_size=(512, 512), return_path=False):
        self.image_paths = image_paths
        self.label_paths = label_paths
        self.transform = transform
        self.image_size = image_size
        self.return_path = bool(return_path)

    def __len__(self):
        return len(self.image_paths)

    def __getitem__(self, idx):
        image = self.load_image(self.image_paths[idx])
        mask = self.create_mask_from_labelme(self.label_paths[idx], image.size)
        image = image.resize(self.image_size, Image.LANCZOS)
        mask = np.array(Image.fromarray(mask).resize(self.image_size, Image.NEAREST))
        image = self.transform(image) if self.transform else ToTensor()(image)
        mask = torch.from_numpy(mask).float().unsqueeze(0)
        if self.return_path:
            return image, mask, self.image_paths[idx]
        return image, mask

    def load_image(self, image_path):
        try:
            return Image.open(image_path).convert("RGB")
        except Exception:
            return Image.new("RGB", (512, 512), (128, 128, 128))

    def create_mask_from_labelme(self, label_path, image_size):
        try:
            with open(label_path, "r", encoding="utf-8") as f:
                label_data = json.load(f)
            return rasterize_shapes(label_data.get("shapes", []), image_size)
        except Exception:
            return np.zeros((image_size[1], image_size[0]), dtype=np.uint8)


def prepare_dataset(images_dir, labels_dir, batch_size=4, image_size=(512, 512), num_workers=0, pin_memory=False):
    """Sorted listing, image/JSON pairing by basename, first 80 % train / last 20 % val,
    train shuffled.  `image_size` / `num_workers` / `pin_memory` are extensions (reference: 512, 0, False);
    with workers + pinned memory + data.DevicePrefetcher the decode/rasterise/resize work leaves the step's critical path."""
    image_files, label_files = [], []
    for img_file in sorted(os.listdir(images_dir)):
        if img_file.lower().endswith((".png", ".jpg", ".jpeg")):
            label_path = os.path.join(labels_dir, os.path.splitext(img_file)[0] + ".json")
            if os.path.exists(label_path):
                image_files.append(os.path.join(images_dir, img_file))
                label_files.append(label_path)
    print(f"Found {len(image_files)} valid image-label pairs")
    if not image_files:
        return None
    split = int(0.8 * len(image_files))
    tf = Compose([Resize(image_size), ToTensor(), Normalize(IMAGENET_MEAN, IMAGENET_STD)])
    train = CoastalDataset(image_files[:split], label_files[:split], transform=tf, image_size=image_size)
    val = CoastalDataset(image_files[split:], label_files[split:], transform=tf, image_size=image_size)
    kw = dict(num_workers=num_workers, pin_memory=pin_memory, persistent_workers=num_workers > 0)
    return (DataLoader(train, batch_size=batch_size, shuffle=True, **kw),
            DataLoader(val, batch_size=batch_size, shuffle=False, **kw))


# --------------------------------------------------------------------------- synthetic tiles
def synthetic_shapes(size, seed):
    """1-3 convex polygons (Labelme `shapes` list, float vertices) covering roughly half the tile."""
    u = prng.uniform(64, prng.name_seed("polygons", seed))
    n_poly = 1 + int(u[0] * 3) % 3
    shapes, k = [], 1
    for _ in range(n_poly):
        cx, cy = (0.2 + 0.6 * u[k]) * size, (0.2 + 0.6 * u[k + 1]) * size
        rx, ry = (0.25 + 0.35 * u[k + 2]) * size, (0.25 + 0.35 * u[k + 3]) * size
        nv = 5 + int(u[k + 4] * 4)
        k += 5
        ang = np.sort(u[k:k + nv]) * 2.0 * math.pi
        k += nv
        pts = [[float(cx + rx * math.cos(a)), float(cy + ry * math.sin(a))] for a in ang]
        shapes.append({"label": "water", "points": pts, "shape_type": "polygon"})
    return shapes


def synthetic_batch(n, size, seed=1234):
    """(images float32 [n,3,size,size] ~ N(0,1), masks float32 [n,1,size,size] in {0,1})."""
    imgs = torch.from_numpy(prng.normal_f32((n, 3, size, size), prng.name_seed("images", seed)))
    masks = np.stack([rasterize_shapes(synthetic_shapes(size, seed * 1000 + i), (size, size)) for i in range(n)])
    return imgs, torch.from_numpy(masks.astype(np.float32)).unsqueeze(1)


# --------------------------------------------------------------------------- input pipeline off the critical path
class DevicePrefetcher:
    """Iterates a DataLoader one batch ahead: the next (images, masks) pair is copied host->device on a side HIP stream
    (pinned memory when the loader provides it) while the current step computes.  The reference runs its dataset inline
    with `num_workers=0` and a blocking `.to(device)` per batch (/root/reference/Main_Final.py:570-571, 708-709); at
    hundreds of images/s that becomes the bottleneck.  Drop-in: `for images, masks in
    DevicePrefetcher(loader, device)`.  On a CPU device it degrades to a plain pass-through."""

    def __init__(self, loader, device):
        self.loader = loader
        self.device = torch.device(device)
        self.cuda = self.device.type == "cuda"
        self.stream = torch.cuda.Stream(device=self.device) if self.cuda else None

    def __len__(self):
        return len(self.loader)

    def _stage(self, batch):
        if not self.cuda:
            return tuple(t.to(self.device) if torch.is_tensor(t) else t for t in batch)
        with torch.cuda.stream(self.stream):
            return tuple(t.to(self.device, non_blocking=True) if torch.is_tensor(t) else t for t in batch)

    def __iter__(self):
        it = iter(self.loader)
        try:
            nxt = self._stage(next(it))
        except StopIteration:
            return
        while True:
            if self.cuda:
                torch.cuda.current_stream(self.device).wait_stream(self.stream)
                for t in nxt:
                    if torch.is_tensor(t):
                        t.record_stream(torch.cuda.current_stream(self.device))
            cur = nxt
            try:
                nxt = self._stage(next(it))
            except StopIteration:
                yield cur
                return
            yield cur
