"""Seeded synthetic inputs for the BASELINE.json configurations.

Everything is generated on the CPU with an explicit ``torch.Generator`` so that this container, the GPU box and the
golden-vector generator see bit-identical inputs.
"""
from __future__ import annotations

import math
from typing import List, Tuple

import torch


def blob_image(seed: int, size: int = 1024, n_blobs: int = 8) -> Tuple[torch.Tensor, Tuple[float, float]]:
    """One [3, size, size] float image in 0..255 (sum of random 2-D Gaussians + N(0, 8) noise, replicated to RGB with
    small per-channel gains) and the (x, y) centre of the brightest blob, used as the positive click."""
    g = torch.Generator().manual_seed(1000 + seed)
    ys = torch.arange(size, dtype=torch.float32)[:, None]
    xs = torch.arange(size, dtype=torch.float32)[None, :]
    img = torch.zeros(size, size)
    best, best_xy = -1.0, (size / 2.0, size / 2.0)
    for _ in range(n_blobs):
        cx, cy = (torch.rand(2, generator=g) * 0.8 + 0.1).tolist()
        sig = float(torch.rand(1, generator=g)) * 0.08 + 0.03
        amp = float(torch.rand(1, generator=g)) * 160 + 60
        img += amp * torch.exp(-((xs - cx * size) ** 2 + (ys - cy * size) ** 2) / (2 * (sig * size) ** 2))
        if amp > best:
            best, best_xy = amp, (cx * size, cy * size)
    img = img + 8.0 * torch.randn(size, size, generator=g)
    img = img.clamp(0, 255)
    gains = torch.tensor([1.0, 0.92, 0.85])[:, None, None]
    return (img[None] * gains).contiguous(), best_xy


def normalize_image(img255: torch.Tensor) -> torch.Tensor:
    """``load_video_frames_from_data`` normalisation (utils/misc.py:215-244): /255, ImageNet mean/std."""
    mean = torch.tensor([0.485, 0.456, 0.406])[:, None, None]
    std = torch.tensor([0.229, 0.224, 0.225])[:, None, None]
    return (img255 / 255.0 - mean) / std


def image_batch(seeds: List[int], size: int = 1024):
    imgs, clicks = [], []
    for s in seeds:
        im, xy = blob_image(s, size)
        imgs.append(normalize_image(im))
        clicks.append(xy)
    pts = torch.tensor(clicks, dtype=torch.float32)[:, None, :]  # [B, 1, 2] (x, y) pixels
    labels = torch.ones(len(seeds), 1, dtype=torch.int32)
    return torch.stack(imgs), pts, labels


def blob_volume(seed: int, n_slices: int = 64, size: int = 1024, n_objects: int = 1, normalised: bool = True):
    """A [T, 3, size, size] normalised volume of 3-D Gaussian-blob "organs" and, per object, the per-slice bounding box
    (x0, y0, x1, y1) of its iso-surface (None where the organ does not cut the slice).  normalised=False returns the 0..255
    intensities instead (what `SAM2VideoPredictor.val_init_state` expects)."""
    g = torch.Generator().manual_seed(5000 + seed)
    ys = torch.arange(size, dtype=torch.float32)[:, None]
    xs = torch.arange(size, dtype=torch.float32)[None, :]
    organs = []
    for _ in range(n_objects):
        c = (torch.rand(3, generator=g) * 0.5 + 0.25).tolist()
        r = (torch.rand(3, generator=g) * 0.12 + 0.1).tolist()
        organs.append((c, r))
    vol = torch.empty(n_slices, 3, size, size)
    boxes = [[None] * n_slices for _ in range(n_objects)]
    gains = torch.tensor([1.0, 0.92, 0.85])[:, None, None]
    for t in range(n_slices):
        z = (t + 0.5) / n_slices
        sl = 30.0 + 6.0 * torch.randn(size, size, generator=g)
        for o, ((cx, cy, cz), (rx, ry, rz)) in enumerate(organs):
            dz = (z - cz) / rz
            d2 = ((xs / size - cx) / rx) ** 2 + ((ys / size - cy) / ry) ** 2 + dz * dz
            sl = sl + 150.0 * torch.exp(-1.5 * d2)
            if abs(dz) < 1.0:
                s = math.sqrt(1.0 - dz * dz)
                boxes[o][t] = ((cx - rx * s) * size, (cy - ry * s) * size, (cx + rx * s) * size, (cy + ry * s) * size)
        raw = sl.clamp(0, 255)[None] * gains
        vol[t] = normalize_image(raw) if normalised else raw
    return vol, boxes
