"""Shared test helpers (CPU side)."""
import json
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def load_npz(name):
    return dict(np.load(os.path.join(GOLDEN, name)))


def load_meta():
    with open(os.path.join(GOLDEN, "meta.json")) as f:
        return json.load(f)


def sub(t: torch.Tensor, n: int = 4096) -> np.ndarray:
    """Same deterministic strided subsample as tests/golden/make_golden.py:sub."""
    f = t.detach().float().contiguous().flatten()
    step = max(1, f.numel() // n)
    return f[::step][:n].numpy().copy()


def rel_err(a, b) -> float:
    a = torch.as_tensor(np.asarray(a), dtype=torch.float64)
    b = torch.as_tensor(np.asarray(b), dtype=torch.float64)
    return float((a - b).norm() / (b.norm() + 1e-30))


def max_abs(a, b) -> float:
    a = torch.as_tensor(np.asarray(a), dtype=torch.float64)
    b = torch.as_tensor(np.asarray(b), dtype=torch.float64)
    return float((a - b).abs().max())


def mask_iou(a, b) -> float:
    a = torch.as_tensor(np.asarray(a)) > 0
    b = torch.as_tensor(np.asarray(b)) > 0
    u = (a | b).sum().item()
    return 1.0 if u == 0 else (a & b).sum().item() / u
