"""Device-side `eval_seg` (func_3d/utils.py:139-214, func_2d/utils.py:505-580): same signature and
return values, but the maps stay on the GPU -- one kernel counts |P & G|, |P|, |G| for every threshold, batch element and class, and
only those 3*T*B*C integers cross PCIe (the reference thresholds, copies and reduces both maps on the CPU once per threshold)."""
from __future__ import annotations

import numpy as np
import torch

from . import ops
from ._lib import check, lib

F32 = torch.float32


def seg_counts(pred: torch.Tensor, true_mask_p: torch.Tensor, threshold) -> np.ndarray:
    """int64 counts [T, b, c, 3] = (|pred>t & gt>t|, |pred>t|, |gt>t|)."""
    b, c, h, w = pred.shape
    p = pred.to(F32).contiguous()
    g = true_mask_p.to(device=p.device, dtype=F32).contiguous()
    assert g.shape == p.shape, "pred and mask must have the same shape"
    th = [float(t) for t in threshold]
    out = []
    for i in range(0, len(th), 8):
        part = torch.tensor(th[i: i + 8], dtype=F32, device=p.device)
        counts = torch.zeros(len(th[i: i + 8]), b * c, 3, dtype=torch.int32, device=p.device)
        check(lib().msam2_seg_counts(ops._p(p), ops._p(g), ops._p(part), part.numel(), b * c, h * w, ops._p(counts), ops._stream()))
        out.append(counts)
    return torch.cat(out).view(len(th), b, c, 3).cpu().numpy().astype(np.int64)


def eval_seg(pred: torch.Tensor, true_mask_p: torch.Tensor, threshold):
    """Drop-in for the reference's eval_seg: pred / true_mask_p [b, c, h, w] on the GPU, threshold = iterable of floats."""
    k = seg_counts(pred, true_mask_p, threshold)
    T, b, c, _ = k.shape
    inter, ps, gs = k[..., 0], k[..., 1], k[..., 2]
    union = ps + gs - inter
    iou = ((inter + 1e-6) / (union + 1e-6)).mean(axis=1)                                  # [T, c], float64 like numpy in the reference
    eps = np.float32(0.0001)
    dice_each = (np.float32(2) * inter.astype(np.float32) + eps) / (ps.astype(np.float32) + gs.astype(np.float32) + eps)
    dice = np.zeros((T, c), dtype=np.float32)
    for i in range(b):                                                                    # float32 running sum, like dice_coeff
        dice = dice + dice_each[:, i]
    dice = dice / np.float32(b)
    ious = [float(iou[:, i].sum()) / T for i in range(c)]
    dices = [float(dice[:, i].astype(np.float64).sum()) / T for i in range(c)]
    if c == 1:
        return ious[0], dices[0]
    return tuple(np.array(ious + dices))
