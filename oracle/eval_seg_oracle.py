"""CPU restatement (numpy) of the reference's segmentation metric -- TEST INFRASTRUCTURE ONLY (never imported by the product).

Follows func_3d/utils.py:139-214 (`eval_seg`, `iou`) and 216-238 (`dice_coeff`, `DiceCoeff.forward`) line by line:
  per threshold th: binarise prediction and ground truth with `> th`;
  IoU  = mean over the batch of (|P & G| + 1e-6) / (|P | G| + 1e-6)            (numpy, float64);
  Dice = mean over the batch of (2 |P & G| + 1e-4) / (|P| + |G| + 1e-4)        (torch float32);
  the results are averaged over the thresholds; c == 1 -> (iou, dice); c == 2 -> (iou_0, iou_1, dice_0, dice_1);
  c > 2 -> (iou_0.., dice_0..).
Pinned by tests/golden/eval_seg.npz, generated from the reference's own function by tests/golden/make_golden.py.
"""
import numpy as np


def iou(outputs: np.ndarray, labels: np.ndarray) -> float:
    """func_3d/utils.py:205-214"""
    smooth = 1e-6
    inter = (outputs & labels).sum((1, 2))
    union = (outputs | labels).sum((1, 2))
    return float(((inter + smooth) / (union + smooth)).mean())


def dice_coeff(inp: np.ndarray, target: np.ndarray) -> float:
    """func_3d/utils.py:216-238 (float32 arithmetic like torch)"""
    s = np.float32(0.0)
    eps = np.float32(0.0001)
    for a, b in zip(inp, target):
        inter = np.float32(np.dot(a.reshape(-1).astype(np.float32), b.reshape(-1).astype(np.float32)))
        union = np.float32(a.astype(np.float32).sum()) + np.float32(b.astype(np.float32).sum()) + eps
        s = s + (np.float32(2) * inter + eps) / union
    return float(s / np.float32(len(inp)))


def eval_seg(pred: np.ndarray, true_mask_p: np.ndarray, threshold):
    """func_3d/utils.py:139-203; pred / true_mask_p float [b, c, h, w]."""
    b, c, h, w = pred.shape
    ious = [0.0] * c
    dices = [0.0] * c
    for th in threshold:
        g = (true_mask_p > th)
        p = (pred > th)
        for i in range(c):
            ious[i] += iou(p[:, i].astype("int32"), g[:, i].astype("int32"))
            dices[i] += dice_coeff(p[:, i].astype(np.float32), g[:, i].astype(np.float32))
    n = len(threshold)
    if c == 1:
        return ious[0] / n, dices[0] / n
    return tuple(np.array(ious + dices) / n)
