"""eval_seg: the CPU oracle against the reference's own outputs (tests/golden/eval_seg.npz) and known answers; the GPU path
(medical_sam2_amd.metrics.eval_seg) against the oracle, with bit-exact integer counts."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import eval_seg_oracle as E  # noqa: E402
from helpers import load_npz  # noqa: E402


def _inputs(c):
    g = torch.Generator().manual_seed(700 + c)
    pred = torch.rand(3, c, 40, 48, generator=g)
    mask = (torch.rand(3, c, 40, 48, generator=g) > 0.6).float() * torch.rand(3, c, 40, 48, generator=g).clamp(min=0.2)
    mask[0, 0] = 0.0
    return pred, mask


@pytest.mark.parametrize("c", [1, 2, 3])
def test_oracle_matches_reference(c):
    g = load_npz("eval_seg.npz")
    pred, mask = _inputs(c)
    got = np.array(E.eval_seg(pred.numpy(), mask.numpy(), g[f"c{c}_thresholds"].tolist()), dtype=np.float64)
    assert got.shape == g[f"c{c}_result"].shape
    assert np.allclose(got, g[f"c{c}_result"], rtol=1e-6, atol=1e-7), (got, g[f"c{c}_result"])


def test_oracle_known_answers():
    p = np.zeros((1, 1, 4, 4), dtype=np.float32)
    t = np.zeros((1, 1, 4, 4), dtype=np.float32)
    p[0, 0, :2] = 1.0          # 8 pixels
    t[0, 0, 1:3] = 1.0         # 8 pixels, 4 shared
    iou, dice = E.eval_seg(p, t, (0.5,))
    assert abs(iou - (4 + 1e-6) / (12 + 1e-6)) < 1e-9 and abs(dice - (8 + 1e-4) / (16 + 1e-4)) < 1e-6
    iou, dice = E.eval_seg(np.zeros_like(p), np.zeros_like(t), (0.5,))     # both empty: the smoothing terms decide
    assert abs(iou - 1.0) < 1e-9 and abs(dice - 1.0) < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("c,shape", [(1, (3, 40, 48)), (2, (3, 40, 48)), (3, (3, 40, 48)), (4, (2, 257, 129)), (1, (1, 1024, 1024))])
def test_gpu_eval_seg_vs_oracle(c, shape):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import medical_sam2_amd.metrics as M
    b, h, w = shape
    g = torch.Generator().manual_seed(900 + c + h)
    pred = torch.rand(b, c, h, w, generator=g)
    mask = (torch.rand(b, c, h, w, generator=g) > 0.55).float() * torch.rand(b, c, h, w, generator=g)
    mask[0, 0] = 0.0
    th = (0.1, 0.3, 0.5, 0.7, 0.9)
    counts = M.seg_counts(pred.cuda(), mask.cuda(), th)
    for ti, t in enumerate(th):
        P, G = pred.numpy() > t, mask.numpy() > t
        assert np.array_equal(counts[ti, :, :, 0], (P & G).sum((2, 3)))
        assert np.array_equal(counts[ti, :, :, 1], P.sum((2, 3))) and np.array_equal(counts[ti, :, :, 2], G.sum((2, 3)))
    got = np.array(M.eval_seg(pred.cuda(), mask.cuda(), th), dtype=np.float64)
    ref = np.array(E.eval_seg(pred.numpy(), mask.numpy(), th), dtype=np.float64)
    assert np.allclose(got, ref, rtol=2e-6, atol=1e-7), (got, ref)
    many = tuple(np.linspace(0.05, 0.95, 11))                     # more than one launch's worth of thresholds
    assert np.allclose(np.array(M.eval_seg(pred.cuda(), mask.cuda(), many)), np.array(E.eval_seg(pred.numpy(), mask.numpy(), many)), rtol=2e-6)
