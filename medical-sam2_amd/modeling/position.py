"""Position encodings (sam2_train/modeling/position_encoding.py) as device-generated, cached tables."""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.nn as nn

from .. import ops
from .common import F32, nchw_view


class PositionEmbeddingSine(nn.Module):
    """position_encoding.py:16-112; forward(x) depends only on x's spatial size, so the table is generated once per
    (H, W) by msam2_sine_pos_2d and broadcast over the batch (the reference caches it the same way, 80-82,111)."""

    def __init__(self, num_pos_feats, temperature: int = 10000, normalize: bool = True, scale: Optional[float] = None):
        super().__init__()
        assert num_pos_feats % 2 == 0, "Expecting even model width"
        assert normalize and scale is None, "SAM2 configs use normalize=True, scale=2*pi"
        self.num_pos_feats = num_pos_feats // 2
        self.temperature = temperature
        self.cache = {}

    def table(self, h: int, w: int, device) -> torch.Tensor:
        key = (h, w, str(device))
        if key not in self.cache:
            self.cache[key] = ops.sine_pos_2d(h, w, 2 * self.num_pos_feats, device, float(self.temperature))
        return self.cache[key]  # [h*w, C] fp32

    @torch.no_grad()
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        B, _, H, W = x.shape
        return nchw_view(self.table(H, W, x.device), 1, H, W).expand(B, -1, -1, -1)


class PositionEmbeddingRandom(nn.Module):
    """position_encoding.py:115-158 (random Fourier features); the grid form is generated on device."""

    def __init__(self, num_pos_feats: int = 64, scale: Optional[float] = None) -> None:
        super().__init__()
        if scale is None or scale <= 0.0:
            scale = 1.0
        self.register_buffer("positional_encoding_gaussian_matrix", scale * torch.randn((2, num_pos_feats)))

    def grid_tokens(self, size: Tuple[int, int]) -> torch.Tensor:
        h, w = size
        return ops.fourier_pe_grid(self.positional_encoding_gaussian_matrix.to(F32), h, w)  # [h*w, C]

    def forward(self, size: Tuple[int, int]) -> torch.Tensor:
        h, w = size
        return nchw_view(self.grid_tokens(size), 1, h, w)[0]  # C x H x W
