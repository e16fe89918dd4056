"""Shared host-side helpers for the drop-in modules: a version-checked cache of kernel-ready (bf16 / repacked) weights
and tiny tensor-view utilities.  Nothing here computes on activations."""
from __future__ import annotations

from typing import Callable, Dict, Sequence, Tuple

import torch

from .. import ops

OP16, F32 = ops.OP16, torch.float32  # OP16: the library 16-bit MFMA operand dtype (fp16 by default)


class WeightCache:
    """Kernel-ready copies of parameters (bf16 casts, fused/concatenated or permuted layouts), rebuilt only when the
    source parameter storage or its in-place version counter changes (i.e. after load_state_dict / an optimizer step).
    The repacking is pure data movement done once per weight update -- not part of the per-slice path."""

    def __init__(self):
        self._c: Dict[str, Tuple[tuple, object]] = {}
        self._casts: Dict[str, torch.Tensor] = {}      # key -> the single parameter a plain 16-bit cast was built from (refresh_casts)

    @staticmethod
    def _sig(params: Sequence[torch.Tensor]) -> tuple:
        return tuple((p.data_ptr(), p._version, str(p.device), p.dtype) for p in params)

    def get(self, key: str, params: Sequence[torch.Tensor], build: Callable[[], object], cast_of: torch.Tensor = None):
        sig = self._sig(params)
        hit = self._c.get(key)
        if hit is None or hit[0] != sig:
            old = None if hit is None else hit[1]
            if cast_of is not None and key in self._casts and isinstance(old, torch.Tensor) and old.numel() == cast_of.numel() \
                    and old.device == cast_of.device:
                # A plain 16-bit cast is refreshed IN PLACE: its storage is stable for the life of the cache entry.  `refresh_casts`
                # inside a captured training step bakes these pointers into the hipGraph; replacing the tensor here (an eager reader
                # between two replays, after `mark_updated`) would hand the old block back to the allocator while the graph still
                # writes weights into it and runs GEMMs out of it.
                with torch.no_grad():
                    old.copy_(cast_of.detach().reshape(old.shape))
                hit = (sig, old)
            else:
                with torch.no_grad():
                    hit = (sig, build())
            self._c[key] = hit
            if cast_of is not None:
                self._casts[key] = cast_of
        return hit[1]

    def tensors(self):
        """every cached kernel-ready tensor (GraphedStep pins them for the life of its graph)"""
        for _, val in self._c.values():
            for t in (val if isinstance(val, (tuple, list)) else (val,)):
                if isinstance(t, torch.Tensor):
                    yield t

    def clear(self):
        self._c.clear()
        self._casts.clear()


def cached_weight_tensors(model: torch.nn.Module) -> list:
    """every kernel-ready weight tensor the modules of `model` currently hold in their caches"""
    out = []
    for mod in model.modules():
        wc = getattr(mod, "_wc", None)
        if isinstance(wc, WeightCache):
            out.extend(wc.tensors())
    return out


def refresh_casts(model: torch.nn.Module) -> int:
    """After an optimiser step every plain 16-bit weight copy (`w_bf16` of one parameter) of every module is stale; rebuilt lazily that is
    one conversion kernel per weight (141 per training iteration, ~6 us each: latency, not bytes).  This refreshes all stale ones IN
    PLACE with one multi-tensor copy and re-stamps their cache entries; fused / permuted layouts still rebuild lazily.  Returns the number
    of copies refreshed.  Safe to call at any time (e.g. at the head of a training step, inside a captured graph)."""
    import os
    if os.environ.get("MSAM2_NO_REFRESH"):
        return 0
    dsts, srcs, stamp = [], [], []
    for mod in model.modules():
        wc = getattr(mod, "_wc", None)
        if not isinstance(wc, WeightCache):
            continue
        for key, src in wc._casts.items():
            hit = wc._c.get(key)
            if hit is None:
                continue
            sig = WeightCache._sig((src,))
            if hit[0] == sig or hit[0][0][0] != sig[0][0] or hit[1].numel() != src.numel():
                continue                                  # fresh, or the parameter was re-allocated: the lazy path handles that
            dsts.append(hit[1])
            srcs.append(src.detach().reshape(hit[1].shape))
            stamp.append((wc, key, sig, hit[1]))
    if dsts:
        with torch.no_grad():
            torch._foreach_copy_(dsts, srcs)
        for wc, key, sig, val in stamp:
            wc._c[key] = (sig, val)
    return len(dsts)


def w_bf16(cache: WeightCache, key: str, *weights: torch.Tensor) -> torch.Tensor:
    """bf16 [sum(out_i), in] matrix from one or more nn.Linear / 1x1-conv weights stacked along the output dim."""
    if len(weights) == 1:                                  # one conversion pass, no concatenation copy (re-done after every optimiser step)
        w0 = weights[0]
        return cache.get(key, weights, lambda: w0.detach().reshape(w0.shape[0], -1).to(OP16).contiguous(), cast_of=w0)
    return cache.get(key, weights, lambda: torch.cat([w.detach().reshape(w.shape[0], -1) for w in weights], 0).to(OP16).contiguous())


def v_f32(cache: WeightCache, key: str, *vecs: torch.Tensor) -> torch.Tensor:
    """fp32 vector from one or more parameter vectors (biases, norm weights).  A single fp32 contiguous parameter is used in place (a
    view: nothing to repack after an optimiser step); several are concatenated into a cached copy."""
    if len(vecs) == 1 and vecs[0].dtype == F32 and vecs[0].is_contiguous():
        return vecs[0].detach().reshape(-1)
    return cache.get(key, vecs, lambda: torch.cat([v.detach().reshape(-1) for v in vecs], 0).to(F32).contiguous())


def tokens_of(x_nchw: torch.Tensor) -> torch.Tensor:
    """[B,C,H,W] -> token-major [B*H*W, C].  Free when the tensor is an NCHW view of NHWC memory (what this package's
    modules return); foreign channels-first tensors are re-laid-out once by torch (data movement only)."""
    B, C, H, W = x_nchw.shape
    t = x_nchw.permute(0, 2, 3, 1)
    if not t.is_contiguous():
        t = t.contiguous()
    return t.reshape(B * H * W, C)


def nchw_view(tokens: torch.Tensor, B: int, H: int, W: int) -> torch.Tensor:
    """token-major [B*H*W, C] -> [B,C,H,W] view (no copy)."""
    return tokens.reshape(B, H, W, tokens.shape[-1]).permute(0, 3, 1, 2)


def to_bf16(x2d: torch.Tensor) -> torch.Tensor:
    """bf16 copy of a [rows, C] tensor through the add_cast kernel (no-op for bf16 input)."""
    if x2d.dtype == OP16:
        return x2d
    return ops.add_cast(x2d.unsqueeze(0), None, 1.0, OP16)[0]


def attn_splits(B: int, H: int, Lq: int, Lk: int) -> int:
    """Split-KV factor: enough workgroups (128 queries each) to cover the 256 CUs twice, at least 8 key tiles per split.
    Inside `parallel.batch_invariant_splits()` (the 3-D propagation chain, volume.segment_volume) the factor is the one a batch of ONE
    object gets: the split boundaries -- and with them the rounding of the 16-bit partials -- then do not depend on how many objects a
    rank carries, so an object-sharded chain reproduces the single-rank bits."""
    from .. import parallel
    if parallel.BATCH_INVARIANT_SPLITS:
        B = 1
    wgs = B * H * ((Lq + 127) // 128)
    if wgs >= 512 or Lk < 1024:
        return 1
    return max(1, min((512 + wgs - 1) // wgs, Lk // 256, 16))
