#!/usr/bin/env python3
"""BASELINE.json configs[2]: sam2_hiera_s, synthetic 64-slice volume at 1024^2, bbox prompt every 2 slices, propagation through the
rest with the memory bank (1 GPU).  Prints slices/s for eager launches and for hipGraph replays of the per-slice forward (graphs.GraphedPropagation: one graph per
memory-bank bucket, pointer tail padded, key count on the device)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import medical_sam2_amd.build_sam as bs  # noqa: E402
import medical_sam2_amd.synthetic as syn  # noqa: E402
import medical_sam2_amd.volume as vol  # noqa: E402
import medical_sam2_amd.weights as wts  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n_obj = int(sys.argv[2]) if len(sys.argv) > 2 else 1
S = 1024
dev = torch.device("cuda", 0)
torch.set_grad_enabled(False)
m = bs.build_sam2("sam2_hiera_s", device="cpu", hydra_overrides_extra=[f"++model.image_size={S}"])
m.load_state_dict(wts.init_weights("hiera_s", 0), strict=True)
m = m.to(dev).eval()
volume, boxes = syn.blob_volume(0, n_slices=T, size=S, n_objects=n_obj)
volume = volume.to(dev)


def box_at(t):
    return torch.tensor([[float(v) for v in (boxes[o][t] or (S * 0.3, S * 0.3, S * 0.6, S * 0.6))] for o in range(n_obj)], device=dev)


prompts = {t: {"boxes": box_at(t)} for t in range(0, T, 2)}
vol.segment_volume(m, volume[:4], {0: prompts[0], 2: prompts[2]}, padded_bank=False)  # warm-up (weight packing, tables, code objects)
torch.cuda.synchronize()


def timed(label, **kw):
    st = {}
    t0 = time.perf_counter()
    masks = vol.segment_volume(m, volume, prompts, fill_hole_area=8, stats=st, **kw)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    fg = sum(float((v > 0).float().mean()) for v in masks.values()) / len(masks)
    print(f"volume T={T} n_obj={n_obj} {label}: {dt:.3f} s -> {T / dt:.1f} slices/s (mean foreground fraction {fg:.3f}) {st}", flush=True)
    return masks


eager = timed("eager launches, bank re-assembled per slice", padded_bank=False)
cache = {}
first = timed("hipGraph replay, first volume (captures inside)", graphs=True, graph_cache=cache)
again = timed("hipGraph replay, graphs kept from the previous volume", graphs=True, graph_cache=cache)
padded = timed("eager launches, assembled bank kept per bucket (default)", padded_bank=True)
same = all(torch.equal(first[t], padded[t]) and torch.equal(again[t], padded[t]) for t in first)
worst = max(float((eager[t] - padded[t]).abs().max()) for t in eager)
flips = sum(int(((eager[t] > 0) != (padded[t] > 0)).sum()) for t in eager)
print(f"graphed == padded eager bit for bit: {same}; padded vs un-padded eager: max |dlogit| {worst:.4f}, {flips} flipped mask pixels")
