#!/usr/bin/env python3
"""Which (weight seed, volume seed) give the config-3 parity tests real masks?  VERDICT r2 weak item 1: with weight seed 0 / volume seed 0
the sampled propagated slices of the 64-slice 1024^2 volume have NO foreground, so the IoU clause compared empty with empty.  Prints, per
seed pair, the foreground pixel counts of the HIP chain's low-res masks on the sampled slices (GPU only)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import medical_sam2_amd.build_sam as bs  # noqa: E402
import medical_sam2_amd.synthetic as syn  # noqa: E402
import medical_sam2_amd.volume as vol  # noqa: E402
import medical_sam2_amd.weights as wts  # noqa: E402

S = 1024
dev = torch.device("cuda", 0)
torch.set_grad_enabled(False)
T = int(sys.argv[1]) if len(sys.argv) > 1 else 64
wseeds = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else list(range(6))
vseeds = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0, 1]
sample = (5, T // 2 + 1, T - 1)
m = bs.build_sam2("sam2_hiera_s", device="cpu", hydra_overrides_extra=[f"++model.image_size={S}", "++model.binarize_mask_from_pts_for_mem_enc=true"])
for vs in vseeds:
    volume, boxes = syn.blob_volume(vs, n_slices=T, size=S, n_objects=1)
    volume = volume.to(dev)
    box_at = lambda t: torch.tensor([[float(v) for v in (boxes[0][t] or (S * 0.3, S * 0.3, S * 0.6, S * 0.6))]], device=dev)
    prompts = {t: {"boxes": box_at(t)} for t in range(0, T, 2)}
    for ws in wseeds:
        m.load_state_dict(wts.init_weights("hiera_s", ws), strict=True)
        m = m.to(dev).eval()
        masks = vol.segment_volume(m, volume, prompts, fill_hole_area=0)
        fg = {t: int((masks[t] > 0).sum()) for t in sample}
        allfg = sorted(int((masks[t] > 0).sum()) for t in range(1, T, 2))
        print(f"T={T} weights seed {ws} volume seed {vs}: fg on sampled propagated slices {fg}; over all propagated: min {allfg[0]} median {allfg[len(allfg) // 2]} "
              f"max {allfg[-1]}; box present on {sum(1 for t in range(T) if boxes[0][t])} slices", flush=True)
        m = m.cpu()
