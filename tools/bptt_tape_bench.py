#!/usr/bin/env python3
"""training_3d: the full tape against the bounded one (volume_forward_saved(bounded_tape=True): the memory attention's intermediates are
re-created slice by slice in the backward) -- bytes held after the forward, peak bytes, time of forward + backward.
hiera_t at 256^2, N slices, 2 objects, one box-prompted slice, random weights, dropout 0.1."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import medical_sam2_amd.build_sam as bs  # noqa: E402
import medical_sam2_amd.synthetic as syn  # noqa: E402
import medical_sam2_amd.training_3d as t3  # noqa: E402
import medical_sam2_amd.weights as wts  # noqa: E402
from medical_sam2_amd.training import upsampled_mask_loss  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 16
S, n = 256, 2
dev = "cuda"
m = bs.build_sam2("sam2_hiera_t", device="cpu", hydra_overrides_extra=[f"++model.image_size={S}"])
m.load_state_dict(wts.init_weights("hiera_t", 0), strict=True)
m = m.to(dev).train()
volume, boxes = syn.blob_volume(3, n_slices=N, size=S, n_objects=n)
volume = volume.to(dev)
dflt = (S * 0.3, S * 0.3, S * 0.6, S * 0.6)
prompts = {0: {"boxes": torch.tensor([[float(v) for v in (boxes[o][0] or dflt)] for o in range(n)], device=dev)}}
targets = {t: torch.zeros(n, 1, S, S, device=dev) for t in range(N)}
for t in range(N):
    for o in range(n):
        b = boxes[o][t]
        if b is not None:
            x0, y0, x1, y1 = [int(round(float(v))) for v in b]
            targets[t][o, :, max(y0, 0): y1 + 1, max(x0, 0): x1 + 1] = 1.0
ma = m.memory_attention


def run(bounded):
    ma.dropout_seed, ma._dropout_calls = 5, 0
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    t0 = time.perf_counter()
    with torch.no_grad():
        tape, low = t3.volume_forward_saved(m, volume, prompts, bounded_tape=bounded)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        held = torch.cuda.memory_allocated() - base
        d = {t: upsampled_mask_loss(low[t], targets[t], 0, 2.0)[1] / (N - 1) for t in range(1, N)}
        g = t3.volume_backward(m, tape, d)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return held, torch.cuda.max_memory_allocated() - base, t1 - t0, t2 - t1


run(False)  # warm
for bounded in (False, True, False, True):
    held, peak, tf, tb = run(bounded)
    print(f"{N} slices, bounded_tape={bounded}: held after forward {held / 2**20:8.1f} MiB, peak {peak / 2**20:8.1f} MiB, forward {tf * 1e3:7.1f} ms, backward {tb * 1e3:7.1f} ms", flush=True)
