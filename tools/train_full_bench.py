#!/usr/bin/env python3
"""One whole 2-D training iteration INCLUDING the image encoder's backward at BASELINE configs[1] (bench.py's `train_iteration` side
figure, alone, for rocprofv3): python tools/train_full_bench.py [frozen | hiera_b+]   (hiera_b+ = BASELINE configs[4]'s model, 4 slices per GPU)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

torch.set_grad_enabled(False)
dev = torch.device("cuda", 0)
if "hiera_b+" in sys.argv[1:]:
    import medical_sam2_amd.build_sam as bs
    import medical_sam2_amd.weights as wts
    m = bs.build_sam2("sam2_hiera_b+", device="cpu", hydra_overrides_extra=["++model.image_size=1024"])
    m.load_state_dict(wts.init_weights("hiera_b+", 0), strict=True)
    m = m.to(dev).eval()
else:
    m = bench.build_model(dev)
imgs, pts, labels, bank, sampled = bench.make_inputs(dev, 4, 0)
memory, memory_pos = bench.assemble_memory(m, bank, sampled)
print(bench.train_iteration(m, imgs, pts, labels, memory, memory_pos, dev, full="frozen" not in sys.argv[1:]))
