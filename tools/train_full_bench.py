#!/usr/bin/env python3
"""One whole 2-D training iteration INCLUDING the image encoder's backward at BASELINE configs[1] (bench.py's `train_iteration` side
figure, alone, for rocprofv3): python tools/train_full_bench.py [frozen]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

torch.set_grad_enabled(False)
dev = torch.device("cuda", 0)
m = bench.build_model(dev)
imgs, pts, labels, bank, sampled = bench.make_inputs(dev, 4, 0)
memory, memory_pos = bench.assemble_memory(m, bank, sampled)
print(bench.train_iteration(m, imgs, pts, labels, memory, memory_pos, dev, full=len(sys.argv) < 2))
