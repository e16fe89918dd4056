#!/usr/bin/env python3
"""kv64 split pass alone at the bench shape, a few split counts.  Usage: python tools/kv64_quick.py [label]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import medical_sam2_amd.ops as ops  # noqa: E402
from tools.kernel_bench import timeit  # noqa: E402

B, Lq, Lk = 4, 4096, 16384
g = torch.Generator().manual_seed(0)
q = torch.randn(B, 1, Lq, 256, generator=g).to(ops.OP16).cuda()
k = torch.randn(B, 1, Lk, 256, generator=g).to(ops.OP16).cuda()
m = torch.randn(B, 1, Lk, 64, generator=g).to(ops.OP16).cuda()
alg = 4.0 * B * Lq * Lk * 256
out = []
for rnd in range(3):
    for sp in (4, 8):
        ws = ops.attention_workspace(B, 1, Lq, 256, sp, "cuda")
        t = timeit(lambda: ops.attention_kv64(q, k, m, splits=sp, workspace=ws, defer_merge=True))
        out.append(f"s{sp}:{t * 1e6:6.1f}us/{alg * 0.625 / t / 1e12:5.0f}TF")
print((sys.argv[1] if len(sys.argv) > 1 else "default"), "  ".join(out), flush=True)
