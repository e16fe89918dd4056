#!/usr/bin/env python3
"""Memory cross-attention at the benchmark's shape (B=4 objects, Lq=4096, Lk=16384): the 256-wide formulation
(attn_glds_kernel<256>) against the value-folded kernel (attn_kv64_kernel), split pass alone and with the merge, interleaved rounds in
one process.  Usage: python tools/attn_kv64_bench.py [B Lq Lk]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import medical_sam2_amd.ops as ops  # noqa: E402
from tools.kernel_bench import timeit  # noqa: E402

B, Lq, Lk = [int(x) for x in sys.argv[1:4]] if len(sys.argv) > 3 else (4, 4096, 16384)
g = torch.Generator().manual_seed(0)
q = torch.randn(B, 1, Lq, 256, generator=g).to(ops.OP16).cuda()
k = torch.randn(B, 1, Lk, 256, generator=g).to(ops.OP16).cuda()
v = torch.randn(B, 1, Lk, 256, generator=g).to(ops.OP16).cuda()
m = torch.randn(B, 1, Lk, 64, generator=g).to(ops.OP16).cuda()
alg = 4.0 * B * Lq * Lk * 256
for rnd in range(3):
    for sp in (4, 6, 8, 12):
        ws = ops.attention_workspace(B, 1, Lq, 256, sp, "cuda")
        t_old = timeit(lambda: ops.attention(q, k, v, splits=sp, workspace=ws, defer_merge=True))
        t_new = timeit(lambda: ops.attention_kv64(q, k, m, splits=sp, workspace=ws, defer_merge=True))
        t_newm = timeit(lambda: ops.attention_kv64(q, k, m, splits=sp, workspace=ws))
        print(f"round {rnd} splits {sp:2d}: d256 {t_old * 1e6:7.1f} us ({alg / t_old / 1e12:6.0f} TF/s)   kv64 {t_new * 1e6:7.1f} us "
              f"({alg / t_new / 1e12:6.0f} TF/s on the 256-wide flops, {alg * 0.625 / t_new / 1e12:6.0f} executed)   kv64+merge {t_newm * 1e6:7.1f} us", flush=True)
