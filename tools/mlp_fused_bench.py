#!/usr/bin/env python3
"""Fused LN + MLP block (msam2_ln_mlp_residual_fwd) against the three-launch path at the Hiera stage-1 / stage-2 / stage-3 shapes (B = 4, 1024^2)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import medical_sam2_amd.ops as ops  # noqa: E402
from tools.win_attn_bench import timeit  # noqa: E402  (graph-replay timer)
g = torch.Generator().manual_seed(0)
for dim, T in ((96, 262144), (192, 65536), (384, 16384)):
    x = torch.randn(T, dim, generator=g).cuda()
    lw, lb = torch.ones(dim).cuda(), torch.zeros(dim).cuda()
    w1 = (torch.randn(4 * dim, dim, generator=g) / dim ** 0.5).to(ops.OP16).cuda()
    w2 = (torch.randn(dim, 4 * dim, generator=g) / (4 * dim) ** 0.5).to(ops.OP16).cuda()
    b1, b2 = torch.zeros(4 * dim).cuda(), torch.zeros(dim).cuda()
    w2p = ops.mlp_fused_permute_w2(w2)
    def three():
        xn = ops.layernorm(x, lw, lb, 1e-6)
        h = ops.gemm(xn, w1, b1, act=ops.ACT_GELU)
        return ops.gemm(h, w2, b2, residual=x, out_dtype=torch.float32)
    t3 = timeit(three)
    tf = timeit(lambda: ops.ln_mlp_residual(x, lw, lb, 1e-6, w1, b1, w2p, b2))
    fl = 4.0 * T * dim * 4 * dim
    print(f"dim {dim:3d} T {T:6d}: three launches {t3 * 1e6:7.1f} us   fused {tf * 1e6:7.1f} us ({fl / tf / 1e12:5.0f} TF/s, {T * dim * 8 / tf / 1e12:4.2f} TB/s of x in + out)", flush=True)
