#!/usr/bin/env python3
"""Hiera windowed attention at the benchmark's shapes (B = 4, hiera_s): the whole-window kernel (attn_win_kernel) against the tiled
register-staged one (MSAM2_WIN_V1=1), with the algorithmic bytes (q, k, v, o once, 16-bit) each launch moves.  GPU only."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import medical_sam2_amd.ops as ops  # noqa: E402


def timeit(fn, n=10, reps=5):
    """kernel time without host launch overhead: n calls captured into one graph, replayed"""
    fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (n * reps) * 1e-3


B = 4
# (name, H=W of the token image, dim_out, heads, window, q-pool)
CASES = [("block0", 256, 96, 1, 8, False), ("block1", 256, 192, 2, 8, True), ("block2", 128, 192, 2, 4, False),
         ("block3", 128, 384, 4, 4, True), ("stage3", 64, 384, 4, 14, False), ("block14", 64, 768, 8, 14, True), ("block15", 32, 768, 8, 7, False)]
g = torch.Generator().manual_seed(0)
for name, hw, dim, heads, ws, pool in (CASES if __name__ == "__main__" else []):
    T = B * hw * hw
    qkv = (torch.randn(T, 3 * dim, generator=g) * 0.5).to(ops.OP16).cuda()
    bias = torch.randn(3 * dim, generator=g).cuda()
    qp = (torch.randn(T // 4, dim, generator=g) * 0.5).to(ops.OP16).cuda() if pool else None
    tq = T // 4 if pool else T
    by = 2.0 * dim * (tq + 2 * T + tq)
    res = []
    for v1 in ("1", "0"):
        os.environ["MSAM2_WIN_V1"] = v1
        os.environ["MSAM2_NO_TINYWIN"] = v1          # the tiled pass also switches the tiny-window kernel (16 / 64-key windows) off
        t = timeit(lambda: ops.window_attention(qkv, B, hw, hw, heads, ws, bias, q_pooled=qp))
        res.append(t)
    print(f"{name:8s} tokens {T:7d} dim {dim:4d} heads {heads} ws {ws:2d} pool {int(pool)}: tiled {res[0] * 1e6:7.1f} us ({by / res[0] / 1e12:5.2f} TB/s)   "
          f"whole-window / tiny-window {res[1] * 1e6:7.1f} us ({by / res[1] / 1e12:5.2f} TB/s = {by / res[1] / 8e12 * 100:4.1f} % of 8 TB/s)", flush=True)
