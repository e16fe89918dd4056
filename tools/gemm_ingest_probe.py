#!/usr/bin/env python3
"""Where the A operand comes from sets a GEMM tile's time: the fc2 shape of Hiera stage 3 (K = 1536, fp32 output + residual,
128 x 128 tiles) with LONE tiles (N = 128: one tile per CU on 16 - 128 CUs) and as the product launch (N = 384: 384 tiles on 256 CUs),
A served from (a) the XCD's L2 (a small A re-read by every replay), (b) the Infinity Cache (one 50 MB A, every launch), (c) HBM
(launches rotate over 8 such operands = 400 MB > the 256 MB cache).  Times are graph replays (no host launch overhead)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import medical_sam2_amd.ops as ops  # noqa: E402
from tools.win_attn_bench import timeit  # noqa: E402

K = 1536
g = torch.Generator().manual_seed(0)


def operands(M, N, n_a):
    a = [torch.randn(M, K, generator=g).to(ops.OP16).cuda() for _ in range(n_a)]
    w = (torch.randn(N, K, generator=g) * 0.03).to(ops.OP16).cuda()
    b = torch.randn(N, generator=g).cuda()
    res = torch.randn(M, N, generator=g).cuda()
    out = torch.empty(M, N, dtype=torch.float32, device="cuda")
    return a, w, b, res, out


def case(label, M, N, n_a):
    a, w, b, res, out = operands(M, N, n_a)

    def fn():
        for x in a:
            ops.gemm(x, w, b, residual=res, out=out, out_dtype=torch.float32)
    t = timeit(fn, n=max(1, 8 // n_a)) / n_a
    tiles = (M // 128) * ((N + 127) // 128)
    per_cu = -(-tiles // 256)
    abytes, wbytes = 128 * K * 2, 128 * K * 2
    print(f"{label:58s} M {M:6d} N {N:4d}: {tiles:4d} tiles ({per_cu} per CU at most)  {t * 1e6:7.1f} us per launch", flush=True)
    return t


print("model (DESIGN 3.5): per tile 393 KB of A + 393 KB of W; L2 ~70 GB/s per CU, Infinity Cache ~33, HBM ~23 => lone tile 11.2 / 17.5 / 22.7 us + epilogue")
case("lone tiles, A 6 MB re-read by every launch (L2)", 2048, 128, 1)
case("lone tiles, A 50 MB, one operand (Infinity Cache)", 16384, 128, 1)
case("lone tiles, A rotating over 8 x 50 MB (HBM)", 16384, 128, 8)
case("product shape, A 50 MB, one operand (Infinity Cache)", 16384, 384, 1)
case("product shape, A rotating over 8 x 50 MB (HBM)", 16384, 384, 8)
case("product shape at an eighth of the rows (L2)", 2048, 384, 1)
