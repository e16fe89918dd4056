#!/usr/bin/env python3
"""Micro-benchmarks of the two MFMA kernels at the shapes of the benchmark workload (GPU only).
Usage: python tools/kernel_bench.py [gemm|attn|all]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import medical_sam2_amd.ops as ops  # noqa: E402

DEV = "cuda"


def timeit(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


def gemm_shapes(B=4):
    T1, T2, T3, T4 = B * 65536, B * 16384, B * 4096, B * 1024
    return [
        ("patch_embed", T1, 96, 160, "f32"), ("b0.qkv", T1, 288, 96, "bf16"), ("b0.proj", T1, 96, 96, "f32"),
        ("b0.fc1", T1, 384, 96, "bf16"), ("b0.fc2", T1, 96, 384, "f32"), ("b1.qkv", T1, 576, 96, "bf16"),
        ("b1.short", T1, 192, 96, "f32"), ("b2.qkv", T2, 576, 192, "bf16"), ("b2.fc1", T2, 768, 192, "bf16"),
        ("b2.fc2", T2, 192, 768, "f32"), ("b3.qkv", T2, 1152, 192, "bf16"),
        ("s3.qkv", T3, 1152, 384, "bf16"), ("s3.proj", T3, 384, 384, "f32"), ("s3.fc1", T3, 1536, 384, "bf16"),
        ("s3.fc2", T3, 384, 1536, "f32"), ("s4.qkv", T4, 2304, 768, "bf16"), ("s4.fc1", T4, 3072, 768, "bf16"),
        ("s4.fc2", T4, 768, 3072, "f32"), ("neck0", T1, 256, 96, "f32"), ("conv_s0", T1, 32, 256, "f32"),
        ("ma.qkv", T3, 768, 256, "bf16"), ("ma.kproj", B * 16384, 256, 64, "bf16"), ("ma.fc1", T3, 2048, 256, "bf16"),
        ("ma.fc2", T3, 256, 2048, "f32"), ("dec.up1", T3, 256, 256, "bf16"), ("dec.up2", B * 16384, 128, 64, "bf16"),
    ]


def bench_gemm():
    g = torch.Generator().manual_seed(0)
    tot = 0.0
    print(f"{'name':12s} {'M':>8s} {'N':>5s} {'K':>5s} {'us':>9s} {'TF/s':>8s} {'GB/s':>8s}")
    for name, M, N, K, od in gemm_shapes():
        a = torch.randn(M, K, generator=g).to(ops.OP16).to(DEV)
        w = (torch.randn(N, K, generator=g) * 0.05).to(ops.OP16).to(DEV)
        bias = torch.randn(N, generator=g).to(DEV)
        out = torch.empty(M, N, dtype=torch.float32 if od == "f32" else ops.OP16, device=DEV)
        res = torch.randn(M, N, generator=g).to(DEV) if od == "f32" else None
        t = timeit(lambda: ops.gemm(a, w, bias, act=1 if "fc1" in name else 0, residual=res, out=out))
        by = 2 * M * K + 2 * N * K + M * N * (4 if od == "f32" else 2) + (M * N * 4 if res is not None else 0)
        print(f"{name:12s} {M:8d} {N:5d} {K:5d} {t * 1e6:9.1f} {2 * M * N * K / t / 1e12:8.1f} {by / t / 1e9:8.0f}")
        tot += t
    print(f"sum {tot * 1e3:.3f} ms")


def bench_attn():
    g = torch.Generator().manual_seed(0)
    from medical_sam2_amd.modeling.common import attn_splits
    print(f"{'shape':40s} {'splits':>6s} {'us':>9s} {'TF/s':>8s}")
    for (B, H, Lq, Lk, D) in [(4, 1, 4096, 4096, 256), (4, 1, 4096, 16384, 256), (1, 1, 4096, 28704, 256), (4, 4, 4096, 4096, 96),
                              (1, 4, 4096, 4096, 96)]:
        q = torch.randn(B, H, Lq, D, generator=g).to(ops.OP16).to(DEV)
        k = torch.randn(B, H, Lk, D, generator=g).to(ops.OP16).to(DEV)
        v = torch.randn(B, H, Lk, D, generator=g).to(ops.OP16).to(DEV)
        for sp in sorted({1, attn_splits(B, H, Lq, Lk), 2, 4, 8}):
            t = timeit(lambda: ops.attention(q, k, v, splits=sp))
            print(f"B{B} H{H} Lq{Lq} Lk{Lk} D{D}".ljust(40) + f" {sp:6d} {t * 1e6:9.1f} {4 * B * H * Lq * Lk * D / t / 1e12:8.1f}")
    # windowed stage-3 block: 4 images, 64x64 tokens, 4 heads, ws 14
    B, Hh, heads, D = 4, 64, 4, 96
    qkv = torch.randn(B * Hh * Hh, 3 * heads * D, generator=g).to(ops.OP16).to(DEV)
    bias = torch.randn(3 * heads * D, generator=g).to(DEV)
    t = timeit(lambda: ops.window_attention(qkv, B, Hh, Hh, heads, 14, bias))
    print(f"window ws14 B4 64x64 h4".ljust(40) + f" {1:6d} {t * 1e6:9.1f} {4 * B * 25 * heads * 196 * 196 * D / t / 1e12:8.1f}")
    B, Hh, heads = 4, 256, 1
    qkv = torch.randn(B * Hh * Hh, 3 * heads * D, generator=g).to(ops.OP16).to(DEV)
    bias = torch.randn(3 * heads * D, generator=g).to(DEV)
    t = timeit(lambda: ops.window_attention(qkv, B, Hh, Hh, heads, 8, bias))
    print(f"window ws8 B4 256x256 h1".ljust(40) + f" {1:6d} {t * 1e6:9.1f} {4 * B * 1024 * heads * 64 * 64 * D / t / 1e12:8.1f}  ({qkv.numel() * 2 * 4 / 3 / t / 1e9:.0f} GB/s)")


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("gemm", "all"):
        bench_gemm()
    if what in ("attn", "all"):
        bench_attn()
