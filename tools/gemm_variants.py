#!/usr/bin/env python3
"""Experiment driver: every GEMM kernel variant (MSAM2_GEMM_VARIANT) on the benchmark step's hot shapes, checked against fp32 matmul."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.kernel_bench import timeit
import medical_sam2_amd.ops as ops
shapes = [(16384, 1536, 384), (16384, 384, 1536), (16384, 1152, 384), (16384, 384, 384), (16384, 256, 2048), (16384, 2048, 256),
          (16384, 256, 256), (16384, 768, 256), (262144, 576, 96), (262144, 288, 96), (262144, 384, 96), (262144, 96, 384),
          (65536, 768, 192), (65536, 576, 192), (65536, 1152, 192), (4096, 3072, 768), (4096, 768, 3072), (4096, 2304, 768),
          (8192, 8192, 8192)]
variants = sys.argv[1:] or ["", "2", "5", "6", "7", "8", "9"]
print(f"{'M':>7s} {'N':>5s} {'K':>5s} " + " ".join(f"{'v' + (v or 'dflt'):>14s}" for v in variants))
for M, N, K in shapes:
    a = torch.randn(M, K, device="cuda", dtype=ops.OP16)
    w = (torch.randn(N, K, device="cuda") * 0.05).to(ops.OP16)
    bias = torch.randn(N, device="cuda")
    out = torch.empty(M, N, device="cuda", dtype=ops.OP16)
    ref = (a[:512].float() @ w.float().t() + bias)
    cells = []
    for v in variants:
        if v: os.environ["MSAM2_GEMM_VARIANT"] = v
        else: os.environ.pop("MSAM2_GEMM_VARIANT", None)
        out.zero_()
        ops.gemm(a, w, bias, out=out)
        err = ((out[:512].float() - ref).abs().max() / ref.abs().max()).item()
        tail = (out[-256:].float() - (a[-256:].float() @ w.float().t() + bias)).abs().max().item() / ref.abs().max().item()
        t = timeit(lambda: ops.gemm(a, w, bias, out=out), n=30)
        cells.append(f"{t*1e6:6.1f}us {2*M*N*K/t/1e12:4.0f}TF" + ("" if max(err, tail) < 2e-3 else f" ERR{max(err,tail):.1e}"))
    print(f"{M:7d} {N:5d} {K:5d} " + " ".join(f"{c:>14s}" for c in cells), flush=True)
