#!/usr/bin/env python3
"""Calibration only (never on the product path): what the vendor GEMM (hipBLASLt via torch.matmul) reaches on the step's hot shapes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.kernel_bench import timeit
import medical_sam2_amd.ops as ops
shapes = [(16384, 1536, 384), (16384, 384, 1536), (16384, 1152, 384), (16384, 384, 384), (16384, 256, 2048), (16384, 2048, 256),
          (262144, 576, 96), (65536, 192, 768), (65536, 768, 192), (16384, 256, 256), (4096, 3072, 768), (8192, 8192, 8192)]
for M, N, K in shapes:
    a = torch.randn(M, K, device="cuda", dtype=torch.float16)
    w = torch.randn(N, K, device="cuda", dtype=torch.float16) * 0.05
    out = torch.empty(M, N, device="cuda", dtype=torch.float16)
    t = timeit(lambda: torch.matmul(a, w.t(), out=out), n=30)
    t2 = timeit(lambda: ops.gemm(a, w, out=out), n=30)
    print(f"{M:7d} {N:5d} {K:5d}  blas {t*1e6:8.1f} us {2*M*N*K/t/1e12:7.1f} TF   ours {t2*1e6:8.1f} us {2*M*N*K/t2/1e12:7.1f} TF")
