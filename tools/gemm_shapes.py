#!/usr/bin/env python3
"""Experiment: output-pattern sensitivity of the GEMM (same bytes, different N)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.kernel_bench import timeit
import medical_sam2_amd.ops as ops
for M, N, K, od in [(1179648, 128, 96, ops.OP16), (589824, 256, 96, ops.OP16), (262144, 576, 96, ops.OP16), (131072, 1152, 96, ops.OP16),
                    (262144, 512, 96, ops.OP16), (262144, 640, 96, ops.OP16), (262144, 576, 96, torch.float32), (262144, 96, 96, torch.float32), (786432, 96, 96, ops.OP16)]:
    a = torch.randn(M, K, device="cuda", dtype=ops.OP16)
    w = (torch.randn(N, K, device="cuda") * 0.05).to(ops.OP16)
    out = torch.empty(M, N, device="cuda", dtype=od)
    t = timeit(lambda: ops.gemm(a, w, None, out=out), n=20)
    by = M * K * 2 + M * N * out.element_size()
    print(f"{M:8d} {N:5d} {K:4d} {str(od)[6:]:8s} {t*1e6:8.1f} us  {by/t/1e12:5.2f} TB/s  {2*M*N*K/t/1e12:6.1f} TF", flush=True)
