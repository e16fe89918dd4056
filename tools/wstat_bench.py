#!/usr/bin/env python3
"""W-stationary GEMM (gemm_wstat_kernel) against the tiled kernels at the shapes it serves: correctness (bit-equal integers, max error vs
the default path on random data) and time.  GPU only."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import medical_sam2_amd.ops as ops  # noqa: E402
from tools.kernel_bench import timeit  # noqa: E402

dev = "cuda"
g = torch.Generator().manual_seed(0)
shapes = [(16384, 1536, 384, 1), (16384, 1152, 384, 0), (16384, 2048, 256, 2), (16384, 1024, 256, 1), (16384, 384, 384, 0), (65536, 768, 256, 0),
          (8192, 1536, 384, 1), (32768, 1152, 384, 0)]
for M, N, K, act in shapes:
    a = torch.randn(M, K, generator=g).to(ops.OP16).to(dev)
    w = (torch.randn(N, K, generator=g) * 0.05).to(ops.OP16).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    out0, out1 = torch.empty(M, N, dtype=ops.OP16, device=dev), torch.empty(M, N, dtype=ops.OP16, device=dev)
    os.environ["MSAM2_GEMM_WSTAT"] = "0"
    ops.gemm(a, w, b, act=act, out=out0)
    t0 = timeit(lambda: ops.gemm(a, w, b, act=act, out=out0), n=30)
    os.environ["MSAM2_GEMM_WSTAT"] = "1"
    ops.gemm(a, w, b, act=act, out=out1)
    t1 = timeit(lambda: ops.gemm(a, w, b, act=act, out=out1), n=30)
    torch.cuda.synchronize()
    d = (out0.float() - out1.float()).abs().max().item()
    # integers: exact
    ai = torch.randint(-3, 4, (M, K), generator=g).to(ops.OP16).to(dev)
    wi = torch.randint(-3, 4, (N, K), generator=g).to(ops.OP16).to(dev)
    ref = (ai.float() @ wi.float().t())
    got = ops.gemm(ai, wi, None, act=0, out_dtype=ops.OP16).float()
    exact = bool(torch.equal(got, ref.to(ops.OP16).float()))
    fl = 2.0 * M * N * K
    print(f"M={M} N={N} K={K} act={act}: tiled {t0 * 1e6:7.1f} us ({fl / t0 / 1e12:6.1f} TF/s)  w-stationary {t1 * 1e6:7.1f} us ({fl / t1 / 1e12:6.1f} TF/s)  "
          f"x{t0 / t1:.2f}  max|d| {d:.4g}  integers exact: {exact}", flush=True)
