#!/usr/bin/env python3
"""Calibration: plain fill / copy bandwidth on buffers of the GEMM output sizes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.kernel_bench import timeit
for shape, dt in [((262144, 576), torch.float16), ((262144, 576), torch.float32), ((262144, 96), torch.float32), ((16384, 1536), torch.float16)]:
    x = torch.randn(*shape, device="cuda").to(dt); y = torch.empty_like(x)
    nb = x.numel() * x.element_size()
    t = timeit(lambda: y.fill_(1.0), n=20); print(shape, dt, f"fill {t*1e6:7.1f} us {nb/t/1e12:5.2f} TB/s", end="  ")
    t = timeit(lambda: y.copy_(x), n=20); print(f"copy {t*1e6:7.1f} us {2*nb/t/1e12:5.2f} TB/s (r+w)")
