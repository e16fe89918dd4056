"""The pooled GEMMs of Hiera's q-pool blocks (msam2_gemm_qkv_pool2x2, msam2_gemm_pool2x2) at the step's shapes, stand-alone graph replays.
A/B between library builds through MSAM2_LIB_PATH."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import medical_sam2_amd.ops as ops
from tools.win_attn_bench import timeit
g = torch.Generator().manual_seed(0)
B = 4
for HW, K, width in ((256, 96, 192), (128, 192, 384), (64, 384, 768)):
    M = B * HW * HW
    a = torch.randn(M, K, generator=g).to(ops.OP16).cuda()
    w = (torch.randn(3 * width, K, generator=g) * 0.05).to(ops.OP16).cuda()
    b = torch.randn(3 * width, generator=g).cuda()
    t = timeit(lambda: ops.gemm_qkv_pool2x2(a, w, b, B, HW, HW, width), n=20)
    wp = (torch.randn(width, K, generator=g) * 0.05).to(ops.OP16).cuda()
    bp = torch.randn(width, generator=g).cuda()
    t2 = timeit(lambda: ops.gemm_pool2x2(a, wp, bp, B, HW, HW), n=20)
    print(f"  qkv_pool2x2 {M} x {3 * width} x {K}: {t * 1e6:6.1f} us    pool2x2 {M} x {width} x {K}: {t2 * 1e6:6.1f} us", flush=True)
