import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.kernel_bench import timeit
import medical_sam2_amd.ops as ops
for M, N, K, f32 in [(65536, 192, 768, True), (65536, 192, 192, True), (262144, 192, 96, True), (65536, 192, 384, False)]:
    a = torch.randn(M, K, device="cuda", dtype=ops.OP16); w = (torch.randn(N, K, device="cuda") * 0.05).to(ops.OP16)
    out = torch.empty(M, N, device="cuda", dtype=torch.float32 if f32 else ops.OP16); res = torch.randn(M, N, device="cuda") if f32 else None
    for env in (None, "1"):
        if env: os.environ["MSAM2_GEMM_DMA_ANY_N"] = "1"
        else: os.environ.pop("MSAM2_GEMM_DMA_ANY_N", None)
        t = timeit(lambda: ops.gemm(a, w, None, residual=res, out=out), n=20)
        print(M, N, K, "dma-any" if env else "default", f"{t*1e6:.1f} us")
