"""gemm_tt (weight-gradient GEMM on k-major operands) at the training step's shapes; split policy via MSAM2_TT_MINK / MSAM2_TT_WGS."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import medical_sam2_amd.backward as B, medical_sam2_amd.ops as ops
shapes = [(16384, 256, 256), (16384, 768, 256), (16384, 2048, 256), (16384, 256, 2048), (65536, 256, 64), (65536, 128, 256), (262144, 32, 128), (28, 256, 256)]
tot = 0.0
for K, M, N in shapes:
    a = torch.randn(K, M, device="cuda").to(ops.OP16); b = torch.randn(K, N, device="cuda").to(ops.OP16)
    for _ in range(3): B.gemm_tt(a, b, a_colsum=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): B.gemm_tt(a, b, a_colsum=True)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    tot += dt
    print(f"K={K:6d} M={M:5d} N={N:5d}: {dt * 1e6:7.1f} us  {2.0 * K * M * N / dt / 1e12:6.1f} TFLOP/s")
print(f"sum {tot * 1e6:.1f} us")
