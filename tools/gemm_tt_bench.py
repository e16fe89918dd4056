"""gemm_tt (weight-gradient GEMM on k-major operands) at the training iteration's shapes (dW [out, in] = dY^T X over the tokens of 4 slices
at 1024^2: Hiera stages 1-4, memory attention, mask decoder); HIP-event timing of back-to-back launches.
Split policy via MSAM2_TT_MINK (k-tiles per split, at least) / MSAM2_TT_WGS (workgroups aimed at); MSAM2_GEMM_TT_V1=1: the register-staged kernel."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import medical_sam2_amd.backward as B, medical_sam2_amd.ops as ops
# (tokens K, out M, in N, count per iteration)
shapes = [(262144, 288, 96, 1), (262144, 96, 96, 1), (262144, 384, 96, 1), (262144, 96, 384, 1), (65536, 576, 192, 2), (65536, 192, 192, 2),
          (65536, 768, 192, 2), (65536, 192, 768, 2), (16384, 1152, 384, 11), (16384, 384, 384, 11), (16384, 1536, 384, 11), (16384, 384, 1536, 11),
          (4096, 2304, 768, 2), (4096, 768, 768, 2), (4096, 3072, 768, 2), (4096, 768, 3072, 2),
          (16384, 256, 256, 20), (16384, 2048, 256, 4), (16384, 256, 2048, 4), (65536, 256, 64, 8)]
tot = 0.0
fl = 0.0
for K, M, N, cnt in shapes:
    a = torch.randn(K, M, device="cuda").to(ops.OP16); b = torch.randn(K, N, device="cuda").to(ops.OP16)
    for _ in range(3): B.gemm_tt(a, b, a_colsum=True)
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): B.gemm_tt(a, b, a_colsum=True)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 20 * 1e-3)
    dt = sorted(ts)[1]
    tot += dt * cnt; fl += 2.0 * K * M * N * cnt
    print(f"K={K:6d} M={M:5d} N={N:5d} x{cnt:2d}: {dt * 1e6:7.1f} us  {2.0 * K * M * N / dt / 1e12:6.1f} TFLOP/s", flush=True)
print(f"per iteration: {tot * 1e3:.2f} ms, {fl / tot / 1e12:.0f} TFLOP/s")
