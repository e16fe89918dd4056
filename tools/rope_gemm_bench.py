"""The RoPE-fused projections of the memory attention (msam2_gemm_rope) at the step's shapes, stand-alone (30 launches per graph replay),
next to the same GEMM without the rotation.  A/B between library builds through MSAM2_LIB_PATH."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import medical_sam2_amd.ops as ops
from tools.win_attn_bench import timeit
g = torch.Generator().manual_seed(0)
tab = ops.rope_table(64, 256, 10000.0, "cuda")
# (M, N, K, rope_cols, rows_per_batch, n_rope): self-attention q|k|v, cross-attention q, the bank's keys (4 x 4096 per slice)
for M, N, K, rc, rpb, nr in ((16384, 768, 256, 512, 4096, 4096), (16384, 256, 256, 256, 4096, 4096), (65536, 256, 64, 256, 16384, 16384)):
    a = torch.randn(M, K, generator=g).to(ops.OP16).cuda()
    w = (torch.randn(N, K, generator=g) * 0.05).to(ops.OP16).cuda()
    b = torch.randn(N, generator=g).cuda()
    out = torch.empty(M, N, dtype=ops.OP16, device="cuda")
    t_r = timeit(lambda: ops.gemm_rope(a, w, b, tab, rope_cols=rc, head_dim=256, rows_per_batch=rpb, n_rope=nr, out=out), n=30)
    t_p = timeit(lambda: ops.gemm(a, w, b, out=out), n=30)
    by = 2.0 * (M * K + M * N)
    print(f"  {M} x {N} x {K}: with RoPE {t_r * 1e6:6.1f} us ({by / t_r / 1e12:4.2f} TB/s of A + C)   plain {t_p * 1e6:6.1f} us", flush=True)
