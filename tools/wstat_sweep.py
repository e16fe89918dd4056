import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import medical_sam2_amd.ops as ops
from tools.kernel_bench import timeit
g = torch.Generator().manual_seed(0)
for (M, N, K, act) in [(16384, 1152, 384, 0), (16384, 1536, 384, 1), (16384, 1536, 384, 0), (16384, 2048, 256, 2), (16384, 1024, 256, 1), (16384, 768, 256, 0), (16384, 384, 384, 0),
                       (8192, 1152, 384, 0), (8192, 1536, 384, 1), (32768, 1152, 384, 0), (32768, 1536, 384, 1), (65536, 768, 256, 0)]:
    a = torch.randn(M, K, generator=g).to(ops.OP16).cuda(); w = (torch.randn(N, K, generator=g) * 0.05).to(ops.OP16).cuda(); b = torch.randn(N, generator=g).cuda()
    out = torch.empty(M, N, dtype=ops.OP16, device="cuda")
    res = []
    ref = None
    for mode in ("0", "1", "2"):
        os.environ["MSAM2_GEMM_WSTAT"] = mode
        for nt in ("1000000000000", "0"):
            os.environ["MSAM2_NT_BYTES"] = nt
            res.append(timeit(lambda: ops.gemm(a, w, b, act=act, out=out), n=20) * 1e6)
        torch.cuda.synchronize()
        if ref is None:
            ref = out.clone()
        else:
            assert torch.equal(ref, out), (mode, M, N, K, act, float((ref.float() - out.float()).abs().max()))
    print(f"M={M} N={N} K={K} act={act}: tiled {res[0]:.1f} (nt {res[1]:.1f})  wstat {res[2]:.1f} (nt {res[3]:.1f})  wstat+apf {res[4]:.1f} (nt {res[5]:.1f}) us", flush=True)
