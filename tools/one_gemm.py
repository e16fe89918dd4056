import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import medical_sam2_amd.ops as ops
M, N, K = [int(x) for x in sys.argv[1:4]]
g = torch.Generator().manual_seed(0)
a = torch.randn(M, K, generator=g).to(ops.OP16).cuda()
w = (torch.randn(N, K, generator=g) * 0.05).to(ops.OP16).cuda()
b = torch.randn(N, generator=g).cuda()
out = torch.empty(M, N, dtype=ops.OP16, device="cuda")
for _ in range(10):
    ops.gemm(a, w, b, out=out)
torch.cuda.synchronize()
