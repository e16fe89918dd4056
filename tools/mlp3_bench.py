import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.kernel_bench import timeit
import medical_sam2_amd.ops as ops
for G, B in [(6, 4), (6, 1), (1, 4), (1, 1)]:
    hs = torch.randn(B, 9, 256, device="cuda")
    tok = torch.arange(G, dtype=torch.int32, device="cuda")
    w = [torch.randn(G, 256, 256, device="cuda").to(ops.OP16) for _ in range(3)]
    b = [torch.randn(G, 256, device="cuda") for _ in range(3)]
    od = torch.full((G,), 32, dtype=torch.int32, device="cuda"); sg = torch.zeros(G, dtype=torch.int32, device="cuda")
    t = timeit(lambda: ops.token_mlp3(hs, tok, w[0], b[0], w[1], b[1], w[2], b[2], od, sg), n=50)
    print(G, B, f"{t*1e6:.1f} us")
