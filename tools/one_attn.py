import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import medical_sam2_amd.ops as ops
B, H, Lq, Lk, D, sp = [int(x) for x in sys.argv[1:7]]
g = torch.Generator().manual_seed(0)
q = torch.randn(B, H, Lq, D, generator=g).to(ops.OP16).cuda()
k = torch.randn(B, H, Lk, D, generator=g).to(ops.OP16).cuda()
v = torch.randn(B, H, Lk, D, generator=g).to(ops.OP16).cuda()
for _ in range(5):
    ops.attention(q, k, v, splits=sp)
torch.cuda.synchronize()
