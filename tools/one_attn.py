"""One attention launch x5 (the PMC passes in profiles/ run on it).
usage: one_attn.py B H Lq Lk D splits     (D = 96 / 128 / 256: msam2_attention_fwd; D = kv64: msam2_attention_kv64_fwd)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import medical_sam2_amd.ops as ops
B, H, Lq, Lk = [int(x) for x in sys.argv[1:5]]
kv64 = sys.argv[5] == "kv64"
D, sp = (256 if kv64 else int(sys.argv[5])), int(sys.argv[6])
g = torch.Generator().manual_seed(0)
q = torch.randn(B, H, Lq, D, generator=g).to(ops.OP16).cuda()
k = torch.randn(B, H, Lk, D, generator=g).to(ops.OP16).cuda()
v = torch.randn(B, H, Lk, 64 if kv64 else D, generator=g).to(ops.OP16).cuda()
for _ in range(5):
    (ops.attention_kv64 if kv64 else ops.attention)(q, k, v, splits=sp)
torch.cuda.synchronize()
