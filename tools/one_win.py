"""One Hiera windowed-attention launch x5 at the benchmark's shape (the PMC passes in profiles/ run on it).
usage: one_win.py [stage3|block0]      (B = 4, hiera_s: stage-3 blocks 64^2 tokens, dim 384, 4 heads, 14x14 windows; block 0 256^2 x 96, 8x8)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import medical_sam2_amd.ops as ops
case = sys.argv[1] if len(sys.argv) > 1 else "stage3"
hw, dim, heads, ws = {"stage3": (64, 384, 4, 14), "block0": (256, 96, 1, 8)}[case]
B = 4
g = torch.Generator().manual_seed(0)
T = B * hw * hw
qkv = (torch.randn(T, 3 * dim, generator=g) * 0.5).to(ops.OP16).cuda()
bias = torch.randn(3 * dim, generator=g).cuda()
for _ in range(5):
    ops.window_attention(qkv, B, hw, hw, heads, ws, bias)
torch.cuda.synchronize()
