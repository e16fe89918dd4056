"""Gradient errors of the backward pass with the bf16 operand build (MSAM2_LIB_PATH=.../libmsam2_hip_bf16.so) against fp32 autograd."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import sam2_oracle as O
import medical_sam2_amd.ops as ops, medical_sam2_amd.backward as bwd
print("operand type:", ops.OP16)
rnd = lambda *s, seed=0: torch.randn(*s, generator=torch.Generator().manual_seed(seed))
rel = lambda a, b: ((a.detach().cpu().double() - b.detach().cpu().double()).norm() / b.detach().cpu().double().norm()).item()
for B, H, Lq, Lk, D in ((1, 1, 1024, 4096, 256), (2, 2, 256, 300, 96), (2, 3, 33, 300, 128)):
    with torch.enable_grad():
        q16 = lambda t: t.to(ops.OP16).float()
        q = q16(rnd(B, H, Lq, D, seed=1)).requires_grad_(True); k = q16(rnd(B, H, Lk, D, seed=2)).requires_grad_(True)
        v = q16(rnd(B, H, Lk, D, seed=3)).requires_grad_(True); do = q16(rnd(B, H, Lq, D, seed=4))
        O.softmax_attention(q, k, v).backward(do)
    d = lambda t: t.detach().to(ops.OP16).cuda()
    with torch.no_grad():
        dq, dk, dv = bwd.attention_backward(d(q), d(k), d(v), do.cuda())
    print(f"attention backward B={B} H={H} Lq={Lq} Lk={Lk} D={D}: dq {rel(dq, q.grad):.2e} dk {rel(dk, k.grad):.2e} dv {rel(dv, v.grad):.2e}")
