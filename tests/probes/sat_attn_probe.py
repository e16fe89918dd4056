"""Probe (GPU): attention / GEMM / LayerNorm outputs of the loaded library build on fixed inputs, saved for an offline bitwise comparison."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import medical_sam2_amd.ops as ops
import medical_sam2_amd.backward as B_
torch.set_grad_enabled(False)
rnd = lambda *s, seed=0, scale=1.0: (torch.randn(*s, generator=torch.Generator().manual_seed(seed)) * scale)
out = {}
for name, (B, H, Lq, Lk, D) in {"d256_lse": (2, 1, 256, 260, 256), "d256_big": (1, 1, 512, 2100, 256), "d96": (1, 4, 256, 256, 96)}.items():
    q, k, v = (rnd(B, H, L, D, seed=s).to(ops.OP16).cuda() for L, s in ((Lq, 1), (Lk, 2), (Lk, 3)))
    out[name + "_plain"] = ops.attention(q, k, v).float().cpu()
    out[name + "_split3"] = ops.attention(q, k, v, splits=3).float().cpu()
    o, lse = B_.attention_forward_lse(q, k, v)
    out[name + "_lse_o"], out[name + "_lse"] = o.float().cpu(), lse.cpu()
x = rnd(300, 256, seed=5).cuda()
out["ln"] = ops.layernorm(x, torch.ones(256, device="cuda"), torch.zeros(256, device="cuda"), 1e-5).float().cpu()
a, w = rnd(300, 256, seed=6).to(ops.OP16).cuda(), rnd(768, 256, seed=7, scale=0.06).to(ops.OP16).cuda()
out["gemm16"] = ops.gemm(a, w, None).float().cpu()
out["gemm32"] = ops.gemm(a, w, None, out_dtype=torch.float32).cpu()
torch.save(out, os.path.join(ROOT, "gpurun_out", f"sat_attn_{sys.argv[1]}.pt"))
print("saved")
