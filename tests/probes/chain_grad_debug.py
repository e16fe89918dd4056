"""Where does the gradient error of the memory-attention -> decoder chain come from (debug aid)."""
import os, sys, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import sam2_oracle as O
import medical_sam2_amd.build_sam as bs, medical_sam2_amd.weights as wts, medical_sam2_amd.training as T, medical_sam2_amd.ops as ops
import medical_sam2_amd.backward as bwd
DEV = "cuda"
rnd = lambda *s, seed=0, scale=1.0: torch.randn(*s, generator=torch.Generator().manual_seed(seed)) * scale
rel = lambda a, b: ((a.detach().cpu().double() - b.detach().cpu().double()).norm() / b.detach().cpu().double().norm()).item()
m = bs.build_sam2("sam2_hiera_t", device="cpu", hydra_overrides_extra=["++model.image_size=256"])
sd = wts.init_weights("hiera_t", 0)
m.load_state_dict(sd, strict=True)
mod, dec = m.memory_attention.to(DEV).eval(), m.sam_mask_decoder.to(DEV).eval()
cfg = O.model_config("hiera_t", 256)
P = {k: v.clone().float().requires_grad_(k.startswith("memory_attention.") or k.startswith("sam_mask_decoder.")) for k, v in sd.items()}
B, E, C, n_ptr = 2, 16, 256, 4
L, Nk = E * E, E * E + 4
q16 = lambda t: t.to(ops.OP16).float()
curr, curr_pos = rnd(L, B, C, seed=140).requires_grad_(True), rnd(L, B, C, seed=141)
memory, memory_pos = rnd(Nk, B, 64, seed=142), rnd(Nk, B, 64, seed=143)
pe, sparse, dense = rnd(1, C, E, E, seed=144), rnd(B, 2, C, seed=145), rnd(1, C, seed=146, scale=0.3)
f0, f1 = q16(rnd(B, 32, 4 * E, 4 * E, seed=147)), q16(rnd(B, 64, 2 * E, 2 * E, seed=148))
target = (rnd(B, 4, 4 * E, 4 * E, seed=149) > 0.4).float()
y = O.memory_attention(P, cfg, curr, memory, curr_pos, memory_pos, n_ptr)
y.retain_grad()
emb = y.permute(1, 2, 0).reshape(B, C, E, E)
masks, _, _, _ = O.mask_decoder_predict(P, emb, pe, sparse, dense.view(1, C, 1, 1).expand(B, C, E, E), [f0, f1])
loss = F.binary_cross_entropy_with_logits(masks, target)
loss.backward()
d = lambda t: t.detach().to(DEV)
tm = lambda t: d(t).permute(0, 2, 3, 1).reshape(-1, t.shape[1]).contiguous()
with torch.no_grad():
    # 1. HIP memory-attention backward fed with the EXACT upstream gradient
    for s in (1.0, 2.0 ** 14):
        dcurr, _, _, g = bwd.memory_attention_backward(mod, d(curr), d(curr_pos), d(memory), d(memory_pos), n_ptr, d(y.grad) * s)
        num = sum((g[k].cpu().double() / s - P["memory_attention." + k].grad.double()).pow(2).sum().item() for k in g if not k.endswith("k_proj.bias"))
        den = sum(P["memory_attention." + k].grad.double().pow(2).sum().item() for k in g if not k.endswith("k_proj.bias"))
        print(f"memattn bwd with exact dy x {s:g}: group rel {(num / den) ** 0.5:.4f}, dcurr rel {rel(dcurr / s, curr.grad):.4f}; max|dy| {y.grad.abs().max().item():.3e}")
    # 2. the decoder's d_src against autograd's dL/dy
    ysrc = d(y).transpose(0, 1).reshape(B * L, C)
    src = ops.add_cast(ysrc.view(B, L, C), d(dense).view(1, 1, C).expand(B, L, C), 1.0, torch.float32).view(B * L, C)
    mk, _, _, _ = dec.predict_masks_tokens(src, tm(pe), d(sparse), tm(f0).to(ops.OP16), tm(f1).to(ops.OP16), B, E, E)
    print("masks rel", rel(mk, masks))
    l, dm = T.bce_with_logits(mk, d(target), 1.0)
    for s in (1.0, 2.0 ** 11):
        d_src, _, _ = bwd.mask_decoder_backward(dec, src, tm(pe), d(sparse), tm(f0).to(ops.OP16), tm(f1).to(ops.OP16), B, E, E, dm * s)
        print(f"decoder d_src (x{s:g}) rel vs autograd dL/dy: {rel(d_src.view(B, L, C).transpose(0, 1) / s, y.grad):.4f}")
    args = (d(curr), d(curr_pos), d(memory), d(memory_pos), n_ptr, tm(pe), d(sparse), tm(f0).to(ops.OP16), tm(f1).to(ops.OP16), B, E, E, d(target))
    for ms in (None, 1.0):
        loss2, sc, scm, g_dec, g_mem, dcurr = T.memory_decoder_loss_grads(mod, dec, *args, dense_tokens=d(dense), mem_scale=ms)
        num = sum((g_mem[k].cpu().double() / scm - P["memory_attention." + k].grad.double()).pow(2).sum().item() for k in g_mem if not k.endswith("k_proj.bias"))
        den = sum(P["memory_attention." + k].grad.double().pow(2).sum().item() for k in g_mem if not k.endswith("k_proj.bias"))
        worst = sorted(((rel(g_mem[k] / scm, P["memory_attention." + k].grad), k) for k in g_mem if not k.endswith("k_proj.bias")), reverse=True)[:4]
        print(f"chain mem_scale={ms}: scale {sc} scale_mem {scm} group rel {(num / den) ** 0.5:.4f} dcurr rel {rel(dcurr / scm, curr.grad):.4f} worst {worst}")
    yp, st = bwd.memory_attention_forward_saved(mod, d(curr), d(curr_pos), d(memory), d(memory_pos), n_ptr)
    yf = mod(curr=[d(curr)], curr_pos=[d(curr_pos)], memory=d(memory), memory_pos=d(memory_pos), num_obj_ptr_tokens=n_ptr)
    print("forward_saved y vs oracle:", rel(yp, y), " module forward vs oracle:", rel(yf, y), " x_last vs module:", yp.shape, yf.shape)
    def grp(g, s):
        num = sum((g[k].cpu().double() / s - P["memory_attention." + k].grad.double()).pow(2).sum().item() for k in g if not k.endswith("k_proj.bias"))
        den = sum(P["memory_attention." + k].grad.double().pow(2).sum().item() for k in g if not k.endswith("k_proj.bias"))
        return (num / den) ** 0.5
    # A: state, then decoder fwd+bwd in between, then backward with the EXACT dy
    yp, st = bwd.memory_attention_forward_saved(mod, d(curr), d(curr_pos), d(memory), d(memory_pos), n_ptr)
    srcp = ops.add_cast(yp.transpose(0, 1).reshape(B * L, C).view(B, L, C), d(dense).view(1, 1, C).expand(B, L, C), 1.0, torch.float32).view(B * L, C)
    mk, _, _, _ = dec.predict_masks_tokens(srcp, tm(pe), d(sparse), tm(f0).to(ops.OP16), tm(f1).to(ops.OP16), B, E, E)
    l, dm = T.bce_with_logits(mk, d(target), 1.0)
    d_src, _, _ = bwd.mask_decoder_backward(dec, srcp, tm(pe), d(sparse), tm(f0).to(ops.OP16), tm(f1).to(ops.OP16), B, E, E, dm * 2048)
    print("A: d_src (product y) vs autograd:", rel(d_src.view(B, L, C).transpose(0, 1) / 2048, y.grad))
    _, _, _, gA = bwd.memory_attention_backward_saved(mod, st, d(y.grad) * 2048)
    print("A: backward_saved after the decoder ran, exact dy:", grp(gA, 2048))
    yp, st = bwd.memory_attention_forward_saved(mod, d(curr), d(curr_pos), d(memory), d(memory_pos), n_ptr)
    _, _, _, gB = bwd.memory_attention_backward_saved(mod, st, d_src.view(B, L, C).transpose(0, 1))
    print("B: fresh state, the decoder's d_src:", grp(gB, 2048))
    def dsrc_of(yy, tag):
        sr = ops.add_cast(yy.transpose(0, 1).reshape(B * L, C).view(B, L, C), d(dense).view(1, 1, C).expand(B, L, C), 1.0, torch.float32).view(B * L, C)
        mk, _, _, _ = dec.predict_masks_tokens(sr, tm(pe), d(sparse), tm(f0).to(ops.OP16), tm(f1).to(ops.OP16), B, E, E)
        l, dm = T.bce_with_logits(mk, d(target), 1.0)
        ds, _, _ = bwd.mask_decoder_backward(dec, sr, tm(pe), d(sparse), tm(f0).to(ops.OP16), tm(f1).to(ops.OP16), B, E, E, dm * 2048)
        print(f"{tag}: masks rel {rel(mk, masks):.5f}  d_src rel {rel(ds.view(B, L, C).transpose(0, 1) / 2048, y.grad):.4f}")
    yo = d(y)
    dsrc_of(yo, "oracle y")
    dsrc_of(yp, "product y (view)")
    dsrc_of(yp.contiguous().clone(), "product y (copy)")
    n = torch.randn_like(yo); n = n * (5e-4 * yo.norm() / n.norm())
    dsrc_of(yo + n, "oracle y + 5e-4 noise")
    dsrc_of(yo, "oracle y again")
with torch.enable_grad():
    yc = yp.detach().cpu().float().contiguous().requires_grad_(True)
    e2 = yc.permute(1, 2, 0).reshape(B, C, E, E)
    P2 = {k: v.detach() for k, v in P.items()}
    mk2, _, _, _ = O.mask_decoder_predict(P2, e2, pe, sparse, dense.view(1, C, 1, 1).expand(B, C, E, E), [f0, f1])
    l2 = F.binary_cross_entropy_with_logits(mk2, target)
    g_at_yp = torch.autograd.grad(l2, yc)[0]
print("autograd dL/dy at the product's y vs at the oracle's y:", rel(g_at_yp, y.grad), " |yp - y|/|y| =", rel(yp, y))
with torch.no_grad():
    sr = ops.add_cast(yp.transpose(0, 1).reshape(B * L, C).view(B, L, C), d(dense).view(1, 1, C).expand(B, L, C), 1.0, torch.float32).view(B * L, C)
    mk, _, _, _ = dec.predict_masks_tokens(sr, tm(pe), d(sparse), tm(f0).to(ops.OP16), tm(f1).to(ops.OP16), B, E, E)
    l, dm = T.bce_with_logits(mk, d(target), 1.0)
    ds, _, _ = bwd.mask_decoder_backward(dec, sr, tm(pe), d(sparse), tm(f0).to(ops.OP16), tm(f1).to(ops.OP16), B, E, E, dm * 2048)
    print("HIP d_src at the product's y vs autograd AT THAT y:", rel(ds.view(B, L, C).transpose(0, 1) / 2048, g_at_yp))
