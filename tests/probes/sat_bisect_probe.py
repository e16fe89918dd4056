"""Probe (GPU): dumps the intermediate tensors of test_memory_decoder_loss_grads for the loaded library build, so that two builds can be
compared bit for bit offline (python tests/probes/sat_bisect_probe.py <tag>; MSAM2_LIB_PATH selects the build)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import medical_sam2_amd.backward as B_
import medical_sam2_amd.build_sam as bs
import medical_sam2_amd.ops as ops
import medical_sam2_amd.training as T
import medical_sam2_amd.weights as wts
torch.set_grad_enabled(False)
DEV = "cuda"
rnd = lambda *s, seed=0, scale=1.0: torch.randn(*s, generator=torch.Generator().manual_seed(seed)) * scale
sd = wts.init_weights("hiera_t", 0)
m = bs.build_sam2("sam2_hiera_t", device="cpu", hydra_overrides_extra=["++model.image_size=256"])
m.load_state_dict(sd, strict=True)
m = m.to(DEV).eval()
mod, dec = m.memory_attention, m.sam_mask_decoder
Bn, E, C, n_ptr = 2, 16, 256, 4
L, Nk = E * E, E * E + 4
q16 = lambda t: t.to(ops.OP16).float()
curr, curr_pos = rnd(L, Bn, C, seed=140), rnd(L, Bn, C, seed=141)
memory, memory_pos = rnd(Nk, Bn, 64, seed=142), rnd(Nk, Bn, 64, seed=143)
pe, sparse, dense = rnd(1, C, E, E, seed=144), rnd(Bn, 2, C, seed=145), rnd(1, C, seed=146, scale=0.3)
f0, f1 = q16(rnd(Bn, 32, 4 * E, 4 * E, seed=147)), q16(rnd(Bn, 64, 2 * E, 2 * E, seed=148))
target = (rnd(Bn, 4, 4 * E, 4 * E, seed=149) > 0.4).float()
d = lambda t: t.detach().to(DEV)
tm = lambda t: d(t).permute(0, 2, 3, 1).reshape(-1, t.shape[1]).contiguous()
out = {}
for rep in range(2):
    y_hip, _ = B_.memory_attention_forward_saved(mod, d(curr), d(curr_pos), d(memory), d(memory_pos), n_ptr)
    out[f"y_hip{rep}"] = y_hip.float().cpu()
    args = (d(curr), d(curr_pos), d(memory), d(memory_pos), n_ptr, tm(pe), d(sparse), tm(f0).to(ops.OP16), tm(f1).to(ops.OP16), Bn, E, E, d(target))
    aux = {}
    loss, scale, scale_mem, g_dec, g_mem, dcurr = T.memory_decoder_loss_grads(mod, dec, *args, dense_tokens=d(dense), aux=aux)
    out[f"loss{rep}"] = loss.cpu(); out[f"scales{rep}"] = torch.tensor([scale, scale_mem])
    out[f"masks{rep}"] = aux["masks"].cpu(); out[f"dcurr{rep}"] = dcurr.float().cpu()
    out[f"d_feat_s1_{rep}"] = aux["d_feat_s1"].float().cpu()
    for k in ("layers.3.linear2.weight", "layers.0.self_attn.q_proj.weight", "layers.2.cross_attn_image.k_proj.weight", "norm.weight"):
        out[f"gmem.{k}.{rep}"] = g_mem[k].cpu()
    for k in ("transformer.layers.0.self_attn.q_proj.weight", "output_upscaling.0.weight"):
        out[f"gdec.{k}.{rep}"] = g_dec[k].cpu()
torch.save(out, os.path.join(ROOT, "gpurun_out", f"sat_probe_{sys.argv[1]}.pt"))
print("saved", sys.argv[1], {k: float(v.double().abs().sum()) for k, v in out.items() if k.endswith("0") or k.endswith(".0")})
