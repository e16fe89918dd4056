"""Probe: decoder gradient of the prompt loss on the box-prompted slice 0 of the BPTT fixture: HIP vs oracle autograd on the HIP tape's inputs."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import sam2_oracle as O
import medical_sam2_amd.weights as wts
import medical_sam2_amd.training_3d as t3
from medical_sam2_amd.training import upsampled_mask_loss
import test_bptt_gpu as tb

m, G, meta, volume, prompts, targets = tb._case()
S = meta["image_size"]
with torch.no_grad():
    tape, low = t3.volume_forward_saved(m, volume, prompts)
for t in (0, 3):
    fr = tape["frames"][t]
    n, h, w = tape["n"], fr["h"], fr["w"]
    with torch.no_grad():
        l_t, d_t = upsampled_mask_loss(low[t], targets[t], 0, meta["pos_weight"])
        g_hip = t3.volume_backward(m, tape, {t: d_t})["decoder"]
    sd = wts.init_weights("hiera_t", 0)
    pre = "sam_mask_decoder"
    P = {k: v.clone().float().requires_grad_(k.startswith(pre)) for k, v in sd.items()}
    C = 256
    dense = sd["sam_prompt_encoder.no_mask_embed.weight"].float().reshape(1, C, 1, 1)
    src = fr["src"].float().cpu().view(n, h * w, C).permute(0, 2, 1).reshape(n, C, h, w)
    emb = src - dense
    pe = fr["pe"].float().cpu().view(1, h * w, C).permute(0, 2, 1).reshape(1, C, h, w)
    f0 = fr["f0"].float().cpu().view(n, 4 * h, 4 * w, -1).permute(0, 3, 1, 2)
    f1 = fr["f1"].float().cpu().view(n, 2 * h, 2 * w, -1).permute(0, 3, 1, 2)
    with torch.enable_grad():
        masks, iou, toks, obj = O.mask_decoder_predict(P, emb, pe, fr["sparse"].float().cpu(), dense.expand(n, C, h, w), [f0, f1])
        sel = fr["mask_sel"].cpu()
        lowm = masks[torch.arange(n), sel][:, None]
        up = torch.nn.functional.interpolate(lowm, size=(S, S), mode="bilinear", align_corners=False)
        loss = torch.nn.BCEWithLogitsLoss(pos_weight=torch.ones(1) * meta["pos_weight"])(up, targets[t].cpu())
        loss.backward()
    print(f"slice {t}: sparse tokens {fr['sparse'].shape[1]}, loss hip {float(l_t):.5f} oracle {float(loss):.5f}, low-res |d| max {float((lowm.detach() - low[t].float().cpu()).abs().max()):.4f}")
    errs = {}
    for k, g in g_hip.items():
        ref = P[f"{pre}.{k}"].grad
        if ref is None or ref.abs().sum() == 0 or k.endswith("k_proj.bias") or ".q_proj." in k or ".k_proj." in k:
            continue
        errs[k] = float((g.cpu().double() - ref.double()).norm() / ref.double().norm())
    w_ = sorted(errs.items(), key=lambda kv: -kv[1])
    print("   HIP vs oracle-on-HIP-inputs: median", round(float(np.median(list(errs.values()))), 4), "worst", [(k[-40:], round(v, 3)) for k, v in w_[:4]])
    errs2 = {}
    for k in G:
        if k.startswith(f"prompt_t{t}.sam_mask_decoder."):
            pn = k.split(".", 2)[2]
            if pn.endswith("k_proj.bias") or ".q_proj." in pn or ".k_proj." in pn:
                continue
            a = torch.from_numpy(tb.sub((P[f"{pre}.{pn}"].grad / len(prompts)).float(), 256)).double(); b = torch.from_numpy(np.asarray(G[k])).double()
            errs2[pn] = float((a - b).norm() / b.norm())
    w2 = sorted(errs2.items(), key=lambda kv: -kv[1])
    print("   oracle-on-HIP-inputs vs reference fixture: median", round(float(np.median(list(errs2.values()))), 4), "worst", [(k[-40:], round(v, 3)) for k, v in w2[:4]])

# sparse prompt embeddings and image embedding of slice 0: HIP tape vs oracle / reference-equivalent fp32 path
from medical_sam2_amd.volume import box_point_inputs
sd = wts.init_weights("hiera_t", 0)
P = {k: v.clone().float() for k, v in sd.items()}
cfg = O.model_config("hiera_t", S)
pin = box_point_inputs(prompts[0]["boxes"].cpu())
se_ref, _ = O.prompt_encoder(P, cfg, (pin["point_coords"], pin["point_labels"]), None, None)
se_hip = tape["frames"][0]["sparse"].float().cpu()
print("slice 0 sparse: shapes", tuple(se_hip.shape), tuple(se_ref.shape), "rel err per token", [round(float((se_hip[:, i] - se_ref[:, i]).norm() / se_ref[:, i].norm().clamp_min(1e-9)), 5) for i in range(se_ref.shape[1])])
print("box coords", pin["point_coords"].tolist(), pin["point_labels"].tolist())

# are the tape's saved feature tensors of slice 0 still slice 0's (or were they overwritten by a later forward_image)?
from medical_sam2_amd.modeling.common import to_bf16, tokens_of
with torch.no_grad():
    for t in (0, 3, 1):
        bo = m.forward_image(volume[t][None])
        bo = {"backbone_fpn": [f.expand(n, -1, -1, -1) for f in bo["backbone_fpn"]], "vision_pos_enc": [q.expand(n, -1, -1, -1) for q in bo["vision_pos_enc"]]}
        _, feats, pos, sizes = m._prepare_backbone_features(bo)
        hr = [f.permute(1, 2, 0).view(n, -1, *s) for f, s in zip(feats[:-1], sizes[:-1])]
        f0, f1 = to_bf16(tokens_of(hr[0])).clone(), to_bf16(tokens_of(hr[1])).clone()
        fr = tape["frames"][t]
        print(f"slice {t}: saved f0 == fresh {bool(torch.equal(fr['f0'], f0))}  f1 == fresh {bool(torch.equal(fr['f1'], f1))}  top == fresh {bool(torch.equal(fr['top'], feats[-1]))}")

# every decoder input of slices 0 and 3: HIP tape vs the oracle's fp32 values
with torch.no_grad():
    for t in (0, 3):
        fr = tape["frames"][t]
        h, w = fr["h"], fr["w"]
        bo = O.forward_image(P, cfg, volume[t][None].cpu())
        bo = {"backbone_fpn": [f.expand(n, -1, -1, -1) for f in bo["backbone_fpn"]], "vision_pos_enc": [q.expand(n, -1, -1, -1) for q in bo["vision_pos_enc"]]}
        feats, pos, sizes = O.prepare_backbone_features(bo)
        pix = O.prepare_memory_conditioned_features(P, cfg, t, True, feats[-1], pos[-1], sizes[-1], {"cond_frame_outputs": {}, "non_cond_frame_outputs": {}}, 5)
        dense = sd["sam_prompt_encoder.no_mask_embed.weight"].float().reshape(1, 256, 1, 1)
        src_ref = (pix + dense).flatten(2).permute(0, 2, 1).reshape(n * h * w, 256)
        rel = lambda a, b: float((a.float().cpu() - b).norm() / b.norm())
        hr = [x.permute(1, 2, 0).reshape(x.shape[1], x.shape[2], *s) for x, s in zip(feats[:-1], sizes[:-1])]
        f0_ref = hr[0].permute(0, 2, 3, 1).reshape(-1, hr[0].shape[1]); f1_ref = hr[1].permute(0, 2, 3, 1).reshape(-1, hr[1].shape[1])
        pe_ref = O.dense_pe(P, h, w).flatten(2).permute(0, 2, 1).reshape(h * w, 256)
        print(f"slice {t}: src {rel(fr['src'], src_ref):.5f}  f0 {rel(fr['f0'], f0_ref):.5f}  f1 {rel(fr['f1'], f1_ref):.5f}  pe {rel(fr['pe'], pe_ref):.6f}")

# which input moves the slice-0 gradient?  oracle autograd with the HIP tape's inputs swapped for the fp32 ones, one at a time
def dec_grads(emb, pe4, sparse, dense4, f0_, f1_, t, sel):
    Pg = {k: v.clone().float().requires_grad_(k.startswith("sam_mask_decoder")) for k, v in sd.items()}
    with torch.enable_grad():
        masks, _, _, _ = O.mask_decoder_predict(Pg, emb, pe4, sparse, dense4, [f0_, f1_])
        lowm = masks[torch.arange(n), sel][:, None]
        up = torch.nn.functional.interpolate(lowm, size=(S, S), mode="bilinear", align_corners=False)
        loss = torch.nn.BCEWithLogitsLoss(pos_weight=torch.ones(1) * meta["pos_weight"])(up, targets[t].cpu())
        loss.backward()
    return float(loss), {k: v.grad for k, v in Pg.items() if v.grad is not None and v.grad.abs().sum() > 0 and ".q_proj." not in k and ".k_proj." not in k}

def gdiff(a, b):
    return float(np.median([float((a[k] - b[k]).norm() / b[k].norm()) for k in b]))

for t in (0,):
    fr = tape["frames"][t]
    h, w, C = fr["h"], fr["w"], 256
    with torch.no_grad():
        bo = O.forward_image(P, cfg, volume[t][None].cpu())
        bo = {"backbone_fpn": [f.expand(n, -1, -1, -1) for f in bo["backbone_fpn"]], "vision_pos_enc": [q.expand(n, -1, -1, -1) for q in bo["vision_pos_enc"]]}
        feats, pos, sizes = O.prepare_backbone_features(bo)
        pix_ref = O.prepare_memory_conditioned_features(P, cfg, t, True, feats[-1], pos[-1], sizes[-1], {"cond_frame_outputs": {}, "non_cond_frame_outputs": {}}, 5)
        hr = [x.permute(1, 2, 0).reshape(x.shape[1], x.shape[2], *s) for x, s in zip(feats[:-1], sizes[:-1])]
    dense4 = sd["sam_prompt_encoder.no_mask_embed.weight"].float().reshape(1, C, 1, 1).expand(n, C, h, w)
    src_h = fr["src"].float().cpu().view(n, h * w, C).permute(0, 2, 1).reshape(n, C, h, w) - dense4
    pe_h = fr["pe"].float().cpu().view(1, h * w, C).permute(0, 2, 1).reshape(1, C, h, w)
    f0_h = fr["f0"].float().cpu().view(n, 4 * h, 4 * w, -1).permute(0, 3, 1, 2)
    f1_h = fr["f1"].float().cpu().view(n, 2 * h, 2 * w, -1).permute(0, 3, 1, 2)
    sp_h = fr["sparse"].float().cpu()
    sel = fr["mask_sel"].cpu()
    l_ref, g_ref = dec_grads(pix_ref, O.dense_pe(P, h, w), sp_h, dense4, hr[0], hr[1], t, sel)
    l_hip, g_hipin = dec_grads(src_h, pe_h, sp_h, dense4, f0_h, f1_h, t, sel)
    print(f"slice {t}: loss fp32 inputs {l_ref:.5f}, HIP inputs {l_hip:.5f}; gradient median rel diff {gdiff(g_hipin, g_ref):.4f}")
    for name, args in (("src", (src_h, O.dense_pe(P, h, w), sp_h, dense4, hr[0], hr[1])), ("pe", (pix_ref, pe_h, sp_h, dense4, hr[0], hr[1])),
                       ("f0", (pix_ref, O.dense_pe(P, h, w), sp_h, dense4, f0_h, hr[1])), ("f1", (pix_ref, O.dense_pe(P, h, w), sp_h, dense4, hr[0], f1_h))):
        l_, g_ = dec_grads(*args, t, sel)
        print(f"   only {name} from the HIP tape: loss {l_:.5f}, gradient median rel diff {gdiff(g_, g_ref):.4f}")
    d = src_h - pix_ref
    dm = d.mean(dim=(2, 3), keepdim=True)
    print("   |d|/|src|", float(d.norm() / pix_ref.norm()), " token-mean part", float(dm.expand_as(d).norm() / pix_ref.norm()), " residual", float((d - dm).norm() / pix_ref.norm()),
          " |no_mem_embed|/|src| per token", float(sd["no_mem_embed"].float().norm() / (pix_ref.norm() / (n * h * w) ** 0.5)))
    for name, src_x in (("token-mean part only", pix_ref + dm), ("residual only", pix_ref + d - dm)):
        l_, g_ = dec_grads(src_x, O.dense_pe(P, h, w), sp_h, dense4, hr[0], hr[1], t, sel)
        print(f"   src = fp32 + {name}: loss {l_:.5f}, gradient median rel diff {gdiff(g_, g_ref):.4f}")
