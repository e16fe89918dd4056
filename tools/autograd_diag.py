import os, sys, copy, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch.nn.functional as F
from test_bptt_gpu import GROUP_OF, _case, rel_sub
import medical_sam2_amd.training as tr, medical_sam2_amd.training_3d as t3
from medical_sam2_amd.volume import box_point_inputs
freeze = len(sys.argv) > 1 and sys.argv[1] == "freeze"
net, G, meta, volume, prompts, targets = _case()
if freeze:
    net.image_encoder.requires_grad_(False); net.sam_mask_decoder.conv_s0.requires_grad_(False); net.sam_mask_decoder.conv_s1.requires_grad_(False)
twin = copy.deepcopy(net)
T, S, n = meta["n_slices"], meta["image_size"], meta["n_objects"]
lossfunc = torch.nn.BCEWithLogitsLoss(pos_weight=torch.ones([1]).cuda() * meta["pos_weight"])
od = {"cond_frame_outputs": {}, "non_cond_frame_outputs": {}}
seg = {}
for t in sorted(prompts) + [t for t in range(T) if t not in prompts]:
    bo = net.forward_image(volume[t][None])
    bo = {"backbone_fpn": [f.expand(n, -1, -1, -1) for f in bo["backbone_fpn"]], "vision_pos_enc": [q.expand(n, -1, -1, -1) for q in bo["vision_pos_enc"]]}
    _, feats, pos, sizes = net._prepare_backbone_features(bo)
    pin = None
    if t in prompts:
        pr = prompts[t]
        pin = box_point_inputs(pr["boxes"]) if "boxes" in pr else {"point_coords": pr["point_coords"], "point_labels": pr["point_labels"]}
    cur = net.track_step(frame_idx=t, is_init_cond_frame=t in prompts, current_vision_feats=feats, current_vision_pos_embeds=pos, feat_sizes=sizes,
                         point_inputs=pin, mask_inputs=None, output_dict=od, num_frames=T)
    od["cond_frame_outputs" if t in prompts else "non_cond_frame_outputs"][t] = cur
    seg[t] = F.interpolate(cur["pred_masks"], size=(S, S), mode="bilinear", align_corners=False)
npl = 0
for t in range(T):
    if t not in prompts:
        for o in range(n):
            npl = npl + lossfunc(seg[t][o][None], targets[t][o][None])
npl = npl / (T - len(prompts)) / n
npl.backward(retain_graph=True)
named = dict(net.named_parameters())
out = {}
with torch.no_grad():
    opts = {"decoder": tr.DecoderAdam(twin.sam_mask_decoder, lr=0.0)}
    t3.train_step_3d(twin, opts, volume, prompts, targets, pos_weight=meta["pos_weight"], grads_out=out)
    tape, low = t3.volume_forward_saved(twin, volume, prompts)
for t in range(T):
    print("slice", t, "low-res logits: autograd path vs explicit path max|d|", float((od["cond_frame_outputs" if t in prompts else "non_cond_frame_outputs"][t]["pred_masks"].detach() - low[t]).abs().max()))
for t in range(T):
    a = od["cond_frame_outputs" if t in prompts else "non_cond_frame_outputs"][t]
    b = tape["frames"][t]["out"]
    r = lambda x, y: float((x.detach().float() - y.float()).norm() / y.float().norm())
    print("slice", t, "maskmem rel", r(a["maskmem_features"], b["maskmem_features"]), "obj_ptr rel", r(a["obj_ptr"], b["obj_ptr"]),
          "high-res max|d|", float((a["pred_masks_high_res"].detach() - b["pred_masks_high_res"]).abs().max()))
prefix = {v: k for k, v in GROUP_OF.items()}
for grp, gd in out["non_prompt"].items():
    errs = []
    for pname, g in gd.items():
        if pname.startswith("conv_s"): continue
        got = named[f"{prefix[grp]}.{pname}"].grad
        errs.append((float((got.double()-g.double()).norm()/g.double().norm().clamp_min(1e-30)), pname, float(g.norm())))
    errs.sort(reverse=True)
    print(grp, "worst", [(round(e,4), p, round(nn,6)) for e,p,nn in errs[:4]], "median", errs[len(errs)//2][0])
fx = {}
for k in G:
    if k.startswith("non_prompt."):
        _, gname, pname = k.split(".", 2)
        if pname.endswith("k_proj.bias") or meta["grad_stats"][k]["abs_sum"] < 1e-7: continue
        e_a = rel_sub(named[f"{gname}.{pname}"].grad, G[k])
        e_e = rel_sub(out["non_prompt"][GROUP_OF[gname]][pname], G[k])
        fx[gname+"."+pname] = (e_a, e_e)
w = sorted(fx.items(), key=lambda kv: -kv[1][0])[:6]
print("vs fixture (autograd, explicit):", [(k, round(a,4), round(b,4)) for k,(a,b) in w])
