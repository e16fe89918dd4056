"""TIGHT parity of the loaded library -- fp16 operands by default, bf16 when this file runs in the bf16 child suite -- against the oracle
with the SAME operand rounding emulated (`oracle.operand_rounding(dtype)`: every matrix-product operand rounded to the 16-bit type,
fp32 accumulation, fp32 everything else).  VERDICT r3 item 2: against the reference's fp32 goldens the bf16 bars have to admit bf16's
own rounding (features 2e-2, flips 0.5 % of the map, chained gradients "median < 0.3") and would pass a real kernel bug of a few per cent;
with the rounding on both sides what is left is the ORDER of roundings (where an intermediate is stored in 16 bits, folded weights,
running-maximum softmax) -- the bound is the same few 1e-3 for both operand types and a kernel bug can no longer hide in it.

Slices after the first are TEACHER-FORCED: the oracle's `track_step` for slice t runs on the HIP chain's own memory bank (its outputs up
to t-1), so the binarisation of a border pixel upstream (sam2_base.py:686-688, a step function) is on both sides or on neither.

STATED TOLERANCE (asserted below; measured values in gpurun_out/operand_rounding_report_<dtype>.json):
  FPN features relative L2 <= FEAT; memory-conditioned features <= FEAT; mask logits max |d| <= LOGIT_MAX, mean |d| <= LOGIT_MEAN;
  low-res mask pixels on the other side of 0: <= FLIPS per slice, and every flipped pixel has |logit| <= LOGIT_MAX in the oracle;
  object pointer / new memory relative L2 <= PTR."""
import json
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import medical_sam2_amd.synthetic as syn  # noqa: E402
import medical_sam2_amd.weights as wts  # noqa: E402
from helpers import max_abs, rel_err  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda"
# one set of bars for BOTH operand types (the point of the emulation)
FEAT, PTR, LOGIT_MAX, LOGIT_MEAN, FLIPS = 2.5e-3, 4e-3, 0.04, 0.006, 4
REPORT = {}


def _dump(dtype):
    os.makedirs("gpurun_out", exist_ok=True)
    with open(f"gpurun_out/operand_rounding_report_{str(dtype).split('.')[-1]}.json", "w") as f:
        json.dump(REPORT, f, indent=1)


def _chain(model, S, T, weights_seed, image_seed, tag, flips_bar=FLIPS):
    import medical_sam2_amd.build_sam as bs
    import medical_sam2_amd.ops as ops
    from oracle import sam2_oracle as O
    m = bs.build_sam2("sam2_" + model, device="cpu", hydra_overrides_extra=[f"++model.image_size={S}", "++model.binarize_mask_from_pts_for_mem_enc=true"])
    W = wts.init_weights(model, weights_seed)
    m.load_state_dict(W, strict=True)
    m = m.to(DEV).eval()
    cfg = O.model_config(model, S)
    od = {"cond_frame_outputs": {}, "non_cond_frame_outputs": {}}
    c = lambda t: t.detach().float().cpu()
    pick = lambda o: {"maskmem_features": c(o["maskmem_features"]), "maskmem_pos_enc": [c(o["maskmem_pos_enc"][0])], "obj_ptr": c(o["obj_ptr"])}
    worst = dict(fpn=0.0, mem_feat=0.0, max=0.0, mean=0.0, flips=0, flip_logit=0.0, ptr=0.0, mem=0.0)
    with torch.no_grad():
        for t in range(T):
            img, pts, labels = syn.image_batch([image_seed + t], S)
            bo = m.forward_image(img.to(DEV))
            _, feats, pos, sizes = m._prepare_backbone_features(bo)
            pin = {"point_coords": pts.to(DEV), "point_labels": labels.to(DEV)} if t == 0 else None
            seen = {}
            real = m.memory_attention.forward
            if t > 0:
                def spy(*a, **kw):
                    seen["out"] = real(*a, **kw)
                    return seen["out"]
                m.memory_attention.forward = spy
            try:
                cur = m.track_step(frame_idx=t, is_init_cond_frame=(t == 0), current_vision_feats=feats, current_vision_pos_embeds=pos,
                                   feat_sizes=sizes, point_inputs=pin, mask_inputs=None, output_dict=od, num_frames=T)
            finally:
                m.memory_attention.forward = real
            # ---- the oracle on the same inputs, its matrix products on operands of the library's type; slice t > 0 on the HIP bank
            with O.operand_rounding(ops.OP16):
                bo_o = O.forward_image(W, cfg, img)
                f_o, p_o, s_o = O.prepare_backbone_features(bo_o)
                od_o = {"cond_frame_outputs": {u: pick(o) for u, o in od["cond_frame_outputs"].items()},
                        "non_cond_frame_outputs": {u: pick(o) for u, o in od["non_cond_frame_outputs"].items()}}
                col = {}
                ref = O.track_step(W, cfg, t, t == 0, f_o, p_o, s_o, None if pin is None else {"point_coords": pts, "point_labels": labels},
                                   None, od_o, T, collect=col)
            (od["cond_frame_outputs"] if t == 0 else od["non_cond_frame_outputs"])[t] = cur
            rep = {f"fpn{l}": rel_err(c(bo["backbone_fpn"][l]), bo_o["backbone_fpn"][l]) for l in range(3)}
            if t > 0:
                got_mem = c(seen["out"])                                            # [HW, B, C] memory-conditioned features
                rep["mem_feat"] = rel_err(got_mem, col["pix_feat_with_mem"].flatten(2).permute(2, 0, 1))
            g, r = c(cur["pred_masks"]).numpy(), ref["pred_masks"].numpy()
            flipped = (g > 0) != (r > 0)
            rep.update(max_abs=max_abs(g, r), mean_abs=float(np.abs(g - r).mean()), flips=int(flipped.sum()),
                       flip_logit=float(np.abs(r[flipped]).max()) if flipped.any() else 0.0, fg=int((r > 0).sum()),
                       ptr=rel_err(c(cur["obj_ptr"]), ref["obj_ptr"]), mem=rel_err(c(cur["maskmem_features"]), ref["maskmem_features"]))
            REPORT[f"{tag}_t{t}"] = rep
            worst = dict(fpn=max(worst["fpn"], *(rep[f"fpn{l}"] for l in range(3))), mem_feat=max(worst["mem_feat"], rep.get("mem_feat", 0.0)),
                         max=max(worst["max"], rep["max_abs"]), mean=max(worst["mean"], rep["mean_abs"]), flips=max(worst["flips"], rep["flips"]),
                         flip_logit=max(worst["flip_logit"], rep["flip_logit"]), ptr=max(worst["ptr"], rep["ptr"]), mem=max(worst["mem"], rep["mem"]))
    REPORT[f"{tag}_worst"] = worst
    _dump(ops.OP16)
    print(tag, str(ops.OP16), worst)
    assert worst["fpn"] <= FEAT and worst["mem_feat"] <= FEAT, worst
    assert worst["max"] <= LOGIT_MAX and worst["mean"] <= LOGIT_MEAN, worst
    assert worst["flips"] <= flips_bar and worst["flip_logit"] <= LOGIT_MAX, worst
    assert worst["ptr"] <= PTR and worst["mem"] <= PTR, worst


def test_chain_hiera_s_1024_vs_operand_rounding_oracle():
    """BASELINE configs[1]'s model and size: a prompted slice and two propagated ones (the inputs of chain_hiera_s_1024.npz)"""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    _chain("hiera_s", 1024, 3, 0, 10, "s1024", flips_bar=2 * FLIPS)       # 16x the pixels of the 256^2 cases


def test_chain_hiera_t_256_vs_operand_rounding_oracle():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    _chain("hiera_t", 256, 3, 0, 10, "t256")


def test_chain_hiera_bplus_256_vs_operand_rounding_oracle():
    """configs[4]'s model (head dim 56 zero-padded to 64; weight seed of chain_hiera_bplus_256.npz: real masks)"""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from helpers import load_meta
    meta = load_meta()["chain_hiera_bplus_256"]
    _chain("hiera_b+", 256, 3, meta.get("weights_seed", 0), meta.get("image_seed_base", 10), "b256")
