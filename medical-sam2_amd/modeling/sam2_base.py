"""Drop-in SAM2Base (sam2_train/modeling/sam2_base.py): same constructor, attributes, state-dict keys and method
signatures/returns (forward_image, _prepare_backbone_features, _prepare_memory_conditioned_features, _forward_sam_heads,
_use_mask_as_output, _encode_new_memory, track_step), with every tensor computation on the MI355X kernels.

`image_size` is honoured as given (the fork's hard-coded 256 at sam2_base.py:159-160 is the special case
image_size=256; SURVEY.md shim S1)."""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from .. import ops
from .common import OP16, F32, WeightCache, nchw_view, to_bf16, tokens_of, v_f32, w_bf16
from .encoder import MLP
from .sam_heads import MaskDecoder, PromptEncoder, TwoWayTransformer

NO_OBJ_SCORE = -1024.0


def select_closest_cond_frames(frame_idx, cond_frame_outputs, max_cond_frame_num):
    """sam2_utils.py:15-57: cond frames attended by frame_idx (all of them when max_cond_frame_num == -1)."""
    if max_cond_frame_num == -1 or len(cond_frame_outputs) <= max_cond_frame_num:
        return cond_frame_outputs, {}
    assert max_cond_frame_num >= 2, "we should allow using 2+ conditioning frames"
    chosen = {}
    before = max((t for t in cond_frame_outputs if t < frame_idx), default=None)
    if before is not None:
        chosen[before] = cond_frame_outputs[before]
    after = min((t for t in cond_frame_outputs if t >= frame_idx), default=None)
    if after is not None:
        chosen[after] = cond_frame_outputs[after]
    rest = sorted((t for t in cond_frame_outputs if t not in chosen), key=lambda t: abs(t - frame_idx))
    for t in rest[: max_cond_frame_num - len(chosen)]:
        chosen[t] = cond_frame_outputs[t]
    return chosen, {t: v for t, v in cond_frame_outputs.items() if t not in chosen}


class SAM2Base(nn.Module):
    def __init__(self, image_encoder, memory_attention, memory_encoder, num_maskmem=7, image_size=512, backbone_stride=16,
                 sigmoid_scale_for_mem_enc=1.0, sigmoid_bias_for_mem_enc=0.0, binarize_mask_from_pts_for_mem_enc=False,
                 use_mask_input_as_output_without_sam=False, max_cond_frames_in_attn=-1, directly_add_no_mem_embed=False,
                 use_high_res_features_in_sam=False, multimask_output_in_sam=False, multimask_min_pt_num=1, multimask_max_pt_num=1,
                 multimask_output_for_tracking=False, use_multimask_token_for_obj_ptr: bool = False, iou_prediction_use_sigmoid=False,
                 memory_temporal_stride_for_eval=1, add_all_frames_to_correct_as_cond=False, non_overlap_masks_for_mem_enc=False,
                 use_obj_ptrs_in_encoder=False, max_obj_ptrs_in_encoder=16, add_tpos_enc_to_obj_ptrs=True,
                 proj_tpos_enc_in_obj_ptrs=False, only_obj_ptrs_in_the_past_for_eval=False, pred_obj_scores: bool = False,
                 pred_obj_scores_mlp: bool = False, fixed_no_obj_ptr: bool = False, soft_no_obj_ptr: bool = False,
                 use_mlp_for_obj_ptr_proj: bool = False, sam_mask_decoder_extra_args=None, compile_image_encoder: bool = False):
        super().__init__()
        assert use_high_res_features_in_sam and use_obj_ptrs_in_encoder and pred_obj_scores and fixed_no_obj_ptr and \
            use_mlp_for_obj_ptr_proj and directly_add_no_mem_embed and not add_tpos_enc_to_obj_ptrs and not soft_no_obj_ptr and \
            not proj_tpos_enc_in_obj_ptrs, \
            "HIP path implements the model wiring of sam2_hiera_{t,s}.yaml"
        self.image_encoder = image_encoder
        self.use_high_res_features_in_sam = use_high_res_features_in_sam
        self.num_feature_levels = 3
        self.use_obj_ptrs_in_encoder = use_obj_ptrs_in_encoder
        self.max_obj_ptrs_in_encoder = max_obj_ptrs_in_encoder
        self.mask_downsample = nn.Conv2d(1, 1, kernel_size=4, stride=4)
        self.add_tpos_enc_to_obj_ptrs = add_tpos_enc_to_obj_ptrs
        self.proj_tpos_enc_in_obj_ptrs = proj_tpos_enc_in_obj_ptrs
        self.only_obj_ptrs_in_the_past_for_eval = only_obj_ptrs_in_the_past_for_eval
        self.memory_attention = memory_attention
        self.hidden_dim = memory_attention.d_model
        self.memory_encoder = memory_encoder
        self.mem_dim = self.hidden_dim
        if hasattr(self.memory_encoder, "out_proj") and hasattr(self.memory_encoder.out_proj, "weight"):
            self.mem_dim = self.memory_encoder.out_proj.weight.shape[0]
        self.num_maskmem = num_maskmem
        self.maskmem_tpos_enc = nn.Parameter(torch.zeros(num_maskmem, 1, 1, self.mem_dim))
        nn.init.trunc_normal_(self.maskmem_tpos_enc, std=0.02)
        self.no_mem_embed = nn.Parameter(torch.zeros(1, 1, self.hidden_dim))
        self.no_mem_pos_enc = nn.Parameter(torch.zeros(1, 1, self.hidden_dim))
        nn.init.trunc_normal_(self.no_mem_embed, std=0.02)
        nn.init.trunc_normal_(self.no_mem_pos_enc, std=0.02)
        self.directly_add_no_mem_embed = directly_add_no_mem_embed
        self.sigmoid_scale_for_mem_enc = sigmoid_scale_for_mem_enc
        self.sigmoid_bias_for_mem_enc = sigmoid_bias_for_mem_enc
        self.binarize_mask_from_pts_for_mem_enc = binarize_mask_from_pts_for_mem_enc
        self.non_overlap_masks_for_mem_enc = non_overlap_masks_for_mem_enc
        self.memory_temporal_stride_for_eval = memory_temporal_stride_for_eval
        self.use_mask_input_as_output_without_sam = use_mask_input_as_output_without_sam
        self.multimask_output_in_sam = multimask_output_in_sam
        self.multimask_min_pt_num = multimask_min_pt_num
        self.multimask_max_pt_num = multimask_max_pt_num
        self.multimask_output_for_tracking = multimask_output_for_tracking
        self.use_multimask_token_for_obj_ptr = use_multimask_token_for_obj_ptr
        self.iou_prediction_use_sigmoid = iou_prediction_use_sigmoid
        self.image_size = image_size
        self.backbone_stride = backbone_stride
        self.sam_mask_decoder_extra_args = sam_mask_decoder_extra_args
        self.pred_obj_scores = pred_obj_scores
        self.pred_obj_scores_mlp = pred_obj_scores_mlp
        self.fixed_no_obj_ptr = fixed_no_obj_ptr
        self.soft_no_obj_ptr = soft_no_obj_ptr
        self.no_obj_ptr = nn.Parameter(torch.zeros(1, self.hidden_dim))
        nn.init.trunc_normal_(self.no_obj_ptr, std=0.02)
        self.use_mlp_for_obj_ptr_proj = use_mlp_for_obj_ptr_proj
        self._build_sam_heads()
        self.add_all_frames_to_correct_as_cond = add_all_frames_to_correct_as_cond
        self.max_cond_frames_in_attn = max_cond_frames_in_attn
        self._wc = WeightCache()

    @property
    def device(self):
        return next(self.parameters()).device

    def forward(self, *args, **kwargs):
        raise NotImplementedError("Please use the corresponding methods in SAM2VideoPredictor for inference.")

    def _build_sam_heads(self):
        """sam2_base.py:202-250."""
        self.sam_prompt_embed_dim = self.hidden_dim
        self.sam_image_embedding_size = self.image_size // self.backbone_stride
        E = self.sam_image_embedding_size
        self.sam_prompt_encoder = PromptEncoder(embed_dim=self.sam_prompt_embed_dim, image_embedding_size=(E, E),
                                                input_image_size=(self.image_size, self.image_size), mask_in_chans=16)
        self.sam_mask_decoder = MaskDecoder(
            num_multimask_outputs=3,
            transformer=TwoWayTransformer(depth=2, embedding_dim=self.sam_prompt_embed_dim, mlp_dim=2048, num_heads=8),
            transformer_dim=self.sam_prompt_embed_dim, iou_head_depth=3, iou_head_hidden_dim=256,
            use_high_res_features=self.use_high_res_features_in_sam, iou_prediction_use_sigmoid=self.iou_prediction_use_sigmoid,
            pred_obj_scores=self.pred_obj_scores, pred_obj_scores_mlp=self.pred_obj_scores_mlp,
            use_multimask_token_for_obj_ptr=self.use_multimask_token_for_obj_ptr, **(self.sam_mask_decoder_extra_args or {}))
        self.obj_ptr_proj = MLP(self.hidden_dim, self.hidden_dim, self.hidden_dim, 3)
        self.obj_ptr_tpos_proj = nn.Identity()

    # ---------------------------------------------------------------------------------------------------------------
    def forward_image(self, img_batch: torch.Tensor):
        """sam2_base.py:464-476.  train() + grad mode: the grad-carrying form (autograd.py)."""
        from .. import autograd as ag
        if ag.active(self):
            return ag.forward_image(self, img_batch)
        if not self.use_high_res_features_in_sam:
            return self.image_encoder(img_batch)
        # conv_s0 / conv_s1 (sam2_base.py:470-475) are folded into the neck's lateral convs of levels 0 / 1 (FpnNeck.forward)
        dec = self.sam_mask_decoder
        backbone_out = self.image_encoder(img_batch, post_convs={0: dec.conv_s0, 1: dec.conv_s1})
        return backbone_out

    def _prepare_backbone_features(self, backbone_out):
        """sam2_base.py:478-492 (views only)."""
        backbone_out = backbone_out.copy()
        assert len(backbone_out["backbone_fpn"]) == len(backbone_out["vision_pos_enc"]) >= self.num_feature_levels
        feature_maps = backbone_out["backbone_fpn"][-self.num_feature_levels:]
        vision_pos_embeds = backbone_out["vision_pos_enc"][-self.num_feature_levels:]
        feat_sizes = [(x.shape[-2], x.shape[-1]) for x in vision_pos_embeds]
        vision_feats = [x.flatten(2).permute(2, 0, 1) for x in feature_maps]
        vision_pos_embeds = [x.flatten(2).permute(2, 0, 1) for x in vision_pos_embeds]
        return backbone_out, vision_feats, vision_pos_embeds, feat_sizes

    # ---------------------------------------------------------------------------------------------------------------
    def _forward_sam_heads(self, backbone_features, point_inputs=None, mask_inputs=None, high_res_features=None,
                           multimask_output=False):
        """sam2_base.py:252-410.  Returns the same 7-tuple; the multimask high-res maps are up-sampled from all candidates
        (as the reference does) and the best one is a view into them."""
        from .. import autograd as ag
        if ag.active(self):
            return ag.forward_sam_heads(self, backbone_features, point_inputs, mask_inputs, high_res_features, multimask_output)
        B = backbone_features.size(0)
        device = backbone_features.device
        E = self.sam_image_embedding_size
        assert backbone_features.size(1) == self.sam_prompt_embed_dim and backbone_features.size(2) == E and backbone_features.size(3) == E
        if point_inputs is not None:
            sam_point_coords, sam_point_labels = point_inputs["point_coords"], point_inputs["point_labels"]
            assert sam_point_coords.size(0) == B and sam_point_labels.size(0) == B
        else:
            sam_point_coords = torch.zeros(B, 1, 2, device=device)
            sam_point_labels = -torch.ones(B, 1, dtype=torch.int32, device=device)
        if mask_inputs is not None:
            assert len(mask_inputs.shape) == 4 and mask_inputs.shape[:2] == (B, 1)
            if tuple(mask_inputs.shape[-2:]) != tuple(self.sam_prompt_encoder.mask_input_size):
                f = mask_inputs.shape[-1] // self.sam_prompt_encoder.mask_input_size[-1]
                assert f >= 1 and mask_inputs.shape[-1] == f * self.sam_prompt_encoder.mask_input_size[-1], \
                    "mask prompts are down-sampled by an integer factor"
                sam_mask_prompt = ops.aa_downsample(mask_inputs.to(F32).contiguous(), f)
            else:
                sam_mask_prompt = mask_inputs
        else:
            sam_mask_prompt = None
        sparse, dense = self.sam_prompt_encoder(points=(sam_point_coords, sam_point_labels), boxes=None, masks=sam_mask_prompt)
        dec = self.sam_mask_decoder
        masks, ious, mask_tokens, obj = dec.predict_masks(backbone_features, self.sam_prompt_encoder.get_dense_pe(), sparse, dense,
                                                          False, None, high_res_features)
        dyn = dec.dynamic_multimask_via_stability and not self.training
        low_res_masks, sel, iou_sel = ops.select_mask(masks, ious, obj.reshape(B).contiguous(), multimask_output, dyn,
                                                      dec.dynamic_multimask_stability_delta, dec.dynamic_multimask_stability_thresh)
        S = self.image_size
        if multimask_output:
            low_res_multimasks = masks[:, 1:].contiguous()
            ops.gate_no_obj_(low_res_multimasks, obj.reshape(B).contiguous(), NO_OBJ_SCORE)
            ious_out = ious[:, 1:]
        else:
            low_res_multimasks, ious_out = low_res_masks, iou_sel
        high_res_masks = ops.bilinear_upsample(low_res_masks, S, S)
        high_res_multimasks = ops.bilinear_upsample(low_res_multimasks, S, S) if multimask_output else high_res_masks
        if multimask_output and dec.use_multimask_token_for_obj_ptr:
            sam_output_token = ops.gather_rows(mask_tokens.contiguous(), sel)
        else:
            sam_output_token = ops.gather_rows(mask_tokens.contiguous(), None)
        obj_ptr = self.obj_ptr_proj.run_tokens(sam_output_token)
        ops.obj_ptr_mix_(obj_ptr, obj.reshape(B).contiguous(), v_f32(self._wc, "nop", self.no_obj_ptr))
        return low_res_multimasks, high_res_multimasks, ious_out, low_res_masks, high_res_masks, obj_ptr, obj

    def _use_mask_as_output(self, backbone_features, high_res_features, mask_inputs):
        """sam2_base.py:412-462."""
        B = mask_inputs.shape[0]
        mf = mask_inputs.to(F32).contiguous()
        # +-10 logits and their anti-aliased 1/4-scale version
        high_res_masks = _affine(mf, 20.0, -10.0)
        low_res_masks = ops.aa_downsample(mf, 4, 20.0, -10.0)
        ious = torch.ones(B, 1, device=mf.device, dtype=F32)
        S = mf.shape[-1]
        cols = ops.space_to_depth(mf.reshape(B * S * S, 1), B, S, S, 4)
        wm = self._wc.get("mdw", [self.mask_downsample.weight], lambda: self.mask_downsample.weight.detach().reshape(1, 16).to(OP16).contiguous())
        md = ops.gemm(cols, wm, v_f32(self._wc, "mdb", self.mask_downsample.bias), out_dtype=F32).reshape(B, 1, S // 4, S // 4)
        _, _, _, _, _, obj_ptr, _ = self._forward_sam_heads(backbone_features=backbone_features, mask_inputs=md,
                                                            high_res_features=high_res_features)
        lam = ops.any_positive(mf)  # [B,1] fp32 in {0,1}
        object_score_logits = _affine(lam, 20.0, -10.0)
        ops.obj_ptr_mix_(obj_ptr, object_score_logits.reshape(B).contiguous(), v_f32(self._wc, "nop", self.no_obj_ptr))
        return low_res_masks, high_res_masks, ious, low_res_masks, high_res_masks, obj_ptr, object_score_logits

    # ---------------------------------------------------------------------------------------------------------------
    def _select_memory(self, frame_idx, output_dict, num_frames, track_in_reverse=False):
        """The host half of sam2_base.py:494-663: which stored slices this one attends to.  Returns (spatial, ptrs): spatial =
        [(t_pos, stored output)] in key order (conditioning slices first, t_pos 0), ptrs = the object pointers in key order."""
        assert len(output_dict["cond_frame_outputs"]) > 0
        cond_outputs = output_dict["cond_frame_outputs"]
        selected, unselected = select_closest_cond_frames(frame_idx, cond_outputs, self.max_cond_frames_in_attn)
        t_pos_and_prevs = [(0, out) for out in selected.values()]
        r = self.memory_temporal_stride_for_eval
        for t_pos in range(1, self.num_maskmem):
            t_rel = self.num_maskmem - t_pos
            if t_rel == 1:
                prev_idx = frame_idx - t_rel if not track_in_reverse else frame_idx + t_rel
            elif not track_in_reverse:
                prev_idx = ((frame_idx - 2) // r) * r - (t_rel - 2) * r
            else:
                prev_idx = -(-(frame_idx + 2) // r) * r + (t_rel - 2) * r
            out = output_dict["non_cond_frame_outputs"].get(prev_idx, None)
            if out is None:
                out = unselected.get(prev_idx, None)
            t_pos_and_prevs.append((t_pos, out))
        spatial = [(t_pos, prev) for t_pos, prev in t_pos_and_prevs if prev is not None]
        # object pointers (past-only in eval; no temporal encoding: add_tpos_enc_to_obj_ptrs=False)
        max_ptrs = min(num_frames, self.max_obj_ptrs_in_encoder)
        if not self.training and self.only_obj_ptrs_in_the_past_for_eval:
            ptr_cond = {t: o for t, o in selected.items() if (t >= frame_idx if track_in_reverse else t <= frame_idx)}
        else:
            ptr_cond = selected
        ptrs = [o["obj_ptr"] for o in ptr_cond.values()]
        for t_diff in range(1, max_ptrs):
            t = frame_idx + t_diff if track_in_reverse else frame_idx - t_diff
            if t < 0 or (num_frames is not None and t >= num_frames):
                break
            out = output_dict["non_cond_frame_outputs"].get(t, unselected.get(t, None))
            if out is not None:
                ptrs.append(out["obj_ptr"])
        return spatial, ptrs

    def _assemble_memory(self, spatial, ptrs, B: int, H: int, W: int, device):
        """The bank as ONE [N_k, B, 64] buffer + its position encoding (sam2_base.py:565-638): spatial memories (+ their temporal
        encoding on the position side), then the object pointers as C // mem_dim tokens each, zero position encoding.  `ptrs`: a list
        of [B, C] pointers or the padded form (ptr_bank [capacity, B, C] fp32, key_count) -- see _prepare_memory_conditioned_features.
        Returns (memory, memory_pos, number of pointer tokens, key_count or None)."""
        C = self.hidden_dim
        split = C // self.mem_dim
        HW = H * W
        key_count = None
        if isinstance(ptrs, tuple):
            ptr_bank, key_count = ptrs
            assert ptr_bank.dim() == 3 and ptr_bank.shape[1] == B and ptr_bank.shape[2] == C and ptr_bank.dtype == F32
            n_ptr_tok = ptr_bank.shape[0] * split
        else:
            ptr_bank, n_ptr_tok = None, len(ptrs) * split
        n_sp = len(spatial) * HW
        Nk = n_sp + n_ptr_tok
        memory = torch.empty(Nk, B, self.mem_dim, dtype=F32, device=device)
        memory_pos = torch.zeros(Nk, B, self.mem_dim, dtype=F32, device=device) if n_ptr_tok else torch.empty(Nk, B, self.mem_dim, dtype=F32, device=device)
        for i, (t_pos, prev) in enumerate(spatial):
            feats = prev["maskmem_features"].to(device, non_blocking=True)          # [B, 64, H, W]
            enc = prev["maskmem_pos_enc"][-1].to(device)
            tpos = self.maskmem_tpos_enc[self.num_maskmem - t_pos - 1].detach().to(F32)  # [1,1,64]
            sl = slice(i * HW, (i + 1) * HW)
            ops.add_cast_into(memory[sl], feats.flatten(2).permute(2, 0, 1), None, 1.0)
            ops.add_cast_into(memory_pos[sl], enc.flatten(2).permute(2, 0, 1), tpos.expand(HW, B, self.mem_dim), 1.0)
        if ptr_bank is not None:
            # [capacity, B, C] -> capacity * (C // mem_dim) tokens of mem_dim, one strided fp32 copy (exact)
            cap = ptr_bank.shape[0]
            memory[n_sp:].view(cap, split, B, self.mem_dim).copy_(ptr_bank.view(cap, B, split, self.mem_dim).permute(0, 2, 1, 3))
        else:
            for j, ptr in enumerate(ptrs):
                # [B, C] -> (C // mem_dim) tokens of mem_dim (sam2_base.py:626-632)
                sl = slice(n_sp + j * split, n_sp + (j + 1) * split)
                ops.add_cast_into(memory[sl], ptr.to(F32).reshape(B, split, self.mem_dim).permute(1, 0, 2), None, 1.0)
        return memory, memory_pos, n_ptr_tok, key_count

    def _prepare_memory_conditioned_features(self, frame_idx, is_init_cond_frame, current_vision_feats, current_vision_pos_embeds,
                                             feat_sizes, output_dict, num_frames, track_in_reverse=False, memory_selection=None):
        """sam2_base.py:494-663: memory-bank selection is host logic (dict lookups, `_select_memory`); the bank itself is assembled
        into one [N_k, B, 64] buffer by strided copy kernels and handed to memory attention.

        memory_selection = (spatial, ptrs) skips the selection (the caller made it).  `ptrs` is then either the list of pointers or a
        PADDED bank (ptr_bank [capacity, B, C] fp32, key_count int32 device scalar = n_spatial*HW + n_ptrs*C/mem_dim): the launch
        shapes depend on the capacity only, the attention kernel reads the number of valid keys from the device -- the form a
        hipGraph of the per-slice forward is captured in (graphs.GraphedPropagation); rows of the bank past the valid pointers are
        never attended to."""
        B = current_vision_feats[-1].size(1)
        C = self.hidden_dim
        H, W = feat_sizes[-1]
        device = current_vision_feats[-1].device
        if self.num_maskmem == 0:
            return current_vision_feats[-1].permute(1, 2, 0).view(B, C, H, W)
        if is_init_cond_frame:
            # directly_add_no_mem_embed (sam2_base.py:640-644)
            y = ops.add_cast(current_vision_feats[-1].transpose(0, 1), self.no_mem_embed.detach().to(F32).expand(B, H * W, C), 1.0, F32)
            return y.view(B, H * W, C).transpose(0, 1).permute(1, 2, 0).view(B, C, H, W)
        if memory_selection is not None and memory_selection[0] == "assembled":
            # ("assembled", memory, memory_pos, number of pointer tokens, key_count): the caller keeps the bank assembled between slices
            # and re-writes only the entries that changed (graphs.GraphedPropagation)
            _, memory, memory_pos, n_ptr_tok, key_count = memory_selection
        else:
            spatial, ptrs = memory_selection if memory_selection is not None else \
                self._select_memory(frame_idx, output_dict, num_frames, track_in_reverse)
            memory, memory_pos, n_ptr_tok, key_count = self._assemble_memory(spatial, ptrs, B, H, W, device)
        pix = self.memory_attention(curr=current_vision_feats, curr_pos=current_vision_pos_embeds, memory=memory,
                                    memory_pos=memory_pos, num_obj_ptr_tokens=n_ptr_tok, key_count=key_count)
        return pix.permute(1, 2, 0).view(B, C, H, W)

    def _encode_new_memory(self, current_vision_feats, feat_sizes, pred_masks_high_res, is_mask_from_pts):
        """sam2_base.py:665-703; the sigmoid / binarise + scale + bias is fused into the first down-sampler conv."""
        from .. import autograd as ag
        if ag.active(self):
            return ag.encode_new_memory(self, current_vision_feats, feat_sizes, pred_masks_high_res, is_mask_from_pts)
        B = current_vision_feats[-1].size(1)
        C = self.hidden_dim
        H, W = feat_sizes[-1]
        if self.non_overlap_masks_for_mem_enc and not self.training:
            pred_masks_high_res = self._apply_non_overlapping_constraints(pred_masks_high_res)
        top = current_vision_feats[-1]  # [HW, B, C]
        pix_tokens = ops.add_cast(top.transpose(0, 1), None, 1.0, OP16).view(B * H * W, C)
        binarize = self.binarize_mask_from_pts_for_mem_enc and is_mask_from_pts and not self.training
        y = self.memory_encoder.run(pix_tokens, pred_masks_high_res, 2 if binarize else 1, float(self.sigmoid_scale_for_mem_enc),
                                    float(self.sigmoid_bias_for_mem_enc), B, H, W)
        maskmem_features = nchw_view(y, B, H, W)
        maskmem_pos_enc = [self.memory_encoder.position_encoding(maskmem_features).to(maskmem_features.dtype)]
        return maskmem_features, maskmem_pos_enc

    def track_step(self, frame_idx, is_init_cond_frame, current_vision_feats, current_vision_pos_embeds, feat_sizes, point_inputs,
                   mask_inputs, output_dict, num_frames, track_in_reverse=False, run_mem_encoder=True, prev_sam_mask_logits=None,
                   memory_selection=None):
        """sam2_base.py:705-800.  memory_selection: see _prepare_memory_conditioned_features (not part of the reference's signature).
        In train() mode with gradients enabled -- the state the reference's training loops put the net in -- the step runs through
        `autograd.track_step`: the same control flow with every module behind a torch.autograd.Function, so the returned tensors carry a
        graph and `loss.backward()` reaches the parameters (func_3d/function.py:176-186)."""
        from .. import autograd as ag
        if ag.active(self) and memory_selection is None:
            return ag.track_step(self, frame_idx, is_init_cond_frame, current_vision_feats, current_vision_pos_embeds, feat_sizes,
                                 point_inputs, mask_inputs, output_dict, num_frames, track_in_reverse, run_mem_encoder, prev_sam_mask_logits)
        current_out = {"point_inputs": point_inputs, "mask_inputs": mask_inputs}
        if len(current_vision_feats) > 1:
            high_res_features = [x.permute(1, 2, 0).view(x.size(1), x.size(2), *s)
                                 for x, s in zip(current_vision_feats[:-1], feat_sizes[:-1])]
        else:
            high_res_features = None
        if mask_inputs is not None and self.use_mask_input_as_output_without_sam:
            pix_feat = current_vision_feats[-1].permute(1, 2, 0)
            pix_feat = pix_feat.view(-1, self.hidden_dim, *feat_sizes[-1])
            sam_outputs = self._use_mask_as_output(pix_feat, high_res_features, mask_inputs)
        else:
            pix_feat_with_mem = self._prepare_memory_conditioned_features(
                frame_idx=frame_idx, is_init_cond_frame=is_init_cond_frame, current_vision_feats=current_vision_feats[-1:],
                current_vision_pos_embeds=current_vision_pos_embeds[-1:], feat_sizes=feat_sizes[-1:], output_dict=output_dict,
                num_frames=num_frames, track_in_reverse=track_in_reverse, memory_selection=memory_selection)
            if prev_sam_mask_logits is not None:
                assert point_inputs is not None and mask_inputs is None
                mask_inputs = prev_sam_mask_logits
            multimask_output = self._use_multimask(is_init_cond_frame, point_inputs)
            sam_outputs = self._forward_sam_heads(backbone_features=pix_feat_with_mem, point_inputs=point_inputs,
                                                  mask_inputs=mask_inputs, high_res_features=high_res_features,
                                                  multimask_output=multimask_output)
        _, _, _, low_res_masks, high_res_masks, obj_ptr, _ = sam_outputs
        current_out["pred_masks"] = low_res_masks
        current_out["pred_masks_high_res"] = high_res_masks
        current_out["obj_ptr"] = obj_ptr
        if run_mem_encoder and self.num_maskmem > 0:
            maskmem_features, maskmem_pos_enc = self._encode_new_memory(
                current_vision_feats=current_vision_feats, feat_sizes=feat_sizes, pred_masks_high_res=high_res_masks,
                is_mask_from_pts=(point_inputs is not None))
            current_out["maskmem_features"] = maskmem_features
            current_out["maskmem_pos_enc"] = maskmem_pos_enc
        else:
            current_out["maskmem_features"] = None
            current_out["maskmem_pos_enc"] = None
        return current_out

    def _apply_non_overlapping_constraints(self, pred_masks):
        """sam2_base.py:812-830."""
        if pred_masks.size(0) == 1:
            return pred_masks
        return ops.non_overlap(pred_masks)

    def _use_multimask(self, is_init_cond_frame, point_inputs):
        """sam2_base.py:802-810."""
        num_pts = 0 if point_inputs is None else point_inputs["point_labels"].size(1)
        return (self.multimask_output_in_sam and (is_init_cond_frame or self.multimask_output_for_tracking)
                and (self.multimask_min_pt_num <= num_pts <= self.multimask_max_pt_num))


# -- tiny helpers on [B, ...] fp32 tensors, all through the add/cast kernel -----------------------------------------------
def _affine(x: torch.Tensor, scale: float, bias: float) -> torch.Tensor:
    """x * scale + bias via out = b + alpha * a with a broadcast constant b."""
    flat = x.reshape(1, -1, 1)
    const = torch.full((1, 1, 1), bias, dtype=F32, device=x.device).expand(1, flat.shape[1], 1)
    return ops.add_cast(const, flat, scale, F32).reshape(x.shape)
