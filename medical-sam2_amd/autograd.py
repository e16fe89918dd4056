"""torch.autograd bridge: the reference's training loops run against the drop-in modules as they are written.

The reference trains with autograd (`func_3d/function.py:130-191`: `train_add_new_* -> train_propagate_in_video ->
non_prompt_loss.backward(retain_graph=True); optimizer2.step(); prompt_loss.backward(); optimizer1.step()`, optimisers built at
`train_3d.py:34-54`; `func_2d/function.py:70-259`: `forward_image -> memory_attention -> sam_mask_decoder -> losses.backward();
optimizer.step()`).  The HIP path has no autograd graph of its own -- every module's backward is an explicit function (`backward.py`,
`backward_encoder.py`).  This file connects the two at MODULE granularity: one `torch.autograd.Function` per trained module whose
`forward` runs the HIP forward and whose `backward` calls the explicit HIP backward and hands the parameter gradients back to
autograd, which accumulates them into `.grad` -- so `loss.backward()` and any `torch.optim` optimiser work unchanged.  Everything between
the modules (memory-bank concatenation, mask selection, bilinear resizes, object-pointer mixing, per-object consolidation) is plain
differentiable torch glue, so back-propagation through time along the propagation chain is autograd's own bookkeeping.

When it is active: `torch.is_grad_enabled()` and the model is in `train()` mode -- what the reference's loops establish with
`net.train()`; `eval()` or `torch.no_grad()` keep the inference launches (bit-identical to what they were).

Numerics: the 16-bit backward operands run under a power-of-two scale chosen per call from max|upstream gradient| (one host read per
module and call -- this path is eager, not graph-captured) and are un-scaled in fp32, so `.grad` holds TRUE gradients.

Not differentiated (as on the explicit path): the prompt encoder (no optimiser of the reference's 3-D loop trains it; it
runs under `torch.no_grad()` in the 2-D loop), the IoU / object-score heads (no loss of either loop reaches them through the mask
logits: a gradient arriving there RAISES -- `_RaiseOnGrad` -- instead of being dropped), a mask PROMPT's pointer path (`_use_mask_as_output` is evaluated as a constant).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

from . import backward as bwd
from . import ops
from .ops import F32, OP16

NO_OBJ_SCORE = -1024.0


def active(model) -> bool:
    """the autograd bridge applies: grad mode on and the module in train() mode (module docstring)"""
    return torch.is_grad_enabled() and model.training


def _pow2(t: torch.Tensor) -> float:
    """power of two that brings max|t| to [2^-4, 2^-3] (one host read)"""
    a = float(t.detach().abs().max().item())
    return 2.0 ** (-3 - math.ceil(math.log2(a))) if a > 0 and math.isfinite(a) else 1.0


def _named(module) -> List[str]:
    return [n for n, _ in module.named_parameters()]


def _stamp(ctx, params):
    """remember the in-place version of every parameter at forward time (see `_check`)"""
    ctx.param_versions = tuple(p._version for p in params)


def _check(ctx, params, what: str):
    """The explicit backward passes re-read the module's CURRENT weights (the decoder's even recomputes its forward from them).  torch's
    own autograd raises when a tensor saved for backward was modified in place; this bridge does the same instead of silently
    differentiating a different function: an optimiser step between a module's forward and its backward is an error."""
    now = tuple(p._version for p in params)
    if now != ctx.param_versions:
        n = sum(1 for a, b in zip(now, ctx.param_versions) if a != b)
        raise RuntimeError(f"{what}: {n} parameter(s) were modified in place (an optimiser step?) between this module's forward and its "
                           "backward; the HIP backward reads the current weights, so the gradient would belong to a different function")


class _RaiseOnGrad(torch.autograd.Function):
    """identity whose backward raises: outputs that the HIP path does not differentiate (`mark_non_differentiable` dropped a
    gradient arriving there silently, so a loss term on the IoU prediction trained its head with zero gradient and no diagnostic)"""

    @staticmethod
    def forward(ctx, x, what):
        ctx.what = what
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        raise RuntimeError(f"a gradient reached {ctx.what}, which the HIP backward does not differentiate (DESIGN.md 7.4): no loss of the "
                           "reference's 2-D / 3-D training loops does; detach() the value if the term is meant as a constant")


def _param_grads(names: List[str], params, grads: Dict[str, torch.Tensor], inv: float):
    out = []
    for n, p in zip(names, params):
        g = grads.get(n)
        out.append(None if (g is None or not p.requires_grad) else (g.to(F32) * inv).reshape(p.shape))
    return out


# ---------------------------------------------------------------------------------------------------------------------
class MemoryAttentionFn(torch.autograd.Function):
    """`MemoryAttention.forward` (memory_attention.py:119-169): HIP forward that keeps its intermediates, explicit HIP backward."""

    @staticmethod
    def forward(ctx, module, n_ptr_tok, curr, curr_pos, memory, memory_pos, *params):
        with torch.no_grad():
            y, state = bwd.memory_attention_forward_saved(module, curr.detach(), curr_pos.detach(), memory.detach().to(F32),
                                                          memory_pos.detach().to(F32), int(n_ptr_tok), dropout=module.next_dropout())
        ctx.module, ctx.state, ctx.names, ctx.params = module, state, _named(module), params
        _stamp(ctx, params)
        ctx.set_materialize_grads(False)
        return y

    @staticmethod
    def backward(ctx, dy):
        if dy is None:
            return (None,) * (6 + len(ctx.params))
        _check(ctx, ctx.params, "MemoryAttention")
        with torch.no_grad():
            s = _pow2(dy)
            st = dict(ctx.state)
            st["ctxs"] = list(st["ctxs"])               # the explicit backward releases its state as it goes: keep ours for retain_graph=True
            dcurr, dmem, dmem_pos, grads = bwd.memory_attention_backward_saved(ctx.module, st, dy.to(F32) * s)
            inv = 1.0 / s
            return (None, None, dcurr.to(F32) * inv, None, dmem.to(F32) * inv, dmem_pos.to(F32) * inv, *_param_grads(ctx.names, ctx.params, grads, inv))


def memory_attention(module, curr, curr_pos, memory, memory_pos, num_obj_ptr_tokens: int):
    params = tuple(module.parameters())
    return MemoryAttentionFn.apply(module, num_obj_ptr_tokens, curr, curr_pos, memory, memory_pos, *params)


# ---------------------------------------------------------------------------------------------------------------------
class MaskDecoderFn(torch.autograd.Function):
    """`MaskDecoder.predict_masks` on token-major inputs (mask_decoder.py:171-266).  Outputs: masks [B, nm, 4h, 4w], IoU predictions,
    mask tokens [B, nm, C], object-score logits.  The backward takes gradients on the masks and on the mask tokens (the object-pointer
    path, sam2_base.py:376-388)."""

    @staticmethod
    def forward(ctx, dec, B, h, w, src, pe, sparse, f0, f1, *params):
        with torch.no_grad():
            f0_16, f1_16 = ops.add_cast(f0.detach().unsqueeze(0), None, 1.0, OP16)[0], ops.add_cast(f1.detach().unsqueeze(0), None, 1.0, OP16)[0]
            src_c, sp = src.detach().to(F32).contiguous(), sparse.detach().to(F32).contiguous()
            masks, ious, tokens, obj = dec.predict_masks_tokens(src_c, pe.detach(), sp, f0_16, f1_16, B, h, w)
        ctx.args = (dec, src_c, pe.detach(), sp, f0_16, f1_16, B, h, w)
        ctx.names, ctx.params = _named(dec), params
        _stamp(ctx, params)
        ctx.need_hr = f0.requires_grad or f1.requires_grad
        ctx.set_materialize_grads(False)
        return masks, ious, tokens.contiguous(), obj

    @staticmethod
    def backward(ctx, d_masks, d_ious, d_tokens, d_obj):
        n_in = 9 + len(ctx.params)
        if d_ious is not None or d_obj is not None:     # (`mask_decoder` below wraps both outputs so that this is reported at the source)
            raise RuntimeError("MaskDecoder: a gradient reached the IoU prediction / object-score logits, which the HIP backward does not differentiate")
        if d_masks is None and d_tokens is None:
            return (None,) * n_in
        _check(ctx, ctx.params, "MaskDecoder")
        dec = ctx.args[0]
        with torch.no_grad():
            dev = ctx.args[1].device
            B, h, w = ctx.args[6:9]
            nm, C = dec.num_mask_tokens, dec.transformer_dim
            # One scale for both upstream gradients, chosen from the MASK gradient (it enters as a 16-bit GEMM operand right away); the token
            # gradient is added in fp32 before anything is rounded and only has to stay inside the 16-bit range -- if it would not, the two
            # are back-propagated separately (the decoder backward is linear).  Same rule as training_3d.volume_backward.
            a_m = float(d_masks.abs().max().item()) if d_masks is not None else 0.0
            a_t = float(d_tokens.abs().max().item()) if d_tokens is not None else 0.0
            if a_m == 0.0 and a_t == 0.0:
                return (None,) * n_in
            dm = d_masks.to(F32).contiguous() if a_m > 0 else torch.zeros(B, nm, 4 * h, 4 * w, dtype=F32, device=dev)
            dt = d_tokens.to(F32).contiguous() if a_t > 0 else None
            s_d = _pow2(dm) if a_m > 0 else _pow2(dt)
            aux: dict = {}
            if a_m > 0 and a_t * s_d > 1024.0:
                d_src, d_sp, g = bwd.mask_decoder_backward(*ctx.args, dm * s_d, aux=aux)
                s_t = _pow2(dt)
                aux2: dict = {}
                d_src2, d_sp2, g2 = bwd.mask_decoder_backward(*ctx.args, torch.zeros_like(dm), aux=aux2, d_mask_tokens=dt * s_t)
                r = s_d / s_t
                d_src, d_sp = d_src + d_src2 * r, d_sp + d_sp2 * r
                g = {k: v + g2[k] * r for k, v in g.items()}
                for k in ("d_feat_s0", "d_feat_s1"):
                    aux[k] = aux[k].to(F32) + aux2[k].to(F32) * r
            else:
                d_src, d_sp, g = bwd.mask_decoder_backward(*ctx.args, dm * s_d, aux=aux, d_mask_tokens=None if dt is None else dt * s_d)
            inv = 1.0 / s_d
            d_f0 = aux["d_feat_s0"].to(F32) * inv if ctx.need_hr else None
            d_f1 = aux["d_feat_s1"].to(F32) * inv if ctx.need_hr else None
            return (None, None, None, None, d_src.to(F32) * inv, None, d_sp.to(F32) * inv, d_f0, d_f1, *_param_grads(ctx.names, ctx.params, g, inv))


def mask_decoder(dec, src_tokens, pe_tokens, sparse, f0_tokens, f1_tokens, B: int, h: int, w: int):
    masks, ious, tokens, obj = MaskDecoderFn.apply(dec, B, h, w, src_tokens, pe_tokens, sparse, f0_tokens, f1_tokens, *tuple(dec.parameters()))
    # the two heads the HIP backward does not differentiate: usable as values (comparisons, selection, logging, .detach()), an error as
    # soon as a loss back-propagates into them
    return masks, _RaiseOnGrad.apply(ious, "the IoU prediction head"), tokens, _RaiseOnGrad.apply(obj, "the object-score head")


# ---------------------------------------------------------------------------------------------------------------------
class MemoryEncoderFn(torch.autograd.Function):
    """`MemoryEncoder.forward` on the slice's pixel features and its predicted mask (memory_encoder.py:138-181, sam2_base.py:665-703)."""

    @staticmethod
    def forward(ctx, enc, mode, sc, bi, B, H, W, top, mask_high, *params):
        from .modeling.common import nchw_view
        with torch.no_grad():
            pix = ops.add_cast(top.detach().transpose(0, 1), None, 1.0, OP16).view(B * H * W, -1)
            m = mask_high.detach().to(F32).contiguous()
            y = enc.run(pix, m, mode, sc, bi, B, H, W)
        ctx.args = (enc, pix, m, mode, sc, bi, B, H, W)
        ctx.names, ctx.params = _named(enc), params
        _stamp(ctx, params)
        ctx.set_materialize_grads(False)
        return nchw_view(y, B, H, W)

    @staticmethod
    def backward(ctx, dy):
        if dy is None:
            return (None,) * (9 + len(ctx.params))
        _check(ctx, ctx.params, "MemoryEncoder")
        from .modeling.common import tokens_of
        enc, pix, m, mode, sc, bi, B, H, W = ctx.args
        with torch.no_grad():
            d = tokens_of(dy.to(F32))
            s = _pow2(d)
            d_pix, g, d_mask = bwd.memory_encoder_backward(enc, pix, m, mode, sc, bi, B, H, W, (d * s).contiguous(), need_dmask=True)
            inv = 1.0 / s
            d_top = (d_pix.to(F32) * inv).view(B, H * W, -1).transpose(0, 1)
            return (None, None, None, None, None, None, None, d_top, d_mask.to(F32) * inv, *_param_grads(ctx.names, ctx.params, g, inv))


class TokenMLPFn(torch.autograd.Function):
    """`obj_ptr_proj` (a 3-layer MLP, sam2_utils.py:108-132) on the selected SAM output tokens (sam2_base.py:386-388)."""

    @staticmethod
    def forward(ctx, mlp, token, *params):
        with torch.no_grad():
            t = token.detach().to(F32).contiguous()
            y = mlp.run_tokens(t)
            ctx.tok16 = ops.add_cast(t.unsqueeze(0), None, 1.0, OP16)[0]
        ctx.mlp, ctx.names, ctx.params = mlp, _named(mlp), params
        _stamp(ctx, params)
        ctx.set_materialize_grads(False)
        return y

    @staticmethod
    def backward(ctx, dy):
        if dy is None:
            return (None,) * (2 + len(ctx.params))
        _check(ctx, ctx.params, "obj_ptr_proj")
        with torch.no_grad():
            s = _pow2(dy)
            g: dict = {}
            d_tok = bwd.mlp_layers_backward(ctx.mlp, ctx.tok16, (dy.to(F32) * s).contiguous(), "p", g)
            inv = 1.0 / s
            return (None, d_tok.to(F32) * inv, *_param_grads(ctx.names, ctx.params, {k[2:]: v for k, v in g.items()}, inv))


class BilinearUpsampleFn(torch.autograd.Function):
    """F.interpolate(mode="bilinear", align_corners=False) on [n,1,h,w] fp32 maps (sam2_base.py:367-373, sam2_video_predictor.py:724-744)
    by the library's kernel and its exact adjoint -- the same bits as the inference path produces, so a training forward and an
    inference forward of the same weights agree bit for bit."""

    @staticmethod
    def forward(ctx, x, H, W):
        ctx.shape = x.shape
        ctx.set_materialize_grads(False)
        with torch.no_grad():
            return ops.bilinear_upsample(x.detach().to(F32).contiguous(), H, W)

    @staticmethod
    def backward(ctx, dy):
        if dy is None:
            return None, None, None
        from ._lib import check, lib
        n, c, h, w = ctx.shape
        with torch.no_grad():
            d = dy.to(F32).contiguous()
            dx = torch.empty(n * c, h, w, dtype=F32, device=d.device)
            check(lib().msam2_bilinear_upsample_bwd(ops._p(d), ops._p(dx), n * c, h, w, d.shape[-2], d.shape[-1], ops._stream()))
        return dx.view(n, c, h, w), None, None


def bilinear_upsample(x: torch.Tensor, H: int, W: int) -> torch.Tensor:
    if tuple(x.shape[-2:]) == (H, W):
        return x
    return BilinearUpsampleFn.apply(x, H, W)


# ---------------------------------------------------------------------------------------------------------------------
class ImageEncoderFn(torch.autograd.Function):
    """`SAM2Base.forward_image` (sam2_base.py:464-476: Hiera trunk + FPN neck + conv_s0 / conv_s1): outputs the feature levels."""

    @staticmethod
    def forward(ctx, model, imgs, *params):
        from . import backward_encoder as be
        with torch.no_grad():
            out, state = be.image_encoder_forward_saved(model, imgs.detach())
        ctx.model, ctx.state, ctx.params = model, state, params
        ctx.names = ["image_encoder." + n for n in _named(model.image_encoder)]
        if model.use_high_res_features_in_sam:
            ctx.names += [f"sam_mask_decoder.{c}.{k}" for c in ("conv_s0", "conv_s1") for k in ("weight", "bias")]
        _stamp(ctx, params)
        ctx.set_materialize_grads(False)
        return tuple(out["backbone_fpn"])

    @staticmethod
    def backward(ctx, *d_levels):
        from . import backward_encoder as be
        from .modeling.common import tokens_of
        if all(d is None for d in d_levels):
            return (None,) * (2 + len(ctx.params))
        _check(ctx, ctx.params, "ImageEncoder")
        with torch.no_grad():
            d_fpn = [None if d is None else tokens_of(d.to(F32)) for d in d_levels]
            # a frozen trunk (`freeze_untrained`; only conv_s0 / conv_s1, which live in the decoder's optimiser group, still need gradients):
            # the neck level of the backward is all there is to do
            need_trunk = any(p.requires_grad for p in ctx.model.image_encoder.trunk.parameters())
            g = be.image_encoder_backward(ctx.model, ctx.state, d_fpn, [1.0] * len(d_fpn), None, trunk_grads=need_trunk)
            return (None, None, *_param_grads(ctx.names, ctx.params, g, 1.0))


def freeze_untrained(model, optimizers) -> int:
    """`requires_grad_(False)` on every parameter of `model` that none of `optimizers` (torch.optim objects) owns; returns how many
    were frozen.  The reference's 3-D loop (train_3d.py:34-54) builds its two optimisers over the memory / SAM layers only but never
    clears `requires_grad` on the rest, so autograd -- there and through this bridge -- differentiates the image encoder for every
    slice and throws the result away.  Calling this once after the optimisers are built stops the encoder's backward behind the FPN neck
    (conv_s0 / conv_s1 sit in the decoder's group and still get their gradients; with those frozen too `forward_image` is a plain
    graph-free forward): same updates, a fraction of the work."""
    owned = {id(p) for opt in optimizers for grp in opt.param_groups for p in grp["params"]}
    n = 0
    for p in model.parameters():
        if id(p) not in owned and p.requires_grad:
            p.requires_grad_(False)
            n += 1
    return n


def forward_image(model, imgs: torch.Tensor) -> dict:
    """grad-carrying `SAM2Base.forward_image`; the position tables are constants"""
    params = list(model.image_encoder.parameters())
    if model.use_high_res_features_in_sam:
        dec = model.sam_mask_decoder
        params += [dec.conv_s0.weight, dec.conv_s0.bias, dec.conv_s1.weight, dec.conv_s1.bias]
    if not any(p.requires_grad for p in params):
        with torch.no_grad():                          # a frozen encoder (train_3d.py:34-37 leaves it out of both optimisers): plain forward
            return model.forward_image(imgs)
    feats = ImageEncoderFn.apply(model, imgs, *params)
    with torch.no_grad():                              # sine position maps: input-independent constants (image_encoder.py:101-133)
        pos = [model.image_encoder.neck.position_encoding(f).to(f.dtype) for f in feats]
    return {"vision_features": feats[-1], "vision_pos_enc": pos, "backbone_fpn": list(feats)}


# ---------------------------------------------------------------------------------------------------------------------
# differentiable glue (plain torch): what sits between the modules in SAM2Base.track_step
# ---------------------------------------------------------------------------------------------------------------------
def assemble_memory(model, spatial, ptrs, n: int):
    """sam2_base.py:565-638 in differentiable form: (memory [Nk, n, 64], memory_pos, number of pointer tokens)"""
    C, md = model.hidden_dim, model.mem_dim
    split = C // md
    mem, pos = [], []
    for t_pos, prev in spatial:
        feats = prev["maskmem_features"].to(F32)                                  # [n, 64, H, W]
        enc = prev["maskmem_pos_enc"][-1].to(F32)
        mem.append(feats.flatten(2).permute(2, 0, 1))
        pos.append(enc.flatten(2).permute(2, 0, 1) + model.maskmem_tpos_enc[model.num_maskmem - t_pos - 1].to(F32))
    for p in ptrs:                                                                # [n, C] -> C // mem_dim tokens of mem_dim, no position
        tok = p.to(F32).reshape(n, split, md).permute(1, 0, 2)
        mem.append(tok)
        pos.append(torch.zeros_like(tok))
    return torch.cat(mem, 0), torch.cat(pos, 0), len(ptrs) * split


def encode_new_memory(model, current_vision_feats, feat_sizes, pred_masks_high_res, is_mask_from_pts: bool):
    """grad-carrying `SAM2Base._encode_new_memory` (sam2_base.py:665-703)"""
    B = current_vision_feats[-1].size(1)
    H, W = feat_sizes[-1]
    binarize = model.binarize_mask_from_pts_for_mem_enc and is_mask_from_pts and not model.training
    enc = model.memory_encoder
    y = MemoryEncoderFn.apply(enc, 2 if binarize else 1, float(model.sigmoid_scale_for_mem_enc), float(model.sigmoid_bias_for_mem_enc), B, H, W,
                              current_vision_feats[-1], pred_masks_high_res, *tuple(enc.parameters()))
    with torch.no_grad():
        pos = [enc.position_encoding(y).to(y.dtype)]
    return y, pos


def forward_sam_heads(model, backbone_features, point_inputs=None, mask_inputs=None, high_res_features=None, multimask_output=False):
    """grad-carrying `SAM2Base._forward_sam_heads` (sam2_base.py:252-410): same 7-tuple"""
    from .modeling.common import tokens_of
    B = backbone_features.size(0)
    dev = backbone_features.device
    E, S = model.sam_image_embedding_size, model.image_size
    C = model.hidden_dim
    if point_inputs is not None:
        coords, labels = point_inputs["point_coords"], point_inputs["point_labels"]
    else:
        coords, labels = torch.zeros(B, 1, 2, device=dev), -torch.ones(B, 1, dtype=torch.int32, device=dev)
    with torch.no_grad():                                # prompt encoder: not trained by the reference's loops (module docstring)
        if mask_inputs is not None and tuple(mask_inputs.shape[-2:]) != tuple(model.sam_prompt_encoder.mask_input_size):
            f = mask_inputs.shape[-1] // model.sam_prompt_encoder.mask_input_size[-1]
            mask_prompt = ops.aa_downsample(mask_inputs.detach().to(F32).contiguous(), f)
        else:
            mask_prompt = None if mask_inputs is None else mask_inputs.detach()
        sparse, dense = model.sam_prompt_encoder(points=(coords, labels), boxes=None, masks=mask_prompt)
        pe = tokens_of(model.sam_prompt_encoder.get_dense_pe().to(F32))[: E * E]
        dense_tok = tokens_of(dense.to(F32).expand(B, C, E, E)).view(B, E * E, C)
    dec = model.sam_mask_decoder
    src = tokens_of(backbone_features.to(F32)).view(B, E * E, C) + dense_tok       # image embedding + dense prompt (mask_decoder.py:231)
    f0, f1 = (tokens_of(f) for f in high_res_features)
    masks, ious, mask_tokens, obj = mask_decoder(dec, src.reshape(B * E * E, C), pe, sparse.to(F32), f0, f1, B, E, E)
    objv = obj.reshape(B)
    with torch.no_grad():                                # WHICH mask: arg-max / stability logic on values, no gradient (mask_decoder.py:147-168)
        dyn = dec.dynamic_multimask_via_stability and not model.training
        _, sel, iou_sel = ops.select_mask(masks.detach().contiguous(), ious.contiguous(), objv.contiguous(), multimask_output, dyn,
                                          dec.dynamic_multimask_stability_delta, dec.dynamic_multimask_stability_thresh)
        sel = sel.long()
    ar = torch.arange(B, device=dev)
    alive = (objv > 0).view(B, 1, 1, 1)
    low_res_masks = torch.where(alive, masks[ar, sel].unsqueeze(1), torch.full((), NO_OBJ_SCORE, device=dev))   # sam2_base.py:354-363
    if multimask_output:
        low_res_multimasks = torch.where(alive, masks[:, 1:], torch.full((), NO_OBJ_SCORE, device=dev))
        ious_out = ious[:, 1:]
    else:
        low_res_multimasks, ious_out = low_res_masks, iou_sel
    high_res_masks = bilinear_upsample(low_res_masks, S, S)
    high_res_multimasks = bilinear_upsample(low_res_multimasks, S, S) if multimask_output else high_res_masks
    tok_sel = sel if (multimask_output and dec.use_multimask_token_for_obj_ptr) else torch.zeros(B, dtype=torch.long, device=dev)
    token = mask_tokens[ar, tok_sel]
    mlp = model.obj_ptr_proj
    obj_ptr = TokenMLPFn.apply(mlp, token, *tuple(mlp.parameters()))
    if model.pred_obj_scores:                            # hard gate with fixed_no_obj_ptr (sam2_base.py:389-400; soft_no_obj_ptr is off in every YAML)
        lam = (objv > 0).to(F32).view(B, 1)
        if model.fixed_no_obj_ptr:
            obj_ptr = lam * obj_ptr
        obj_ptr = obj_ptr + (1 - lam) * model.no_obj_ptr.to(F32)
    return low_res_multimasks, high_res_multimasks, ious_out, low_res_masks, high_res_masks, obj_ptr, obj


def track_step(model, frame_idx, is_init_cond_frame, current_vision_feats, current_vision_pos_embeds, feat_sizes, point_inputs, mask_inputs,
               output_dict, num_frames, track_in_reverse=False, run_mem_encoder=True, prev_sam_mask_logits=None):
    """grad-carrying `SAM2Base.track_step` (sam2_base.py:705-800): the same control flow as the inference method, every module through
    its autograd Function, the glue in torch"""
    current_out = {"point_inputs": point_inputs, "mask_inputs": mask_inputs}
    n = current_vision_feats[-1].size(1)
    C = model.hidden_dim
    H, W = feat_sizes[-1]
    high_res = [x.permute(1, 2, 0).reshape(x.size(1), x.size(2), *s) for x, s in zip(current_vision_feats[:-1], feat_sizes[:-1])] \
        if len(current_vision_feats) > 1 else None
    if mask_inputs is not None and model.use_mask_input_as_output_without_sam:
        with torch.no_grad():                            # a mask prompt's outputs are constants here (module docstring)
            pix_feat = current_vision_feats[-1].detach().permute(1, 2, 0).reshape(-1, C, H, W)
            sam_outputs = model._use_mask_as_output(pix_feat, [h.detach() for h in high_res], mask_inputs)
    else:
        top, top_pos = current_vision_feats[-1], current_vision_pos_embeds[-1]
        if model.num_maskmem == 0:
            pix = top
        elif is_init_cond_frame:
            pix = top.to(F32) + model.no_mem_embed.to(F32)                       # directly_add_no_mem_embed (sam2_base.py:640-644)
        else:
            spatial, ptrs = model._select_memory(frame_idx, output_dict, num_frames, track_in_reverse)
            memory, memory_pos, n_ptr_tok = assemble_memory(model, spatial, ptrs, n)
            pix = memory_attention(model.memory_attention, top, top_pos, memory, memory_pos, n_ptr_tok)
        pix_feat_with_mem = pix.permute(1, 2, 0).reshape(n, C, H, W)
        if prev_sam_mask_logits is not None:
            assert point_inputs is not None and mask_inputs is None
            mask_inputs = prev_sam_mask_logits
        sam_outputs = forward_sam_heads(model, pix_feat_with_mem, point_inputs, mask_inputs, high_res,
                                        model._use_multimask(is_init_cond_frame, point_inputs))
    _, _, _, low_res_masks, high_res_masks, obj_ptr, _ = sam_outputs
    current_out.update(pred_masks=low_res_masks, pred_masks_high_res=high_res_masks, obj_ptr=obj_ptr)
    if run_mem_encoder and model.num_maskmem > 0:
        mm, mm_pos = encode_new_memory(model, current_vision_feats, feat_sizes, high_res_masks, point_inputs is not None)
        current_out.update(maskmem_features=mm, maskmem_pos_enc=mm_pos)
    else:
        current_out.update(maskmem_features=None, maskmem_pos_enc=None)
    return current_out
