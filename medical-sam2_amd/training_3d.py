"""Training step of the 3-D flow with back-propagation through time along the propagation chain -- what the reference's 3-D loop
differentiates with autograd (func_3d/function.py:58-191): `net.train()`, prompts on every `prompt_freq`-th slice through
`train_add_new_bbox / train_add_new_points`, `train_propagate_in_video` WITHOUT inference_mode (sam2_video_predictor.py:424-451, 1125-1236),
BCEWithLogitsLoss(pos_weight) on every slice's video-resolution mask logits, `non_prompt_loss.backward()` for the memory groups, then
`prompt_loss.backward()`, two Adam optimisers (train_3d.py:34-54: `sam_layers` = mask decoder at lr 1e-4; `mem_layers` = obj_ptr_proj +
memory encoder + memory attention + mask_downsample at lr 1e-8; image and prompt encoders frozen).

There is no autograd graph on the HIP path.  The forward pass keeps a TAPE: per slice the decoder's inputs, the selected mask token, the
memory attention's saved state and -- the part that makes it a chain -- WHICH earlier slices' memories and object pointers the slice
attended to.  The backward pass walks the slices in reverse processing order; slice t receives
    dL/d pred_masks[t]      from its own loss term,
    dL/d maskmem[t]         from every later slice that attended to its memory   -> memory encoder backward -> d(high-res mask) ->
                            bilinear up-sampling adjoint -> added to dL/d pred_masks[t]   (and the memory encoder's gradients),
    dL/d obj_ptr[t]         from every later slice that used its pointer          -> obj_ptr_proj backward -> d(SAM output token),
runs the mask decoder backward on (d masks, d token), and -- for a propagated slice -- the memory attention backward, whose d memory is
scattered to the slices the bank was assembled from.  Every link runs on 16-bit MFMA operands under its own power-of-two scale
(chosen from max|upstream|, one host read per link: this path is not graph-captured) and is un-scaled in fp32, so the returned
gradients are TRUE gradients.  Pinned by tests/golden/grads_bptt_t256.npz (torch.autograd on the reference, 5 slices, 2 objects).

What is not differentiated, as in the reference's optimisers: image encoder, prompt encoder, `no_mem_embed`, `no_obj_ptr`,
`maskmem_tpos_enc` (none of them is in `sam_layers` / `mem_layers`); the IoU and object-score heads see no mask loss.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch

from . import backward as bwd
from . import ops
from ._lib import check, lib
from .ops import F32, OP16, _p, _stream
from .training import DecoderAdam, upsampled_mask_loss

GROUPS = ("decoder", "memory_attention", "memory_encoder", "obj_ptr_proj")
# train_step_3d(bounded_tape=None): volumes longer than this many slices train on the bounded tape (the memory attention of a propagated
# slice is run a second time in the backward: +1 forward of 4 layers per slice against O(slices x keys) of saved projections)
BOUNDED_TAPE_FROM = 16


def _pow2(t: torch.Tensor) -> float:
    """power of two that brings max|t| to [2^-4, 2^-3] (one host read)"""
    a = float(t.abs().max().item())
    return 2.0 ** (-3 - math.ceil(math.log2(a))) if a > 0 and math.isfinite(a) else 1.0


def _acc(dst: Dict[str, torch.Tensor], src: Dict[str, torch.Tensor], inv_scale: float, prefix: str = ""):
    for k, v in src.items():
        k = prefix + k
        g = v.to(F32) * inv_scale
        dst[k] = g if k not in dst else dst[k] + g


def _prompt_points(pr: dict):
    if "boxes" in pr:
        from .volume import box_point_inputs
        return box_point_inputs(pr["boxes"])
    return {"point_coords": pr["point_coords"], "point_labels": pr["point_labels"]}


@torch.no_grad()
def volume_forward_saved(model, volume: torch.Tensor, prompts: Dict[int, dict], bounded_tape: bool = False):
    """The chain of `volume.segment_volume` (single rank) with a tape.  volume [T,3,S,S] normalised, prompts {slice: {"boxes": [n,4]} |
    {"point_coords", "point_labels"}}.  model.training decides dropout, mask binarisation for the memory encoder, the pointer selection
    and the dynamic multimask fallback exactly as in `track_step`.  Returns (tape, {slice: low-res mask logits [n,1,S/4,S/4]}).
    bounded_tape: the memory attention's per-layer intermediates of a propagated slice (its projections of EVERY bank key, four layers:
    O(slices x keys) over a volume -- func_3d/function.py:130-191 trains over `video_length` slices) are NOT kept; the tape keeps what
    selects and re-creates them -- the bank selection (references to outputs the tape holds anyway) and the dropout stream position --
    and `volume_backward` runs that slice's memory attention forward again, one slice at a time.  Same kernels on the same inputs: the
    gradients are those of the full tape (tests/test_bptt_gpu.py)."""
    from .modeling.common import to_bf16, tokens_of, v_f32
    T, S = volume.shape[0], model.image_size
    cond_ids = sorted(prompts)
    first = prompts[cond_ids[0]]
    n = (first["boxes"] if "boxes" in first else first["point_coords"]).shape[0]
    dev = volume.device
    dec, ma = model.sam_mask_decoder, model.memory_attention
    C = model.hidden_dim
    od = {"cond_frame_outputs": {}, "non_cond_frame_outputs": {}}
    owner: Dict[int, int] = {}                           # id(stored output dict / pointer tensor) -> slice
    tape = {"frames": {}, "order": [], "n": n, "T": T}
    dense = model.sam_prompt_encoder.no_mask_embed.weight.detach().reshape(1, -1).to(F32)
    order = cond_ids + [t for t in range(T) if t not in prompts]
    for t in order:
        is_cond = t in prompts
        bo = model.forward_image(volume[t][None])
        bo = {"backbone_fpn": [f.expand(n, -1, -1, -1) for f in bo["backbone_fpn"]],
              "vision_pos_enc": [q.expand(n, -1, -1, -1) for q in bo["vision_pos_enc"]]}
        _, feats, pos, sizes = model._prepare_backbone_features(bo)
        h, w = sizes[-1]
        L = h * w
        hr = [f.permute(1, 2, 0).view(n, -1, *s) for f, s in zip(feats[:-1], sizes[:-1])]
        f0, f1 = to_bf16(tokens_of(hr[0])), to_bf16(tokens_of(hr[1]))
        pe = tokens_of(model.sam_prompt_encoder.get_dense_pe().to(F32))[:L]
        fr = {"cond": is_cond, "h": h, "w": w, "f0": f0, "f1": f1, "pe": pe, "top": feats[-1]}
        if is_cond:
            pin = _prompt_points(prompts[t])
            coords, labels = pin["point_coords"], pin["point_labels"]
            src = ops.add_cast(feats[-1].transpose(0, 1), model.no_mem_embed.detach().to(F32).expand(n, L, C), 1.0, F32)   # sam2_base.py:640-644
        else:
            pin = None
            coords = torch.zeros(n, 1, 2, device=dev)
            labels = -torch.ones(n, 1, dtype=torch.int32, device=dev)
            spatial, ptrs = model._select_memory(t, od, T)
            memory, memory_pos, n_ptr_tok, _ = model._assemble_memory(spatial, ptrs, n, h, w, dev)
            drop = ma.next_dropout()
            y, state = bwd.memory_attention_forward_saved(ma, feats[-1], pos[-1], memory, memory_pos, n_ptr_tok, dropout=drop)
            src = y.transpose(0, 1)
            fr.update(spatial=[owner[id(o)] for _, o in spatial], ptrs=[owner[id(p)] for p in ptrs], n_ptr_tok=n_ptr_tok)
            # what re-creates the state: the selection itself (references, no copies), the dropout sub-stream of this forward; the
            # position encoding of the top level is the same for every slice and kept once
            fr.update(mem_sel=(spatial, ptrs), drop=drop)
            tape.setdefault("top_pos", pos[-1])
            if bounded_tape:
                del state
            else:
                fr.update(state=state)
        src = ops.add_cast(src.reshape(n, L, C), dense.view(1, 1, C).expand(n, L, C), 1.0, F32).view(n * L, C)
        se, _ = model.sam_prompt_encoder(points=(coords, labels), boxes=None, masks=None)
        se = se.to(F32)
        masks, ious, mask_tokens, obj = dec.predict_masks_tokens(src, pe, se, f0, f1, n, h, w)
        multimask = model._use_multimask(is_cond, pin)
        dyn = dec.dynamic_multimask_via_stability and not model.training
        objv = obj.reshape(n).contiguous()
        low, sel, _ = ops.select_mask(masks, ious, objv, multimask, dyn, dec.dynamic_multimask_stability_delta,
                                      dec.dynamic_multimask_stability_thresh)
        high = ops.bilinear_upsample(low, S, S)
        tok_sel = sel if (multimask and dec.use_multimask_token_for_obj_ptr) else None
        token = ops.gather_rows(mask_tokens.contiguous(), tok_sel)
        tok16 = to_bf16(token.contiguous())
        ptr = model.obj_ptr_proj.run_tokens(token)
        ops.obj_ptr_mix_(ptr, objv, v_f32(model._wc, "nop", model.no_obj_ptr))
        mm, mm_pos = model._encode_new_memory(current_vision_feats=feats, feat_sizes=sizes, pred_masks_high_res=high, is_mask_from_pts=pin is not None)
        binarize = model.binarize_mask_from_pts_for_mem_enc and pin is not None and not model.training
        cur = {"pred_masks": low, "pred_masks_high_res": high, "obj_ptr": ptr, "maskmem_features": mm, "maskmem_pos_enc": mm_pos,
               "point_inputs": pin, "mask_inputs": None}
        (od["cond_frame_outputs"] if is_cond else od["non_cond_frame_outputs"])[t] = cur
        owner[id(cur)], owner[id(ptr)] = t, t
        fr.update(src=src, sparse=se, mask_sel=sel.long(), tok_sel=None if tok_sel is None else tok_sel.long(), obj=objv, tok16=tok16,
                  high=high, mode=2 if binarize else 1, out=cur)
        tape["frames"][t] = fr
        tape["order"].append(t)
    tape["output_dict"] = od
    return tape, {t: tape["frames"][t]["out"]["pred_masks"] for t in sorted(tape["frames"])}


@torch.no_grad()
def volume_backward(model, tape: dict, d_low: Dict[int, torch.Tensor]) -> Dict[str, Dict[str, torch.Tensor]]:
    """Back-propagation through time.  d_low: {slice: dL/d pred_masks [n,1,S/4,S/4] fp32} for the slices that carry a loss term.
    Returns TRUE gradients {"decoder" | "memory_attention" | "memory_encoder" | "obj_ptr_proj": {parameter name: fp32 gradient}}."""
    n = tape["n"]
    dec, ma, enc = model.sam_mask_decoder, model.memory_attention, model.memory_encoder
    C, md = model.hidden_dim, model.mem_dim
    split = C // md
    S = model.image_size
    sc, bi = float(model.sigmoid_scale_for_mem_enc), float(model.sigmoid_bias_for_mem_enc)
    grads: Dict[str, Dict[str, torch.Tensor]] = {g: {} for g in GROUPS}
    d_mem: Dict[int, torch.Tensor] = {}                  # slice -> d maskmem rows [n*L, 64]
    d_ptr: Dict[int, torch.Tensor] = {}                  # slice -> d obj_ptr [n, C]
    for t in reversed(tape["order"]):
        fr = tape["frames"][t]
        h, w = fr["h"], fr["w"]
        L = h * w
        dev = fr["src"].device
        alive = (fr["obj"] > 0).to(F32)                  # NO_OBJ_SCORE fill (sam2_base.py:354-363) and the hard pointer gate (389-400)
        dl = d_low.get(t)
        dl = None if dl is None else dl.to(F32).clone()
        # 1. memory of this slice was attended to later: memory encoder backward, continue into the mask
        if t in d_mem:
            s_e = _pow2(d_mem[t])
            pix = ops.add_cast(fr["top"].transpose(0, 1), None, 1.0, OP16).view(n * L, C)
            _, g_enc, dmask = bwd.memory_encoder_backward(enc, pix, fr["high"], fr["mode"], sc, bi, n, h, w, (d_mem[t] * s_e).contiguous(),
                                                          need_dmask=True)
            _acc(grads["memory_encoder"], g_enc, 1.0 / s_e)
            h4, w4 = fr["out"]["pred_masks"].shape[-2:]
            dlow_e = torch.empty(n, h4, w4, dtype=F32, device=dev)
            check(lib().msam2_bilinear_upsample_bwd(_p(dmask.contiguous()), _p(dlow_e), n, h4, w4, S, S, _stream()))
            dlow_e = dlow_e.view(n, 1, h4, w4) / s_e
            dl = dlow_e if dl is None else dl + dlow_e
        # 2. pointer of this slice was used later: obj_ptr_proj backward -> d(selected SAM token)
        d_tok = None
        if t in d_ptr:
            dp = d_ptr[t] * alive.view(n, 1)
            s_p = _pow2(dp)
            g_ptr: dict = {}
            d_token = bwd.mlp_layers_backward(model.obj_ptr_proj, fr["tok16"], (dp * s_p).contiguous(), "p", g_ptr)
            _acc(grads["obj_ptr_proj"], {k[2:]: v for k, v in g_ptr.items()}, 1.0 / s_p)
            d_tok = d_token.to(F32) / s_p                                                # [n, C]
        if dl is None and d_tok is None:
            continue
        # 3. mask decoder
        nm = dec.num_mask_tokens
        h4, w4 = fr["out"]["pred_masks"].shape[-2:]
        ar = torch.arange(n, device=dev)
        d_masks = torch.zeros(n, nm, h4, w4, dtype=F32, device=dev)
        if dl is not None:
            d_masks[ar, fr["mask_sel"]] = (dl * alive.view(n, 1, 1, 1))[:, 0]
        d_mtok = None
        if d_tok is not None:
            d_mtok = torch.zeros(n, nm, C, dtype=F32, device=dev)
            d_mtok[ar, fr["tok_sel"] if fr["tok_sel"] is not None else torch.zeros(n, dtype=torch.long, device=dev)] = d_tok
        # One scale for both upstream gradients, chosen from the MASK gradient: it is the one that enters as a 16-bit GEMM operand right
        # away (d_masks against the up-scaled features), and a mean-reduced BCE gradient scaled by anything much larger's maximum would
        # land in fp16's subnormals.  The token gradient is added in fp32 to the
        # hyper-network path's result before anything is rounded, so it only has to stay inside the 16-bit RANGE: if it would not
        # (ratio beyond 2^13), the two upstream gradients are back-propagated separately -- the decoder backward is linear.
        a_m = float(d_masks.abs().max().item()) if dl is not None else 0.0
        a_t = float(d_mtok.abs().max().item()) if d_mtok is not None else 0.0
        s_d = _pow2(d_masks) if a_m > 0 else _pow2(d_mtok)
        args = (dec, fr["src"], fr["pe"], fr["sparse"], fr["f0"], fr["f1"], n, h, w)
        if a_m > 0 and a_t * s_d > 1024.0:
            d_src, _, g_dec = bwd.mask_decoder_backward(*args, d_masks * s_d)
            s_t = _pow2(d_mtok)
            d_src2, _, g_dec2 = bwd.mask_decoder_backward(*args, torch.zeros_like(d_masks), d_mask_tokens=d_mtok * s_t)
            r = s_d / s_t
            d_src = d_src + d_src2 * r
            g_dec = {k: v + g_dec2[k] * r for k, v in g_dec.items()}
        else:
            d_src, _, g_dec = bwd.mask_decoder_backward(*args, d_masks * s_d, d_mask_tokens=None if d_mtok is None else d_mtok * s_d)
        _acc(grads["decoder"], g_dec, 1.0 / s_d)
        if fr["cond"]:
            continue                                     # src = features + no_mem_embed: nothing trained upstream
        # 4. memory attention; its d memory goes back to the slices the bank was assembled from
        s_m = _pow2(d_src)
        state = fr.get("state")
        if state is None:                                # bounded tape: this slice's memory attention forward again (see volume_forward_saved)
            spatial, ptrs = fr["mem_sel"]
            memory, memory_pos, n_ptr_tok, _ = model._assemble_memory(spatial, ptrs, n, h, w, dev)
            _, state = bwd.memory_attention_forward_saved(ma, fr["top"], tape["top_pos"], memory, memory_pos, n_ptr_tok, dropout=fr["drop"])
            del memory, memory_pos
        _, dmemory, _, g_mem = bwd.memory_attention_backward_saved(ma, state, (d_src * s_m).view(n, L, C).transpose(0, 1))
        del state
        _acc(grads["memory_attention"], g_mem, 1.0 / (s_d * s_m))
        dmemory = dmemory.to(F32) / (s_d * s_m)                                          # [Nk, n, 64]
        for i, u in enumerate(fr["spatial"]):
            g = dmemory[i * L:(i + 1) * L].transpose(0, 1).reshape(n * L, md)
            d_mem[u] = g.contiguous() if u not in d_mem else d_mem[u] + g
        base = len(fr["spatial"]) * L
        for j, u in enumerate(fr["ptrs"]):
            g = dmemory[base + j * split: base + (j + 1) * split].transpose(0, 1).reshape(n, C)   # tokens of 64 -> [n, C] (sam2_base.py:626-632)
            d_ptr[u] = g.contiguous() if u not in d_ptr else d_ptr[u] + g
    return grads


@torch.no_grad()
def train_step_3d(model, optimizers: Dict[str, DecoderAdam], volume: torch.Tensor, prompts: Dict[int, dict], targets: Dict[int, torch.Tensor],
                  pos_weight: float = 2.0, grads_out: Optional[dict] = None, data_parallel: bool = False, group=None,
                  bounded_tape: Optional[bool] = None):
    """One iteration of func_3d/function.py:58-191 on one volume.  targets {slice: [n,1,S,S] in {0,1}} for every slice.  optimizers maps
    "decoder" (the reference's optimizer1 / `sam_layers`) and "memory_attention" / "memory_encoder" / "obj_ptr_proj" (optimizer2 /
    `mem_layers`) to DecoderAdam instances over the respective module; missing groups are left alone.
    As in the reference the memory groups step on the gradient of the NON-PROMPT loss alone, the decoder on non-prompt + prompt
    (function.py:176-186: non_prompt_loss.backward(), optimizer2.step(), prompt_loss.backward(), optimizer1.step(); the second backward
    does not reach the memory groups).  Returns {"loss", "prompt_loss", "non_prompt_loss"} (floats) and fills grads_out (TRUE gradients
    per group: "non_prompt" / "prompt") when given.
    data_parallel: one process per GPU, every rank on its OWN volume (volumes share nothing: SURVEY 8(e) row 1): the per-group
    gradients -- true gradients, no loss scale to agree on -- are summed over the ranks in place (`parallel.allreduce_gradients_async`,
    all groups in flight together) and averaged inside the Adam kernel; the returned losses stay this rank's.
    bounded_tape: see `volume_forward_saved`; None = on for volumes of more than BOUNDED_TAPE_FROM slices."""
    T = volume.shape[0]
    if bounded_tape is None:
        bounded_tape = T > BOUNDED_TAPE_FROM
    tape, low = volume_forward_saved(model, volume, prompts, bounded_tape=bounded_tape)
    cond = set(prompts)
    n_c, n_nc = len(cond), T - len(cond)
    d_np, d_p = {}, {}
    loss_p = torch.zeros(1, dtype=F32, device=volume.device)
    loss_np = torch.zeros(1, dtype=F32, device=volume.device)
    for t in range(T):
        l_t, d_t = upsampled_mask_loss(low[t], targets[t].to(volume.device), 0, pos_weight)   # mean over objects and pixels
        if t in cond:
            loss_p += l_t / n_c
            d_p[t] = d_t / n_c
        else:
            loss_np += l_t / n_nc
            d_np[t] = d_t / n_nc
    g_np = volume_backward(model, tape, d_np) if n_nc else {g: {} for g in GROUPS}
    g_p = volume_backward(model, tape, d_p)
    if grads_out is not None:
        grads_out["non_prompt"], grads_out["prompt"] = g_np, g_p
    g_dec = dict(g_p["decoder"])
    for k, v in g_np["decoder"].items():
        g_dec[k] = g_dec[k] + v if k in g_dec else v
    step_grads = {"decoder": g_dec, "memory_attention": g_np["memory_attention"], "memory_encoder": g_np["memory_encoder"],
                  "obj_ptr_proj": g_np["obj_ptr_proj"]}
    inv_world = 1.0
    if data_parallel:
        from . import parallel
        # every rank on its OWN volume: which groups / parameters got a gradient depends on that volume (all slices prompted -> no
        # memory groups; a pointer never attended to -> no obj_ptr_proj), so the collective schedule must not follow the local
        # dictionaries: the ranks first agree on the union of the parameter names per group, absent ones are zero-filled,
        # and every rank then issues the same all-reduces over the same lists in the fixed GROUPS order.
        mods = {"decoder": model.sam_mask_decoder, "memory_attention": model.memory_attention, "memory_encoder": model.memory_encoder,
                "obj_ptr_proj": model.obj_ptr_proj}
        step_grads = parallel.union_gradient_keys(step_grads, mods, GROUPS, group)
        pend = [(grp, parallel.allreduce_gradients_async(step_grads[grp], group)) for grp in GROUPS]
        for grp, pnd in pend:
            step_grads[grp], inv_world = pnd.wait()
    for grp in ("memory_attention", "memory_encoder", "obj_ptr_proj", "decoder"):
        if grp in optimizers and step_grads[grp]:
            optimizers[grp].step(step_grads[grp], grad_scale=inv_world)
    lp, lnp = float(loss_p.item()), float(loss_np.item())
    return {"loss": (lp * n_c + lnp * n_nc) / T, "prompt_loss": lp, "non_prompt_loss": lnp}
