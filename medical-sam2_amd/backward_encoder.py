"""Backward of the image encoder on the HIP path: Hiera trunk (patch embedding, position embedding, every MultiScaleBlock with
windowed / global attention, q-pool and pooled shortcut) and the FPN neck with the folded conv_s0 / conv_s1 -- the part of the 2-D
training loop the reference differentiates at func_2d/function.py:70-72 ("Train image encoder": `net.forward_image` under grad) with
`AdamW` over every `net.parameters()` (train_2d.py:43-47) and `losses.backward()` at func_2d/function.py:246-259.  (The prompt encoder
runs under `torch.no_grad()` there -- func_2d/function.py:140-149 -- so it has no gradient in that loop.)

No autograd graph: `image_encoder_forward_saved` keeps every block's input (the fp32 residual stream, ~0.6 GB at 4 x 1024^2),
`image_encoder_backward` walks the blocks in reverse and recomputes a block's inner activations from its input (LayerNorm, qkv,
attention, MLP hidden) before differentiating it.  Arithmetic runs on the library's kernels: GEMMs (forward kernel on transposed
operands + `msam2_gemm_tt` for the weight gradients), `msam2_layernorm_bwd`, `msam2_act_bwd`, flash-style `msam2_attention_bwd` on
window-partitioned q / k / v, `msam2_maxpool2x2_bwd`, `msam2_sumpool2x2`, `msam2_hiera_pos_embed_bwd`.  Window partition /
un-partition (zero-copy in the forward kernels) is plain data movement here (torch views + one copy per direction).

Pinned by tests/test_backward_encoder_gpu.py against torch.autograd through the oracle's trunk / neck and by the reference's own
`.grad` fixtures (tests/golden/grads_t256.npz, `image_encoder.*`).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch

from . import backward as bwd
from . import ops
from ._lib import check, lib
from .modeling.common import OP16, F32, nchw_view, to_bf16, tokens_of, v_f32, w_bf16
from .ops import _is_bf16, _p, _stream


# ---------------------------------------------------------------------------------------------------------------------
def maxpool2x2_backward(x: torch.Tensor, dy: torch.Tensor, B: int, H: int, W: int) -> torch.Tensor:
    """x [B*H*W, C] (row stride may exceed C; the pool's input), dy fp32 [B*(H/2)*(W/2), C] -> dx fp32 [B*H*W, C]."""
    C = x.shape[1]
    dy = dy.to(F32)
    dy = dy if dy.stride(1) == 1 else dy.contiguous()
    dx = torch.empty(B * H * W, C, dtype=F32, device=x.device)
    check(lib().msam2_maxpool2x2_bwd(_p(x), _is_bf16(x), x.stride(0), _p(dy), dy.stride(0), _p(dx), dx.stride(0), B, H, W, C, _stream()))
    return dx


def sumpool2x2(dy: torch.Tensor, B: int, H: int, W: int) -> torch.Tensor:
    """adjoint of ops.upsample2x_add_: fp32 [B*H*W, C] -> [B*(H/2)*(W/2), C]"""
    dy = dy.to(F32).contiguous()
    C = dy.shape[-1]
    out = torch.empty(B * (H // 2) * (W // 2), C, dtype=F32, device=dy.device)
    check(lib().msam2_sumpool2x2(_p(dy), _p(out), B, H, W, C, _stream()))
    return out


def _windows(img: torch.Tensor, ws: int, fill: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, int, int]:
    """[B, H, W, heads, D] -> [B*nW, heads, ws*ws, D] (contiguous copy), bottom / right padded to window multiples with `fill`
    ([heads, D], default 0) -- window_partition of backbones/utils.py:16-38 on an already projected map."""
    B, H, W, heads, D = img.shape
    Hp, Wp = -(-H // ws) * ws, -(-W // ws) * ws
    if (Hp, Wp) != (H, W):
        buf = (torch.zeros(1, 1, 1, heads, D, dtype=img.dtype, device=img.device) if fill is None else fill.to(img.dtype).view(1, 1, 1, heads, D)
               ).expand(B, Hp, Wp, heads, D).clone()
        buf[:, :H, :W] = img
        img = buf
    win = img.view(B, Hp // ws, ws, Wp // ws, ws, heads, D).permute(0, 1, 3, 5, 2, 4, 6)
    return win.reshape(B * (Hp // ws) * (Wp // ws), heads, ws * ws, D), Hp, Wp


def window_partition(img: torch.Tensor, B: int, H: int, W: int, heads: int, D: int, ws: int, fill: Optional[torch.Tensor] = None):
    """img: [B*H*W, >= heads*D] token rows (row stride free, 16-bit or fp32; a column slice of a wider buffer is fine) ->
    (windows [B*nW, heads, ws*ws, D] contiguous, Hp, Wp): backbones/utils.py:16-38 by one 16-byte-chunk kernel (msam2_window_move);
    padded tokens take `fill` [heads*D] (cast to img's type) or zero."""
    assert img.dim() == 2 and img.stride(1) == 1 and img.shape[0] == B * H * W and img.shape[1] == heads * D
    nwy, nwx = -(-H // ws), -(-W // ws)
    win = torch.empty(B * nwy * nwx, heads, ws * ws, D, dtype=img.dtype, device=img.device)
    f = None if fill is None else fill.to(img.dtype).contiguous()
    check(lib().msam2_window_move(_p(img), img.stride(0), _p(win), _p(f), B, H, W, heads, D, ws, img.element_size(), 1, _stream()))
    return win, nwy * ws, nwx * ws


def window_unpartition_into(win: torch.Tensor, img: torch.Tensor, B: int, H: int, W: int, heads: int, D: int, ws: int) -> None:
    """windows [B*nW, heads, ws*ws, D] contiguous -> the [B*H*W, heads*D] rows of `img` (a column slice of a wider buffer is fine),
    padding cropped: backbones/utils.py:41-62."""
    assert win.is_contiguous() and img.dim() == 2 and img.stride(1) == 1 and img.shape == (B * H * W, heads * D) and img.dtype == win.dtype
    check(lib().msam2_window_move(_p(img), img.stride(0), _p(win), None, B, H, W, heads, D, ws, img.element_size(), 0, _stream()))


def window_unpartition_cvt_into(win: torch.Tensor, img16: torch.Tensor, B: int, H: int, W: int, heads: int, D: int, ws: int) -> None:
    """fp32 windows [B*nW, heads, ws*ws, D] contiguous -> the 16-bit [B*H*W, heads*D] rows of `img16` (a column slice of a wider buffer is
    fine), padding cropped: un-partition and conversion to the GEMM operand type in one pass (`msam2_window_unpartition_cvt`)."""
    assert win.is_contiguous() and win.dtype == F32 and img16.dim() == 2 and img16.stride(1) == 1 and img16.shape == (B * H * W, heads * D)
    assert img16.dtype == ops.OP16
    check(lib().msam2_window_unpartition_cvt(_p(img16), img16.stride(0), _p(win), B, H, W, heads, D, ws, _stream()))


def _unwindows(win: torch.Tensor, B: int, Hp: int, Wp: int, ws: int) -> torch.Tensor:
    """inverse of `_windows` (without the crop): [B*nW, heads, ws*ws, D] (any strides) -> [B, Hp, Wp, heads, D]"""
    _, heads, _, D = win.shape
    return win.reshape(B, Hp // ws, Wp // ws, heads, ws, ws, D).permute(0, 1, 4, 2, 5, 3, 6).reshape(B, Hp, Wp, heads, D)


def _add32(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """fp32 a + b on [rows, C] maps through the add/cast kernel"""
    return ops.add_cast(a.reshape(1, a.shape[0], a.shape[1]), b.reshape(1, b.shape[0], b.shape[1]), 1.0, F32)[0]


def hiera_block_backward(blk, t: torch.Tensor, B: int, H: int, W: int, dy: torch.Tensor) -> Tuple[torch.Tensor, Dict[str, torch.Tensor]]:
    """Backward of `MultiScaleBlock.run` (hieradet.py:136-168 / 58-83): t fp32 [B*H*W, dim] the block's input, dy fp32
    [B*H'*W', dim_out] the gradient of its output.  Returns (dt fp32 [B*H*W, dim], {parameter name relative to the block: gradient})."""
    wc, a = blk._wc, blk.attn
    heads, dim, dim_out = a.num_heads, blk.dim, blk.dim_out
    Dt, Dp, qkv_w, qkv_b, proj_w = blk._packed_attn_weights()
    # Head dims the attention kernels do not take as they are (56 of Hiera-B+) run zero-padded per head to Dp, in the backward as in the
    # forward: a padded q / k channel is zero on both sides of the score product and a padded v channel meets a zero proj column, so the
    # gradients of the padded weight rows / columns are exactly zero and are dropped when the packed gradients are un-padded below.
    D = Dp
    width = heads * D
    scale = Dt ** -0.5
    T = B * H * W
    pool = blk.q_stride is not None
    Hq, Wq = (H // 2, W // 2) if pool else (H, W)
    ws = blk.window_size
    f = lambda key, p: v_f32(wc, key, p)
    g: Dict[str, torch.Tensor] = {}
    # ---- recompute the block's forward from its input
    n1w, n1b = f("n1w", blk.norm1.weight), f("n1b", blk.norm1.bias)
    xn = ops.layernorm(t, n1w, n1b, 1e-6)                                            # 16-bit [T, dim]
    qkv = ops.gemm(xn, qkv_w, qkv_b)                                                 # 16-bit [T, 3*width], image order
    q_img = qkv[:, :width]
    qp = ops.maxpool2x2(q_img, B, H, W) if pool else None                            # 16-bit [Tq, width]
    q_src = (qp if pool else q_img).reshape(B, Hq, Wq, heads, D)
    kv5 = qkv.view(B, H, W, 3, heads, D)
    if ws > 0:
        ws_q = ws // 2 if pool else ws
        bias3 = qkv_b.view(3, width)
        qw, Hqp, Wqp = window_partition(qp if pool else q_img, B, Hq, Wq, heads, D, ws_q)
        kw, Hp, Wp = window_partition(qkv[:, width:2 * width], B, H, W, heads, D, ws, bias3[1])
        vw, _, _ = window_partition(qkv[:, 2 * width:], B, H, W, heads, D, ws, bias3[2])
        q4, k4, v4 = qw, kw, vw                                                      # [B*nW, heads, L, D]
    else:
        q4 = q_src.reshape(B, Hq * Wq, heads, D).permute(0, 2, 1, 3)
        k4 = kv5[:, :, :, 1].reshape(B, H * W, heads, D).permute(0, 2, 1, 3)
        v4 = kv5[:, :, :, 2].reshape(B, H * W, heads, D).permute(0, 2, 1, 3)
    # windowed blocks: the output is written in the [windows, heads, Lq, D] order window_unpartition reads (no re-layout copy behind it);
    # global blocks: attention_forward_lse's default, a [.., heads, Lq, D] view of a token-major [.., Lq, heads, D] buffer
    o4_buf = torch.empty(q4.shape, dtype=q4.dtype, device=q4.device) if ws > 0 else None
    o4, lse = bwd.attention_forward_lse(q4, k4, v4, scale, out=o4_buf)
    if ws > 0:
        o = torch.empty(B * Hq * Wq, width, dtype=o4.dtype, device=o4.device)
        window_unpartition_into(o4, o, B, Hq, Wq, heads, D, ws_q)
    else:
        o = o4.permute(0, 2, 1, 3).reshape(B * Hq * Wq, width)
        o = o if o.is_contiguous() else o.contiguous()
    if dim != dim_out:
        pw, pb = w_bf16(wc, "pw", blk.proj.weight), f("pb", blk.proj.bias)
        pre = ops.gemm(xn, pw, pb, out_dtype=F32)                                    # un-pooled projected shortcut [T, dim_out]
        shortcut = ops.maxpool2x2(pre, B, H, W) if pool else pre
    else:
        shortcut = t
    t_mid = ops.gemm(o, proj_w, f("ob", a.proj.bias), residual=shortcut, out_dtype=F32)
    # ---- MLP + norm2
    mlp = blk.mlp
    n2w, n2b = f("n2w", blk.norm2.weight), f("n2b", blk.norm2.bias)
    xn2 = ops.layernorm(t_mid, n2w, n2b, 1e-6)
    w1, b1 = w_bf16(mlp._wc, "w0", mlp.layers[0].weight), v_f32(mlp._wc, "b0", mlp.layers[0].bias)
    w2, b2 = w_bf16(mlp._wc, "w1", mlp.layers[1].weight), v_f32(mlp._wc, "b1", mlp.layers[1].bias)
    dxn2, g["mlp.layers.0.weight"], g["mlp.layers.0.bias"], g["mlp.layers.1.weight"], g["mlp.layers.1.bias"] = \
        bwd.mlp_backward(xn2, w1, b1, w2, b2, dy, mlp._act_code)
    dt_mid, g["norm2.weight"], g["norm2.bias"] = bwd.layernorm_backward(t_mid, n2w, dxn2, 1e-6, add=dy.to(F32).contiguous())
    # ---- attention output projection
    do, g["attn.proj.weight"], g["attn.proj.bias"] = bwd.linear_backward(o, proj_w, dt_mid)      # do fp32 [Tq, width]
    if Dp != Dt:
        g["attn.proj.weight"] = g["attn.proj.weight"].view(dim_out, heads, Dp)[:, :, :Dt].reshape(dim_out, heads * Dt)
    # ---- attention core
    do_img = do.view(B, Hq, Wq, heads, D)
    dqkv = None
    if ws > 0:
        dow, _, _ = window_partition(do, B, Hq, Wq, heads, D, ws_q)                   # padded queries: zero upstream gradient
        dq4, dk4, dv4 = bwd.attention_backward(q4, k4, v4, dow, scale, o_lse=(o4, lse))
        dq4, dk4, dv4 = (x.to(F32).contiguous() for x in (dq4, dk4, dv4))
        # the three gradients go straight into their column thirds of the fused-qkv gradient (no window -> image copies, no torch.cat),
        # un-partitioned AND converted to the 16-bit GEMM operand type in one pass: the fused-qkv gradient is never an fp32 map
        dqkv = torch.empty(T, 3 * width, dtype=ops.OP16, device=t.device)
        if pool:
            dq_img = torch.empty(B * Hq * Wq, width, dtype=F32, device=t.device)
            window_unpartition_into(dq4, dq_img, B, Hq, Wq, heads, D, ws_q)
        else:
            window_unpartition_cvt_into(dq4, dqkv[:, :width], B, Hq, Wq, heads, D, ws_q)
            dq_img = None
        window_unpartition_cvt_into(dk4, dqkv[:, width:2 * width], B, H, W, heads, D, ws)
        window_unpartition_cvt_into(dv4, dqkv[:, 2 * width:], B, H, W, heads, D, ws)
        pad_bias = None
        if (Hp, Wp) != (H, W):
            # zero-padded tokens carry k = v = bias (the LayerNorm'ed map is padded BEFORE the qkv Linear, hieradet.py:143-150 +
            # utils.py:28-31): their dk / dv flow into the qkv bias, whose k / v thirds are the sums over ALL window tokens = the column
            # sums over the image's tokens that linear_backward returns + the sums over the padded tokens (16 % of a 14 x 14 window
            # grid on 64 x 64 tokens: a small gather-sum instead of two reductions over the whole fp32 window tensors)
            pad_bias = torch.zeros(2, width, dtype=F32, device=t.device)
            check(lib().msam2_window_pad_colsum(_p(dk4), _p(pad_bias[0]), B, H, W, heads, D, ws, _stream()))
            check(lib().msam2_window_pad_colsum(_p(dv4), _p(pad_bias[1]), B, H, W, heads, D, ws, _stream()))
    else:
        do4 = do_img.reshape(B, Hq * Wq, heads, D).permute(0, 2, 1, 3)
        if not pool and D in (64, 96, 128, 256):
            # global block: the flash backward writes dq / dk / dv straight into the column thirds of the fused-qkv gradient
            dqkv = torch.empty(T, 3 * width, dtype=F32, device=t.device)
            third = lambda i: dqkv[:, i * width:(i + 1) * width].view(B, H * W, heads, D).permute(0, 2, 1, 3)
            bwd.attention_backward(q4, k4, v4, do4, scale, o_lse=(o4, lse), out=(third(0), third(1), third(2)))
            dq_img = None
        else:
            dq4, dk4, dv4 = bwd.attention_backward(q4, k4, v4, do4, scale, o_lse=(o4, lse))
            dq_img = dq4.permute(0, 2, 1, 3).reshape(B * Hq * Wq, width)
            dk_img, dv_img = dk4.permute(0, 2, 1, 3).reshape(T, width), dv4.permute(0, 2, 1, 3).reshape(T, width)
        pad_bias = None
    if pool:
        dq_img = maxpool2x2_backward(q_img, dq_img.contiguous(), B, H, W)            # routed to the arg-max of each 2x2 window
    if dqkv is None:
        dqkv = torch.cat([dq_img, dk_img, dv_img], dim=1)                            # fp32 [T, 3*width] (data movement)
    elif dq_img is not None:
        dqkv[:, :width].copy_(dq_img)
    dxn, g["attn.qkv.weight"], g["attn.qkv.bias"] = bwd.linear_backward(xn, qkv_w, dqkv)
    if pad_bias is not None:
        g["attn.qkv.bias"][width:].add_(pad_bias.view(-1))      # image tokens (the GEMM's column sums) + padded tokens
    if Dp != Dt:
        g["attn.qkv.weight"] = g["attn.qkv.weight"].view(3, heads, Dp, dim)[:, :, :Dt].reshape(3 * heads * Dt, dim)
        g["attn.qkv.bias"] = g["attn.qkv.bias"].view(3, heads, Dp)[:, :, :Dt].reshape(3 * heads * Dt)
    # ---- shortcut + norm1
    if dim != dim_out:
        d_pre = maxpool2x2_backward(pre, dt_mid, B, H, W) if pool else dt_mid
        dxn_s, g["proj.weight"], g["proj.bias"] = bwd.linear_backward(xn, pw, d_pre)
        dxn = _add32(dxn, dxn_s)
        d_res = None
    else:
        d_res = dt_mid
    dt, g["norm1.weight"], g["norm1.bias"] = bwd.layernorm_backward(t, n1w, dxn, 1e-6, add=d_res)
    return dt, g


# ---------------------------------------------------------------------------------------------------------------------
def image_encoder_forward_saved(model, imgs: torch.Tensor) -> Tuple[dict, dict]:
    """`SAM2Base.forward_image` that also keeps what the backward needs: every trunk block's input tokens and geometry.
    Returns (backbone_out, state)."""
    enc = model.image_encoder
    trunk = enc.trunk
    B, _, S, _ = imgs.shape
    h = w = S // 4
    x = imgs.to(F32).contiguous()
    cols = ops.im2col_patch(x)
    t = trunk.patch_embed.tokens(x, trunk._pos_tokens(h, w))
    saved, outs = [], []
    for i, blk in enumerate(trunk.blocks):
        saved.append((t, h, w))
        t, h, w = blk.run(t, B, h, w)
        if i == trunk.stage_ends[-1] or (i in trunk.stage_ends and trunk.return_interm_layers):
            outs.append((i, t, h, w))
    xs = [nchw_view(tt, B, hh, ww) for _, tt, hh, ww in outs]
    dec = model.sam_mask_decoder
    post = {0: dec.conv_s0, 1: dec.conv_s1} if model.use_high_res_features_in_sam else None
    feats, pos = enc.neck(xs, post) if post else enc.neck(xs)
    if enc.scalp > 0:
        feats, pos = feats[: -enc.scalp], pos[: -enc.scalp]
    out = {"vision_features": feats[-1], "vision_pos_enc": pos, "backbone_fpn": feats}
    return out, {"B": B, "S": S, "cols": cols, "blocks": saved, "outs": outs, "post": post}


def _lateral_backward(x_tok: torch.Tensor, conv, d_out: torch.Tensor, post=None, need_dx: bool = True):
    """1x1 lateral conv (optionally followed by the folded 1x1 `post` conv: y = Wp (Wl x + bl) + bp) on fp32 tokens x [T, Cin]:
    returns (dx fp32 [T, Cin], {"w","b"} of the lateral conv, {"w","b"} of the post conv or None).  The composed weight is what the
    forward ran (FpnNeck.forward); its gradient is split back onto the two factors in parameter space (tiny matrices, fp32)."""
    Wl = conv.weight.detach().float().reshape(conv.weight.shape[0], -1)
    bl = conv.bias.detach().float()
    x16 = to_bf16(x_tok)
    if post is None:
        dx, dW, db = bwd.linear_backward(x16, Wl.to(OP16).contiguous(), d_out, need_dx=need_dx)
        return dx, {"w": dW.view_as(conv.weight), "b": db}, None
    Wp = post.weight.detach().float().reshape(post.weight.shape[0], -1)
    Wc = (Wp @ Wl).to(OP16).contiguous()
    dx, dWc, db = bwd.linear_backward(x16, Wc, d_out, need_dx=need_dx)                # dWc [Np, Cin], db [Np]
    g_lat = {"w": (Wp.t() @ dWc).view_as(conv.weight), "b": Wp.t() @ db}
    g_post = {"w": (dWc @ Wl.t() + torch.outer(db, bl)).view_as(post.weight), "b": db}
    return dx, g_lat, g_post


def _pow2_scale(amax: float) -> float:
    import math
    return 2.0 ** (-3 - math.ceil(math.log2(amax))) if amax > 0 and math.isfinite(amax) else 1.0


def record_scaled_amax(scales: dict, key: str, scaled: torch.Tensor, calibrating: bool) -> None:
    """Running max|scaled gradient| of one link, kept ON THE DEVICE next to its cached scale (scales["_amax"][key], a 1-element tensor
    created while calibrating and updated in place afterwards, so the update is part of a captured graph).  The scales are calibrated
    once and then frozen for every replay: as gradient magnitudes drift over training, a frozen scale lets 16-bit operands
    saturate or flush to zero without any visible sign -- `scale_drift` reads these slots and says when to calibrate again."""
    slots = scales.setdefault("_amax", {})
    t = scaled.detach()
    if t.dim() == 2 and t.shape[0] >= 4096:
        t = t[::4]                                         # a monitor, not a norm: every 4th row of the large token maps (order of magnitude)
    mn, mx = torch.aminmax(t)                              # one pass, no |t| temporary
    a = torch.maximum(mx, -mn).reshape(1).to(F32)
    if calibrating or key not in slots:
        slots[key] = a.clone()
    else:
        torch.maximum(slots[key], a, out=slots[key])


def scale_drift(scales: dict, lo: float = 2.0 ** -11, hi: float = 2.0 ** 3, reset: bool = True) -> Dict[str, float]:
    """{link: max|scaled gradient| since the last check} for the links that left the safe band (one host read).  Calibration puts the
    maximum at 2^-3; `hi` = 64x growth (fp16 saturates at 2^16, but products inside a block's backward run ahead of its input), `lo` =
    256x shrinkage (the small entries of the operand are then below fp16's normal range).  A non-finite maximum counts as drift."""
    import math
    out = {}
    for key, slot in scales.get("_amax", {}).items():
        v = float(slot.item())
        if not math.isfinite(v) or v > hi or (0.0 < v < lo):
            out[key] = v
        if reset:
            slot.zero_()
    return out


def image_encoder_backward(model, state: dict, d_fpn: List[Optional[torch.Tensor]], d_fpn_scales: Optional[List[float]] = None,
                           scales: Optional[dict] = None, trunk_grads: bool = True) -> Dict[str, torch.Tensor]:
    """d_fpn: gradients of `backbone_fpn` levels 0, 1, 2 as token-major maps ([B*256^2, 32], [B*128^2, 64], [B*64^2, 256] at 1024^2;
    None = no gradient), each multiplied by d_fpn_scales[l] (the loss scale it was computed under; default 1).  Returns
    {parameter name relative to the MODEL: TRUE gradient (loss scales removed)} for `image_encoder.*` and, when the high-res convs are
    folded into the neck, `sam_mask_decoder.conv_s0/conv_s1.*`.

    Loss scaling.  Between blocks the running gradient is fp32; inside a block it is a 16-bit GEMM / attention operand, and its
    magnitude changes by orders of magnitude along the trunk (and between the three entry points), so every block -- and every neck
    level -- runs under its own power-of-two scale that brings max|gradient| to 2^-3.  `scales` (a dict, filled on the first call) caches
    them: the first call calibrates with one host read per block, later calls -- and a hipGraph captured after it -- reuse the cached
    values without any synchronisation.
    trunk_grads=False: stop behind the neck (lateral convs + the folded conv_s0 / conv_s1) -- what is left to do when an optimiser owns the
    mask decoder's high-resolution convs but not the encoder (train_3d.py:34-37)."""
    enc = model.image_encoder
    trunk, neck = enc.trunk, enc.neck
    B = state["B"]
    outs = state["outs"]                                   # [(block index, tokens, h, w)] for the stage ends, finest first
    post = state["post"] or {}
    n = len(neck.convs) - 1
    grads: Dict[str, torch.Tensor] = {}
    td = neck.fpn_top_down_levels
    levels = len(outs)
    d_fpn_scales = list(d_fpn_scales) if d_fpn_scales is not None else [1.0] * len(d_fpn)
    scales = scales if scales is not None else {}
    calibrate = not scales.get("done", False)

    pending_t, pending_s = [], []

    def unscale(name: str, g: torch.Tensor, inv: float):
        """registers a parameter gradient that still carries its link's loss scale (fp32, owned by this call: scaled in place at the end)"""
        g = g if g.dtype == F32 else g.to(F32)
        grads[name] = g
        if inv != 1.0:
            pending_t.append(g)
            pending_s.append(float(inv))

    def rescaled(key: str, parts):
        """sum of (tensor, scale) pairs brought to the cached / calibrated power-of-two scale of `key`; returns (tensor, scale)"""
        if calibrate:
            amax = max(float(t.abs().max().item()) / s for t, s in parts)
            scales[key] = _pow2_scale(amax)
        s_new = scales[key]
        acc = None
        for t, s in parts:
            u = t if t.dtype == F32 else t.to(F32)
            if s_new != s:
                u = u * (s_new / s)                        # fp32, exact power-of-two factors (neighbouring links mostly share a scale: no pass)
            acc = u if acc is None else acc + u
        record_scaled_amax(scales, key, acc, calibrate)
        return acc, s_new

    # ---- neck: output gradients per level, the top-down contribution pushed to the coarser level, then the lateral convs
    is_folded = lambda l: post.get(l) is not None and l not in td and (l == 0 or (l - 1) not in td)
    d_lat: Dict[int, list] = {}
    for lvl in range(levels):
        if lvl < len(d_fpn) and d_fpn[lvl] is not None:
            d_lat[lvl] = [(d_fpn[lvl], d_fpn_scales[lvl])]
    d_stage: Dict[int, Tuple[torch.Tensor, float]] = {}    # gradient of each stage output (tokens, scale), keyed by block index
    for lvl in range(levels):                              # fine -> coarse
        if lvl not in d_lat:
            continue
        g_lvl, s_lvl = rescaled(f"neck{lvl}", d_lat[lvl])
        # the forward adds nearest-2x(out[lvl + 1]) into level lvl when lvl is a top-down level and lvl + 1 took the plain branch
        if lvl in td and (lvl + 1) < levels and not is_folded(lvl + 1):
            _, _, hh, ww = outs[lvl]
            d_lat.setdefault(lvl + 1, []).append((sumpool2x2(g_lvl, B, hh, ww), s_lvl))
        bi, tok, hh, ww = outs[lvl]
        conv = neck.convs[n - lvl].conv
        pc = post.get(lvl)
        folded = is_folded(lvl)
        assert pc is None or folded, "a post conv on a top-down level is not folded in the forward either"
        dx, g_lat, g_post = _lateral_backward(tok, conv, g_lvl, pc if folded else None)
        inv = 1.0 / s_lvl
        unscale(f"image_encoder.neck.convs.{n - lvl}.conv.weight", g_lat["w"], inv)
        unscale(f"image_encoder.neck.convs.{n - lvl}.conv.bias", g_lat["b"], inv)
        if g_post is not None:
            name = "conv_s0" if lvl == 0 else "conv_s1"
            unscale(f"sam_mask_decoder.{name}.weight", g_post["w"], inv)
            unscale(f"sam_mask_decoder.{name}.bias", g_post["b"], inv)
        d_stage[bi] = (dx, s_lvl)
    # ---- trunk blocks in reverse
    run: Optional[Tuple[torch.Tensor, float]] = None
    for i in (range(len(trunk.blocks) - 1, -1, -1) if trunk_grads else ()):
        parts = ([run] if run is not None else []) + ([d_stage[i]] if i in d_stage else [])
        if not parts:
            continue                                       # blocks behind the coarsest level that received a gradient
        dt, s_blk = rescaled(f"block{i}", parts)
        t_in, hh, ww = state["blocks"][i]
        dt, g = hiera_block_backward(trunk.blocks[i], t_in, B, hh, ww, dt)
        inv = 1.0 / s_blk
        for k, v in g.items():
            unscale(f"image_encoder.trunk.blocks.{i}.{k}", v, inv)
        run = (dt, s_blk)
    # ---- patch embedding + position embedding
    if run is not None:
        dt, s_pe = rescaled("patch_embed", [run])
        inv = 1.0 / s_pe
        pe = trunk.patch_embed
        E = pe.proj.weight.shape[0]
        dt16 = bwd._op16(dt)
        dW, db = bwd.gemm_tt(dt16, state["cols"], a_colsum=True)                       # [E, 160] (147 real columns)
        unscale("image_encoder.trunk.patch_embed.proj.weight", dW[:, :147].reshape(pe.proj.weight.shape), inv)
        unscale("image_encoder.trunk.patch_embed.proj.bias", db, inv)
        h = w = state["S"] // 4
        d_table = bwd.colsum(dt.view(B, h * w * E)).view(h * w, E)                     # the table is broadcast over the batch
        dpe, dpw = torch.empty_like(trunk.pos_embed, dtype=F32), torch.empty_like(trunk.pos_embed_window, dtype=F32)
        wsz = trunk.pos_embed_window.shape[-1]
        nb = lib().msam2_hiera_pos_embed_bwd_workspace_bytes(E, trunk.pos_embed.shape[3], h, wsz)
        ws = torch.empty(nb, dtype=torch.uint8, device=d_table.device)
        check(lib().msam2_hiera_pos_embed_bwd(_p(d_table), _p(dpe), _p(dpw), E, trunk.pos_embed.shape[2], trunk.pos_embed.shape[3], h, w,
                                              wsz, _p(ws), nb, _stream()))
        unscale("image_encoder.trunk.pos_embed", dpe, inv)
        unscale("image_encoder.trunk.pos_embed_window", dpw, inv)
    # every parameter gradient leaves its link's power-of-two scale in ONE multi-tensor launch (166 tensors: was a torch kernel each)
    if pending_t:
        torch._foreach_mul_(pending_t, pending_s)
    scales["done"] = True
    return grads
