"""CPU ORACLE (test infrastructure, never shipped on the product path).

A plain fp32 PyTorch-CPU restatement of the reference's per-slice SAM2 forward (Hiera image encoder -> memory
attention -> prompt encoder + two-way mask decoder -> memory encoder) written as pure functions over a flat
``{state_dict_name: tensor}`` dictionary.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this module; the product package (``medical-sam2_amd/``) never does.

Pinned: every function below is checked in ``tests/test_oracle_golden.py`` against golden vectors captured from the
reference's own modules (``tests/golden/make_golden.py`` imports them from ``/root/reference`` with shims S1-S3, S5
of SURVEY.md section 8(c)).  Citations are ``file:line`` relative to the reference root.

Conventions: ``P`` is the weight dict, ``pre`` a key prefix.  Image-like tensors are NCHW at function boundaries
(the reference's layout) and token sequences are ``[L, B, C]`` or ``[B, L, C]`` exactly where the reference uses them.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
NO_OBJ_SCORE = -1024.0  # sam2_train/modeling/sam2_base.py:19


# --------------------------------------------------------------------------------------------------------------------
# configuration (the YAML leaves of sam2_train/sam2_hiera_{t,s}.yaml; b+ = Hiera class defaults hieradet.py:176-201)
# --------------------------------------------------------------------------------------------------------------------
def model_config(name: str = "hiera_s", image_size: int = 1024) -> dict:
    trunk = {
        "hiera_t": dict(embed_dim=96, num_heads=1, stages=(1, 2, 7, 2), global_att_blocks=(5, 7, 9), bkg=(7, 7)),
        "hiera_s": dict(embed_dim=96, num_heads=1, stages=(1, 2, 11, 2), global_att_blocks=(7, 10, 13), bkg=(7, 7)),
        "hiera_b+": dict(embed_dim=112, num_heads=2, stages=(2, 3, 16, 3), global_att_blocks=(12, 16, 20), bkg=(14, 14)),
    }[name]
    return dict(
        name=name,
        image_size=image_size,
        trunk=dict(trunk, window_spec=(8, 4, 14, 7), q_pool=3, q_stride=2),
        d_model=256,
        mem_dim=64,
        num_maskmem=7,
        memattn_layers=4,
        ffn_dim=2048,
        rope_theta=10000.0,
        sigmoid_scale_for_mem_enc=20.0,
        sigmoid_bias_for_mem_enc=-10.0,
        max_obj_ptrs_in_encoder=16,
        # build_sam.py:26-31,56-65 overrides
        dynamic_multimask_via_stability=True,
        dynamic_multimask_stability_delta=0.05,
        dynamic_multimask_stability_thresh=0.98,
        binarize_mask_from_pts_for_mem_enc=True,
        fill_hole_area=8,
        multimask_min_pt_num=0,
        multimask_max_pt_num=1,
    )


def hiera_block_specs(tc: dict) -> List[dict]:
    """Per-block (dim, dim_out, heads, window, q_stride) table; restates the loop at hieradet.py:229-257."""
    stages = tc["stages"]
    depth = sum(stages)
    stage_ends = [sum(stages[:i]) - 1 for i in range(1, len(stages) + 1)]
    q_pool_blocks = [e + 1 for e in stage_ends[:-1]][: tc["q_pool"]]
    dim, heads, cur_stage = tc["embed_dim"], tc["num_heads"], 1
    specs = []
    for i in range(depth):
        dim_out = dim
        window = tc["window_spec"][cur_stage - 1]  # lags one block behind the stage change
        if i in tc["global_att_blocks"]:
            window = 0
        if i - 1 in stage_ends:
            dim_out, heads, cur_stage = dim * 2, heads * 2, cur_stage + 1
        specs.append(dict(dim=dim, dim_out=dim_out, heads=heads, window=window, pool=(i in q_pool_blocks),
                          stage_end=(i in stage_ends)))
        dim = dim_out
    return specs


# --------------------------------------------------------------------------------------------------------------------
# primitives
# --------------------------------------------------------------------------------------------------------------------
# Operand-rounding EMULATION (test infrastructure for the precision argument of DESIGN.md section 4): with OPERAND_DTYPE set (a 16-bit
# torch dtype) every matrix product of the path rounds BOTH operands to that type and accumulates in fp32 -- what an MFMA kernel with
# 16-bit operands does -- while residual streams, LayerNorm, softmax statistics and biases stay fp32.  None (default): plain fp32.
OPERAND_DTYPE = None


def _rop(t: Tensor) -> Tensor:
    return t if OPERAND_DTYPE is None else t.to(OPERAND_DTYPE).to(t.dtype)


class operand_rounding:
    """`with operand_rounding(torch.float16): ...` -- run the oracle with emulated 16-bit matrix-product operands"""

    def __init__(self, dtype):
        self.dtype = dtype

    def __enter__(self):
        global OPERAND_DTYPE
        self.prev, OPERAND_DTYPE = OPERAND_DTYPE, self.dtype

    def __exit__(self, *a):
        global OPERAND_DTYPE
        OPERAND_DTYPE = self.prev
        return False


def lin(P, pre: str, x: Tensor) -> Tensor:
    return _rop(x) @ _rop(P[pre + ".weight"]).t() + P[pre + ".bias"]


def lnorm(P, pre: str, x: Tensor, eps: float) -> Tensor:
    mu = x.mean(-1, keepdim=True)
    var = (x - mu).pow(2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * P[pre + ".weight"] + P[pre + ".bias"]


def lnorm2d(P, pre: str, x: Tensor, eps: float = 1e-6) -> Tensor:
    """Channel LayerNorm on NCHW (sam2_utils.py:137-149)."""
    return lnorm(P, pre, x.permute(0, 2, 3, 1), eps).permute(0, 3, 1, 2)


def gelu(x: Tensor) -> Tensor:
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))  # exact erf form (nn.GELU default)


def softmax_attention(q: Tensor, k: Tensor, v: Tensor, pmask: Optional[Tensor] = None) -> Tensor:
    """softmax(q k^T / sqrt(D)) v on [..., L, D]; what F.scaled_dot_product_attention computes at
    hieradet.py:72-76, transformer.py:258,318 (no attention mask).  pmask: train-mode dropout_p as an explicit multiplier on the
    probabilities (keep / (1 - p) or 0), so that a test can hand over the masks of the implementation under test."""
    if pmask is None and OPERAND_DTYPE is None and q.shape[-2] * k.shape[-2] > (1 << 28):
        # the reference's own call at these sites (hieradet.py:72-76, transformer.py:258,318): on the CPU a streaming kernel that never
        # materialises the [Lq, Lk] scores -- 8x faster than the blocked form below at the 1.06 M-key bank of BASELINE configs[3]
        # (1.05 s against 8.6 s per 512 query rows on 8 cores), equal to it to 2e-8
        return torch.nn.functional.scaled_dot_product_attention(q, k, v)
    if pmask is None and q.shape[-2] * k.shape[-2] > (1 << 28) and q.shape[-2] > 512:
        # a score matrix of > 2^28 entries per head (the 1.06 M-key bank of BASELINE configs[3]: 17 GB in fp32): query rows are
        # independent, so the rows are processed in blocks of 512 -- the same arithmetic per row
        return torch.cat([softmax_attention(q[..., i:i + 512, :], k, v) for i in range(0, q.shape[-2], 512)], dim=-2)
    s = (_rop(q) @ _rop(k).transpose(-1, -2)) / math.sqrt(q.shape[-1])
    s = s - s.amax(-1, keepdim=True)
    p = s.exp()
    l = p.sum(-1, keepdim=True)
    if pmask is not None:
        p = p * pmask
    return (_rop(p) @ _rop(v)) / l


def mlp(P, pre: str, x: Tensor, n_layers: int, act, sigmoid_out: bool = False) -> Tensor:
    """sam2_utils.py:108-132."""
    for i in range(n_layers):
        x = lin(P, f"{pre}.layers.{i}", x)
        if i < n_layers - 1:
            x = act(x)
    return torch.sigmoid(x) if sigmoid_out else x


def sine_pos_2d(h: int, w: int, num_pos_feats: int, temperature: float = 10000.0) -> Tensor:
    """PositionEmbeddingSine.forward (position_encoding.py:78-112) for one image -> [C, h, w]."""
    npf = num_pos_feats // 2
    eps, scale = 1e-6, 2 * math.pi
    y = torch.arange(1, h + 1, dtype=torch.float32)
    x = torch.arange(1, w + 1, dtype=torch.float32)
    y = y / (y[-1] + eps) * scale
    x = x / (x[-1] + eps) * scale
    i = torch.arange(npf, dtype=torch.float32)
    dim_t = temperature ** (2 * torch.div(i, 2, rounding_mode="floor") / npf)
    px = x[:, None] / dim_t  # [w, npf]
    py = y[:, None] / dim_t  # [h, npf]

    def interleave(p):
        return torch.stack((p[:, 0::2].sin(), p[:, 1::2].cos()), dim=2).flatten(1)

    px, py = interleave(px), interleave(py)
    pos = torch.cat((py[:, None, :].expand(h, w, npf), px[None, :, :].expand(h, w, npf)), dim=2)
    return pos.permute(2, 0, 1).contiguous()


# --------------------------------------------------------------------------------------------------------------------
# Hiera trunk  (a-1 .. a-7)
# --------------------------------------------------------------------------------------------------------------------
def hiera_pos_embed(P, pre: str, h: int, w: int) -> Tensor:
    """hieradet.py:269-277 -> [1, h, w, C]; input independent."""
    bkg = F.interpolate(P[pre + ".pos_embed"], size=(h, w), mode="bicubic")
    win = P[pre + ".pos_embed_window"]
    win = win.tile([a // b for a, b in zip(bkg.shape, win.shape)])
    return (bkg + win).permute(0, 2, 3, 1)


def to_windows(x: Tensor, ws: int) -> Tuple[Tensor, Tuple[int, int]]:
    """backbones/utils.py:16-38: zero pad bottom/right to a multiple of ws, split into ws x ws windows."""
    B, H, W, C = x.shape
    ph, pw = (ws - H % ws) % ws, (ws - W % ws) % ws
    if ph or pw:
        x = F.pad(x, (0, 0, 0, pw, 0, ph))
    Hp, Wp = H + ph, W + pw
    x = x.reshape(B, Hp // ws, ws, Wp // ws, ws, C).permute(0, 1, 3, 2, 4, 5)
    return x.reshape(-1, ws, ws, C), (Hp, Wp)


def from_windows(xw: Tensor, ws: int, pad_hw: Tuple[int, int], hw: Tuple[int, int]) -> Tensor:
    """backbones/utils.py:41-62."""
    Hp, Wp = pad_hw
    H, W = hw
    B = xw.shape[0] // ((Hp // ws) * (Wp // ws))
    x = xw.reshape(B, Hp // ws, Wp // ws, ws, ws, -1).permute(0, 1, 3, 2, 4, 5).reshape(B, Hp, Wp, -1)
    return x[:, :H, :W, :]


def maxpool2x2_nhwc(x: Tensor) -> Tensor:
    """do_pool with MaxPool2d(2,2) (hieradet.py:23-34) on NHWC."""
    B, H, W, C = x.shape
    return x.reshape(B, H // 2, 2, W // 2, 2, C).amax(dim=(2, 4))


def multiscale_attention(P, pre: str, x: Tensor, heads: int, pool: bool) -> Tensor:
    """hieradet.py:58-83 on windows x: [Bw, h, w, dim]."""
    Bw, h, w, _ = x.shape
    qkv = lin(P, pre + ".qkv", x).reshape(Bw, h * w, 3, heads, -1)
    q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]  # [Bw, L, heads, D]
    if pool:
        q = maxpool2x2_nhwc(q.reshape(Bw, h, w, -1))
        h, w = q.shape[1:3]
        q = q.reshape(Bw, h * w, heads, -1)
    o = softmax_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2))
    o = o.transpose(1, 2).reshape(Bw, h, w, -1)
    return lin(P, pre + ".proj", o)


def multiscale_block(P, pre: str, x: Tensor, spec: dict) -> Tensor:
    """hieradet.py:136-168; x NHWC."""
    shortcut = x
    xn = lnorm(P, pre + ".norm1", x, 1e-6)
    if spec["dim"] != spec["dim_out"]:
        shortcut = lin(P, pre + ".proj", xn)
        if spec["pool"]:
            shortcut = maxpool2x2_nhwc(shortcut)
    ws = spec["window"]
    H, W = xn.shape[1:3]
    pad_hw = (H, W)
    if ws > 0:
        xn, pad_hw = to_windows(xn, ws)
    a = multiscale_attention(P, pre + ".attn", xn, spec["heads"], spec["pool"])
    if spec["pool"]:
        ws = ws // 2
        H, W = shortcut.shape[1:3]
        pad_hw = (H + (ws - H % ws) % ws, W + (ws - W % ws) % ws) if ws > 0 else (H, W)
    if spec["window"] > 0:
        a = from_windows(a, ws, pad_hw, (H, W))
    x = shortcut + a
    h = lnorm(P, pre + ".norm2", x, 1e-6)
    h = lin(P, pre + ".mlp.layers.1", gelu(lin(P, pre + ".mlp.layers.0", h)))
    return x + h


def hiera_trunk(P, cfg: dict, img: Tensor, pre: str = "image_encoder.trunk", collect: Optional[dict] = None) -> List[Tensor]:
    """Hiera.forward (hieradet.py:279-295): [B,3,S,S] -> list of NCHW stage outputs (high to low resolution)."""
    x = F.conv2d(_rop(img), _rop(P[pre + ".patch_embed.proj.weight"]), P[pre + ".patch_embed.proj.bias"], stride=4, padding=3)
    x = x.permute(0, 2, 3, 1)
    x = x + hiera_pos_embed(P, pre, x.shape[1], x.shape[2])
    outs = []
    for i, spec in enumerate(hiera_block_specs(cfg["trunk"])):
        x = multiscale_block(P, f"{pre}.blocks.{i}", x, spec)
        if collect is not None:
            collect[f"block{i}"] = x
        if spec["stage_end"]:
            outs.append(x.permute(0, 3, 1, 2))
    return outs


# --------------------------------------------------------------------------------------------------------------------
# FPN neck + forward_image (a-8, a-9, a-10)
# --------------------------------------------------------------------------------------------------------------------
def conv1x1(P, pre: str, x: Tensor) -> Tensor:
    return F.conv2d(_rop(x), _rop(P[pre + ".weight"]), P[pre + ".bias"])


def fpn_neck(P, cfg: dict, xs: Sequence[Tensor], pre: str = "image_encoder.neck") -> Tuple[List[Tensor], List[Tensor]]:
    """FpnNeck.forward (image_encoder.py:101-133): top-down only on levels 2,3, nearest x2, sum fuse."""
    n = len(xs) - 1
    out, pos = [None] * len(xs), [None] * len(xs)
    prev = None
    for i in range(n, -1, -1):
        lat = conv1x1(P, f"{pre}.convs.{n - i}.conv", xs[i])
        if i in (2, 3) and prev is not None:
            prev = lat + F.interpolate(prev.float(), scale_factor=2.0, mode="nearest")
        else:
            prev = lat
        out[i] = prev
        pos[i] = sine_pos_2d(prev.shape[-2], prev.shape[-1], cfg["d_model"])[None].expand(prev.shape[0], -1, -1, -1)
    return out, pos


def forward_image(P, cfg: dict, img: Tensor, collect: Optional[dict] = None) -> dict:
    """ImageEncoder.forward (image_encoder.py:29-42, scalp=1) + SAM2Base.forward_image (sam2_base.py:464-476)."""
    feats, pos = fpn_neck(P, cfg, hiera_trunk(P, cfg, img, collect=collect))
    feats, pos = feats[:-1], pos[:-1]
    feats = list(feats)
    feats[0] = conv1x1(P, "sam_mask_decoder.conv_s0", feats[0])
    feats[1] = conv1x1(P, "sam_mask_decoder.conv_s1", feats[1])
    return {"vision_features": feats[-1], "vision_pos_enc": list(pos), "backbone_fpn": feats}


def prepare_backbone_features(bo: dict):
    """sam2_base.py:478-492: NCHW -> [HW, B, C] for the three levels."""
    feats = [x.flatten(2).permute(2, 0, 1) for x in bo["backbone_fpn"]]
    pos = [x.flatten(2).permute(2, 0, 1) for x in bo["vision_pos_enc"]]
    sizes = [(x.shape[-2], x.shape[-1]) for x in bo["vision_pos_enc"]]
    return feats, pos, sizes


# --------------------------------------------------------------------------------------------------------------------
# memory attention (a-12, a-13)
# --------------------------------------------------------------------------------------------------------------------
def axial_rope_table(head_dim: int, end_x: int, end_y: int, theta: float = 10000.0) -> Tuple[Tensor, Tensor]:
    """compute_axial_cis (position_encoding.py:174-183) as (cos, sin) of shape [end_x*end_y, head_dim/2]:
    the first head_dim/4 pairs rotate with the x coordinate, the rest with y."""
    f = 1.0 / (theta ** (torch.arange(0, head_dim, 4)[: head_dim // 4].float() / head_dim))
    t = torch.arange(end_x * end_y, dtype=torch.float32)
    tx, ty = t % end_x, torch.div(t, end_x, rounding_mode="floor")
    ang = torch.cat([torch.outer(tx, f), torch.outer(ty, f)], dim=-1)
    return ang.cos(), ang.sin()


def rope_rotate(x: Tensor, cos: Tensor, sin: Tensor) -> Tensor:
    """apply_rotary_enc (position_encoding.py:194-216) on [..., L, D]; adjacent channels (2i, 2i+1) form a pair."""
    xr, xi = x[..., 0::2], x[..., 1::2]
    return torch.stack((xr * cos - xi * sin, xr * sin + xi * cos), dim=-1).flatten(-2)


def rope_attention(P, pre: str, q: Tensor, k: Tensor, v: Tensor, heads: int, theta: float,
                   num_k_exclude_rope: int = 0, pmask: Optional[Tensor] = None) -> Tensor:
    """RoPEAttention.forward (transformer.py:288-331), batch-first [B, L, C]; the table is recomputed for the query
    grid (302-305) and tiled over the keys (rope_k_repeat)."""
    fold = OPERAND_DTYPE is not None and heads == 1 and v.shape[-1] != P[pre + ".v_proj.weight"].shape[0]
    v_in = v
    q, k = lin(P, pre + ".q_proj", q), lin(P, pre + ".k_proj", k)
    v = None if fold else lin(P, pre + ".v_proj", v)
    B, Lq, C = q.shape
    D = C // heads
    if fold:
        # Operand-rounding emulation of the memory cross-attention AS THE HIP PATH EVALUATES IT (attn_kv64x2_kernel):
        # the values carry no RoPE and softmax rows sum to one, so P (M Wv^T + bv) Wo^T + bo = (P M) (Wo Wv)^T + (Wo bv + bo) exactly;
        # the kernel contracts P with the 64-channel memory rows M themselves and one K = 64 GEMM applies the composed weight (multiplied
        # in fp32, rounded once).  Same function, different ROUNDING POINTS (P M is rounded where v and o would be) -- with plain fp32
        # operands the two forms agree to round-off (tests/test_oracle_golden.py), so this branch only exists under `operand_rounding`.
        side = int(round(math.sqrt(Lq)))
        cos, sin = axial_rope_table(D, side, side, theta)
        qh, kh = q[:, None], k[:, None]
        qh = rope_rotate(qh, cos, sin)
        n_rope = kh.shape[-2] - num_k_exclude_rope
        if n_rope > 0:
            r = n_rope // Lq
            kh = torch.cat([rope_rotate(kh[:, :, :n_rope], cos.repeat(r, 1), sin.repeat(r, 1)), kh[:, :, n_rope:]], dim=2)
        o64 = softmax_attention(qh, kh, v_in[:, None], pmask)[:, 0]                      # [B, Lq, 64]
        Wc = P[pre + ".out_proj.weight"] @ P[pre + ".v_proj.weight"]
        bc = P[pre + ".out_proj.weight"] @ P[pre + ".v_proj.bias"] + P[pre + ".out_proj.bias"]
        return _rop(o64) @ _rop(Wc).t() + bc

    def split(t):
        return t.reshape(B, t.shape[1], heads, D).transpose(1, 2)

    q, k, v = split(q), split(k), split(v)
    side = int(round(math.sqrt(Lq)))
    cos, sin = axial_rope_table(D, side, side, theta)
    q = rope_rotate(q, cos, sin)
    n_rope = k.shape[-2] - num_k_exclude_rope
    if n_rope > 0:
        r = n_rope // Lq
        k = torch.cat([rope_rotate(k[:, :, :n_rope], cos.repeat(r, 1), sin.repeat(r, 1)), k[:, :, n_rope:]], dim=2)
    o = softmax_attention(q, k, v, pmask).transpose(1, 2).reshape(B, Lq, C)
    return lin(P, pre + ".out_proj", o)


def memory_attention(P, cfg: dict, curr: Tensor, memory: Tensor, curr_pos: Tensor, memory_pos: Tensor,
                     num_obj_ptr_tokens: int = 0, pre: str = "memory_attention", masks: Optional[dict] = None) -> Tensor:
    """MemoryAttention.forward (memory_attention.py:119-169), layers 17-99; seq-first in/out.  masks = None: eval mode.  Train mode:
    masks[(layer, site)] are the dropout multipliers (keep / (1 - p) or 0) of the six sites of a layer -- "sa_attn" [B,1,L,L] and
    "ca_attn" [B,1,L,Nk] on the attention probabilities (transformer.py:317-318), "drop1" / "drop2" / "drop3" [B,L,C] on the residual
    branches (63, 80, 98) and "ffn" [B,L,hidden] inside the FFN (97)."""
    one = lambda l, site: 1.0 if masks is None else masks[(l, site)]
    pm = lambda l, site: None if masks is None else masks[(l, site)]
    x = (curr + 0.1 * curr_pos).transpose(0, 1)
    mem, mpos = memory.transpose(0, 1), memory_pos.transpose(0, 1)
    for l in range(cfg["memattn_layers"]):
        lp = f"{pre}.layers.{l}"
        t = lnorm(P, lp + ".norm1", x, 1e-5)
        x = x + one(l, "drop1") * rope_attention(P, lp + ".self_attn", t, t, t, 1, cfg["rope_theta"], pmask=pm(l, "sa_attn"))
        t = lnorm(P, lp + ".norm2", x, 1e-5)
        x = x + one(l, "drop2") * rope_attention(P, lp + ".cross_attn_image", t, mem + mpos, mem, 1, cfg["rope_theta"],
                                                  num_k_exclude_rope=num_obj_ptr_tokens, pmask=pm(l, "ca_attn"))
        t = lnorm(P, lp + ".norm3", x, 1e-5)
        x = x + one(l, "drop3") * lin(P, lp + ".linear2", one(l, "ffn") * torch.relu(lin(P, lp + ".linear1", t)))
    return lnorm(P, pre + ".norm", x, 1e-5).transpose(0, 1)


# --------------------------------------------------------------------------------------------------------------------
# prompt encoder (a-14)
# --------------------------------------------------------------------------------------------------------------------
def random_fourier_pe(P, coords01: Tensor, pre: str = "sam_prompt_encoder.pe_layer") -> Tensor:
    """PositionEmbeddingRandom._pe_encoding (position_encoding.py:130-137)."""
    c = (2 * coords01 - 1) @ P[pre + ".positional_encoding_gaussian_matrix"]
    c = 2 * math.pi * c
    return torch.cat([c.sin(), c.cos()], dim=-1)


def dense_pe(P, h: int, w: int) -> Tensor:
    """PromptEncoder.get_dense_pe (prompt_encoder.py:68-77) -> [1, 256, h, w]."""
    ys = (torch.arange(h, dtype=torch.float32) + 0.5) / h
    xs = (torch.arange(w, dtype=torch.float32) + 0.5) / w
    grid = torch.stack([xs[None, :].expand(h, w), ys[:, None].expand(h, w)], dim=-1)
    return random_fourier_pe(P, grid).permute(2, 0, 1)[None]


def prompt_encoder(P, cfg: dict, points: Optional[Tuple[Tensor, Tensor]], boxes: Optional[Tensor],
                   masks: Optional[Tensor], pre: str = "sam_prompt_encoder") -> Tuple[Tensor, Tensor]:
    """PromptEncoder.forward (prompt_encoder.py:140-190) with shim S2 (dense output at image_embedding_size)."""
    S = cfg["image_size"]
    E = S // 16
    bs = points[0].shape[0] if points is not None else (boxes.shape[0] if boxes is not None else
                                                         (masks.shape[0] if masks is not None else 1))
    sparse = torch.empty(bs, 0, cfg["d_model"])
    if points is not None:
        xy, lab = points
        xy = xy + 0.5
        if boxes is None:
            xy = torch.cat([xy, torch.zeros(bs, 1, 2)], dim=1)
            lab = torch.cat([lab, -torch.ones(bs, 1, dtype=lab.dtype)], dim=1)
        e = random_fourier_pe(P, xy.float() / S)
        e = torch.where((lab == -1)[..., None], P[pre + ".not_a_point_embed.weight"].expand_as(e), e)
        for c in range(4):
            e = e + (lab == c)[..., None] * P[f"{pre}.point_embeddings.{c}.weight"]
        sparse = torch.cat([sparse, e], dim=1)
    if boxes is not None:
        c = random_fourier_pe(P, (boxes + 0.5).reshape(-1, 2, 2) / S)
        c = c + torch.stack([P[pre + ".point_embeddings.2.weight"][0], P[pre + ".point_embeddings.3.weight"][0]])
        sparse = torch.cat([sparse, c], dim=1)
    if masks is not None:
        md = pre + ".mask_downscaling"
        d = F.conv2d(masks, P[md + ".0.weight"], P[md + ".0.bias"], stride=2)
        d = gelu(lnorm2d(P, md + ".1", d))
        d = F.conv2d(d, P[md + ".3.weight"], P[md + ".3.bias"], stride=2)
        d = gelu(lnorm2d(P, md + ".4", d))
        dense = F.conv2d(d, P[md + ".6.weight"], P[md + ".6.bias"])
    else:
        dense = P[pre + ".no_mask_embed.weight"].reshape(1, -1, 1, 1).expand(bs, -1, E, E)
    return sparse, dense


# --------------------------------------------------------------------------------------------------------------------
# two-way transformer + mask decoder (a-15, a-16, a-17)
# --------------------------------------------------------------------------------------------------------------------
def mh_attention(P, pre: str, q: Tensor, k: Tensor, v: Tensor, heads: int = 8) -> Tensor:
    """Attention.forward (transformer.py:239-263)."""
    q, k, v = lin(P, pre + ".q_proj", q), lin(P, pre + ".k_proj", k), lin(P, pre + ".v_proj", v)
    B, _, C = q.shape

    def split(t):
        return t.reshape(B, t.shape[1], heads, C // heads).transpose(1, 2)

    o = softmax_attention(split(q), split(k), split(v)).transpose(1, 2).reshape(B, q.shape[1], C)
    return lin(P, pre + ".out_proj", o)


def two_way_transformer(P, pre: str, src: Tensor, pos_src: Tensor, tokens: Tensor) -> Tuple[Tensor, Tensor]:
    """TwoWayTransformer.forward (transformer.py:74-118) with its two TwoWayAttentionBlocks (165-196)."""
    keys = src.flatten(2).permute(0, 2, 1)
    kpe = pos_src.flatten(2).permute(0, 2, 1)
    queries, qpe = tokens, tokens
    for i in range(2):
        lp = f"{pre}.layers.{i}"
        if i == 0:
            queries = mh_attention(P, lp + ".self_attn", queries, queries, queries)
        else:
            q = queries + qpe
            queries = queries + mh_attention(P, lp + ".self_attn", q, q, queries)
        queries = lnorm(P, lp + ".norm1", queries, 1e-5)
        queries = queries + mh_attention(P, lp + ".cross_attn_token_to_image", queries + qpe, keys + kpe, keys)
        queries = lnorm(P, lp + ".norm2", queries, 1e-5)
        queries = queries + mlp(P, lp + ".mlp", queries, 2, torch.relu)
        queries = lnorm(P, lp + ".norm3", queries, 1e-5)
        keys = keys + mh_attention(P, lp + ".cross_attn_image_to_token", keys + kpe, queries + qpe, queries)
        keys = lnorm(P, lp + ".norm4", keys, 1e-5)
    queries = queries + mh_attention(P, pre + ".final_attn_token_to_image", queries + qpe, keys + kpe, keys)
    return lnorm(P, pre + ".norm_final_attn", queries, 1e-5), keys


def mask_decoder_predict(P, image_embeddings: Tensor, image_pe: Tensor, sparse: Tensor, dense: Tensor,
                         high_res: Sequence[Tensor], pre: str = "sam_mask_decoder"):
    """MaskDecoder.predict_masks (mask_decoder.py:170-267) with cell_nums=None (shim S3)."""
    out_tok = torch.cat([P[pre + ".obj_score_token.weight"], P[pre + ".iou_token.weight"],
                         P[pre + ".mask_tokens.weight"]], dim=0)
    tokens = torch.cat([out_tok[None].expand(sparse.shape[0], -1, -1), sparse], dim=1)
    src = image_embeddings + dense
    b, c, h, w = src.shape
    hs, keys = two_way_transformer(P, pre + ".transformer", src, image_pe, tokens)
    iou_tok, mask_toks = hs[:, 1], hs[:, 2:6]
    src = keys.transpose(1, 2).reshape(b, c, h, w)
    feat_s0, feat_s1 = high_res
    up = pre + ".output_upscaling"
    u = F.conv_transpose2d(_rop(src), _rop(P[up + ".0.weight"]), P[up + ".0.bias"], stride=2) + feat_s1
    u = gelu(lnorm2d(P, up + ".1", u))
    u = gelu(F.conv_transpose2d(_rop(u), _rop(P[up + ".3.weight"]), P[up + ".3.bias"], stride=2) + feat_s0)
    hyper = torch.stack([mlp(P, f"{pre}.output_hypernetworks_mlps.{i}", mask_toks[:, i], 3, torch.relu)
                         for i in range(4)], dim=1)
    b, c, h, w = u.shape
    masks = (hyper @ u.reshape(b, c, h * w)).reshape(b, -1, h, w)
    iou = mlp(P, pre + ".iou_prediction_head", iou_tok, 3, torch.relu, sigmoid_out=True)
    obj = mlp(P, pre + ".pred_obj_score_head", hs[:, 0], 3, torch.relu)
    return masks, iou, mask_toks, obj


def mask_decoder(P, cfg: dict, image_embeddings, image_pe, sparse, dense, multimask_output: bool, high_res):
    """MaskDecoder.forward (mask_decoder.py:110-168) in eval mode."""
    masks, iou, mask_toks, obj = mask_decoder_predict(P, image_embeddings, image_pe, sparse, dense, high_res)
    if multimask_output:
        masks, iou = masks[:, 1:], iou[:, 1:]
    elif cfg["dynamic_multimask_via_stability"]:
        masks, iou = dynamic_multimask_via_stability(cfg, masks, iou)
    else:
        masks, iou = masks[:, 0:1], iou[:, 0:1]
    toks = mask_toks[:, 1:] if multimask_output else mask_toks[:, 0:1]  # use_multimask_token_for_obj_ptr=True
    return masks, iou, toks, obj


def dynamic_multimask_via_stability(cfg: dict, all_masks: Tensor, all_iou: Tensor):
    """mask_decoder.py:269-317."""
    d = cfg["dynamic_multimask_stability_delta"]
    mm, mi = all_masks[:, 1:], all_iou[:, 1:]
    best = mi.argmax(-1)
    ar = torch.arange(mi.shape[0])
    best_m, best_i = mm[ar, best][:, None], mi[ar, best][:, None]
    single, single_i = all_masks[:, 0:1], all_iou[:, 0:1]
    flat = single.flatten(-2)
    area_i = (flat > d).sum(-1).float()
    area_u = (flat > -d).sum(-1).float()
    stab = torch.where(area_u > 0, area_i / area_u, torch.ones_like(area_u))
    ok = stab >= cfg["dynamic_multimask_stability_thresh"]
    return (torch.where(ok[..., None, None].expand_as(single), single, best_m),
            torch.where(ok.expand_as(single_i), single_i, best_i))


def forward_sam_heads(P, cfg: dict, backbone_features: Tensor, point_inputs: Optional[dict], mask_inputs: Optional[Tensor],
                      high_res_features: Sequence[Tensor], multimask_output: bool):
    """SAM2Base._forward_sam_heads (sam2_base.py:252-410)."""
    B = backbone_features.shape[0]
    S = cfg["image_size"]
    if point_inputs is not None:
        xy, lab = point_inputs["point_coords"], point_inputs["point_labels"]
    else:
        xy, lab = torch.zeros(B, 1, 2), -torch.ones(B, 1, dtype=torch.int32)
    mask_prompt = None
    if mask_inputs is not None:
        mask_prompt = mask_inputs
        if tuple(mask_inputs.shape[-2:]) != (S // 4, S // 4):
            mask_prompt = F.interpolate(mask_inputs.float(), size=(S // 4, S // 4), align_corners=False,
                                        mode="bilinear", antialias=True)
    sparse, dense = prompt_encoder(P, cfg, (xy, lab), None, mask_prompt)
    E = S // 16
    low_multi, ious, toks, obj = mask_decoder(P, cfg, backbone_features, dense_pe(P, E, E), sparse, dense,
                                              multimask_output, high_res_features)
    appearing = obj > 0
    low_multi = torch.where(appearing[:, None, None], low_multi, torch.full_like(low_multi, NO_OBJ_SCORE)).float()
    high_multi = F.interpolate(low_multi, size=(S, S), mode="bilinear", align_corners=False)
    tok = toks[:, 0]
    if multimask_output:
        best = ious.argmax(-1)
        ar = torch.arange(B)
        low, high = low_multi[ar, best][:, None], high_multi[ar, best][:, None]
        if toks.shape[1] > 1:
            tok = toks[ar, best]
    else:
        low, high = low_multi, high_multi
    ptr = mlp(P, "obj_ptr_proj", tok, 3, torch.relu)
    lam = appearing.float()
    ptr = lam * ptr + (1 - lam) * P["no_obj_ptr"]  # fixed_no_obj_ptr=True
    return low_multi, high_multi, ious, low, high, ptr, obj


# --------------------------------------------------------------------------------------------------------------------
# memory encoder (a-19)
# --------------------------------------------------------------------------------------------------------------------
def memory_encoder(P, cfg: dict, pix_feat: Tensor, mask_for_mem: Tensor, pre: str = "memory_encoder"):
    """MemoryEncoder.forward (memory_encoder.py:158-181) with skip_mask_sigmoid=True; MaskDownSampler 17-58,
    CXBlock 62-117."""
    m = mask_for_mem
    ds = pre + ".mask_downsampler.encoder"
    for j in range(4):
        m = F.conv2d(m, P[f"{ds}.{3 * j}.weight"], P[f"{ds}.{3 * j}.bias"], stride=2, padding=1)
        m = gelu(lnorm2d(P, f"{ds}.{3 * j + 1}", m))
    m = F.conv2d(m, P[f"{ds}.12.weight"], P[f"{ds}.12.bias"])
    x = conv1x1(P, pre + ".pix_feat_proj", pix_feat) + m
    for j in range(2):
        lp = f"{pre}.fuser.layers.{j}"
        h = F.conv2d(x, P[lp + ".dwconv.weight"], P[lp + ".dwconv.bias"], padding=3, groups=x.shape[1])
        h = lnorm2d(P, lp + ".norm", h).permute(0, 2, 3, 1)
        h = lin(P, lp + ".pwconv2", gelu(lin(P, lp + ".pwconv1", h))) * P[lp + ".gamma"]
        x = x + h.permute(0, 3, 1, 2)
    x = conv1x1(P, pre + ".out_proj", x)
    pos = sine_pos_2d(x.shape[-2], x.shape[-1], cfg["mem_dim"])[None].expand(x.shape[0], -1, -1, -1)
    return x, pos


def encode_new_memory(P, cfg: dict, vision_feat_top: Tensor, feat_size: Tuple[int, int], pred_masks_high_res: Tensor,
                      is_mask_from_pts: bool):
    """SAM2Base._encode_new_memory (sam2_base.py:665-703), eval mode; vision_feat_top is [HW, B, C]."""
    B, C = vision_feat_top.shape[1], vision_feat_top.shape[2]
    pix = vision_feat_top.permute(1, 2, 0).reshape(B, C, *feat_size)
    if cfg["binarize_mask_from_pts_for_mem_enc"] and is_mask_from_pts:
        m = (pred_masks_high_res > 0).float()
    else:
        m = torch.sigmoid(pred_masks_high_res)
    m = m * cfg["sigmoid_scale_for_mem_enc"] + cfg["sigmoid_bias_for_mem_enc"]
    return memory_encoder(P, cfg, pix, m)


# --------------------------------------------------------------------------------------------------------------------
# memory-bank assembly + track_step (a-11, a-21)
# --------------------------------------------------------------------------------------------------------------------
def prepare_memory_conditioned_features(P, cfg: dict, frame_idx: int, is_init_cond_frame: bool, vision_feat_top: Tensor,
                                        vision_pos_top: Tensor, feat_size: Tuple[int, int], output_dict: dict,
                                        num_frames: int, collect: Optional[dict] = None) -> Tensor:
    """SAM2Base._prepare_memory_conditioned_features (sam2_base.py:494-663) for the YAML's settings: all cond frames
    attended (max_cond_frames_in_attn=-1), stride 1, forward tracking, eval mode, pointers past-only, no tpos on
    pointers, directly_add_no_mem_embed."""
    B, C = vision_feat_top.shape[1], cfg["d_model"]
    H, W = feat_size
    if is_init_cond_frame:
        return (vision_feat_top + P["no_mem_embed"]).permute(1, 2, 0).reshape(B, C, H, W)
    nm = cfg["num_maskmem"]
    cond = output_dict["cond_frame_outputs"]
    noncond = output_dict["non_cond_frame_outputs"]
    picks = [(0, o) for o in cond.values()]
    for t_pos in range(1, nm):
        picks.append((t_pos, noncond.get(frame_idx - (nm - t_pos))))
    mem, mem_pos = [], []
    for t_pos, prev in picks:
        if prev is None:
            continue
        mem.append(prev["maskmem_features"].flatten(2).permute(2, 0, 1))
        enc = prev["maskmem_pos_enc"][-1].flatten(2).permute(2, 0, 1)
        mem_pos.append(enc + P["maskmem_tpos_enc"][nm - t_pos - 1])
    ptrs = [o["obj_ptr"] for t, o in cond.items() if t <= frame_idx]
    for t_diff in range(1, min(num_frames, cfg["max_obj_ptrs_in_encoder"])):
        t = frame_idx - t_diff
        if t < 0:
            break
        if t in noncond:
            ptrs.append(noncond[t]["obj_ptr"])
    n_ptr_tokens = 0
    if ptrs:
        md = cfg["mem_dim"]
        op = torch.stack(ptrs, dim=0).reshape(-1, B, C // md, md).permute(0, 2, 1, 3).flatten(0, 1)
        mem.append(op)
        mem_pos.append(torch.zeros_like(op))
        n_ptr_tokens = op.shape[0]
    memory, memory_pos = torch.cat(mem, dim=0), torch.cat(mem_pos, dim=0)
    if collect is not None:
        collect["memory_shape"] = tuple(memory.shape)
        collect["num_obj_ptr_tokens"] = n_ptr_tokens
    out = memory_attention(P, cfg, vision_feat_top, memory, vision_pos_top, memory_pos, n_ptr_tokens)
    return out.permute(1, 2, 0).reshape(B, C, H, W)


def use_mask_as_output(P, cfg: dict, pix_feat: Tensor, high_res_features, mask_inputs: Tensor):
    """SAM2Base._use_mask_as_output (sam2_base.py:412-462)."""
    mf = mask_inputs.float()
    high = mf * 20.0 - 10.0
    low = F.interpolate(high, size=(high.shape[-2] // 4, high.shape[-1] // 4), align_corners=False, mode="bilinear",
                        antialias=True)
    ious = torch.ones(mask_inputs.shape[0], 1)
    md = F.conv2d(mf, P["mask_downsample.weight"], P["mask_downsample.bias"], stride=4)
    ptr = forward_sam_heads(P, cfg, pix_feat, None, md, high_res_features, False)[5]
    lam = (mask_inputs.flatten(1).float() > 0).any(dim=1)[..., None].float()
    obj = 20.0 * lam - 10.0
    ptr = lam * ptr + (1 - lam) * P["no_obj_ptr"]
    return low, high, ious, low, high, ptr, obj


def track_step(P, cfg: dict, frame_idx: int, is_init_cond_frame: bool, vision_feats: Sequence[Tensor],
               vision_pos: Sequence[Tensor], feat_sizes: Sequence[Tuple[int, int]], point_inputs: Optional[dict],
               mask_inputs: Optional[Tensor], output_dict: dict, num_frames: int, run_mem_encoder: bool = True,
               collect: Optional[dict] = None) -> dict:
    """SAM2Base.track_step (sam2_base.py:705-800)."""
    high_res = [x.permute(1, 2, 0).reshape(x.shape[1], x.shape[2], *s) for x, s in zip(vision_feats[:-1], feat_sizes[:-1])]
    if mask_inputs is not None:
        pix = vision_feats[-1].permute(1, 2, 0).reshape(-1, cfg["d_model"], *feat_sizes[-1])
        sam = use_mask_as_output(P, cfg, pix, high_res, mask_inputs)
    else:
        pix = prepare_memory_conditioned_features(P, cfg, frame_idx, is_init_cond_frame, vision_feats[-1], vision_pos[-1],
                                                  feat_sizes[-1], output_dict, num_frames, collect=collect)
        if collect is not None:
            collect["pix_feat_with_mem"] = pix
        n_pts = 0 if point_inputs is None else point_inputs["point_labels"].shape[1]
        multimask = cfg["multimask_min_pt_num"] <= n_pts <= cfg["multimask_max_pt_num"]  # sam2_base.py:802-810
        sam = forward_sam_heads(P, cfg, pix, point_inputs, None, high_res, multimask)
    _, _, _, low, high, ptr, _ = sam
    out = {"pred_masks": low, "pred_masks_high_res": high, "obj_ptr": ptr, "maskmem_features": None, "maskmem_pos_enc": None}
    if run_mem_encoder:
        feats, pos = encode_new_memory(P, cfg, vision_feats[-1], feat_sizes[-1], high, point_inputs is not None)
        out["maskmem_features"], out["maskmem_pos_enc"] = feats, [pos]
    return out


# --------------------------------------------------------------------------------------------------------------------
# hole filling (a-20): labels/areas come from oracle/cc_oracle.c (or the python mirror below for tiny cases)
# --------------------------------------------------------------------------------------------------------------------
def fill_holes_in_mask_scores(mask: Tensor, max_area: int, cc_fn) -> Tensor:
    """utils/misc.py:247-258; cc_fn(uint8 [N,1,H,W]) -> (labels, areas)."""
    labels, areas = cc_fn((mask <= 0).to(torch.uint8))
    return torch.where((labels > 0) & (areas <= max_area), torch.full_like(mask, 0.1), mask)
