"""State-dict contract of the reference SAM2 model and a deterministic, name-keyed weight initialiser.

The key names and shapes restate what ``SAM2Base(...).state_dict()`` holds when built from the reference's YAML
(``sam2_train/sam2_hiera_s.yaml``) so that ``build_sam.py:79-89``'s strict checkpoint load works
unchanged against the drop-in modules.  ``tests/test_oracle_golden.py`` checks this table against the key/shape list
captured from the reference itself.

No checkpoints are available offline, so tests and the benchmark use ``init_weights``: every tensor is drawn from a
generator seeded by ``sha256(f"{seed}:{name}")`` -- identical on every machine, independent of construction order.
"""
from __future__ import annotations

import hashlib
from collections import OrderedDict
from typing import Dict, Tuple

import torch


def trunk_config(name: str) -> dict:
    """Hiera hyper-parameters: ``sam2_hiera_{t,s}.yaml:11-15``; b+ = class defaults ``hieradet.py:176-201`` with the
    upstream embed_dim/num_heads (assumption recorded in SURVEY.md section 0.3)."""
    table = {
        "hiera_t": dict(embed_dim=96, num_heads=1, stages=(1, 2, 7, 2), global_att_blocks=(5, 7, 9), bkg=(7, 7)),
        "hiera_s": dict(embed_dim=96, num_heads=1, stages=(1, 2, 11, 2), global_att_blocks=(7, 10, 13), bkg=(7, 7)),
        "hiera_b+": dict(embed_dim=112, num_heads=2, stages=(2, 3, 16, 3), global_att_blocks=(12, 16, 20), bkg=(14, 14)),
    }
    return dict(table[name], window_spec=(8, 4, 14, 7), q_pool=3, q_stride=2)


def block_table(tc: dict):
    """(dim, dim_out, heads, window, pool, stage_end) per block; the rule of ``hieradet.py:229-257``."""
    stages = tc["stages"]
    ends = [sum(stages[:i]) - 1 for i in range(1, len(stages) + 1)]
    pool_blocks = [e + 1 for e in ends[:-1]][: tc["q_pool"]]
    dim, heads, stage = tc["embed_dim"], tc["num_heads"], 1
    rows = []
    for i in range(sum(stages)):
        window = 0 if i in tc["global_att_blocks"] else tc["window_spec"][stage - 1]
        dim_out = dim
        if i - 1 in ends:
            dim_out, heads, stage = dim * 2, heads * 2, stage + 1
        rows.append(dict(dim=dim, dim_out=dim_out, heads=heads, window=window, pool=i in pool_blocks, stage_end=i in ends))
        dim = dim_out
    return rows


def state_dict_spec(model: str = "hiera_s") -> "OrderedDict[str, Tuple[int, ...]]":
    tc = trunk_config(model)
    S: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()

    def lin(pre, out_f, in_f):
        S[pre + ".weight"] = (out_f, in_f)
        S[pre + ".bias"] = (out_f,)

    def norm(pre, c):
        S[pre + ".weight"] = (c,)
        S[pre + ".bias"] = (c,)

    def conv(pre, out_c, in_c, k):
        S[pre + ".weight"] = (out_c, in_c, k, k)
        S[pre + ".bias"] = (out_c,)

    S["maskmem_tpos_enc"] = (7, 1, 1, 64)
    S["no_mem_embed"] = (1, 1, 256)
    S["no_mem_pos_enc"] = (1, 1, 256)
    S["no_obj_ptr"] = (1, 256)
    # image encoder
    E = tc["embed_dim"]
    t = "image_encoder.trunk"
    S[t + ".pos_embed"] = (1, E, *tc["bkg"])
    S[t + ".pos_embed_window"] = (1, E, 8, 8)
    conv(t + ".patch_embed.proj", E, 3, 7)
    rows = block_table(tc)
    for i, r in enumerate(rows):
        b = f"{t}.blocks.{i}"
        norm(b + ".norm1", r["dim"])
        lin(b + ".attn.qkv", 3 * r["dim_out"], r["dim"])
        lin(b + ".attn.proj", r["dim_out"], r["dim_out"])
        norm(b + ".norm2", r["dim_out"])
        lin(b + ".mlp.layers.0", 4 * r["dim_out"], r["dim_out"])
        lin(b + ".mlp.layers.1", r["dim_out"], 4 * r["dim_out"])
        if r["dim"] != r["dim_out"]:
            lin(b + ".proj", r["dim_out"], r["dim"])
    chans = [rows[i]["dim_out"] for i, r in enumerate(rows) if r["stage_end"]][::-1]
    for j, c in enumerate(chans):
        conv(f"image_encoder.neck.convs.{j}.conv", 256, c, 1)
    conv("mask_downsample", 1, 1, 4)
    # memory attention
    for l in range(4):
        p = f"memory_attention.layers.{l}"
        for a, kv in (("self_attn", 256), ("cross_attn_image", 64)):
            lin(f"{p}.{a}.q_proj", 256, 256)
            lin(f"{p}.{a}.k_proj", 256, kv)
            lin(f"{p}.{a}.v_proj", 256, kv)
            lin(f"{p}.{a}.out_proj", 256, 256)
        lin(p + ".linear1", 2048, 256)
        lin(p + ".linear2", 256, 2048)
        for n in ("norm1", "norm2", "norm3"):
            norm(f"{p}.{n}", 256)
    norm("memory_attention.norm", 256)
    # memory encoder
    me = "memory_encoder"
    cin = 1
    for j in range(4):
        conv(f"{me}.mask_downsampler.encoder.{3 * j}", cin * 4, cin, 3)
        norm(f"{me}.mask_downsampler.encoder.{3 * j + 1}", cin * 4)
        cin *= 4
    conv(f"{me}.mask_downsampler.encoder.12", 256, 256, 1)
    conv(f"{me}.pix_feat_proj", 256, 256, 1)
    for j in range(2):
        f = f"{me}.fuser.layers.{j}"
        S[f + ".gamma"] = (256,)
        S[f + ".dwconv.weight"] = (256, 1, 7, 7)
        S[f + ".dwconv.bias"] = (256,)
        norm(f + ".norm", 256)
        lin(f + ".pwconv1", 1024, 256)
        lin(f + ".pwconv2", 256, 1024)
    conv(f"{me}.out_proj", 64, 256, 1)
    # prompt encoder
    pe = "sam_prompt_encoder"
    S[pe + ".pe_layer.positional_encoding_gaussian_matrix"] = (2, 128)
    for j in range(4):
        S[f"{pe}.point_embeddings.{j}.weight"] = (1, 256)
    S[pe + ".not_a_point_embed.weight"] = (1, 256)
    conv(pe + ".mask_downscaling.0", 4, 1, 2)
    norm(pe + ".mask_downscaling.1", 4)
    conv(pe + ".mask_downscaling.3", 16, 4, 2)
    norm(pe + ".mask_downscaling.4", 16)
    conv(pe + ".mask_downscaling.6", 256, 16, 1)
    S[pe + ".no_mask_embed.weight"] = (1, 256)
    # mask decoder
    md = "sam_mask_decoder"

    def attn(pre, internal):
        lin(pre + ".q_proj", internal, 256)
        lin(pre + ".k_proj", internal, 256)
        lin(pre + ".v_proj", internal, 256)
        lin(pre + ".out_proj", 256, internal)

    for l in range(2):
        p = f"{md}.transformer.layers.{l}"
        attn(p + ".self_attn", 256)
        norm(p + ".norm1", 256)
        attn(p + ".cross_attn_token_to_image", 128)
        norm(p + ".norm2", 256)
        lin(p + ".mlp.layers.0", 2048, 256)
        lin(p + ".mlp.layers.1", 256, 2048)
        norm(p + ".norm3", 256)
        norm(p + ".norm4", 256)
        attn(p + ".cross_attn_image_to_token", 128)
    attn(f"{md}.transformer.final_attn_token_to_image", 128)
    norm(f"{md}.transformer.norm_final_attn", 256)
    S[md + ".iou_token.weight"] = (1, 256)
    S[md + ".mask_tokens.weight"] = (4, 256)
    S[md + ".obj_score_token.weight"] = (1, 256)
    S[md + ".output_upscaling.0.weight"] = (256, 64, 2, 2)
    S[md + ".output_upscaling.0.bias"] = (64,)
    norm(md + ".output_upscaling.1", 64)
    S[md + ".output_upscaling.3.weight"] = (64, 32, 2, 2)
    S[md + ".output_upscaling.3.bias"] = (32,)
    conv(md + ".conv_s0", 32, 256, 1)
    conv(md + ".conv_s1", 64, 256, 1)
    for j in range(4):
        h = f"{md}.output_hypernetworks_mlps.{j}"
        lin(h + ".layers.0", 256, 256)
        lin(h + ".layers.1", 256, 256)
        lin(h + ".layers.2", 32, 256)
    lin(md + ".iou_prediction_head.layers.0", 256, 256)
    lin(md + ".iou_prediction_head.layers.1", 256, 256)
    lin(md + ".iou_prediction_head.layers.2", 4, 256)
    lin(md + ".pred_obj_score_head.layers.0", 256, 256)
    lin(md + ".pred_obj_score_head.layers.1", 256, 256)
    lin(md + ".pred_obj_score_head.layers.2", 1, 256)
    for j in range(3):
        lin(f"obj_ptr_proj.layers.{j}", 256, 256)
    return S


_NORM_TAGS = (".norm", ".norm1", ".norm2", ".norm3", ".norm4", ".norm_final_attn", ".mask_downscaling.1",
              ".mask_downscaling.4", ".output_upscaling.1", ".encoder.1", ".encoder.4", ".encoder.7", ".encoder.10")


def _is_norm_weight(name: str) -> bool:
    return name.endswith(".weight") and name[: -len(".weight")].endswith(_NORM_TAGS)


def init_tensor(name: str, shape: Tuple[int, ...], seed: int = 0) -> torch.Tensor:
    """SURVEY.md Appendix D convention: dim>=2 -> randn/sqrt(fan_in); norm weights -> 1+0.1 randn; else 0.02 randn."""
    g = torch.Generator().manual_seed(int(hashlib.sha256(f"{seed}:{name}".encode()).hexdigest()[:8], 16))
    r = torch.randn(shape, generator=g, dtype=torch.float32)
    if _is_norm_weight(name):
        return 1.0 + 0.1 * r
    if name.endswith("positional_encoding_gaussian_matrix"):
        return r  # unit-variance random Fourier frequencies, as position_encoding.py:120-127 draws them
    if len(shape) >= 2:
        fan_in = 1
        for d in shape[1:]:
            fan_in *= d
        return r / max(fan_in, 1) ** 0.5
    return 0.02 * r


def init_weights(model: str = "hiera_s", seed: int = 0) -> Dict[str, torch.Tensor]:
    return OrderedDict((k, init_tensor(k, shp, seed)) for k, shp in state_dict_spec(model).items())
