"""Drop-in for sam2_train/build_sam.py:15-89: `build_sam2(config_file, ckpt_path, device, mode, hydra_overrides_extra,
apply_postprocessing)` without Hydra.  `config_file` is either one of the built-in names ("sam2_hiera_t", "sam2_hiera_s",
"sam2_hiera_b+") or a path to a YAML in the reference's schema (nested `_target_` nodes), which is instantiated through
`modeling.TARGETS`.  Checkpoints are loaded strictly from `ckpt["model"]`, like the reference."""
from __future__ import annotations

import copy
import logging
import os
import re
from typing import Any, Dict, List

import torch

from .modeling import TARGETS
from .weights import trunk_config

_P = "sam2_train.modeling."
_FLOAT = re.compile(r"[-+]?(\d+\.?\d*|\.\d+)[eE][-+]?\d+")


def builtin_config(name: str) -> Dict[str, Any]:
    """The YAML leaves of sam2_train/sam2_hiera_{t,s}.yaml as a dict (b+ uses the Hiera class defaults)."""
    tc = trunk_config(name.replace("sam2_", ""))
    pos = lambda n: {"_target_": _P + "position_encoding.PositionEmbeddingSine", "num_pos_feats": n, "normalize": True,
                     "scale": None, "temperature": 10000}
    rope = lambda **kw: dict({"_target_": _P + "sam.transformer.RoPEAttention", "rope_theta": 10000.0, "feat_sizes": [32, 32],
                              "embedding_dim": 256, "num_heads": 1, "downsample_rate": 1, "dropout": 0.1}, **kw)
    dims = [tc["embed_dim"] * 2 ** i for i in range(4)][::-1]
    return {"model": {
        "_target_": _P + "sam2_base.SAM2Base",
        "image_encoder": {
            "_target_": _P + "backbones.image_encoder.ImageEncoder", "scalp": 1,
            "trunk": {"_target_": _P + "backbones.hieradet.Hiera", "embed_dim": tc["embed_dim"], "num_heads": tc["num_heads"],
                      "stages": list(tc["stages"]), "global_att_blocks": list(tc["global_att_blocks"]),
                      "window_pos_embed_bkg_spatial_size": list(tc["bkg"])},
            "neck": {"_target_": _P + "backbones.image_encoder.FpnNeck", "position_encoding": pos(256), "d_model": 256,
                     "backbone_channel_list": dims, "fpn_top_down_levels": [2, 3], "fpn_interp_model": "nearest"}},
        "memory_attention": {
            "_target_": _P + "memory_attention.MemoryAttention", "d_model": 256, "pos_enc_at_input": True, "num_layers": 4,
            "layer": {"_target_": _P + "memory_attention.MemoryAttentionLayer", "activation": "relu", "dim_feedforward": 2048,
                      "dropout": 0.1, "pos_enc_at_attn": False, "self_attention": rope(), "d_model": 256,
                      "pos_enc_at_cross_attn_keys": True, "pos_enc_at_cross_attn_queries": False,
                      "cross_attention": rope(rope_k_repeat=True, kv_in_dim=64)}},
        "memory_encoder": {
            "_target_": _P + "memory_encoder.MemoryEncoder", "out_dim": 64, "position_encoding": pos(64),
            "mask_downsampler": {"_target_": _P + "memory_encoder.MaskDownSampler", "kernel_size": 3, "stride": 2, "padding": 1},
            "fuser": {"_target_": _P + "memory_encoder.Fuser", "num_layers": 2,
                      "layer": {"_target_": _P + "memory_encoder.CXBlock", "dim": 256, "kernel_size": 7, "padding": 3,
                                "layer_scale_init_value": 1e-6, "use_dwconv": True}}},
        "num_maskmem": 7, "image_size": 1024, "sigmoid_scale_for_mem_enc": 20.0, "sigmoid_bias_for_mem_enc": -10.0,
        "use_mask_input_as_output_without_sam": True, "directly_add_no_mem_embed": True, "use_high_res_features_in_sam": True,
        "multimask_output_in_sam": True, "iou_prediction_use_sigmoid": True, "use_obj_ptrs_in_encoder": True,
        "add_tpos_enc_to_obj_ptrs": False, "only_obj_ptrs_in_the_past_for_eval": True, "pred_obj_scores": True,
        "pred_obj_scores_mlp": True, "fixed_no_obj_ptr": True, "multimask_output_for_tracking": True,
        "use_multimask_token_for_obj_ptr": True, "multimask_min_pt_num": 0, "multimask_max_pt_num": 1,
        "use_mlp_for_obj_ptr_proj": True, "compile_image_encoder": False}}


def instantiate(node: Any) -> Any:
    """Minimal `hydra.utils.instantiate(cfg, _recursive_=True)`: dicts with `_target_` become objects, depth first."""
    if isinstance(node, dict):
        kw = {k: instantiate(v) for k, v in node.items() if k != "_target_"}
        if "_target_" in node:
            if node["_target_"] not in TARGETS:
                raise KeyError(f"no MI355X drop-in registered for {node['_target_']}")
            return TARGETS[node["_target_"]](**kw)
        return kw
    if isinstance(node, (list, tuple)):
        return [instantiate(v) for v in node]
    if isinstance(node, str) and _FLOAT.fullmatch(node):
        return float(node)  # PyYAML reads "1e-6" (no dot) as a string; OmegaConf reads it as a float
    return node


def _parse_value(text: str) -> Any:
    import yaml
    return yaml.safe_load(text)


def apply_overrides(cfg: Dict[str, Any], overrides: List[str]) -> None:
    """`++a.b.c=value` / `a.b=value` overrides (the subset of Hydra's grammar build_sam.py uses)."""
    for ov in overrides:
        key, _, val = ov.lstrip("+").partition("=")
        node = cfg
        parts = key.split(".")
        for p in parts[:-1]:
            node = node.setdefault(p, {})
        node[parts[-1]] = _parse_value(val)


def _load_config(config_file: str) -> Dict[str, Any]:
    base = os.path.basename(config_file).replace(".yaml", "")
    if not os.path.exists(config_file) and base in ("sam2_hiera_t", "sam2_hiera_s", "sam2_hiera_b+"):
        return builtin_config(base)
    import yaml
    with open(config_file) as f:
        return yaml.safe_load(f)


def build_sam2(config_file, ckpt_path=None, device="cuda", mode="eval", hydra_overrides_extra=[], apply_postprocessing=True):
    overrides = list(hydra_overrides_extra)
    if apply_postprocessing:
        overrides += ["++model.sam_mask_decoder_extra_args.dynamic_multimask_via_stability=true",
                      "++model.sam_mask_decoder_extra_args.dynamic_multimask_stability_delta=0.05",
                      "++model.sam_mask_decoder_extra_args.dynamic_multimask_stability_thresh=0.98"]
    cfg = copy.deepcopy(_load_config(config_file))
    apply_overrides(cfg, overrides)
    model = instantiate(cfg["model"])
    _load_checkpoint(model, ckpt_path)
    model = model.to(device)
    if mode == "eval":
        model.eval()
    return model


def build_sam2_video_predictor(config_file, ckpt_path=None, device="cuda", mode="eval", hydra_overrides_extra=[], apply_postprocessing=True):
    """sam2_train/build_sam.py:43-76: same config with `model._target_` re-pointed at the video predictor and, with
    post-processing, dynamic multimask fallback + binarised click masks for the memory encoder + `fill_hole_area=8`."""
    from .video_predictor import SAM2VideoPredictor
    TARGETS.setdefault("sam2_train.sam2_video_predictor.SAM2VideoPredictor", SAM2VideoPredictor)
    overrides = ["++model._target_=sam2_train.sam2_video_predictor.SAM2VideoPredictor"]
    extra = list(hydra_overrides_extra)
    if apply_postprocessing:
        extra += ["++model.sam_mask_decoder_extra_args.dynamic_multimask_via_stability=true",
                  "++model.sam_mask_decoder_extra_args.dynamic_multimask_stability_delta=0.05",
                  "++model.sam_mask_decoder_extra_args.dynamic_multimask_stability_thresh=0.98",
                  "++model.binarize_mask_from_pts_for_mem_enc=true",
                  "++model.fill_hole_area=8"]
    cfg = copy.deepcopy(_load_config(config_file))
    apply_overrides(cfg, overrides + extra)
    model = instantiate(cfg["model"])
    _load_checkpoint(model, ckpt_path)
    model = model.to(device)
    if mode == "eval":
        model.eval()
    return model


def _load_checkpoint(model, ckpt_path):
    if ckpt_path is not None:
        sd = torch.load(ckpt_path, map_location="cpu")["model"]
        missing_keys, unexpected_keys = model.load_state_dict(sd)
        if missing_keys:
            logging.error(missing_keys)
            raise RuntimeError()
        if unexpected_keys:
            logging.error(unexpected_keys)
            raise RuntimeError()
        logging.info("Loaded checkpoint sucessfully")
