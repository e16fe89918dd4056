"""3-D volume driver for BASELINE.json configs 3/4: bbox (or click) prompts on every `prompt_freq`-th slice, propagation through
the remaining slices with the memory bank -- the work `func_3d/function.py:226-274` drives through `SAM2VideoPredictor`
(`val_init_state` -> `train_add_new_bbox` -> `propagate_in_video`), restated against the mirrored `SAM2Base` surface only
(`forward_image`, `_prepare_backbone_features`, `track_step`).

Differences from the reference flow that do not change results: every slice is encoded once (the reference encodes conditioning
slices twice, sam2_video_predictor.py:1378-1380), and conditioning slices run the memory encoder in the same `track_step` call.

Multi-GPU (one process per GPU): conditioning slices are independent -> sharded contiguously over the ranks, then ONE RCCL
all-gather of their memories (`parallel.gather_cond_memories`); the propagation chain is sequential in the slice index and is
replicated, or sharded over objects when there are at least as many objects as ranks.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
import torch.distributed as dist

from . import ops
from .parallel import gather_cond_memories, shard_range


class _SliceEncoder:
    """Backbone features of the slices in the order they will be consumed, encoded `batch` slices per forward_image call (the
    trunk's kernels are far better filled at 8 slices than at 1; every kernel is batch-invariant, so the features are the ones a
    slice-by-slice pass produces) and dropped once consumed: at most `batch` slices (17 MB each at 1024^2) are resident."""

    def __init__(self, model, volume: torch.Tensor, order: List[int], batch: int):
        self.model, self.volume, self.order, self.batch = model, volume, list(order), max(1, int(batch))
        self.pos = 0
        self.cache: Dict[int, dict] = {}

    def get(self, t: int, n_obj: int):
        """features of slice t expanded (views) over the objects like sam2_video_predictor.py:1284-1296"""
        if t not in self.cache:
            assert self.pos < len(self.order) and self.order[self.pos] == t, "slices must be consumed in the announced order"
            ids = self.order[self.pos: self.pos + self.batch]
            self.pos += len(ids)
            bo = self.model.forward_image(self.volume[ids] if len(ids) > 1 else self.volume[ids[0]][None])
            for j, u in enumerate(ids):
                self.cache[u] = {"backbone_fpn": [f[j: j + 1] for f in bo["backbone_fpn"]],
                                 "vision_pos_enc": [p[:1] for p in bo["vision_pos_enc"]]}
        one = self.cache.pop(t)
        bo = {"backbone_fpn": [f.expand(n_obj, -1, -1, -1) for f in one["backbone_fpn"]],
              "vision_pos_enc": [p.expand(n_obj, -1, -1, -1) for p in one["vision_pos_enc"]]}
        _, feats, pos, sizes = self.model._prepare_backbone_features(bo)
        return feats, pos, sizes


def box_point_inputs(boxes: torch.Tensor) -> dict:
    """[n,4] (x0,y0,x1,y1) -> the two-corner point prompt with labels 2/3 (sam2_video_predictor.py:330-345)."""
    n = boxes.shape[0]
    return {"point_coords": boxes.reshape(n, 2, 2).float(),
            "point_labels": torch.tensor([[2, 3]], dtype=torch.int32, device=boxes.device).expand(n, 2).contiguous()}


@torch.no_grad()
def segment_volume(model, volume: torch.Tensor, prompts: Dict[int, dict], fill_hole_area: int = 0, group=None,
                   shard_objects: bool = True, encode_batch: int = 8) -> Dict[int, torch.Tensor]:
    """volume: [T,3,S,S] normalised slices on the GPU; prompts: {slice_idx: {"boxes": [n,4]} | {"point_coords", "point_labels"}}
    for the conditioning slices (same n objects everywhere).  Returns {slice_idx: low-res mask logits [n,1,S/4,S/4]}.
    encode_batch: slices per image-encoder call (results do not depend on it)."""
    T = volume.shape[0]
    cond_ids = sorted(prompts)
    assert cond_ids, "at least one conditioning slice is needed"
    first = prompts[cond_ids[0]]
    n_obj = (first["boxes"] if "boxes" in first else first["point_coords"]).shape[0]
    distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    rank = dist.get_rank(group) if distributed else 0
    world = dist.get_world_size(group) if distributed else 1

    # 1. conditioning slices: independent -> this rank's contiguous share
    b, e = shard_range(len(cond_ids), rank, world)
    empty = {"cond_frame_outputs": {}, "non_cond_frame_outputs": {}}
    local = {}
    enc = _SliceEncoder(model, volume, [cond_ids[i] for i in range(b, e)], encode_batch)
    for i in range(b, e):
        t = cond_ids[i]
        pr = prompts[t]
        pin = box_point_inputs(pr["boxes"]) if "boxes" in pr else {"point_coords": pr["point_coords"], "point_labels": pr["point_labels"]}
        feats, pos, sizes = enc.get(t, n_obj)
        local[t] = model.track_step(frame_idx=t, is_init_cond_frame=True, current_vision_feats=feats, current_vision_pos_embeds=pos,
                                    feat_sizes=sizes, point_inputs=pin, mask_inputs=None, output_dict=empty, num_frames=T)
    # 2. the one exchange step
    cond = gather_cond_memories(local, cond_ids, group) if distributed else local
    masks: Dict[int, torch.Tensor] = {t: o["pred_masks"] for t, o in local.items()}

    # 3. propagation (sequential in t).  Objects never interact (non_overlap_masks off), so with enough objects each rank
    # carries a slice of the object batch through the chain; otherwise the chain is replicated.
    ob, oe = (shard_range(n_obj, rank, world) if (distributed and shard_objects and n_obj >= world) else (0, n_obj))
    sl = slice(ob, oe)
    if (ob, oe) != (0, n_obj):
        cond = {t: _slice_objects(o, sl) for t, o in cond.items()}
    output_dict = {"cond_frame_outputs": cond, "non_cond_frame_outputs": {}}
    enc = _SliceEncoder(model, volume, [t for t in range(T) if t not in cond], encode_batch)
    for t in range(T):
        if t in cond:
            continue
        feats, pos, sizes = enc.get(t, oe - ob)
        cur = model.track_step(frame_idx=t, is_init_cond_frame=False, current_vision_feats=feats, current_vision_pos_embeds=pos,
                               feat_sizes=sizes, point_inputs=None, mask_inputs=None, output_dict=output_dict, num_frames=T)
        output_dict["non_cond_frame_outputs"][t] = cur
        masks[t] = cur["pred_masks"]
    if fill_hole_area > 0:
        for t in masks:
            masks[t] = ops.fill_holes_(masks[t].contiguous().clone(), fill_hole_area)
    return masks


def _slice_objects(o: dict, sl: slice) -> dict:
    return {"maskmem_features": o["maskmem_features"][sl], "maskmem_pos_enc": [p[sl] for p in o["maskmem_pos_enc"]],
            "obj_ptr": o["obj_ptr"][sl], "pred_masks": None if o.get("pred_masks") is None else o["pred_masks"][sl]}
