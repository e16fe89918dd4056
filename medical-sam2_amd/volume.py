"""3-D volume driver for BASELINE.json configs 3/4: bbox (or click) prompts on every `prompt_freq`-th slice, propagation through
the remaining slices with the memory bank -- the work `func_3d/function.py:226-274` drives through `SAM2VideoPredictor`
(`val_init_state` -> `train_add_new_bbox` -> `propagate_in_video`), restated against the mirrored `SAM2Base` surface only
(`forward_image`, `_prepare_backbone_features`, `track_step`).

Differences from the reference flow that do not change results: every slice is encoded once (the reference encodes conditioning
slices twice, sam2_video_predictor.py:1378-1380), and conditioning slices run the memory encoder in the same `track_step` call.

Multi-GPU (one process per GPU):
  1. image encoder of ALL slices + conditioning-slice heads / memory encoding: independent per slice -> every rank takes a contiguous
     share of the slices (`parallel.shard_range`), conditioning or not;
  2. the exchange: all-gather of the conditioning memories / pointers (`parallel.gather_cond_memories`; ranks without a
     conditioning slice join with an empty slab); the non-conditioning slices' backbone features (16.8 MB per slice at 1024^2) travel
     as chunked asynchronous broadcasts from their owners, issued in slice order and waited for chunk by chunk by the chain
     (`parallel.FeatureStream`), so the chain starts after the first chunk and the rest of the 4.3 GB (512 slices) moves under it;
  3. the propagation chain is sequential in the slice index: the ranks form min(n_obj, ranks) GROUPS (`parallel.chain_layout`), each
     carrying its share of the OBJECTS (objects never interact: non_overlap_masks is off); the ranks of a group split the memory
     cross-attention's KEY range among themselves (`parallel.KVSplit` on the group's sub-communicator: each rank's share of the
     split-KV partials, one all-gather of the (max, sum, O') triples per layer, then the same merge kernel).  n_obj >= ranks: pure
     object sharding; one object: pure key split; 1 < n_obj < ranks: the object x key hybrid (one object per group).  The key split
     is bit-identical to the single-rank result by construction; the object shards are too as long as every kernel on the chain picks
     the same variant for the smaller object batch (the split-KV factors are pinned: `parallel.batch_invariant_splits`);
  4. every rank returns ALL slices' masks for ALL objects (one all-gather over the object shards).
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
import torch.distributed as dist

from . import ops
from .graphs import GraphedPropagation, pointer_capacity
from .parallel import (FeatureStream, KVSplit, _is_dist, batch_invariant_splits, chain_layout, chain_subgroup, gather_cond_memories,
                       gather_object_shards, gather_slice_features, shard_range)


class _SliceEncoder:
    """Backbone features of the slices in the order they will be consumed, encoded `batch` slices per forward_image call (the
    trunk's kernels are far better filled at 8 slices than at 1; every kernel is batch-invariant, so the features are the ones a
    slice-by-slice pass produces) and dropped once consumed: at most `batch` slices (17 MB each at 1024^2) are resident."""

    def __init__(self, model, volume: torch.Tensor, order: List[int], batch: int):
        self.model, self.volume, self.order, self.batch = model, volume, list(order), max(1, int(batch))
        self.pos = 0
        self.cache: Dict[int, dict] = {}

    def raw(self, t: int) -> dict:
        """{"backbone_fpn": [3 x [1,C,h,w]], "vision_pos_enc": [3 x [1,C,h,w]]} of slice t"""
        if t not in self.cache:
            assert self.pos < len(self.order) and self.order[self.pos] == t, "slices must be consumed in the announced order"
            ids = self.order[self.pos: self.pos + self.batch]
            self.pos += len(ids)
            bo = self.model.forward_image(self.volume[ids] if len(ids) > 1 else self.volume[ids[0]][None])
            for j, u in enumerate(ids):
                self.cache[u] = {"backbone_fpn": [f[j: j + 1] for f in bo["backbone_fpn"]],
                                 "vision_pos_enc": [p[:1] for p in bo["vision_pos_enc"]]}
        return self.cache.pop(t)


def _expand(model, one: dict, n_obj: int):
    """features of one slice expanded (views) over the objects like sam2_video_predictor.py:1284-1296"""
    bo = {"backbone_fpn": [f.expand(n_obj, -1, -1, -1) for f in one["backbone_fpn"]],
          "vision_pos_enc": [p.expand(n_obj, -1, -1, -1) for p in one["vision_pos_enc"]]}
    _, feats, pos, sizes = model._prepare_backbone_features(bo)
    return feats, pos, sizes


def box_point_inputs(boxes: torch.Tensor) -> dict:
    """[n,4] (x0,y0,x1,y1) -> the two-corner point prompt with labels 2/3 (sam2_video_predictor.py:330-345)."""
    n = boxes.shape[0]
    return {"point_coords": boxes.reshape(n, 2, 2).float(),
            "point_labels": torch.tensor([[2, 3]], dtype=torch.int32, device=boxes.device).expand(n, 2).contiguous()}


@torch.no_grad()
def segment_volume(model, volume: torch.Tensor, prompts: Dict[int, dict], fill_hole_area: int = 0, group=None,
                   shard_objects: bool = True, encode_batch: int = 8, kv_split: bool = True, return_state: bool = False,
                   graphs: bool = False, padded_bank: bool = True, stats: Optional[dict] = None, graph_cache: Optional[dict] = None,
                   pipelined_exchange: bool = True, exchange_chunk: int = 8):
    """volume: [T,3,S,S] normalised slices on the GPU; prompts: {slice_idx: {"boxes": [n,4]} | {"point_coords", "point_labels"}}
    for the conditioning slices (same n objects everywhere).  Returns {slice_idx: low-res mask logits [n,1,S/4,S/4]} for ALL slices
    and ALL objects on every rank.  encode_batch: slices per image-encoder call (results do not depend on it).  shard_objects /
    kv_split: the two ways the propagation chain uses several ranks (module docstring); with both off it is replicated.
    return_state: also return the chain's `output_dict` ({"cond_frame_outputs", "non_cond_frame_outputs"}: per slice the track_step
    outputs with memories and pointers, this rank's object share) -- what a caller needs to continue or to audit the propagation.
    padded_bank (default): the propagation keeps one assembled memory bank per bucket (graphs.GraphedPropagation without capture):
    only the entries that changed since the previous slice are re-written instead of re-assembling all ~35 of them, the pointer tail is
    padded to a fixed capacity and the attention kernel reads the valid key count from the device (measured at 64 slices, one
    object: 245 -> 287 slices/s).  False: the reference-shaped assembly per slice.  graphs: additionally replay the per-slice forward as
    hipGraphs, one per bucket -- bit-identical to the padded-bank launches; not combined with the cross-GPU key split, whose exchange
    runs between the partial pass and the merge (the padded bank itself is).  stats: filled with the replay / capture counts of this
    call.  graph_cache: a dict the caller keeps between volumes of the same shape (slices, objects, prompt schedule): the captured
    graphs are reused, a later volume replays from its first steady-state slice on.
    pipelined_exchange (multi-rank): the non-conditioning slices' features travel as chunked asynchronous broadcasts that the chain waits
    for chunk by chunk (`parallel.FeatureStream`, `exchange_chunk` slices per message) instead of one all-gather before the chain starts."""
    T = volume.shape[0]
    cond_ids = sorted(prompts)
    assert cond_ids, "at least one conditioning slice is needed"
    first = prompts[cond_ids[0]]
    n_obj = (first["boxes"] if "boxes" in first else first["point_coords"]).shape[0]
    distributed = _is_dist(group)      # (a world of one rank counts when parallel.FORCE_SINGLE_RANK_COLLECTIVES is set: RCCL test)
    rank = dist.get_rank(group) if distributed else 0
    world = dist.get_world_size(group) if distributed else 1
    cond_set = set(cond_ids)
    phase_s: Optional[dict] = {} if (stats is not None and stats.get("time_phases")) else None
    import time as _time
    _t = [_time.perf_counter()]

    def _phase(name: str):
        """wall seconds of the phase that just ended (device drained first): only when the caller asked for `time_phases`"""
        if stats is not None:
            stats["phases_done"] = stats.get("phases_done", []) + [name]      # live: what a watchdog reports when the pass hangs
        if phase_s is not None:
            torch.cuda.synchronize(volume.device)
            now = _time.perf_counter()
            phase_s[name] = phase_s.get(name, 0.0) + now - _t[0]
            _t[0] = now

    # 1. this rank's contiguous share of ALL slices: image encoder; conditioning slices also run their heads + memory encoder
    b, e = shard_range(T, rank, world)
    mine = list(range(b, e))
    # conditioning slices first (their memories go into the exchange), then the rest of the share in slice order
    order = [t for t in mine if t in cond_set] + [t for t in mine if t not in cond_set]
    enc = _SliceEncoder(model, volume, order, encode_batch)
    empty = {"cond_frame_outputs": {}, "non_cond_frame_outputs": {}}
    local_cond: Dict[int, dict] = {}
    for t in order:
        if t not in cond_set:
            break
        pr = prompts[t]
        pin = box_point_inputs(pr["boxes"]) if "boxes" in pr else {"point_coords": pr["point_coords"], "point_labels": pr["point_labels"]}
        feats, pos, sizes = _expand(model, enc.raw(t), n_obj)
        local_cond[t] = model.track_step(frame_idx=t, is_init_cond_frame=True, current_vision_feats=feats, current_vision_pos_embeds=pos,
                                         feat_sizes=sizes, point_inputs=pin, mask_inputs=None, output_dict=empty, num_frames=T)
    local_feats: Dict[int, dict] = {t: enc.raw(t) for t in order if t not in cond_set}
    # input-independent position tables of the three feature levels: from any slice this rank encoded
    if local_feats:
        pos_tables = next(iter(local_feats.values()))["vision_pos_enc"]
    else:
        pos_tables = model.forward_image(volume[:1])["vision_pos_enc"]
        pos_tables = [p[:1] for p in pos_tables[-model.num_feature_levels:]]

    _phase("encode_all_slices_and_conditioning_pass")
    # 2. the one exchange step
    if distributed:
        owner = lambda t: next(r for r in range(world) if shard_range(T, r, world)[0] <= t < shard_range(T, r, world)[1])
        like = (n_obj, model.mem_dim, model.sam_image_embedding_size, model.hidden_dim, volume.device)
        cond = gather_cond_memories(local_cond, cond_ids, group, owners=[owner(t) for t in cond_ids], like=like)
        non_cond_ids = [t for t in range(T) if t not in cond_set]
        for o in cond.values():                       # remote entries: the position table of the memory encoder is a constant
            if o["maskmem_pos_enc"] is None:
                o["maskmem_pos_enc"] = [model.memory_encoder.position_encoding(o["maskmem_features"]).to(o["maskmem_features"].dtype)]
        if pipelined_exchange and non_cond_ids:
            # chunked asynchronous broadcasts in slice order: the chain below waits per chunk, the rest travels under it
            feats_all = FeatureStream(local_feats, non_cond_ids, [owner(t) for t in non_cond_ids], group, chunk=exchange_chunk,
                                      pos_tables=pos_tables, device=volume.device)
        else:
            feats_all = gather_slice_features(local_feats, non_cond_ids, [owner(t) for t in non_cond_ids], group)
            for o in feats_all.values():
                if o["vision_pos_enc"] is None:
                    o["vision_pos_enc"] = pos_tables
    else:
        cond, feats_all = local_cond, local_feats

    _phase("exchange_memories_and_features")
    # 3. propagation (sequential in t): groups of ranks per object share (parallel.chain_layout) -- object-sharded when there are at
    #    least as many objects as ranks, ONE key-split group for one object, and in between (1 < n_obj < ranks) one object per group with
    #    the group's ranks splitting the memory cross-attention's key range: the object x key hybrid of SURVEY 8(e) row 3
    layout = chain_layout(n_obj, world, shard_objects) if distributed else [((0, n_obj), (0, 1))]
    (ob, oe), (gb, ge) = layout[rank]
    obj_shard = distributed and (oe - ob) < n_obj
    sl = slice(ob, oe)
    chain_cond = {t: _slice_objects(o, sl) for t, o in cond.items()} if obj_shard else cond
    output_dict = {"cond_frame_outputs": chain_cond, "non_cond_frame_outputs": {}}
    masks: Dict[int, torch.Tensor] = {}
    sub = chain_subgroup(layout, group) if (distributed and kv_split) else None       # collective over `group`: every rank calls it
    from . import parallel as _par
    # (a group of one rank splits nothing -- except under FORCE_SINGLE_RANK_COLLECTIVES, the RCCL test of a one-GPU box, where the
    #  key-split path runs with this rank owning every split so that its exchange goes through the real backend)
    split_ctx = KVSplit(model, sub) if (distributed and kv_split and (ge - gb > 1 or (_par.FORCE_SINGLE_RANK_COLLECTIVES and n_obj == oe - ob))) else None
    if stats is not None:
        stats["chain_layout"] = {"objects": [ob, oe], "group_ranks": [gb, ge], "groups": len({span for _, span in layout}),
                                 "key_split_ranks": ge - gb if split_ctx is not None else 1}
    prop = None
    if graphs or padded_bank:
        graphs = graphs and split_ctx is None
        cap = pointer_capacity(model, len(cond_ids), T)
        ck = (id(model), oe - ob, T, cap, bool(graphs))
        prop = graph_cache.get(ck) if graph_cache is not None else None
        if prop is None:
            prop = GraphedPropagation(model, oe - ob, T, cap, enabled=graphs)
            if graph_cache is not None:
                graph_cache[ck] = prop
        before = (prop.replays, prop.captures, prop.eager_steps)
    try:
        with batch_invariant_splits():     # split-KV factors as for one object: an object-sharded chain reproduces the single-rank bits
            for t in range(T):
                if t in cond_set:
                    continue
                if prop is not None:
                    cur = prop.track(t, feats_all.pop(t), output_dict)
                else:
                    feats, pos, sizes = _expand(model, feats_all.pop(t), oe - ob)
                    cur = model.track_step(frame_idx=t, is_init_cond_frame=False, current_vision_feats=feats, current_vision_pos_embeds=pos,
                                           feat_sizes=sizes, point_inputs=None, mask_inputs=None, output_dict=output_dict, num_frames=T)
                output_dict["non_cond_frame_outputs"][t] = cur
                masks[t] = cur["pred_masks"]
    finally:
        if isinstance(feats_all, FeatureStream):
            feats_all.close()                  # nothing of the exchange stays in flight, whatever ended the chain
        if split_ctx is not None:
            split_ctx.close()
        if prop is not None and stats is not None:
            stats.update(replays=prop.replays - before[0], captures=prop.captures - before[1], eager_steps=prop.eager_steps - before[2],
                         buckets=len(prop.buckets))

    _phase("propagation_chain")
    # 4. everything everywhere: conditioning masks from their owners, propagated masks from the object shards
    if distributed:
        cond_masks = {t: o["pred_masks"] for t, o in cond.items()}            # gathered with the memories (full object batch)
        if obj_shard:
            masks = gather_object_shards(masks, [t for t in range(T) if t not in cond_set], n_obj, group, layout=layout)
    else:
        cond_masks = {t: o["pred_masks"] for t, o in local_cond.items()}
    masks.update(cond_masks)
    if fill_hole_area > 0:
        for t in masks:
            masks[t] = ops.fill_holes_(masks[t].contiguous().clone(), fill_hole_area)
    out = {t: masks[t] for t in sorted(masks)}
    _phase("gather_object_shards_and_hole_filling")
    if phase_s is not None:
        stats["phase_s"] = phase_s
    return (out, output_dict) if return_state else out


def _slice_objects(o: dict, sl: slice) -> dict:
    return {"maskmem_features": o["maskmem_features"][sl], "maskmem_pos_enc": [p[sl] for p in o["maskmem_pos_enc"]],
            "obj_ptr": o["obj_ptr"][sl], "pred_masks": None if o.get("pred_masks") is None else o["pred_masks"][sl]}
