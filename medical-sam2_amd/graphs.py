"""hipGraph replay of the per-slice forward of the 3-D propagation: what `propagate_in_video` runs for every slice without a prompt
(sam2_video_predictor.py:1302-1367 -> SAM2Base.track_step, sam2_base.py:705-800): memory-bank assembly, the four memory-attention
layers, prompt encoder, mask decoder, mask up-sampling, object pointer and the memory encoder -- about 200 launches, many of them a
few microseconds long.

The bank a slice attends to changes from slice to slice, a captured graph cannot.  What changes, and how each part is made static:
  * WHICH stored slices are attended to and their temporal positions -- host logic (`SAM2Base._select_memory`).  The tuple of temporal
    positions is the BUCKET KEY: one graph per distinct tuple.  In steady state (all conditioning slices + the last `num_maskmem - 1`
    tracked ones) the tuple repeats with the period of the prompts, so a volume needs a handful of graphs.
  * the CONTENTS of those memories: every bucket owns its assembled bank (memory rows + position rows, the latter a constant of the
    bucket); before a replay only the entries whose source changed are re-written (conditioning memories once per bucket, the few
    recent memories every slice) -- the 70 strided copies per slice that assemble a 35-entry bank run neither eagerly nor in the graph;
  * the NUMBER of object pointers grows with every slice: the pointer tail of the bank is padded to a fixed capacity and the number of
    valid keys is a device-side scalar the attention kernel reads (`msam2_attention_kv64_dyn_fwd`); launch shapes depend on the
    capacity only.

A key is run eagerly -- in exactly the padded form that is captured, so eager and replayed slices are bit-identical -- until it has
been seen `capture_after` times, then captured (warm-up slices of a volume are not worth a capture).  Weights must not change while
graphs are alive (the kernel-ready 16-bit copies of `WeightCache` are baked into the graph): `track` checks the parameters' version
counters and drops its graphs when one moved.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch

F32 = torch.float32


def pointer_capacity(model, n_cond: int, num_frames: int, multiple: int = 16) -> int:
    """Upper bound of the object pointers one slice attends to (sam2_base.py:571-606: every selected conditioning slice + up to
    max_obj_ptrs_in_encoder - 1 tracked ones), rounded up so the padded tail ends on a 64-token key tile."""
    n = n_cond + max(0, min(num_frames, model.max_obj_ptrs_in_encoder) - 1)
    return max(multiple, -(-n // multiple) * multiple)


class _Bucket:
    __slots__ = ("graph", "outputs", "seen", "memory", "memory_pos", "row_src", "n_ptr_tok")

    def __init__(self):
        self.graph, self.outputs, self.seen = None, None, 0
        self.memory = self.memory_pos = None      # the ASSEMBLED bank of this bucket [N_k capacity, n_obj, 64] fp32 (+ its position side)
        self.row_src, self.n_ptr_tok = [], 0      # the stored output each spatial entry currently mirrors


class GraphedPropagation:
    """Static buffers + per-bucket graphs of the prompt-free `track_step` for a fixed number of objects.

    track(frame_idx, feats_one, output_dict) -> the `current_out` dict track_step returns (fresh tensors, safe to store).
    feats_one: {"backbone_fpn": [levels x [1,C,h,w]], "vision_pos_enc": [levels x [1,C,h,w]]} of the slice (one image, expanded over
    the objects like sam2_video_predictor.py:1284-1296)."""

    KEEP = ("pred_masks", "pred_masks_high_res", "obj_ptr", "maskmem_features")

    def __init__(self, model, n_obj: int, num_frames: int, ptr_capacity: int, capture_after: int = 1, keep_high_res: bool = False,
                 enabled: bool = True):
        self.model, self.n_obj, self.num_frames, self.cap = model, int(n_obj), int(num_frames), int(ptr_capacity)
        assert capture_after >= 1, "a bucket runs eagerly at least once before it is captured (weight packing, tables, code objects)"
        self.capture_after, self.keep_high_res, self.enabled = int(capture_after), keep_high_res, enabled
        self.buckets: Dict[Tuple[int, ...], _Bucket] = {}
        self.feat: Optional[List[torch.Tensor]] = None        # static copies of the slice's feature levels
        self.pos: Optional[List[torch.Tensor]] = None         # position tables (constants)
        self.mem_pos_enc: Optional[torch.Tensor] = None       # the memory encoder's position table (a constant)
        self.ptr_bank: Optional[torch.Tensor] = None          # [capacity, n_obj, C] fp32
        self.key_count: Optional[torch.Tensor] = None         # int32 device scalar
        self._params = [p for p in model.parameters()]
        self._versions = self._version_sum()
        self.replays = self.captures = self.eager_steps = 0

    # ---------------------------------------------------------------------------------------------------------------
    def _version_sum(self) -> int:
        return sum(p._version for p in self._params)

    def _statics(self, feats_one: dict, spatial, device):
        m = self.model
        if self.feat is None:
            self.feat = [torch.empty_like(f) for f in feats_one["backbone_fpn"]]     # layout preserved (channels-last token rows)
            self.pos = [p for p in feats_one["vision_pos_enc"]]
            self.ptr_bank = torch.zeros(self.cap, self.n_obj, m.hidden_dim, dtype=F32, device=device)
            self.key_count = torch.zeros(1, dtype=torch.int32, device=device)
            self.mem_pos_enc = spatial[0][1]["maskmem_pos_enc"][-1].to(device).clone()

    def _fill(self, bucket: _Bucket, feats_one: dict, spatial, ptrs):
        """host -> static buffers: a handful of device copies, no synchronisation"""
        from . import ops
        m, n_obj = self.model, self.n_obj
        for dst, src in zip(self.feat, feats_one["backbone_fpn"]):
            dst.copy_(src)
        n = len(ptrs)
        if n > self.cap:
            raise RuntimeError(f"{n} object pointers exceed the capacity {self.cap} this GraphedPropagation was sized for")
        H, W = spatial[0][1]["maskmem_features"].shape[-2:]
        HW, split = H * W, m.hidden_dim // m.mem_dim
        if n:
            self.ptr_bank[:n].copy_(torch.stack([p.to(F32) for p in ptrs]))
        if bucket.memory is None:
            # first sight of the bucket: the model's own assembly (memory rows, position rows with this bucket's temporal encodings,
            # pointer tail from the padded bank) becomes the bucket's static bank
            bucket.memory, bucket.memory_pos, bucket.n_ptr_tok, _ = m._assemble_memory(spatial, (self.ptr_bank, self.key_count), n_obj, H, W,
                                                                                         self.ptr_bank.device)
            bucket.row_src = [prev for _, prev in spatial]
        else:
            for i, (_, prev) in enumerate(spatial):
                if bucket.row_src[i] is not prev:
                    ops.add_cast_into(bucket.memory[i * HW:(i + 1) * HW], prev["maskmem_features"].flatten(2).permute(2, 0, 1), None, 1.0)
                    bucket.row_src[i] = prev
            if n:
                tail = bucket.memory[len(spatial) * HW:].view(self.cap, split, n_obj, m.mem_dim)
                tail[:n].copy_(self.ptr_bank[:n].view(n, n_obj, split, m.mem_dim).permute(0, 2, 1, 3))
        valid = len(spatial) * HW + n * split
        self.key_count.fill_(valid)
        from . import parallel
        kvs = parallel.current_kv_split()
        if kvs is not None:
            kvs.host_key_count = valid         # lets every rank project only the keys of its own splits (RoPEAttention.proj_rope_key_range)

    def _body(self, bucket: _Bucket) -> dict:
        """the prompt-free track_step on the static buffers (what is captured)"""
        m, n = self.model, self.n_obj
        bo = {"backbone_fpn": [f.expand(n, -1, -1, -1) for f in self.feat], "vision_pos_enc": [p.expand(n, -1, -1, -1) for p in self.pos]}
        _, feats, pos, sizes = m._prepare_backbone_features(bo)
        return m.track_step(frame_idx=-1, is_init_cond_frame=False, current_vision_feats=feats, current_vision_pos_embeds=pos,
                            feat_sizes=sizes, point_inputs=None, mask_inputs=None, output_dict=None, num_frames=self.num_frames,
                            memory_selection=("assembled", bucket.memory, bucket.memory_pos, bucket.n_ptr_tok, self.key_count))

    def _capture(self, key, bucket: _Bucket):
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = self._body(bucket)
        bucket.graph, bucket.outputs = g, out
        self.captures += 1

    # ---------------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def track(self, frame_idx: int, feats_one: dict, output_dict: dict, track_in_reverse: bool = False) -> dict:
        m = self.model
        if m.training:
            raise RuntimeError("GraphedPropagation is an inference path: train-mode dropout draws a host-side stream per forward")
        spatial, ptrs = m._select_memory(frame_idx, output_dict, self.num_frames, track_in_reverse)
        assert spatial, "propagation needs at least one stored memory"
        if self._version_sum() != self._versions:          # a parameter was written: the graphs hold stale weight copies
            self.buckets.clear()
            self._versions = self._version_sum()
        key = tuple(t_pos for t_pos, _ in spatial)
        self._statics(feats_one, spatial, spatial[0][1]["maskmem_features"].device)
        bucket = self.buckets.setdefault(key, _Bucket())
        self._fill(bucket, feats_one, spatial, ptrs)
        bucket.seen += 1
        if bucket.graph is None and self.enabled and bucket.seen > self.capture_after:
            self._capture(key, bucket)
        if bucket.graph is None:
            self.eager_steps += 1
            out = self._body(bucket)                  # fresh tensors: nothing to copy
            cur = {k: out[k] for k in self.KEEP if k != "pred_masks_high_res" or self.keep_high_res}
        else:
            bucket.graph.replay()
            self.replays += 1
            cur = {k: bucket.outputs[k].clone() for k in self.KEEP if k != "pred_masks_high_res" or self.keep_high_res}
        cur["maskmem_pos_enc"] = [self.mem_pos_enc]
        cur["point_inputs"], cur["mask_inputs"] = None, None
        if not self.keep_high_res:
            cur["pred_masks_high_res"] = None
        return cur
