"""Multi-GPU plumbing for the hot path (one process per GPU, torch.distributed; backend "nccl" is RCCL over xGMI on ROCm,
"gloo" is used by the CPU tests).

What shards:
  * 2-D batches / whole volumes: independent replicas, nothing is exchanged (`shard_range`).
  * 3-D volume, image encoder + conditioning-slice heads + conditioning-slice memory encoding: independent per slice, so the
    conditioning slices are sharded contiguously over the ranks and ONE exchange step follows: an all-gather of every
    conditioning slice's `maskmem_features` [n_obj, 64, h, w] and `obj_ptr` [n_obj, 256] (`gather_cond_memories`).
    `maskmem_pos_enc` is an input-independent table and is never sent.
  * the propagation chain over non-conditioning slices is sequential in the slice index (slice t needs t-1 ... t-6), so it is
    replicated (or object-sharded by the caller); it has no collective.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Tuple

import torch
import torch.distributed as dist


def init_distributed(backend: Optional[str] = None, device: Optional[torch.device] = None):
    """Initialise the default process group from the torchrun environment (RANK / WORLD_SIZE / MASTER_*).  Returns
    (rank, world_size); a no-op returning (0, 1) outside torchrun."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world == 1:
        return 0, 1
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if not dist.is_initialized():
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
        dist.init_process_group(backend=backend, **kw)
    return rank, world


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced [begin, end) share of n independent units for `rank` (first n % world ranks get one extra)."""
    q, r = divmod(n, world)
    begin = rank * q + min(rank, r)
    return begin, begin + q + (1 if rank < r else 0)


def max_over_ranks(seconds: float, device: torch.device, group=None) -> float:
    """Wall time of the slowest rank (the benchmark's timing rule)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())


def barrier(device: Optional[torch.device] = None, group=None):
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        if device is not None and device.type == "cuda":
            dist.barrier(group=group, device_ids=[device.index])
        else:
            dist.barrier(group=group)


# Tests on a ONE-GPU box set this to run every collective of this module through the real backend (RCCL) in a world of one rank: the
# calls, their stream ordering against the kernels around them and the ragged / coalesced forms then execute instead of being skipped.
FORCE_SINGLE_RANK_COLLECTIVES = False


def _is_dist(group=None) -> bool:
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or FORCE_SINGLE_RANK_COLLECTIVES)


def _to_nhwc(x: torch.Tensor) -> torch.Tensor:
    """[n,C,h,w] -> channels-last MEMORY [n,h,w,C] (free for this package's feature maps, which are NCHW views of token-major buffers)"""
    return x.permute(0, 2, 3, 1).contiguous()


def _from_nhwc(x: torch.Tensor) -> torch.Tensor:
    """[n,h,w,C] memory -> the [n,C,h,w] view the modules hand around (token-major underneath, as every kernel expects)"""
    return x.permute(0, 3, 1, 2)


def _gather_slabs(mine: List[torch.Tensor], counts: List[int], item_shape, dtype, device, group=None) -> List[List[torch.Tensor]]:
    """One all-gather of a fixed-size slab per rank: rank r contributes counts[r] items of `item_shape` (its `mine` list; may be empty),
    slabs are padded to max(counts).  Returns per rank the list of its items (views into the gathered slabs)."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    cap = max(max(counts), 1)
    slab = torch.zeros((cap,) + tuple(item_shape), dtype=dtype, device=device)
    assert len(mine) == counts[rank]
    for i, t in enumerate(mine):
        slab[i].copy_(t.reshape(item_shape))
    out = [torch.empty_like(slab) for _ in range(world)]
    dist.all_gather(out, slab, group=group)
    return [[out[r][i] for i in range(counts[r])] for r in range(world)]


def gather_cond_memories(local: Dict[int, dict], frame_ids: List[int], group=None, owners: Optional[List[int]] = None,
                         like: Optional[tuple] = None) -> Dict[int, dict]:
    """All-gather the conditioning-slice outputs the propagation chain needs.

    `frame_ids` is the global, ordered list of conditioning slice indices and `owners[i]` the rank that processed frame_ids[i]
    (default: contiguous shares `shard_range(len(frame_ids), r, world)`); a rank passes its own outputs in `local` ({frame_idx:
    track_step output with "maskmem_features", "maskmem_pos_enc", "obj_ptr", "pred_masks"}) -- possibly none at all: it then joins
    the collectives with an empty slab and needs `like` = (n_obj, mem_dim, embedding side, hidden_dim, device) for the shapes.
    Every rank returns the full {frame_idx: output} map.  Three collectives in total (memory features, pointers, low-res masks),
    each moving one fixed-size slab per rank (padded to the largest share), i.e. a few large messages instead of one per slice.
    `maskmem_pos_enc` is an input-independent table: never sent; remote entries take the first local one or, on a rank that owns no
    conditioning slice, None (the caller fills it in: `volume.segment_volume`)."""
    if not _is_dist(group):
        return dict(local)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if owners is None:
        owners = []
        for r in range(world):
            b, e = shard_range(len(frame_ids), r, world)
            owners += [r] * (e - b)
    assert len(owners) == len(frame_ids)
    counts = [sum(1 for o in owners if o == r) for r in range(world)]
    my_ids = [f for f, o in zip(frame_ids, owners) if o == rank]
    assert sorted(my_ids) == sorted(local), "a rank must pass exactly the conditioning slices it owns"
    mine = [local[f] for f in my_ids]
    if mine:
        f0, p0, m0 = mine[0]["maskmem_features"], mine[0]["obj_ptr"], mine[0]["pred_masks"]
        n_, c_, h_, w_ = f0.shape
        shapes = ((n_, h_, w_, c_), tuple(p0.shape), tuple(m0.shape))            # memory features travel channels-last
        dts, device = (f0.dtype, p0.dtype, m0.dtype), f0.device
    else:
        assert like is not None, "a rank without conditioning slices needs `like` to join the exchange"
        n_obj, mem_dim, E, hidden, device = like
        shapes = ((n_obj, E, E, mem_dim), (n_obj, hidden), (n_obj, 1, 4 * E, 4 * E))
        dts = (torch.float32,) * 3
    all_f = _gather_slabs([_to_nhwc(o["maskmem_features"]) for o in mine], counts, shapes[0], dts[0], device, group)
    all_p = _gather_slabs([o["obj_ptr"] for o in mine], counts, shapes[1], dts[1], device, group)
    all_m = _gather_slabs([o["pred_masks"] for o in mine], counts, shapes[2], dts[2], device, group)
    pos = mine[0]["maskmem_pos_enc"] if mine else None
    out: Dict[int, dict] = {}
    seen = [0] * world
    for fid, r in zip(frame_ids, owners):
        i = seen[r]
        seen[r] += 1
        if r == rank:
            out[fid] = local[fid]
        else:
            out[fid] = {"maskmem_features": _from_nhwc(all_f[r][i]), "maskmem_pos_enc": pos, "obj_ptr": all_p[r][i], "pred_masks": all_m[r][i],
                        "pred_masks_high_res": None, "point_inputs": None, "mask_inputs": None}
    return out


def gather_slice_features(local: Dict[int, dict], slice_ids: List[int], owners: List[int], group=None) -> Dict[int, dict]:
    """All-gather the backbone features of the slices in `slice_ids` (owners[i] = rank that encoded slice_ids[i]; `local` = this
    rank's {slice: {"backbone_fpn": [levels x [1,C,h,w]], "vision_pos_enc": [...]}}): one collective per feature level, one padded slab
    per rank.  The position tables are input-independent and stay local (every returned entry shares this rank's own, or None when
    the rank encoded nothing -- the caller then supplies them)."""
    if not _is_dist(group):
        return dict(local)
    if not slice_ids:                       # every slice is a conditioning slice: the same on all ranks, so no collective is skipped one-sidedly
        return {}
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    counts = [sum(1 for o in owners if o == r) for r in range(world)]
    my_ids = [t for t, o in zip(slice_ids, owners) if o == rank]
    assert sorted(my_ids) == sorted(local), "a rank must pass exactly the slices it encoded"
    # level shapes: from a local sample, else from the first owner (one small broadcast of the shape table)
    meta = None
    if my_ids:
        one = local[my_ids[0]]["backbone_fpn"]
        meta = [[f.shape[2], f.shape[3], f.shape[1]] for f in one]              # (h, w, C): features travel channels-last
    objs = [None] * world
    dist.all_gather_object(objs, meta, group=group)
    meta = next(m for m in objs if m is not None)
    sample = next(iter(local.values())) if local else None
    device = sample["backbone_fpn"][0].device if sample else torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
    levels = []
    for lvl, shp in enumerate(meta):
        levels.append(_gather_slabs([_to_nhwc(local[t]["backbone_fpn"][lvl]) for t in my_ids], counts, (1,) + tuple(shp), torch.float32, device, group))
    pos = sample["vision_pos_enc"] if sample else None
    out: Dict[int, dict] = {}
    seen = [0] * world
    for t, r in zip(slice_ids, owners):
        i = seen[r]
        seen[r] += 1
        out[t] = local[t] if r == rank else {"backbone_fpn": [_from_nhwc(levels[lvl][r][i]) for lvl in range(len(meta))], "vision_pos_enc": pos}
    return out


class FeatureStream:
    """PIPELINED exchange of the non-conditioning slices' backbone features: instead of one up-front all-gather of
    every slice's features (4.3 GB per rank at 512 slices) that the propagation chain has to wait for, the slices travel in CHUNKS of
    consecutive slices of one owner, each chunk as one asynchronous broadcast per feature level from the rank that encoded it, issued
    in slice order -- the order the chain consumes them in.  `get(t)` waits for the chunk of slice t only, so the chain starts as soon
    as the first chunk has landed and the rest of the transfer runs under it (on RCCL's own stream).
    BOUNDED: at most `window` chunks are allocated and in flight at any time -- chunk i + window is issued when chunk i has
    been consumed -- so the peak is window x chunk slices (4 x 8 x 16.8 MB = 0.5 GB at 1024^2) instead of the whole volume's features;
    every rank consumes the slices in the same order (the chain is sequential), so every rank issues the broadcasts in the same order.
    `close()` (also the context-manager exit) waits for whatever is still in flight: a rank that stops consuming early leaves no
    outstanding collective behind.
    local: this rank's {slice: {"backbone_fpn": [levels x [1,C,h,w]], "vision_pos_enc": [...]}}; slice_ids / owners: the global, ordered
    list of slices to exchange and the rank that encoded each.  Features travel channels-last; the position tables stay local."""

    def __init__(self, local: Dict[int, dict], slice_ids: List[int], owners: List[int], group=None, chunk: int = 8,
                 pos_tables=None, device: Optional[torch.device] = None, window: int = 4):
        self.group, self.local, self.pos = group, dict(local), pos_tables
        self.rank = dist.get_rank(group)
        world = dist.get_world_size(group)
        my_ids = [t for t, o in zip(slice_ids, owners) if o == self.rank]
        assert sorted(my_ids) == sorted(local), "a rank must pass exactly the slices it encoded"
        meta = None
        if my_ids:
            one = local[my_ids[0]]["backbone_fpn"]
            meta = [[f.shape[2], f.shape[3], f.shape[1]] for f in one]              # (h, w, C) per level
            device = one[0].device
        objs = [None] * world
        dist.all_gather_object(objs, meta, group=group)
        self.meta = next((m for m in objs if m is not None), None)
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
        self.device = device
        # chunks: runs of consecutive list entries with one owner, at most `chunk` slices each
        self.chunks: List[dict] = []
        self.where: Dict[int, tuple] = {}
        i = 0
        while i < len(slice_ids):
            j = i
            while j < len(slice_ids) and owners[j] == owners[i] and j - i < chunk:
                j += 1
            ids = slice_ids[i:j]
            for k, t in enumerate(ids):
                self.where[t] = (len(self.chunks), k)
            self.chunks.append({"ids": ids, "owner": owners[i], "bufs": None, "works": [], "left": len(ids)})
            i = j
        self.issued = 0
        self.peak_chunks_alive = 0
        for _ in range(max(1, int(window))):
            self._issue_next()

    def _issue_next(self):
        """allocate the next chunk's slabs and start its broadcasts (the owner fills its slabs from `local` first)"""
        if self.issued >= len(self.chunks):
            return
        ck = self.chunks[self.issued]
        self.issued += 1
        bufs = []
        for lvl, (h, w, c) in enumerate(self.meta):
            buf = torch.empty(len(ck["ids"]), h, w, c, dtype=torch.float32, device=self.device)
            if ck["owner"] == self.rank:
                for k, t in enumerate(ck["ids"]):
                    buf[k].copy_(_to_nhwc(self.local[t]["backbone_fpn"][lvl])[0])
            src = dist.get_global_rank(self.group, ck["owner"]) if self.group is not None else ck["owner"]
            ck["works"].append(dist.broadcast(buf, src=src, group=self.group, async_op=True))
            bufs.append(buf)
        ck["bufs"] = bufs                      # (the owner serves its own tensors from `local`; its slabs live until the sends are done)
        self.peak_chunks_alive = max(self.peak_chunks_alive, sum(1 for c in self.chunks if c["bufs"] is not None))

    def get(self, t: int) -> dict:
        """features of slice t (waits for its chunk if it has not landed yet); call once per slice, in the announced order"""
        ci, k = self.where[t]
        ck = self.chunks[ci]
        while self.issued <= ci:               # (a consumer that skipped slices: catch up, in order)
            self._issue_next()
        for w in ck["works"]:
            w.wait()
        ck["works"] = []
        ck["left"] -= 1
        if ck["owner"] == self.rank:
            out = self.local.pop(t)
        else:
            out = {"backbone_fpn": [_from_nhwc(b[k: k + 1]) for b in ck["bufs"]], "vision_pos_enc": self.pos}
        if out["vision_pos_enc"] is None:
            out["vision_pos_enc"] = self.pos
        if ck["left"] == 0:
            ck["bufs"] = None                                                       # the views handed out keep their storage alive
            self._issue_next()                                                      # keep `window` chunks in flight
        return out

    def pop(self, t: int) -> dict:
        return self.get(t)

    def close(self):
        """wait for every broadcast that is in flight (every rank issued the same ones) and drop the slabs"""
        for ck in self.chunks[: self.issued]:
            for w in ck["works"]:
                w.wait()
            ck["works"] = []
            ck["bufs"] = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()
        return False


def gather_object_shards(masks: Dict[int, torch.Tensor], slice_ids: List[int], n_obj: int, group=None, layout=None) -> Dict[int, torch.Tensor]:
    """Object-sharded chain -> full object batch on every rank: masks[t] is this rank's [n_local, 1, h, w] share for every t in
    slice_ids; one all-gather of a [len(slice_ids), cap, 1, h, w] slab.  layout (`chain_layout`): the object range of every rank; the
    ranks of one chain group hold identical copies and the group's FIRST rank's copy is taken (default: one rank per group,
    `shard_range(n_obj, rank, world)`)."""
    if not _is_dist(group):
        return dict(masks)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if layout is None:
        layout = [(shard_range(n_obj, r, world), (r, r + 1)) for r in range(world)]
    shares = [objs if r == span[0] else (0, 0) for r, (objs, span) in enumerate(layout)]      # (begin, end) taken from rank r
    cap = max(e - b for (b, e), _ in layout)
    if not slice_ids:
        return {}
    m0 = masks[slice_ids[0]]
    slab = torch.zeros((len(slice_ids), cap) + tuple(m0.shape[1:]), dtype=m0.dtype, device=m0.device)
    for i, t in enumerate(slice_ids):
        slab[i, : masks[t].shape[0]].copy_(masks[t])
    out = [torch.empty_like(slab) for _ in range(world)]
    dist.all_gather(out, slab, group=group)
    return {t: torch.cat([out[r][i, : e - b] for r, (b, e) in enumerate(shares) if e > b], dim=0) for i, t in enumerate(slice_ids)}


# set by `batch_invariant_splits()`: modeling.common.attn_splits then sizes the split-KV factor for a batch of one object
BATCH_INVARIANT_SPLITS = False


class batch_invariant_splits:
    """context: the attention kernels' split-KV factor does not depend on the object batch (see modeling.common.attn_splits)"""

    def __enter__(self):
        global BATCH_INVARIANT_SPLITS
        self.prev, BATCH_INVARIANT_SPLITS = BATCH_INVARIANT_SPLITS, True
        return self

    def __exit__(self, *a):
        global BATCH_INVARIANT_SPLITS
        BATCH_INVARIANT_SPLITS = self.prev
        return False


def chain_layout(n_obj: int, world: int, shard_objects: bool = True) -> List[Tuple[Tuple[int, int], Tuple[int, int]]]:
    """How the sequential propagation chain uses `world` ranks for `n_obj` objects: G = min(n_obj, world) GROUPS
    of consecutive ranks; group g carries the objects shard_range(n_obj, g, G) and owns the ranks shard_range(world, g, G), which split
    the memory cross-attention's KEY range among themselves (`KVSplit` on the group's sub-communicator).
      n_obj >= world : G = world, one rank per group -> pure object sharding (no key split);
      n_obj == 1     : G = 1, all ranks in one group -> pure key split;
      1 < n_obj < world (configs[2]'s n = 2..7 objects on 8 GPUs): one object per group, 8 // n or 8 // n + 1 ranks per object --
      the HYBRID the round-3 build lacked (it fell back to the pure key split and carried every object on every rank).
    Returns, per rank, ((object begin, end), (first rank, one past the last rank of its group)).  shard_objects=False: one group."""
    G = min(n_obj, world) if shard_objects else 1
    out = []
    for g in range(G):
        rb, re = shard_range(world, g, G)
        out += [(shard_range(n_obj, g, G), (rb, re))] * (re - rb)
    return out


_SUBGROUPS: Dict[tuple, object] = {}


def chain_subgroup(layout, group=None):
    """The sub-communicator of this rank's chain group (None for a group of one rank; `group` itself when one group spans the world).
    `dist.new_group` is collective over the parent -- every rank creates EVERY group of the layout, in the same order -- and RCCL
    communicators are expensive to build, so they are cached per (parent, member ranks) for the life of the process."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    spans = sorted({span for _, span in layout})
    if len(spans) == 1 and spans[0] == (0, world):
        return group
    mine = None
    for (rb, re) in spans:
        if re - rb < 2:
            continue
        members = tuple(dist.get_global_rank(group, r) if group is not None else r for r in range(rb, re))
        key = (id(group) if group is not None else None, members)
        if key not in _SUBGROUPS:
            _SUBGROUPS[key] = dist.new_group(ranks=list(members))
        if rb <= rank < re:
            mine = _SUBGROUPS[key]
    return mine


_KV_SPLIT = None


def current_kv_split():
    """The active `KVSplit` context (None outside one): consulted by the memory cross-attention."""
    return _KV_SPLIT


class KVSplit:
    """Cross-GPU split of the memory cross-attention's KEY range (the n_obj < ranks case): while active, every
    rank computes only its share of the split-KV partials -- the SAME partials, over the same key ranges and in the same workspace
    slots, that one rank computes for all splits -- then one all-gather per layer moves the (max, sum, O') triples (64-wide O':
    136 B per query row and split) and the library's merge kernel finishes, so the result is bit-identical to the single-rank run.
    Everything else of the chain (self-attention, heads, memory encoder) is replicated."""

    def __init__(self, model=None, group=None):
        global _KV_SPLIT
        assert _is_dist(group), "KVSplit needs an initialised process group with more than one rank"
        self.group, self.world, self.rank = group, dist.get_world_size(group), dist.get_rank(group)
        self.calls = 0
        self.host_key_count = None         # padded banks: the number of valid keys of the CURRENT slice as the host knows it (the device
        self._prev = _KV_SPLIT             # scalar `key_count` holds the same value): graphs.GraphedPropagation._fill sets it
        _KV_SPLIT = self

    def close(self):
        global _KV_SPLIT
        _KV_SPLIT = self._prev

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()
        return False

    def share(self, splits: int) -> Tuple[int, int]:
        return shard_range(splits, self.rank, self.world)

    def exchange(self, workspace: torch.Tensor, splits: int, rows: int, width: int = 64):
        """workspace (uint8) holds o_part [splits, rows, width] 16-bit then ml_part [splits, rows, 2] fp32 (msam2_attention_fwd's
        layout); this rank has filled the slots of its share.  All-gather the other ranks' slots in place."""
        self.calls += 1
        o_bytes = splits * rows * width * 2
        o = workspace[:o_bytes].view(splits, rows * width * 2)                    # raw bytes: every backend moves uint8
        ml = workspace[o_bytes: o_bytes + splits * rows * 8].view(splits, rows * 8)
        shares = [shard_range(splits, r, self.world) for r in range(self.world)]
        cap = max(e - b for b, e in shares)
        b0, e0 = shares[self.rank]
        for buf in (o, ml):
            send = torch.zeros((cap, buf.shape[1]), dtype=buf.dtype, device=buf.device)
            send[: e0 - b0].copy_(buf[b0:e0])
            recv = [torch.empty_like(send) for _ in range(self.world)]
            dist.all_gather(recv, send, group=self.group)
            for r, (b, e) in enumerate(shares):
                if r != self.rank and e > b:
                    buf[b:e].copy_(recv[r][: e - b])


class _PendingAllReduce:
    """handle of `allreduce_gradients_async`: `wait()` blocks the current stream / host until the sums have landed"""

    def __init__(self, works, grads, inv_world):
        self.works, self.grads, self.inv_world = works, grads, inv_world

    def wait(self):
        for w in self.works:
            w.wait()
        self.works = []
        return self.grads, self.inv_world


def allreduce_gradients_async(grads: Dict[str, torch.Tensor], group=None, bucket_bytes: int = 64 << 20) -> _PendingAllReduce:
    """Start the data-parallel SUM of `grads` over the ranks and return at once (`.wait()` -> ({name: summed gradient}, 1 / world)), so
    that the backward of the NEXT parameter group runs while this group's gradients are on the links (the decoder's 16 MB travel under
    the memory attention's backward, the memory attention's 23 MB under the image encoder's).
    IN PLACE and without a pack pass: the gradients (sorted by name, so every rank issues the same sequence) are handed to the
    backend as coalesced groups of <= `bucket_bytes` (torch.distributed's coalescing manager -> one `allreduce_coalesced` per group) --
    RCCL fuses a group into one launch (ncclGroupStart/End) over the tensors where they are; gloo (the CPU tests) flattens internally.  Non-fp32 or strided entries are made fp32 contiguous first (a copy for
    those entries only).  xGMI rings are per-link bound, so few large groups beat one message per parameter."""
    if not _is_dist(group):
        return _PendingAllReduce([], dict(grads), 1.0)
    world = dist.get_world_size(group)
    names = sorted(grads)
    out = {n: (g if (g.dtype == torch.float32 and g.is_contiguous()) else g.detach().float().contiguous()) for n, g in ((n, grads[n]) for n in names)}
    works, i = [], 0
    if dist.get_backend(group) != "nccl" and any(t.is_cuda for t in out.values()):
        # rehearsal setups (gloo ranks sharing one GPU): gloo has no coalesced all-reduce for device tensors -> one async all-reduce per
        # tensor in the same sorted order (RCCL, the production backend, takes the coalesced groups below)
        works = [dist.all_reduce(out[n], op=dist.ReduceOp.SUM, group=group, async_op=True) for n in names]
        i = len(names)
    while i < len(names):
        j, size = i, 0
        while j < len(names) and (j == i or size + out[names[j]].numel() * 4 <= bucket_bytes):
            size += out[names[j]].numel() * 4
            j += 1
        with dist._coalescing_manager(group=group, async_ops=True) as cm:
            for n in names[i:j]:
                dist.all_reduce(out[n], op=dist.ReduceOp.SUM, group=group)
        works.append(cm)
        i = j
    return _PendingAllReduce(works, out, 1.0 / world)


def union_gradient_keys(step_grads: Dict[str, Dict[str, torch.Tensor]], modules: Dict[str, torch.nn.Module], order, group=None):
    """Make the gradient dictionaries of a data-parallel step the same SHAPE on every rank before any all-reduce is issued: one
    `all_gather_object` of the per-group name lists, then every rank holds, per group (in `order`), the sorted UNION of the names --
    names this rank has no gradient for are zero-filled (shape / device of the module's parameter).  Parameters no rank reached stay
    absent, so the optimiser leaves them alone exactly as torch.optim does for `.grad is None`."""
    if not _is_dist(group):
        return step_grads
    world = dist.get_world_size(group)
    mine = {g: sorted(step_grads.get(g, {})) for g in order}
    every = [None] * world
    dist.all_gather_object(every, mine, group=group)
    out: Dict[str, Dict[str, torch.Tensor]] = {}
    for g in order:
        names = sorted(set().union(*[set(e[g]) for e in every]))
        have = step_grads.get(g, {})
        params = dict(modules[g].named_parameters()) if len(names) != len(have) else {}
        out[g] = {n: (have[n] if n in have else torch.zeros_like(params[n], dtype=torch.float32)) for n in names}
    return out


def allreduce_gradients(grads: Dict[str, torch.Tensor], group=None, bucket_bytes: int = 64 << 20) -> Tuple[Dict[str, torch.Tensor], float]:
    """Data-parallel fine-tuning (the reference's `args.distributed` switch wraps the net in nn.DataParallel, utils.py get_network):
    sum the per-rank gradients of `training.*_loss_grads` over the ranks -- `allreduce_gradients_async(...).wait()`.  Returns
    ({name: summed gradient, in place}, 1 / world): pass the second value on as the optimiser's `grad_scale` (times the inverse loss
    scale), which turns the sum into the mean without another pass over the gradients."""
    return allreduce_gradients_async(grads, group, bucket_bytes).wait()
