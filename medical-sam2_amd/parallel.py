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


def gather_cond_memories(local: Dict[int, dict], frame_ids: List[int], group=None) -> Dict[int, dict]:
    """All-gather the conditioning-slice memories.

    `frame_ids` is the global, ordered list of conditioning slice indices; rank r owns the contiguous share
    `shard_range(len(frame_ids), r, world)` of it and passes its outputs in `local` ({frame_idx: track_step output with
    "maskmem_features", "maskmem_pos_enc", "obj_ptr", "pred_masks"}).  Every rank returns the full {frame_idx: output} map.
    Two collectives in total (features, pointers), each moving one fixed-size slab per rank (shares are padded to the largest
    share), i.e. a few large messages instead of one per slice."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return dict(local)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    shares = [shard_range(len(frame_ids), r, world) for r in range(world)]
    cap = max(e - b for b, e in shares)
    b0, e0 = shares[rank]
    mine = [local[frame_ids[i]] for i in range(b0, e0)]
    assert mine, "every rank must own at least one conditioning slice"
    f0, p0 = mine[0]["maskmem_features"], mine[0]["obj_ptr"]
    feats = torch.zeros((cap,) + tuple(f0.shape), dtype=f0.dtype, device=f0.device)
    ptrs = torch.zeros((cap,) + tuple(p0.shape), dtype=p0.dtype, device=p0.device)
    for i, o in enumerate(mine):
        feats[i].copy_(o["maskmem_features"])
        ptrs[i].copy_(o["obj_ptr"])
    all_f = [torch.empty_like(feats) for _ in range(world)]
    all_p = [torch.empty_like(ptrs) for _ in range(world)]
    dist.all_gather(all_f, feats, group=group)
    dist.all_gather(all_p, ptrs, group=group)
    pos = mine[0]["maskmem_pos_enc"]  # constant table, identical on every rank
    out: Dict[int, dict] = {}
    for r, (b, e) in enumerate(shares):
        for i in range(b, e):
            fid = frame_ids[i]
            if r == rank:
                out[fid] = local[fid]
            else:
                out[fid] = {"maskmem_features": all_f[r][i - b], "maskmem_pos_enc": pos, "obj_ptr": all_p[r][i - b],
                            "pred_masks": None, "pred_masks_high_res": None, "point_inputs": None, "mask_inputs": None}
    return out


def allreduce_gradients(grads: Dict[str, torch.Tensor], group=None, bucket_bytes: int = 64 << 20) -> Tuple[Dict[str, torch.Tensor], float]:
    """Data-parallel fine-tuning (the reference's `args.distributed` switch wraps the net in nn.DataParallel, utils.py get_network):
    sum the per-rank gradients of `training.*_loss_grads` over the ranks with a few LARGE all-reduces instead of one per parameter --
    gradients are packed (sorted by name, so every rank packs identically) into flat fp32 buckets of <= `bucket_bytes` (64 MiB: the
    whole mask decoder is 16 MB, the memory attention 23 MB, i.e. one ring all-reduce each; xGMI rings are per-link bound, so fewer and
    larger messages win).  Returns ({name: view into its bucket}, 1 / world): pass the second value on as the optimiser's `grad_scale`
    (times the inverse loss scale), which turns the sum into the mean without another pass over the gradients."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return dict(grads), 1.0
    world = dist.get_world_size(group)
    names = sorted(grads)
    out: Dict[str, torch.Tensor] = {}
    i = 0
    while i < len(names):
        j, size = i, 0
        while j < len(names) and (j == i or size + grads[names[j]].numel() * 4 <= bucket_bytes):
            size += grads[names[j]].numel() * 4
            j += 1
        flat = torch.cat([grads[n].detach().reshape(-1).float() for n in names[i:j]])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        off = 0
        for n in names[i:j]:
            k = grads[n].numel()
            out[n] = flat[off:off + k].view(grads[n].shape)
            off += k
        i = j
    return out, 1.0 / world
