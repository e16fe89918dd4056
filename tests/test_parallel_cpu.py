"""N>1 path on CPU: two gloo ranks exercise the sharding rule, the max-over-ranks timing rule and the one real exchange step
of the 3-D path (all-gather of conditioning-slice memories)."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_is_a_partition():
    sys.path.insert(0, ROOT)
    from medical_sam2_amd.parallel import shard_range
    for n in (0, 1, 7, 8, 9, 64, 513):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from medical_sam2_amd import parallel as par
    r, w = par.init_distributed(backend="gloo")
    assert (r, w) == (rank, world)
    dev = torch.device("cpu")
    # timing rule
    assert par.max_over_ranks(1.0 + rank, dev) == float(world)
    par.barrier(dev)
    # exchange step: 5 conditioning slices (ragged: 3 + 2), 2 objects
    frame_ids = [0, 2, 4, 6, 8]
    b, e = par.shard_range(len(frame_ids), rank, world)
    pos = [torch.arange(64.0).reshape(1, 64, 1, 1).expand(2, 64, 4, 4)]
    local = {}
    for i in range(b, e):
        f = frame_ids[i]
        g = torch.Generator().manual_seed(f)
        local[f] = {"maskmem_features": torch.randn(2, 64, 4, 4, generator=g), "maskmem_pos_enc": pos,
                    "obj_ptr": torch.randn(2, 256, generator=g), "pred_masks": torch.zeros(2, 1, 16, 16)}
    for f in local:
        local[f]["pred_masks"] = torch.full((2, 1, 16, 16), float(f))
    full = par.gather_cond_memories(local, frame_ids)
    assert sorted(full) == frame_ids
    ok_masks = all(torch.equal(full[f]["pred_masks"], torch.full((2, 1, 16, 16), float(f))) for f in frame_ids)
    ok = True
    for f in frame_ids:
        g = torch.Generator().manual_seed(f)
        ok &= torch.equal(full[f]["maskmem_features"], torch.randn(2, 64, 4, 4, generator=g))
        ok &= torch.equal(full[f]["obj_ptr"], torch.randn(2, 256, generator=g))
    ok &= ok_masks
    # explicit owners with an EMPTY rank (more ranks than conditioning slices): rank 1 owns nothing and joins with `like`
    one = {7: {"maskmem_features": torch.full((2, 64, 4, 4), 7.0), "maskmem_pos_enc": pos, "obj_ptr": torch.full((2, 256), 7.0),
               "pred_masks": torch.full((2, 1, 16, 16), 7.0)}} if rank == 0 else {}
    got = par.gather_cond_memories(one, [7], owners=[0], like=(2, 64, 4, 256, dev))
    ok &= got[7]["maskmem_features"].shape == (2, 64, 4, 4) and got[7]["maskmem_features"].permute(0, 2, 3, 1).is_contiguous() == (rank == 1)
    ok &= sorted(got) == [7] and bool((got[7]["maskmem_features"] == 7).all()) and bool((got[7]["pred_masks"] == 7).all())
    ok &= (got[7]["maskmem_pos_enc"] is None) == (rank == 1)
    # backbone features of the non-conditioning slices: 3 slices, rank 0 encoded two of them
    sl_ids, owners = [1, 3, 5], [0, 0, 1]
    mk = lambda t: {"backbone_fpn": [torch.full((1, 4, 8, 8), float(t)), torch.full((1, 8, 4, 4), 10.0 + t)], "vision_pos_enc": ["pos"]}
    feats = par.gather_slice_features({t: mk(t) for t, o in zip(sl_ids, owners) if o == rank}, sl_ids, owners)
    ok &= sorted(feats) == sl_ids and all(bool((feats[t]["backbone_fpn"][0] == t).all()) and bool((feats[t]["backbone_fpn"][1] == 10.0 + t).all())
                                          and feats[t]["backbone_fpn"][0].shape == (1, 4, 8, 8) for t in sl_ids)
    # the same exchange PIPELINED: chunked async broadcasts in slice order, waited for chunk by chunk (5 slices, chunks of 2, ragged owners)
    sl2, own2 = [1, 3, 5, 7, 9], [0, 0, 0, 1, 1]
    stream = par.FeatureStream({t: mk(t) for t, o in zip(sl2, own2) if o == rank}, sl2, own2, chunk=2, pos_tables=["pos"], device=dev)
    ok &= [len(c["ids"]) for c in stream.chunks] == [2, 1, 2]
    for t in sl2:
        f = stream.pop(t)
        ok &= bool((f["backbone_fpn"][0] == t).all()) and bool((f["backbone_fpn"][1] == 10.0 + t).all()) and f["backbone_fpn"][0].shape == (1, 4, 8, 8)
        ok &= f["vision_pos_enc"] == ["pos"]
    # object shards -> full object batch (3 objects over 2 ranks: 2 + 1)
    ob, oe = par.shard_range(3, rank, world)
    sh = {t: torch.arange(ob, oe, dtype=torch.float32).reshape(-1, 1, 1, 1).expand(-1, 1, 2, 2).contiguous() + 10 * t for t in (0, 1)}
    fullm = par.gather_object_shards(sh, [0, 1], 3)
    ok &= all(torch.equal(fullm[t][:, 0, 0, 0], torch.arange(3.0) + 10 * t) for t in (0, 1))
    # key-split exchange: 5 splits over 2 ranks (3 + 2) of a [splits, rows, 64] 16-bit + [splits, rows, 2] fp32 workspace
    S, rows = 5, 6
    ref_o = torch.arange(S * rows * 64, dtype=torch.int16).reshape(S, rows * 64)
    ref_ml = torch.arange(S * rows * 2, dtype=torch.float32).reshape(S, rows * 2)
    ws = torch.zeros(S * rows * 64 * 2 + S * rows * 8, dtype=torch.uint8)
    with par.KVSplit() as kvs:
        b0, e0 = kvs.share(S)
        ws[: S * rows * 128].view(torch.int16).view(S, -1)[b0:e0] = ref_o[b0:e0]
        ws[S * rows * 128:].view(torch.float32).view(S, -1)[b0:e0] = ref_ml[b0:e0]
        kvs.exchange(ws, S, rows)
        ok &= par.current_kv_split() is kvs
    ok &= par.current_kv_split() is None
    ok &= torch.equal(ws[: S * rows * 128].view(torch.int16).view(S, -1), ref_o) and torch.equal(ws[S * rows * 128:].view(torch.float32).view(S, -1), ref_ml)
    # data-parallel gradient reduction: ragged shapes, two buckets forced by a small bucket size
    gen = lambda r, n, shape: torch.randn(*shape, generator=torch.Generator().manual_seed(100 * r + n))
    shapes = {"b.weight": (7, 5), "a.bias": (3,), "c.gamma": (1, 2, 3), "d.w": (40, 10)}
    mine = {k: gen(rank, i, sh) for i, (k, sh) in enumerate(shapes.items())}
    red, inv = par.allreduce_gradients(mine, bucket_bytes=1024)
    assert inv == 1.0 / world and set(red) == set(shapes)
    for i, (k, sh) in enumerate(shapes.items()):
        want = sum(gen(r, i, sh) for r in range(world))
        ok &= red[k].shape == want.shape and torch.allclose(red[k], want, atol=1e-6)
    # ragged gradient dictionaries (every rank on its own volume -> different groups / parameters reached): the ranks agree on
    # the union first, absent entries are zero-filled, a group nobody reached stays empty, and the reductions then pair up
    mods = {"dec": torch.nn.Linear(3, 2), "mem": torch.nn.Linear(4, 1), "ptr": torch.nn.Linear(2, 2)}
    sg = {"dec": {"weight": torch.full((2, 3), 1.0 + rank)}, "mem": ({"weight": torch.ones(1, 4), "bias": torch.ones(1)} if rank == 1 else {}), "ptr": {}}
    if rank == 0:
        sg["dec"]["bias"] = torch.full((2,), 5.0)
    un = par.union_gradient_keys(sg, mods, ("dec", "mem", "ptr"))
    ok &= [sorted(un[g]) for g in ("dec", "mem", "ptr")] == [["bias", "weight"], ["bias", "weight"], []]
    red = {g: par.allreduce_gradients(un[g])[0] for g in ("dec", "mem", "ptr")}
    ok &= bool((red["dec"]["weight"] == 3.0).all()) and bool((red["dec"]["bias"] == 5.0).all()) and bool((red["mem"]["weight"] == 1.0).all())
    ok &= red["ptr"] == {}
    # a volume whose slices are ALL conditioning slices has no features to exchange (was StopIteration on multi-rank)
    ok &= par.gather_slice_features({}, [], []) == {}
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_two_rank_gloo_exchange():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = sorted(q.get(timeout=5) for _ in range(2))
    assert res == [(0, True), (1, True)]
