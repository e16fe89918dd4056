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
    full = par.gather_cond_memories(local, frame_ids)
    assert sorted(full) == frame_ids
    ok = True
    for f in frame_ids:
        g = torch.Generator().manual_seed(f)
        ok &= torch.equal(full[f]["maskmem_features"], torch.randn(2, 64, 4, 4, generator=g))
        ok &= torch.equal(full[f]["obj_ptr"], torch.randn(2, 256, generator=g))
    # data-parallel gradient reduction: ragged shapes, two buckets forced by a small bucket size
    gen = lambda r, n, shape: torch.randn(*shape, generator=torch.Generator().manual_seed(100 * r + n))
    shapes = {"b.weight": (7, 5), "a.bias": (3,), "c.gamma": (1, 2, 3), "d.w": (40, 10)}
    mine = {k: gen(rank, i, sh) for i, (k, sh) in enumerate(shapes.items())}
    red, inv = par.allreduce_gradients(mine, bucket_bytes=1024)
    assert inv == 1.0 / world and set(red) == set(shapes)
    for i, (k, sh) in enumerate(shapes.items()):
        want = sum(gen(r, i, sh) for r in range(world))
        ok &= red[k].shape == want.shape and torch.allclose(red[k], want, atol=1e-6)
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
