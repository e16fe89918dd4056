"""The RCCL branch of the multi-GPU plumbing, executed on the one GPU of the test box (every multi-rank test used
gloo, so `backend="nccl"` -- RCCL on ROCm -- had never run): a world of ONE rank with `parallel.FORCE_SINGLE_RANK_COLLECTIVES`, in a fresh
child process.  Every collective of `parallel` goes through RCCL (init with device_id, coalesced in-place all-reduce, ragged all-gathers,
object all-gather, barrier with device_ids), is ordered against the HIP kernels around it on the same stream, and the results must equal
the single-process ones; `segment_volume` runs its whole distributed path (exchange + key-split chain) and must return the same bits."""
import os
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _worker(port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    torch.set_grad_enabled(False)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    import medical_sam2_amd.build_sam as bs
    import medical_sam2_amd.parallel as par
    import medical_sam2_amd.synthetic as syn
    import medical_sam2_amd.training as T
    import medical_sam2_amd.volume as vol
    import medical_sam2_amd.weights as wts
    S, NT = 256, 6
    m = bs.build_sam2("sam2_hiera_t", device="cpu", hydra_overrides_extra=[f"++model.image_size={S}"])
    m.load_state_dict(wts.init_weights("hiera_t", 0), strict=True)
    m = m.to(dev).eval()
    volume, boxes = syn.blob_volume(3, n_slices=NT, size=S, n_objects=2)
    box_at = lambda t: torch.tensor([[float(v) for v in (boxes[o][t] or (S * 0.3, S * 0.3, S * 0.6, S * 0.6))] for o in range(2)], device=dev)
    prompts = {t: {"boxes": box_at(t)} for t in (0, 4)}
    single = vol.segment_volume(m, volume.to(dev), prompts, fill_hole_area=8)            # no process group yet
    # ---- RCCL
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    res = {"backend": dist.get_backend()}
    par.FORCE_SINGLE_RANK_COLLECTIVES = True
    par.barrier(dev)
    res["max_over_ranks"] = par.max_over_ranks(1.25, dev) == 1.25
    # 1. in-place coalesced async all-reduce of live gradients (two buckets; one 16-bit and one strided entry take the copy path),
    #    consumed right away by a kernel on the same stream (Adam), against the plain single-process step
    lin_a, lin_b = torch.nn.Linear(64, 48).to(dev), torch.nn.Linear(64, 48).to(dev)
    lin_b.load_state_dict(lin_a.state_dict())
    g = torch.Generator().manual_seed(3)
    grads = {"weight": torch.randn(48, 64, generator=g).to(dev), "bias": torch.randn(48, generator=g).to(dev)}
    want = {k: v.clone() for k, v in grads.items()}
    extra = {"z16": torch.randn(8, 8, generator=g).to(dev).half(), "strided": torch.randn(6, 10, generator=g).to(dev)[:, ::2]}
    pend = par.allreduce_gradients_async({**grads, **extra}, bucket_bytes=4096)
    red, inv = pend.wait()
    oa, ob = T.DecoderAdam(lin_a, lr=1e-2), T.DecoderAdam(lin_b, lr=1e-2)
    oa.step({k: red[k] for k in grads}, grad_scale=inv)
    ob.step(want)
    torch.cuda.synchronize()
    res["allreduce"] = inv == 1.0 and all(torch.equal(red[k], want[k]) for k in want) and torch.equal(red["z16"], extra["z16"].float()) \
        and torch.equal(red["strided"], extra["strided"]) and torch.equal(lin_a.weight, lin_b.weight) and torch.equal(lin_a.bias, lin_b.bias)
    # 2. the exchange helpers: ragged all-gathers of fp32 / raw-byte slabs, object all-gather of the shape table
    pos = [torch.zeros(2, 64, 4, 4, device=dev)]
    local = {f: {"maskmem_features": torch.full((2, 64, 4, 4), float(f), device=dev), "maskmem_pos_enc": pos,
                 "obj_ptr": torch.full((2, 256), float(f), device=dev), "pred_masks": torch.full((2, 1, 16, 16), float(f), device=dev)} for f in (0, 2, 5)}
    full = par.gather_cond_memories(local, [0, 2, 5])
    res["gather_cond"] = sorted(full) == [0, 2, 5] and all(bool((full[f]["obj_ptr"] == f).all()) for f in full)
    mk = lambda t: {"backbone_fpn": [torch.full((1, 4, 8, 8), float(t), device=dev)], "vision_pos_enc": ["pos"]}
    feats = par.gather_slice_features({t: mk(t) for t in (1, 3)}, [1, 3], [0, 0])
    res["gather_feats"] = sorted(feats) == [1, 3] and bool((feats[3]["backbone_fpn"][0] == 3).all())
    sh = {t: torch.full((2, 1, 2, 2), float(t), device=dev) for t in (0, 1)}
    fm = par.gather_object_shards(sh, [0, 1], 2)
    res["gather_objects"] = all(torch.equal(fm[t], sh[t]) for t in (0, 1))
    ws = torch.arange(5 * 6 * 128 + 5 * 6 * 8, dtype=torch.int64, device=dev).to(torch.uint8)
    keep = ws.clone()
    with par.KVSplit() as kvs:
        res["kv_share"] = kvs.share(5) == (0, 5)
        kvs.exchange(ws, 5, 6)
    res["kv_exchange"] = torch.equal(ws, keep)
    un = par.union_gradient_keys({"a": {"weight": want["weight"]}, "b": {}}, {"a": lin_a, "b": lin_b}, ("a", "b"))
    res["union_keys"] = sorted(un["a"]) == ["weight"] and un["b"] == {}
    # 3. the whole distributed path of segment_volume over RCCL: exchange of memories / features, then the key-split chain
    #    (shard_objects=False -> KVSplit with this rank owning every split) and the object-sharded chain
    calls = []
    real = par.KVSplit.exchange
    par.KVSplit.exchange = lambda self, *a, **k: (calls.append(1), real(self, *a, **k))[1]
    try:
        got_kv = vol.segment_volume(m, volume.to(dev), prompts, fill_hole_area=8, shard_objects=False)
    finally:
        par.KVSplit.exchange = real
    got_obj = vol.segment_volume(m, volume.to(dev), prompts, fill_hole_area=8)
    res["volume_kv_split"] = all(torch.equal(got_kv[t], single[t]) for t in range(NT)) and len(calls) > 0
    res["volume_object_shards"] = all(torch.equal(got_obj[t], single[t]) for t in range(NT))
    par.barrier(dev)
    q.put(res)
    dist.destroy_process_group()


def test_every_collective_runs_through_rccl_in_a_world_of_one():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker, args=(29500 + (os.getpid() % 2000) + 2000, q))
    p.start()
    p.join(600)
    assert p.exitcode == 0, p.exitcode
    res = q.get(timeout=5)
    assert res.pop("backend") == "nccl"
    assert all(res.values()), res
