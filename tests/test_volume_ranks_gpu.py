"""Multi-rank `segment_volume` on the HIP path: two ranks (gloo rendezvous on 127.0.0.1, both on the one GPU of the test box) against the
single-rank result, bit for bit, on every slice and for every object -- the slice-sharded encoder + conditioning pass, the exchange of
memories / features, the object-sharded chain (n_obj = 2), the cross-rank key split of the memory cross-attention (n_obj = 1), and a
volume with fewer conditioning slices than ranks (one rank joins the exchange empty-handed)."""
import os
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu
S, T = 256, 8


def _model():
    import medical_sam2_amd.build_sam as bs
    import medical_sam2_amd.weights as wts
    m = bs.build_sam2("sam2_hiera_t", device="cpu", hydra_overrides_extra=[f"++model.image_size={S}"])
    m.load_state_dict(wts.init_weights("hiera_t", 0), strict=True)
    return m.cuda().eval()


def _case(n_obj, cond_slices):
    import medical_sam2_amd.synthetic as syn
    volume, boxes = syn.blob_volume(3, n_slices=T, size=S, n_objects=n_obj)
    def box_at(t):
        return torch.tensor([[float(v) for v in (boxes[o][t] or (S * 0.3, S * 0.3, S * 0.6, S * 0.6))] for o in range(n_obj)])
    return volume.cuda(), {t: {"boxes": box_at(t).cuda()} for t in cond_slices}


CASES = [(1, (0, 4)), (2, (0, 4)), (1, (0,)), (3, (2,))]


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    torch.set_grad_enabled(False)
    torch.cuda.set_device(0)
    import medical_sam2_amd.parallel as par
    import medical_sam2_amd.volume as vol
    m = _model()
    single = [vol.segment_volume(m, *_case(n, c), fill_hole_area=8) for n, c in CASES]     # before the process group exists: one rank
    dist.init_process_group("gloo", rank=rank, world_size=world)
    res = []
    for (n, c), ref in zip(CASES, single):
        real = par.KVSplit.exchange
        calls = []
        par.KVSplit.exchange = lambda self, *a, **k: (calls.append(1), real(self, *a, **k))[1]
        try:
            got = vol.segment_volume(m, *_case(n, c), fill_hole_area=8)
        finally:
            par.KVSplit.exchange = real
        same = sorted(got) == list(range(T)) and all(got[t].shape == ref[t].shape and torch.equal(got[t], ref[t]) for t in range(T))
        res.append((n, len(c), bool(same), len(calls)))
    q.put((rank, res))
    dist.destroy_process_group()


def test_segment_volume_two_ranks_equal_one_rank():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29900 + (os.getpid() % 90)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    out = sorted(q.get(timeout=10) for _ in range(2))
    for rank, res in out:
        for n_obj, n_cond, same, kv_calls in res:
            assert same, (rank, n_obj, n_cond)
            # one object: the chain runs with the cross-rank key split (exchanges happened); >= 2 objects: object shards, no key split
            assert (kv_calls > 0) == (n_obj < 2), (rank, n_obj, kv_calls)
