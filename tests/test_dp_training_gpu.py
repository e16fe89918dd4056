"""Data-parallel fine-tuning on the HIP path: two ranks (gloo rendezvous on 127.0.0.1, both on the one GPU of the test box), each with
half of the batch; the bucketed gradient all-reduce must reproduce the single-process full-batch gradients, and one Adam step must leave
both ranks with identical parameters."""
import os
import sys

import pytest
import torch
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from helpers import all_gather_flat, host_cpu_share, init_test_process_group, rank_device  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _inputs(B, E=16, C=256):
    rnd = lambda *s, seed=0, scale=1.0: torch.randn(*s, generator=torch.Generator().manual_seed(seed)) * scale
    emb, pe, sparse = rnd(B, C, E, E, seed=300), rnd(1, C, E, E, seed=301), rnd(B, 2, C, seed=302)
    f0, f1 = rnd(B, 32, 4 * E, 4 * E, seed=303), rnd(B, 64, 2 * E, 2 * E, seed=304)
    target = (rnd(B, 4, 4 * E, 4 * E, seed=305) > 0.4).float()
    return emb, pe, sparse, f0, f1, target


def _decoder():
    import medical_sam2_amd.build_sam as bs
    import medical_sam2_amd.weights as wts
    m = bs.build_sam2("sam2_hiera_t", device="cpu", hydra_overrides_extra=["++model.image_size=256"])
    m.load_state_dict(wts.init_weights("hiera_t", 0), strict=True)
    return m.sam_mask_decoder.cuda().eval()


def _grads(dec, sl, B_all):
    import medical_sam2_amd.backward as bwd
    import medical_sam2_amd.ops as ops
    import medical_sam2_amd.training as T
    emb, pe, sparse, f0, f1, target = _inputs(B_all)
    E = emb.shape[-1]
    d = lambda t: t[sl].cuda() if t.shape[0] == B_all else t.cuda()
    tm = lambda t: d(t).permute(0, 2, 3, 1).reshape(-1, t.shape[1]).contiguous()
    n = d(emb).shape[0]
    f0t, f1t = tm(f0).to(ops.OP16), tm(f1).to(ops.OP16)
    masks, _, _, _ = dec.predict_masks_tokens(tm(emb), tm(pe), d(sparse), f0t, f1t, n, E, E)
    loss, dm = T.bce_with_logits(masks, d(target), 1.0)
    _, _, g = bwd.mask_decoder_backward(dec, tm(emb), tm(pe), d(sparse), f0t, f1t, n, E, E, dm * 4096.0)
    return loss, {k: v / 4096.0 for k, v in g.items()}, (tm(emb), tm(pe), d(sparse), f0t, f1t, n, E, E, d(target))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    torch.set_grad_enabled(False)
    init_test_process_group(rank, world)
    import medical_sam2_amd.parallel as par
    import medical_sam2_amd.training as T
    B_all = 4
    per = B_all // world
    dec = _decoder()
    _, g_local, args = _grads(dec, slice(rank * per, (rank + 1) * per), B_all)
    g_sum, inv = par.allreduce_gradients(g_local)
    # reference on this rank: the full batch in one process (mean over 4 slices = mean of the two per-rank means)
    _, g_full, _ = _grads(dec, slice(0, B_all), B_all)
    num = sum(((g_sum[k] * inv).double() - g_full[k].double()).pow(2).sum().item() for k in g_full if not k.endswith("k_proj.bias"))
    den = sum(g_full[k].double().pow(2).sum().item() for k in g_full if not k.endswith("k_proj.bias"))
    rel = (num / den) ** 0.5
    # one data-parallel Adam step, then compare the parameters of the two ranks
    opt = T.DecoderAdam(dec, lr=1e-4)
    T.decoder_finetune_step(dec, opt, *args, data_parallel=True)
    flat = torch.cat([p.detach().reshape(-1) for p in dec.parameters()]).cpu()
    gathered = all_gather_flat(flat, world)
    same = all(torch.equal(gathered[0], t) for t in gathered[1:])
    q.put((rank, rel, bool(same)))
    dist.destroy_process_group()


def _joint_inputs(B, E=16, C=256, L=256):
    rnd = lambda *s, seed=0, scale=1.0: torch.randn(*s, generator=torch.Generator().manual_seed(seed)) * scale
    d = dict(curr=rnd(L, B, C, seed=400), curr_pos=rnd(L, B, C, seed=401), memory=rnd(2 * L + 4, B, 64, seed=402),
             memory_pos=rnd(2 * L + 4, B, 64, seed=403), pe=rnd(L, C, seed=404), sparse=rnd(B, 2, C, seed=405),
             f0=rnd(B * 16 * L, 32, seed=406), f1=rnd(B * 4 * L, 64, seed=407),
             target=(rnd(B, 4, 4 * E, 4 * E, seed=408) > 0.4).float())
    # very different gradient magnitudes on the two halves of the batch, so that a per-rank calibration WOULD pick different powers of two
    d["target"][B // 2:] = 1.0
    d["curr"][:, B // 2:] *= 8.0
    return d


def _joint_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    torch.set_grad_enabled(False)
    init_test_process_group(rank, world)
    import medical_sam2_amd.build_sam as bs
    import medical_sam2_amd.ops as ops
    import medical_sam2_amd.training as T
    import medical_sam2_amd.weights as wts
    m = bs.build_sam2("sam2_hiera_t", device="cpu", hydra_overrides_extra=["++model.image_size=256"])
    m.load_state_dict(wts.init_weights("hiera_t", 0), strict=True)
    m = m.cuda().eval()
    B_all, E = 4, 16
    per = B_all // world
    d = _joint_inputs(B_all)
    sl = slice(rank * per, (rank + 1) * per)
    c = lambda t: t.cuda().contiguous()
    L = E * E
    f0 = d["f0"].view(B_all, 16 * L, 32)[sl].reshape(-1, 32)
    f1 = d["f1"].view(B_all, 4 * L, 64)[sl].reshape(-1, 64)
    args = (c(d["curr"][:, sl]), c(d["curr_pos"][:, sl]), c(d["memory"][:, sl]), c(d["memory_pos"][:, sl]), 4, c(d["pe"]), c(d["sparse"][sl]),
            c(f0).to(ops.OP16), c(f1).to(ops.OP16), per, E, E, c(d["target"][sl]))
    # what a per-rank (non-collective) calibration would have chosen on this rank
    _, sc, sc_mem_local, _, _, _ = T.memory_decoder_loss_grads(m.memory_attention, m.sam_mask_decoder, *args)
    om, od = T.DecoderAdam(m.memory_attention, lr=1e-5), T.DecoderAdam(m.sam_mask_decoder, lr=1e-4)
    T.memory_decoder_finetune_step(m.memory_attention, m.sam_mask_decoder, om, od, *args, data_parallel=True)
    shared = om.calibrated_loss_scales[None]
    flat = torch.cat([p.detach().reshape(-1) for mod in (m.memory_attention, m.sam_mask_decoder) for p in mod.parameters()]).cpu()
    gathered = all_gather_flat(flat, world)
    same = all(torch.equal(gathered[0], t) for t in gathered[1:])
    q.put((rank, float(sc_mem_local / sc), float(shared), bool(same), bool(torch.isfinite(flat).all())))
    dist.destroy_process_group()


def test_data_parallel_joint_step_shares_the_loss_scale():
    """memory_decoder_finetune_step(data_parallel=True): the memory group's loss scale is calibrated COLLECTIVELY (MAX of max|d_src|
    over the ranks) -- with per-rank calibration the two halves of this batch pick different powers of two and the summed gradients
    would be wrong; afterwards both ranks hold the same scale and bit-identical parameters."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29800 + (os.getpid() % 90)
    procs = [ctx.Process(target=_joint_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    res = sorted(q.get(timeout=10) for _ in range(2))
    (_, loc0, sh0, same0, fin0), (_, loc1, sh1, same1, fin1) = res
    assert sh0 == sh1 and sh0 == min(loc0, loc1)          # the shared scale is the one of the larger gradient (MAX of amax)
    assert loc0 != loc1, "the test inputs must make a per-rank calibration disagree"
    assert same0 and same1 and fin0 and fin1


def _full_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    torch.set_grad_enabled(False)
    torch.cuda.set_device(rank_device(rank))
    torch.set_num_threads(max(1, host_cpu_share() // world))
    import medical_sam2_amd.build_sam as bs
    import medical_sam2_amd.synthetic as syn
    import medical_sam2_amd.training as T
    import medical_sam2_amd.weights as wts
    rnd = lambda *s, seed=0, scale=1.0: torch.randn(*s, generator=torch.Generator().manual_seed(seed)) * scale
    def model():
        m = bs.build_sam2("sam2_hiera_t", device="cpu", hydra_overrides_extra=["++model.image_size=256"])
        m.load_state_dict(wts.init_weights("hiera_t", 0), strict=True)
        return m.cuda().eval()
    B, S, E = 2, 256, 16
    imgs = torch.stack([syn.normalize_image(syn.blob_image(i, S)[0]) for i in range(B)]).cuda()
    pts = torch.tensor([[[100.0, 120.0]], [[60.0, 200.0]]]).cuda()
    labels = torch.ones(B, 1, dtype=torch.int32).cuda()
    memory, memory_pos = rnd(2 * E * E, B, 64, seed=150, scale=0.5).cuda(), rnd(2 * E * E, B, 64, seed=151).cuda()
    target = (rnd(B, 4, S // 4, S // 4, seed=152) > 0.3).float().cuda()
    # single process, full batch: reference gradients (lr 0: gradients only)
    m0 = model()
    full: dict = {}
    z = [T.DecoderAdam(mod, lr=0.0) for mod in (m0.memory_attention, m0.sam_mask_decoder, m0.image_encoder)]
    T.train_step_2d(m0, z[0], z[1], imgs, pts, labels, memory, memory_pos, target, opt_enc=z[2], grads_out=full)
    init_test_process_group(rank, world)
    sl = slice(rank, rank + 1)
    m = model()
    opts = [T.DecoderAdam(m.memory_attention, lr=1e-5), T.DecoderAdam(m.sam_mask_decoder, lr=1e-4), T.DecoderAdam(m.image_encoder, lr=1e-5)]
    got: dict = {}
    T.train_step_2d(m, opts[0], opts[1], imgs[sl], pts[sl], labels[sl], memory[:, sl].contiguous(), memory_pos[:, sl].contiguous(), target[sl],
                    opt_enc=opts[2], grads_out=got, data_parallel=True)
    # grads_out holds the all-reduced SUMS over the ranks; the full-batch mean gradient is their average
    rels = {}
    for grp in ("image_encoder", "memory_attention", "decoder"):
        num = sum(((got[grp][k].double() / world) - full[grp][k].double()).pow(2).sum().item() for k in full[grp] if not k.endswith("k_proj.bias"))
        den = sum(full[grp][k].double().pow(2).sum().item() for k in full[grp] if not k.endswith("k_proj.bias"))
        rels[grp] = (num / den) ** 0.5
    flat = torch.cat([p.detach().reshape(-1) for p in m.parameters()]).cpu()
    gathered = all_gather_flat(flat, world)
    same = all(torch.equal(gathered[0], t) for t in gathered[1:])
    q.put((rank, rels, bool(same), bool(torch.isfinite(flat).all())))
    dist.destroy_process_group()


def test_data_parallel_full_train_step_two_ranks():
    """BASELINE.json configs[4] in miniature: `train_step_2d(data_parallel=True)` with the image encoder trained -- two ranks with one image
    each; the all-reduced gradients of all three groups (sums, overlapped with the backward) equal the single-process full-batch
    gradients, and both ranks hold bit-identical parameters after the Adam step."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + (os.getpid() % 90)
    procs = [ctx.Process(target=_full_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    res = sorted(q.get(timeout=10) for _ in range(2))
    for rank, rels, same, finite in res:
        # batch-1 halves vs the batch-2 run: the same kernels on the same rows (every kernel is batch-invariant) -> only the 16-bit
        # rounding of the summed gradients differs; the decoder -> memory attention hand-off is ill-conditioned (test_backward_gpu.py)
        assert rels["image_encoder"] < 0.1 and rels["decoder"] < 0.1 and rels["memory_attention"] < 0.1, (rank, rels)
        assert same and finite, rank


def test_data_parallel_decoder_step_two_ranks():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + (os.getpid() % 200)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    res = sorted(q.get(timeout=10) for _ in range(2))
    for rank, rel, same in res:
        assert rel < 1e-2, (rank, rel)          # all-reduced half-batch gradients == full-batch gradients (16-bit operand noise)
        assert same, rank                       # identical parameters on both ranks after the step


def _bptt_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    torch.cuda.set_device(rank_device(rank))
    torch.set_num_threads(max(1, host_cpu_share() // world))
    torch.set_grad_enabled(False)
    import medical_sam2_amd.training as T
    import medical_sam2_amd.training_3d as t3
    import test_bptt_gpu as tb
    # every rank: the fixture's volume, its own time-reversed copy as "its" volume (rank 0: as is, rank 1: slices reversed)
    m, G, meta, volume, prompts, targets = tb._case()
    Tn = volume.shape[0]
    def mine(r):
        if r == 0:
            return volume, prompts, targets
        idx = list(range(Tn - 1, -1, -1))
        return volume[idx], {Tn - 1 - t: p for t, p in prompts.items()}, {Tn - 1 - t: v for t, v in targets.items()}
    # single-process reference: mean of the two ranks' gradients
    ref = []
    for r in range(world):
        out = {}
        t3.train_step_3d(m, {}, *mine(r), pos_weight=meta["pos_weight"], grads_out=out)
        g = {("d." + k): v for k, v in out["prompt"]["decoder"].items()}
        for k, v in out["non_prompt"]["decoder"].items():
            g["d." + k] = g["d." + k] + v
        for grp in ("memory_attention", "memory_encoder", "obj_ptr_proj"):
            g.update({grp[:8] + "." + k: v for k, v in out["non_prompt"][grp].items()})
        ref.append(g)
    mean = {k: (ref[0][k] + ref[1][k]) * 0.5 for k in ref[0]}
    init_test_process_group(rank, world)
    lr = 1e-3
    opts = {"decoder": T.DecoderAdam(m.sam_mask_decoder, lr=lr), "memory_attention": T.DecoderAdam(m.memory_attention, lr=lr),
            "memory_encoder": T.DecoderAdam(m.memory_encoder, lr=lr), "obj_ptr_proj": T.DecoderAdam(m.obj_ptr_proj, lr=lr)}
    before = {k: p.detach().clone() for k, p in m.named_parameters()}
    t3.train_step_3d(m, opts, *mine(rank), pos_weight=meta["pos_weight"], data_parallel=True)
    # first Adam step = -lr * sign-ish(g): compare the update direction with the mean gradient where it is not within rounding of zero
    agree, total = 0, 0
    pre = {"d": "sam_mask_decoder", "memory_a": "memory_attention", "memory_e": "memory_encoder", "obj_ptr_": "obj_ptr_proj"}
    for k, g in mean.items():
        grp, name = k.split(".", 1)
        p = dict(m.named_parameters())[pre[grp] + "." + name]
        delta = (p.detach() - before[pre[grp] + "." + name])
        big = g.abs() > 0.05 * g.abs().max()
        if big.any():
            agree += int((torch.sign(delta[big]) == -torch.sign(g[big])).sum())
            total += int(big.sum())
    flat = torch.cat([p.detach().reshape(-1) for p in m.parameters()]).cpu()
    gathered = all_gather_flat(flat, world)
    q.put((rank, agree / max(total, 1), bool(all(torch.equal(gathered[0], t) for t in gathered[1:])), bool(torch.isfinite(flat).all())))
    dist.destroy_process_group()


def test_data_parallel_train_step_3d_two_ranks():
    """train_step_3d(data_parallel=True), two ranks on different volumes: the Adam update follows the MEAN of the two ranks' chained
    (back-propagation-through-time) gradients and both ranks end with bit-identical parameters."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + (os.getpid() % 90)
    procs = [ctx.Process(target=_bptt_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    for rank, frac, same, finite in sorted(q.get(timeout=10) for _ in range(2)):
        assert same and finite and frac > 0.97, (rank, frac, same, finite)
