"""A captured training step (training.GraphedStep) with EAGER readers between its replays.

`refresh_casts` runs inside the captured `train_step_2d` and copies the updated fp32 weights IN PLACE into 16-bit tensors that were
allocated eagerly and are owned by the modules' `WeightCache` only: the hipGraph bakes those raw pointers.  `GraphedStep` promises
eager code between replays (validation, the predictor, `_encode_new_memory`); after `mark_updated()` such a reader used to get a cache
miss that REPLACED the entry -- the old block went back to the caching allocator and the next replay wrote weights into, and ran
GEMMs out of, whatever owned that block by then.  Now every reader refreshes a plain cast in place (stable storage) and the graph
pins what it read.  This test does exactly the sequence the advisor asked for: replay, eager eval forward, fill the allocator's free
blocks with sentinels, replay -- the sentinels must be intact, the cast pointers unchanged, and the parameters equal to an eager-only
run of the same four iterations."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pytestmark = pytest.mark.gpu
DEV = "cuda"


def rnd(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def _setup():
    import medical_sam2_amd.build_sam as bs
    import medical_sam2_amd.synthetic as syn
    import medical_sam2_amd.training as T
    import medical_sam2_amd.weights as wts
    m = bs.build_sam2("sam2_hiera_t", device="cpu", hydra_overrides_extra=["++model.image_size=256"])
    m.load_state_dict(wts.init_weights("hiera_t", 0), strict=True)
    m = m.to(DEV).eval()
    B, S, E = 2, 256, 16
    imgs = torch.stack([syn.normalize_image(syn.blob_image(i, S)[0]) for i in range(B)])
    pts, labels = torch.tensor([[[100.0, 120.0]], [[60.0, 200.0]]]), torch.ones(B, 1, dtype=torch.int32)
    memory, memory_pos = rnd(2 * E * E, B, 64, seed=150, scale=0.5), rnd(2 * E * E, B, 64, seed=151)
    target = (rnd(B, 4, S // 4, S // 4, seed=152) > 0.3).float()
    args = tuple(t.to(DEV) for t in (imgs, pts, labels, memory, memory_pos, target))
    opts = [T.DecoderAdam(m.memory_attention, lr=1e-5), T.DecoderAdam(m.sam_mask_decoder, lr=1e-4), T.DecoderAdam(m.image_encoder, lr=1e-5)]
    step = lambda sync: T.train_step_2d(m, opts[0], opts[1], *args, sync=sync, opt_enc=opts[2])
    return m, opts, step, args


def _eval_forward(m, imgs):
    """what validation does between two training steps: an eager forward of the encoder + memory encoding on the current weights"""
    bo = m.forward_image(imgs)
    _, feats, pos, sizes = m._prepare_backbone_features(bo)
    hr = torch.zeros(imgs.shape[0], 1, m.image_size, m.image_size, device=imgs.device)
    mem, _ = m._encode_new_memory(current_vision_feats=feats, feat_sizes=sizes, pred_masks_high_res=hr, is_mask_from_pts=True)
    return feats[-1].float().sum() + mem.float().sum()


def test_eager_forward_between_graph_replays_keeps_weight_copies_alive():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import medical_sam2_amd.ops as ops
    import medical_sam2_amd.training as T
    from medical_sam2_amd.modeling.common import WeightCache
    with torch.no_grad():
        # ---- eager-only run: calibrating step, 3 more steps, the same eval forward before the last one
        m_e, _, step_e, args = _setup()
        for it in range(4):
            if it == 3:
                _eval_forward(m_e, args[0])
            step_e(True)
        ref = {k: v.detach().clone() for k, v in m_e.state_dict().items()}
        del m_e, step_e
        # ---- graphed run: calibrating step, GraphedStep (its warm-up call = step 2), replay = step 3, eager forward, replay = step 4
        m, opts, step, args = _setup()
        step(True)
        gs = T.GraphedStep(lambda: step(False), opts, model=m)
        gs.replay()
        casts = {}
        for name, mod in m.named_modules():
            wc = getattr(mod, "_wc", None)
            if isinstance(wc, WeightCache):
                for key in wc._casts:
                    casts[(name, key)] = (wc, wc._c[key][1].data_ptr(), wc._c[key][1].numel())
        assert len(casts) > 100                                    # the trunk's, the memory attention's and the decoder's Linear weights
        torch.cuda.synchronize()
        chk = float(_eval_forward(m, args[0]).item())             # eager readers: every stale cast is refreshed
        assert chk == chk
        moved = [k for k, (wc, ptr, _) in casts.items() if wc._c[k[1]][1].data_ptr() != ptr]
        assert not moved, f"{len(moved)} 16-bit weight copies were re-allocated by an eager reader between replays: {moved[:4]}"
        # whatever an eager reader might have freed is now taken by sentinels of exactly those sizes
        sentinels = [torch.full((n,), 7.0, dtype=ops.OP16, device=DEV) for (_, _, n) in casts.values()]
        torch.cuda.synchronize()
        gs.replay()
        torch.cuda.synchronize()
        bad = sum(int((s != 7.0).sum().item()) for s in sentinels)
        assert bad == 0, f"the replay wrote into {bad} elements of tensors it does not own"
        # ---- the parameters follow the eager-only run: the backward's split-K sums use fp32 atomics, so a gradient element near 0 may
        # change sign between runs and Adam moves that element by lr either way -- per step at most 2 lr, everything else to rounding
        lr = {"memory_attention": 1e-5, "sam_mask_decoder": 1e-4, "image_encoder": 1e-5}
        worst, frac = {}, {}
        for k, v in m.state_dict().items():
            grp = k.split(".")[0]
            if grp not in lr or not v.dtype.is_floating_point:
                assert torch.equal(v, ref[k]), k
                continue
            d = (v - ref[k]).abs()
            worst[grp] = max(worst.get(grp, 0.0), float(d.max().item()) / lr[grp])
            frac.setdefault(grp, []).append((int((d > 0.5 * lr[grp]).sum().item()), d.numel()))
        share = {g: sum(a for a, _ in f) / sum(b for _, b in f) for g, f in frac.items()}
        print("graph vs eager after 4 iterations: worst |d| in units of lr", worst, "share of elements off by > lr/2", share)
        assert all(w <= 8.0 for w in worst.values()), worst       # 4 steps x at most 2 lr
        # The share of such elements is a property of the atomics' noise, not of the weight copies this test is about (a stale or freed
        # copy moves EVERY element of a tensor, and by more than 2 lr per step): measured 0.020 / 0.024 / 0.022 (encoder / memory attention /
        # decoder) on the bf16 library in one run and under 0.02 in others -- the earlier bar of 0.02 sat inside that spread and turned the
        # driver's command red on an unlucky run.  0.05 keeps a factor of two over the measured spread and 20x under a corrupted tensor.
        assert all(s < 0.05 for s in share.values()), share
