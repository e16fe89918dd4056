"""Backward of the image encoder (medical_sam2_amd.backward_encoder: Hiera trunk + FPN neck with the folded conv_s0 / conv_s1) against
torch.autograd through the fp32 oracle (oracle.forward_image) -- the gradients the 2-D training loop needs at func_2d/function.py:70-72,
246-259.  hiera_t at 256^2: windows of 8 / 4 / 14 (16 -> 28 padded) / 7 (8 -> 14 padded), three q-pool blocks, three global blocks.
Tolerances: 16-bit operands on both sides of every backward GEMM and of the attention backward, fp32 accumulation."""
import os
import sys

import pytest
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
from helpers import btol, gradient_step_drops, op16_is_fp16  # noqa: E402
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import sam2_oracle as O  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(autouse=True)
def _grad_on():
    with torch.enable_grad():
        yield


def rnd(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def _model(name="hiera_t", S=256):
    import medical_sam2_amd.build_sam as bs
    import medical_sam2_amd.weights as wts
    m = bs.build_sam2("sam2_" + name, device="cpu", hydra_overrides_extra=[f"++model.image_size={S}"])
    W = wts.init_weights(name, 0)
    m.load_state_dict(W, strict=True)
    return m.to(DEV).eval(), W


def test_adjoint_kernels():
    """maxpool backward (first-maximum routing incl. ties), 2x2 sum pool, position-embedding adjoint vs autograd of the torch ops"""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import torch.nn.functional as F
    import medical_sam2_amd.backward_encoder as be
    import medical_sam2_amd.ops as ops
    from medical_sam2_amd._lib import check, lib
    B, H, W, C = 2, 6, 8, 40
    x = rnd(B, H, W, C, seed=1)
    x[0, 0, 0] = x[0, 0, 1]                                     # ties: the first maximum wins
    xr = x.clone().requires_grad_(True)
    y = F.max_pool2d(xr.permute(0, 3, 1, 2), 2, 2)
    dy = rnd(*y.shape, seed=2)
    y.backward(dy)
    got = be.maxpool2x2_backward(x.reshape(-1, C).to(DEV), dy.permute(0, 2, 3, 1).reshape(-1, C).contiguous().to(DEV), B, H, W)
    assert torch.equal(got.cpu().view(B, H, W, C), xr.grad)
    got16 = be.maxpool2x2_backward(x.reshape(-1, C).to(ops.OP16).to(DEV), dy.permute(0, 2, 3, 1).reshape(-1, C).contiguous().to(DEV), B, H, W)
    x16 = x.to(ops.OP16).float().requires_grad_(True)
    F.max_pool2d(x16.permute(0, 3, 1, 2), 2, 2).backward(dy)
    assert torch.equal(got16.cpu().view(B, H, W, C), x16.grad)
    top = rnd(B, H // 2, W // 2, C, seed=3).requires_grad_(True)
    up = F.interpolate(top.permute(0, 3, 1, 2), scale_factor=2.0, mode="nearest")
    g = rnd(B, C, H, W, seed=4)
    up.backward(g)
    got = be.sumpool2x2(g.permute(0, 2, 3, 1).reshape(-1, C).contiguous().to(DEV), B, H, W)
    assert torch.allclose(got.cpu().view(B, H // 2, W // 2, C), top.grad, atol=1e-6)
    pe, pw = rnd(1, 24, 7, 7, seed=5).requires_grad_(True), rnd(1, 24, 8, 8, seed=6).requires_grad_(True)
    h = w = 64
    table = F.interpolate(pe, size=(h, w), mode="bicubic") + pw.tile([x // y for x, y in zip((1, 24, h, w), pw.shape)])
    d = rnd(h * w, 24, seed=7)
    table.backward(d.t().reshape(1, 24, h, w))
    dpe, dpw = torch.empty(1, 24, 7, 7, device=DEV), torch.empty(1, 24, 8, 8, device=DEV)
    nb = lib().msam2_hiera_pos_embed_bwd_workspace_bytes(24, 7, h, 8)
    ws = torch.empty(nb, dtype=torch.uint8, device=DEV)
    check(lib().msam2_hiera_pos_embed_bwd(ops._p(d.to(DEV)), ops._p(dpe), ops._p(dpw), 24, 7, 7, h, w, 8, ops._p(ws), nb, ops._stream()))
    assert rel(dpw, pw.grad) < 1e-5 and rel(dpe, pe.grad) < 1e-4


def test_window_pad_colsum_is_exact_and_deterministic():
    """msam2_window_pad_colsum (the qkv-bias share of the zero-padded window tokens, hieradet.py:143-150 + utils.py:28-31): equals the masked
    sum over the padded tokens, accumulates into its output, and -- round 4, VERDICT r3 item 1a -- repeated launches give the SAME BITS
    (the first form added one fp32 atomic per window, so the encoder gradient changed in its last bits from run to run)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from medical_sam2_amd._lib import check, lib
    import medical_sam2_amd.ops as ops
    for (B, H, W, heads, D, ws) in ((4, 64, 64, 4, 96, 14), (2, 32, 32, 8, 96, 7), (1, 16, 20, 2, 56, 14), (2, 16, 16, 2, 24, 8)):
        nwy, nwx = -(-H // ws), -(-W // ws)
        win = rnd(B * nwy * nwx, heads, ws * ws, D, seed=ws + D).to(DEV)
        ty, tx = torch.meshgrid(torch.arange(ws), torch.arange(ws), indexing="ij")
        ref = torch.zeros(heads * D, dtype=torch.float64)
        for w in range(B * nwy * nwx):
            wx, wy = w % nwx, (w // nwx) % nwy
            pad = ((wy * ws + ty >= H) | (wx * ws + tx >= W)).reshape(-1)
            ref += win[w].double().cpu()[:, pad].sum(1).reshape(-1)
        outs = []
        for rep in range(3):
            out = torch.full((heads * D,), 1.5, dtype=torch.float32, device=DEV)          # accumulates into its output
            check(lib().msam2_window_pad_colsum(win.data_ptr(), out.data_ptr(), B, H, W, heads, D, ws, torch.cuda.current_stream().cuda_stream))
            outs.append(out.cpu())
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
        err = (outs[0].double() - 1.5 - ref).abs().max().item()
        assert err <= 1e-4 * max(1.0, ref.abs().max().item()), (B, H, W, heads, D, ws, err)


def test_hiera_block_backward_variants():
    """single blocks: windowed with padding (ws 14 on 16x16), q-pool + projection + window, global -- every parameter and the input"""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import medical_sam2_amd.backward_encoder as be
    m, W = _model()
    specs = O.hiera_block_specs(O.model_config("hiera_t", 256)["trunk"])
    B = 2
    worst = {}
    for i in (0, 1, 3, 4, 5, 10, 11):                          # ws 8 | pool 8->4 | pool 4->2 | ws 14 padded | global | pool 14->7 | ws 7 padded
        spec = specs[i]
        pre = f"image_encoder.trunk.blocks.{i}"
        side = {96: 64, 192: 32, 384: 16, 768: 8}[spec["dim"]]
        P = {k: v.clone().requires_grad_(True) for k, v in W.items() if k.startswith(pre + ".")}
        x = rnd(B, side, side, spec["dim"], seed=20 + i).requires_grad_(True)
        y = O.multiscale_block(P, pre, x, spec)
        dy = rnd(*y.shape, seed=40 + i, scale=0.05)
        y.backward(dy)
        blk = m.image_encoder.trunk.blocks[i]
        dt, g = be.hiera_block_backward(blk, x.detach().reshape(-1, spec["dim"]).to(DEV), B, side, side,
                                        dy.reshape(-1, spec["dim_out"]).contiguous().to(DEV))
        errs = {"dx": rel(dt.view_as(x), x.grad)}
        for k, v in g.items():
            errs[k] = rel(v, P[pre + "." + k].grad)
        assert set(g) == {k[len(pre) + 1:] for k in P}, (i, sorted(set(g) ^ {k[len(pre) + 1:] for k in P}))
        worst[i] = max(errs.items(), key=lambda kv: kv[1])
        assert worst[i][1] < btol(3e-2), (i, spec, sorted(errs.items(), key=lambda kv: -kv[1])[:5])
    print("worst per block:", worst)


@pytest.mark.parametrize("name", ["hiera_t", "hiera_b+"])
def test_image_encoder_backward_vs_autograd(name):
    """whole encoder: gradients of a random linear functional of the three FPN outputs w.r.t. every image_encoder parameter and the two
    folded high-res convs of the mask decoder.  hiera_b+ is BASELINE configs[4]'s model (24 blocks, head dim 56 zero-padded to the
    64-wide attention kernels, in the backward as in the forward)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import medical_sam2_amd.backward_encoder as be
    import medical_sam2_amd.synthetic as syn
    m, W = _model(name)
    cfg = O.model_config(name, 256)
    img, _, _ = syn.image_batch([10, 11], 256)
    train = lambda k: k.startswith("image_encoder.") or k.startswith("sam_mask_decoder.conv_s")
    P = {k: (v.clone().requires_grad_(True) if train(k) else v) for k, v in W.items()}
    bo = O.forward_image(P, cfg, img)
    dys = [rnd(*f.shape, seed=60 + l, scale=0.05) for l, f in enumerate(bo["backbone_fpn"])]
    sum((f * d).sum() for f, d in zip(bo["backbone_fpn"], dys)).backward()
    with torch.no_grad():
        out, st = be.image_encoder_forward_saved(m, img.to(DEV))
        for l in range(3):
            assert rel(out["backbone_fpn"][l], bo["backbone_fpn"][l]) < btol(3e-3)
        d_fpn = [d.permute(0, 2, 3, 1).reshape(-1, d.shape[1]).contiguous().to(DEV) for d in dys]
        grads = be.image_encoder_backward(m, st, d_fpn)
    want = {k for k in P if train(k) and P[k].grad is not None and not k.startswith("image_encoder.neck.convs.0.")}   # level 3 is scalped: no gradient... except through the top-down path
    errs = {k: rel(grads[k], P[k].grad) for k in grads}
    missing = {k for k in P if train(k) and P[k].grad is not None and P[k].grad.abs().sum() > 0} - set(grads)
    assert not missing, sorted(missing)[:10]
    num = sum((grads[k].double().cpu() - P[k].grad.double()).pow(2).sum().item() for k in grads)
    den = sum(P[k].grad.double().pow(2).sum().item() for k in grads)
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:8]
    print("overall", (num / den) ** 0.5, "worst", worst)
    assert (num / den) ** 0.5 < btol(2e-2) and worst[0][1] < btol(8e-2, 2.0), worst


def test_train_step_2d_with_image_encoder():
    """The whole 2-D training iteration with the image encoder trained (train_2d.py:43-47, func_2d/function.py:70-72): the encoder's
    gradients of the mask loss -- through decoder -> memory attention -> FPN -> trunk -- against autograd through the whole oracle chain,
    then Adam moves all three groups and the loss goes down."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import torch.nn.functional as F
    import medical_sam2_amd.synthetic as syn
    import medical_sam2_amd.training as T
    m, W = _model()
    cfg = O.model_config("hiera_t", 256)
    B, S, E = 2, 256, 16
    imgs = torch.stack([syn.normalize_image(syn.blob_image(i, S)[0]) for i in range(B)])
    pts = torch.tensor([[[100.0, 120.0]], [[60.0, 200.0]]])
    labels = torch.ones(B, 1, dtype=torch.int32)
    memory, memory_pos = rnd(2 * E * E, B, 64, seed=150, scale=0.5), rnd(2 * E * E, B, 64, seed=151)
    target = (rnd(B, 4, S // 4, S // 4, seed=152) > 0.3).float()
    # ---- oracle: autograd through the whole chain w.r.t. the image encoder (+ conv_s0 / conv_s1)
    train = lambda k: k.startswith("image_encoder.") or k.startswith("sam_mask_decoder.conv_s")
    P = {k: (v.clone().float().requires_grad_(True) if train(k) else v.float()) for k, v in W.items()}
    feats, poss, sizes = O.prepare_backbone_features(O.forward_image(P, cfg, imgs))
    y = O.memory_attention(P, cfg, feats[-1], memory, poss[-1], memory_pos, 0)
    emb = y.permute(1, 2, 0).reshape(B, 256, E, E)
    hr = [f.permute(1, 2, 0).reshape(B, -1, *sz) for f, sz in zip(feats[:-1], sizes[:-1])]
    se, de = O.prompt_encoder(P, cfg, (pts, labels), None, None)
    masks, _, _, _ = O.mask_decoder_predict(P, emb, O.dense_pe(P, E, E), se, de, hr)
    ref_loss = F.binary_cross_entropy_with_logits(masks, target)
    ref_loss.backward()
    # ---- HIP path: one iteration with lr 0 (gradients only), then real steps
    d = lambda t: t.to(DEV)
    args = (d(imgs), d(pts), d(labels), d(memory), d(memory_pos), d(target))
    with torch.no_grad():
        zero = [T.DecoderAdam(mod, lr=0.0) for mod in (m.memory_attention, m.sam_mask_decoder, m.image_encoder)]
        got: dict = {}
        loss0, _ = T.train_step_2d(m, zero[0], zero[1], *args, opt_enc=zero[2], grads_out=got)
    assert abs(loss0 - ref_loss.item()) < 5e-3 * abs(ref_loss.item())
    g = {"image_encoder." + k: v for k, v in got["image_encoder"].items()}
    g.update({"sam_mask_decoder." + k: v for k, v in got["decoder"].items() if k.startswith("conv_s")})
    missing = {k for k in P if train(k) and P[k].grad is not None and P[k].grad.abs().sum() > 0} - set(g)
    assert not missing, sorted(missing)[:10]
    num = sum((g[k].double().cpu() - P[k].grad.double()).pow(2).sum().item() for k in g)
    den = sum(P[k].grad.double().pow(2).sum().item() for k in g)
    errs = sorted(((rel(g[k], P[k].grad), k) for k in g), reverse=True)
    print("encoder gradients of the mask loss: overall", (num / den) ** 0.5, "worst", errs[:6])
    # (the decoder is linearised at the HIP forward's own point: its input gradient is sensitive to the 16-bit forward error --
    #  DESIGN 7.2 -- so the end-to-end bar is looser than the encoder-only test above)
    assert (num / den) ** 0.5 < 0.12, errs[:6]
    by_stage = {}
    for e, k in errs:
        key = k.split(".blocks.")[1].split(".")[0] if ".blocks." in k else k.split(".")[1]
        by_stage[key] = max(by_stage.get(key, 0.0), e)
    print("worst per block / part:", by_stage)
    with torch.no_grad():
        # sign and scale of every group's gradient: a plain step p <- p - eta g sized for a first-order drop of 2 % lowers the loss by
        # about that (curvature takes some of it)
        drops = gradient_step_drops(lambda: T.train_step_2d(m, zero[0], zero[1], *args, opt_enc=zero[2])[0],
                                    {"decoder": m.sam_mask_decoder, "memory_attention": m.memory_attention, "image_encoder": m.image_encoder}, got, loss0)
        print("relative loss drop of a plain gradient step per group (predicted 0.02):", drops)
        assert all(0.005 < d < 0.03 for d in drops.values()), drops
        before = {k: v.detach().clone() for k, v in m.state_dict().items()}
        opts = [T.DecoderAdam(m.memory_attention, lr=1e-5), T.DecoderAdam(m.sam_mask_decoder, lr=1e-4), T.DecoderAdam(m.image_encoder, lr=1e-5)]
        losses = [T.train_step_2d(m, opts[0], opts[1], *args, opt_enc=opts[2])[0] for _ in range(4)]
        moved = {k.split(".")[0] for k, v in m.state_dict().items() if not torch.equal(v, before[k])}
    assert moved == {"memory_attention", "sam_mask_decoder", "image_encoder"}, moved
    assert all(x == x and abs(x) < 1e6 for x in losses), losses
    assert all(torch.isfinite(v).all() for v in m.state_dict().values())
    assert all(o.skipped_elements == 0 for o in opts)


def test_graphed_step_detects_scale_drift_and_recalibrates():
    """the power-of-two loss scales are calibrated once and frozen into the captured graph.  Every link records its scaled
    max|gradient| on the device inside the step and Adam counts skipped non-finite elements; `GraphedStep.drift()` reads them, and with
    an eager step function given a drift triggers a fresh calibration + capture."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import medical_sam2_amd.synthetic as syn
    import medical_sam2_amd.training as T
    m, _ = _model()
    B, S, E = 2, 256, 16
    imgs = torch.stack([syn.normalize_image(syn.blob_image(i, S)[0]) for i in range(B)])
    pts, labels = torch.tensor([[[100.0, 120.0]], [[60.0, 200.0]]]), torch.ones(B, 1, dtype=torch.int32)
    memory, memory_pos = rnd(2 * E * E, B, 64, seed=150, scale=0.5), rnd(2 * E * E, B, 64, seed=151)
    target = (rnd(B, 4, S // 4, S // 4, seed=152) > 0.3).float()
    args = tuple(t.to(DEV) for t in (imgs, pts, labels, memory, memory_pos, target))
    with torch.no_grad():
        opts = [T.DecoderAdam(m.memory_attention, lr=1e-6), T.DecoderAdam(m.sam_mask_decoder, lr=1e-5), T.DecoderAdam(m.image_encoder, lr=1e-6)]
        step = lambda sync: T.train_step_2d(m, opts[0], opts[1], *args, sync=sync, opt_enc=opts[2])
        step(True)                                                # eager: calibrates
        gs = T.GraphedStep(lambda: step(False), opts, eager_fn=lambda: step(True), check_every=2)
        mon = opts[0].scale_monitor["_amax"]
        blocks = opts[2].calibrated_block_scales[None]["_amax"]
        assert "mem_scale" in mon and any(k.startswith("block") for k in blocks) and any(k.startswith("neck") for k in blocks)
        gs.replay()
        assert 2.0 ** -5 < float(mon["mem_scale"].item()) <= 2.0 ** -2        # calibration put the maximum at 2^-3; one step later it still is
        gs.replay()                                               # second replay: the periodic check runs, nothing drifted
        assert gs.recalibrations == 0 and gs.drift() == {}
        assert all(o.skipped_elements == 0 for o in opts)
        # simulate what a drifting gradient does to the monitor (a link 2^9 above its calibration point) -> check() recalibrates
        blocks["block3"].fill_(2.0 ** 6)
        d = gs.check()
        assert list(d) == ["block3"] and gs.recalibrations == 1
        loss = gs.replay()[0]
        assert torch.isfinite(loss).all() and gs.drift() == {}
        # a non-finite gradient element is skipped by Adam AND counted
        bad = {k: torch.full_like(v, float("nan")) for k, v in list(m.sam_mask_decoder.named_parameters())[:1]}
        opts[1].step(bad)
        assert opts[1].skipped_elements == next(iter(bad.values())).numel()
        assert gs.drift().get("skipped_elements") == next(iter(bad.values())).numel()
    assert all(torch.isfinite(v).all() for v in m.state_dict().values())


def test_train_step_2d_hiera_bplus():
    """BASELINE configs[4]'s model through the whole 2-D training iteration (train_2d.py:43-47 trains every parameter): Hiera-B+ -- 24 blocks,
    head dim 56 zero-padded to 64 in the attention forward AND backward, 14 x 14 position embedding -- runs, moves the three groups, stays
    finite, and a plain gradient step of every group lowers the loss by the predicted amount (its encoder gradients are pinned above against autograd)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import medical_sam2_amd.synthetic as syn
    import medical_sam2_amd.training as T
    m, _ = _model("hiera_b+")
    B, S, E = 2, 256, 16
    imgs = torch.stack([syn.normalize_image(syn.blob_image(i, S)[0]) for i in range(B)])
    pts, labels = torch.tensor([[[100.0, 120.0]], [[60.0, 200.0]]]), torch.ones(B, 1, dtype=torch.int32)
    memory, memory_pos = rnd(2 * E * E, B, 64, seed=150, scale=0.5), rnd(2 * E * E, B, 64, seed=151)
    target = (rnd(B, 4, S // 4, S // 4, seed=152) > 0.3).float()
    args = tuple(t.to(DEV) for t in (imgs, pts, labels, memory, memory_pos, target))
    with torch.no_grad():
        zero = [T.DecoderAdam(mod, lr=0.0) for mod in (m.memory_attention, m.sam_mask_decoder, m.image_encoder)]
        got: dict = {}
        loss0, _ = T.train_step_2d(m, zero[0], zero[1], *args, opt_enc=zero[2], grads_out=got)
        # sign and scale of every group's gradient: a plain step p <- p - eta g sized for a first-order drop of 2 % lowers the loss by
        # about that (curvature takes some of it)
        drops = gradient_step_drops(lambda: T.train_step_2d(m, zero[0], zero[1], *args, opt_enc=zero[2])[0],
                                    {"decoder": m.sam_mask_decoder, "memory_attention": m.memory_attention, "image_encoder": m.image_encoder}, got, loss0)
        print("relative loss drop of a plain gradient step per group (predicted 0.02):", drops)
        assert all(0.005 < d < 0.03 for d in drops.values()), drops
        before = {k: v.detach().clone() for k, v in m.state_dict().items()}
        opts = [T.DecoderAdam(m.memory_attention, lr=1e-5), T.DecoderAdam(m.sam_mask_decoder, lr=1e-4), T.DecoderAdam(m.image_encoder, lr=1e-5)]
        losses = [T.train_step_2d(m, opts[0], opts[1], *args, opt_enc=opts[2])[0] for _ in range(4)]
        moved = {k.split(".")[0] for k, v in m.state_dict().items() if not torch.equal(v, before[k])}
    assert moved == {"memory_attention", "sam_mask_decoder", "image_encoder"}, moved
    assert all(x == x and abs(x) < 1e6 for x in losses), losses
    assert all(torch.isfinite(v).all() for v in m.state_dict().values())
    assert all(o.skipped_elements == 0 for o in opts)
