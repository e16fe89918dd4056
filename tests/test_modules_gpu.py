"""Public nn.Module forwards of the mirror (medical_sam2_amd.modeling) against the oracle, module by module.

The end-to-end tests drive the fused `run()` paths; a reference user may also call a sub-module directly (e.g.
`net.image_encoder.trunk(x)`, `net.memory_encoder(pix, masks)`), so every public `forward` gets its own parity check here with the
same arguments and layouts as the reference class it mirrors (file:line in each test).
"""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import sam2_oracle as O  # noqa: E402

import medical_sam2_amd.weights as wts  # noqa: E402
from helpers import rel_err  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda"


# TIGHTER parity through operand-rounding emulation.  Every test of this file feeds the HIP module and the oracle the
# SAME inputs.  The tests marked `tight` run twice: against the plain fp32 oracle at the loose stated tolerance of the operand type (3e-3
# fp16, 2e-2 bf16 = 6 / 5 unit round-offs u) and against the oracle with the library's operand rounding emulated (`O.operand_rounding`:
# both operands of every matrix product rounded to fp16 / bf16, fp32 accumulation; the memory cross-attention in the folded form the
# kernel evaluates) at TIGHT_U = 2 u -- ONE constant for both types: 9.8e-4 for fp16, 7.8e-3 for bf16; measured 0.3-1.2 u.  A kernel
# error of one per cent cannot hide under it, which the loose bf16 bar could not promise.
# Why not tighter: rounding noise decorrelates.  Two evaluations whose inputs differ by d round a fraction d / u of their operands the
# other way, each by one u, so the difference after a rounding stage is sqrt(d u): 1e-7 -> 2e-5 -> 3e-4 -> 1e-3 in bf16.  Three stages
# into a block the HIP path's and the emulation's rounding errors are nearly independent; only single kernels compare exactly
# (tests/test_kernels_gpu.py: bit-exact integer-valued GEMMs, attention against fp64), and whole chains compare at the plain
# rounding level again (tests/test_operand_rounding_gpu.py, bars in units of u).
TIGHT_U = 2.0
MEASURED = {}


@pytest.fixture(scope="module")
def _net():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import medical_sam2_amd.build_sam as bs
    m = bs.build_sam2("sam2_hiera_s", device="cpu", hydra_overrides_extra=["++model.image_size=256"])
    sd = wts.init_weights("hiera_s", 0)
    m.load_state_dict(sd, strict=True)
    return m.to(DEV).eval(), sd, O.model_config("hiera_s", 256)


@pytest.fixture(params=["fp32_oracle", "operand_rounding_oracle"])
def net(request, _net):
    import json
    import medical_sam2_amd.ops as ops
    m, sd, cfg = _net
    tight = request.param == "operand_rounding_oracle"
    if tight and request.node.get_closest_marker("tight") is None:
        pytest.skip("this test's reference is not written against the operand-rounding emulation")
    seen = MEASURED.setdefault(f"{request.node.name}", [])
    global rel_err
    real = rel_err

    def recording(a, b):
        e = real(a, b)
        seen.append(e)
        return e
    rel_err = recording
    try:
        if tight:
            with O.operand_rounding(ops.OP16):
                yield m, sd, cfg, TIGHT_U * (2.0 ** -11 if ops.OP16 == torch.float16 else 2.0 ** -8)
        else:
            yield m, sd, cfg, (3e-3 if ops.OP16 == torch.float16 else 2e-2)
    finally:
        rel_err = real
        os.makedirs("gpurun_out", exist_ok=True)
        with open(f"gpurun_out/modules_report_{str(ops.OP16).split('.')[-1]}.json", "w") as f:
            json.dump({k: [max(v), len(v)] for k, v in MEASURED.items() if v}, f, indent=1)


def rnd(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


@pytest.mark.tight
def test_patch_embed_and_block_forward(net):
    """PatchEmbed.forward (backbones/utils.py:68-95, NHWC out) and MultiScaleBlock.forward (hieradet.py:136-168, NHWC in/out)."""
    m, P, cfg, tol = net
    img = rnd(2, 3, 64, 64, seed=1)
    with torch.no_grad():
        y = m.image_encoder.trunk.patch_embed(img.to(DEV))
    ref = torch.nn.functional.conv2d(O._rop(img), O._rop(P["image_encoder.trunk.patch_embed.proj.weight"]),
                                     P["image_encoder.trunk.patch_embed.proj.bias"], stride=4, padding=3).permute(0, 2, 3, 1)
    assert y.shape == ref.shape and rel_err(y.cpu(), ref) < tol
    specs = O.hiera_block_specs(cfg["trunk"])
    sides = {0: 32, 1: 32, 2: 16, 13: 8, 14: 4}   # a windowed block, the q-pool blocks of stages 2-4 and a global one
    for i, side in sides.items():
        s = specs[i]
        x = rnd(2, side, side, s["dim"], seed=10 + i)
        with torch.no_grad():
            y = m.image_encoder.trunk.blocks[i](x.to(DEV))
        ref = O.multiscale_block(P, f"image_encoder.trunk.blocks.{i}", x, s)
        assert y.shape == ref.shape, (i, y.shape, ref.shape)
        assert rel_err(y.cpu(), ref) < tol, (i, rel_err(y.cpu(), ref))


def test_trunk_neck_encoder_forward(net):
    """Hiera.forward (hieradet.py:280-295), FpnNeck.forward (image_encoder.py:99-134), ImageEncoder.forward (image_encoder.py:26-41)."""
    m, P, cfg, tol = net
    img = rnd(1, 3, 256, 256, seed=3)
    with torch.no_grad():
        xs = m.image_encoder.trunk(img.to(DEV))
        feats, pos = m.image_encoder.neck(xs)
        out = m.image_encoder(img.to(DEV))
    ref_xs = O.hiera_trunk(P, cfg, img)
    assert len(xs) == len(ref_xs) == 4
    for a, b in zip(xs, ref_xs):
        assert a.shape == b.shape and rel_err(a.cpu(), b) < tol
    ref_f, ref_p = O.fpn_neck(P, cfg, ref_xs)
    for a, b in zip(feats, ref_f):
        assert a.shape == b.shape and rel_err(a.cpu(), b) < tol
    for a, b in zip(pos, ref_p):
        assert a.shape == b.shape and rel_err(a.cpu().float(), b) < 1e-4
    ref = O.forward_image(P, cfg, img)
    assert set(out.keys()) == {"vision_features", "vision_pos_enc", "backbone_fpn"}
    assert rel_err(out["vision_features"].cpu(), ref["vision_features"]) < tol
    assert len(out["backbone_fpn"]) == len(ref["backbone_fpn"]) == 3      # scalp = 1


@pytest.mark.tight
def test_attention_modules_forward(net):
    """Attention.forward (transformer.py:239-263), RoPEAttention.forward (288-331), MemoryAttentionLayer.forward
    (memory_attention.py:81-99), TwoWayAttentionBlock / TwoWayTransformer.forward (transformer.py:74-118, 165-196)."""
    m, P, cfg, tol = net
    B, L = 2, 256
    layer = m.memory_attention.layers[0]
    pre = "memory_attention.layers.0"
    x = rnd(B, L, 256, seed=5)
    mem = rnd(B, 2 * L + 4, 64, seed=6)
    with torch.no_grad():
        y = layer.self_attn(x.to(DEV), x.to(DEV), x.to(DEV))
        ref = O.rope_attention(P, pre + ".self_attn", x, x, x, 1, cfg["rope_theta"])
        assert rel_err(y.cpu().float(), ref) < tol
        y = layer.cross_attn_image(x.to(DEV), mem.to(DEV), mem.to(DEV), num_k_exclude_rope=4)
        ref = O.rope_attention(P, pre + ".cross_attn_image", x, mem, mem, 1, cfg["rope_theta"], num_k_exclude_rope=4)
        assert rel_err(y.cpu().float(), ref) < tol
        # one full layer, batch-first like the reference's layer.forward
        pos = rnd(B, 2 * L + 4, 64, seed=7)
        pos[:, -4:] = 0
        y = layer(tgt=x.to(DEV), memory=mem.to(DEV), pos=pos.to(DEV), query_pos=None, num_k_exclude_rope=4)
        t = O.lnorm(P, pre + ".norm1", x, 1e-5)
        r = x + O.rope_attention(P, pre + ".self_attn", t, t, t, 1, cfg["rope_theta"])
        t = O.lnorm(P, pre + ".norm2", r, 1e-5)
        r = r + O.rope_attention(P, pre + ".cross_attn_image", t, mem + pos, mem, 1, cfg["rope_theta"], num_k_exclude_rope=4)
        t = O.lnorm(P, pre + ".norm3", r, 1e-5)
        r = r + O.lin(P, pre + ".linear2", torch.relu(O.lin(P, pre + ".linear1", t)))
        assert rel_err(y.cpu().float(), r) < tol
        # SAM decoder attention (8 heads) and the two-way transformer
        tw = m.sam_mask_decoder.transformer
        q, k = rnd(B, 7, 256, seed=8), rnd(B, L, 256, seed=9)
        y = tw.final_attn_token_to_image(q.to(DEV), k.to(DEV), k.to(DEV))
        ref = O.mh_attention(P, "sam_mask_decoder.transformer.final_attn_token_to_image", q, k, k)
        assert rel_err(y.cpu().float(), ref) < tol
        src, pe = rnd(B, 256, 16, 16, seed=11), rnd(1, 256, 16, 16, seed=12)
        qq, kk = tw(src.to(DEV), pe.to(DEV), q.to(DEV))
        rq, rk = O.two_way_transformer(P, "sam_mask_decoder.transformer", src, pe.expand(B, -1, -1, -1), q)
        assert rel_err(qq.cpu().float(), rq) < tol and rel_err(kk.cpu().float(), rk) < tol


def test_memory_encoder_modules_forward(net):
    """LayerNorm2d (sam2_utils.py:134-147), MLP (sam2_utils.py:104-129), MaskDownSampler.forward (memory_encoder.py:57-58),
    CXBlock.forward (100-117), Fuser.forward (133-139), MemoryEncoder.forward (158-181)."""
    m, P, cfg, tol = net
    me = m.memory_encoder
    pix = rnd(2, 256, 16, 16, seed=13)
    mask = rnd(2, 1, 256, 256, seed=14, scale=4.0)
    with torch.no_grad():
        out = me(pix.to(DEV), mask.to(DEV), skip_mask_sigmoid=True)
        ref, ref_pos = O.memory_encoder(P, cfg, pix, mask)
        assert rel_err(out["vision_features"].cpu().float(), ref) < tol
        assert rel_err(out["vision_pos_enc"][0].cpu().float(), ref_pos) < 1e-4
        out = me(pix.to(DEV), mask.to(DEV))           # sigmoid applied inside, like the reference default
        ref, _ = O.memory_encoder(P, cfg, pix, torch.sigmoid(mask))
        assert rel_err(out["vision_features"].cpu().float(), ref) < tol
        # mask down-sampler alone
        y = me.mask_downsampler(mask.to(DEV))
        r = mask
        ds = "memory_encoder.mask_downsampler.encoder"
        for j in range(4):
            r = torch.nn.functional.conv2d(r, P[f"{ds}.{3 * j}.weight"], P[f"{ds}.{3 * j}.bias"], stride=2, padding=1)
            r = O.gelu(O.lnorm2d(P, f"{ds}.{3 * j + 1}", r))
        r = torch.nn.functional.conv2d(r, P[f"{ds}.12.weight"], P[f"{ds}.12.bias"])
        assert y.shape == r.shape and rel_err(y.cpu().float(), r) < tol
        # one ConvNeXt block and the fuser
        x = rnd(2, 256, 16, 16, seed=15)
        y = me.fuser.layers[0](x.to(DEV))
        lp = "memory_encoder.fuser.layers.0"
        h = torch.nn.functional.conv2d(x, P[lp + ".dwconv.weight"], P[lp + ".dwconv.bias"], padding=3, groups=256)
        h = O.lnorm2d(P, lp + ".norm", h).permute(0, 2, 3, 1)
        h = O.lin(P, lp + ".pwconv2", O.gelu(O.lin(P, lp + ".pwconv1", h))) * P[lp + ".gamma"]
        r = x + h.permute(0, 3, 1, 2)
        assert rel_err(y.cpu().float(), r) < tol
        y2 = me.fuser(x.to(DEV))
        assert y2.shape == x.shape and torch.isfinite(y2).all()
        # LayerNorm2d and the 3-layer MLP heads
        ln = me.mask_downsampler.encoder[1]
        z = rnd(2, 4, 32, 32, seed=16)
        assert rel_err(ln(z.to(DEV)).cpu().float(), O.lnorm2d(P, ds + ".1", z)) < 1e-4
        t = rnd(2, 4, 256, seed=17)
        y = m.sam_mask_decoder.iou_prediction_head(t.to(DEV))
        r = O.mlp(P, "sam_mask_decoder.iou_prediction_head", t, 3, torch.relu, sigmoid_out=cfg.get("iou_prediction_use_sigmoid", True))
        assert rel_err(y.cpu().float(), r) < tol


def test_non_overlapping_constraints(net):
    """SAM2Base._apply_non_overlapping_constraints (sam2_base.py:812-830), bit-exact."""
    m, _, _, _ = net
    x = rnd(5, 1, 64, 64, seed=20, scale=8.0)
    x[1, 0, :8] = x[0, 0, :8]          # ties: the first maximum keeps its score (torch.argmax)
    idx = torch.argmax(x, dim=0, keepdim=True)
    keep = idx == torch.arange(5)[:, None, None, None]
    ref = torch.where(keep, x, torch.clamp(x, max=-10.0))
    y = m._apply_non_overlapping_constraints(x.to(DEV))
    assert torch.equal(y.cpu(), ref)
    one = x[:1].to(DEV)
    assert torch.equal(m._apply_non_overlapping_constraints(one), one)


def test_mask_decoder_cell_nums_prompt_repetition(net):
    """MaskDecoder.forward with `cell_nums` (mask_decoder.py:215-231; the 2-D loop's call, func_2d/function.py:159-168): one image
    embedding, several prompt sets -- repeat_interleave of the embedding, broadcast position / dense embeddings and high-res skips --
    against the oracle on the explicitly repeated inputs; plus the reference's failure modes."""
    m, sd, cfg, tol = net
    E, n = 16, 3
    emb, pe = rnd(1, 256, E, E, seed=601), rnd(1, 256, E, E, seed=602)
    sparse, dense = rnd(n, 2, 256, seed=603), rnd(1, 256, 1, 1, seed=604, scale=0.3).expand(1, 256, E, E)
    hr = [rnd(1, 32, 4 * E, 4 * E, seed=605), rnd(1, 64, 2 * E, 2 * E, seed=606)]
    ref, ref_iou, _, ref_obj = O.mask_decoder_predict(sd, emb.expand(n, -1, -1, -1), pe, sparse, dense.expand(n, -1, -1, -1), [h.expand(n, -1, -1, -1) for h in hr])
    d = lambda t: t.to(DEV)
    dec = m.sam_mask_decoder
    with torch.no_grad():
        masks, iou, toks, obj = dec(image_embeddings=d(emb), image_pe=d(pe), sparse_prompt_embeddings=d(sparse), dense_prompt_embeddings=d(dense),
                                    multimask_output=True, repeat_image=False, cell_nums=torch.tensor([n]), high_res_features=[d(h) for h in hr])
        assert masks.shape == (n, 3, 4 * E, 4 * E) and iou.shape == (n, 3) and toks.shape == (n, 3, 256) and obj.shape == (n, 1)
        assert rel_err(masks.cpu(), ref[:, 1:]) < 2 * tol and rel_err(iou.cpu(), ref_iou[:, 1:]) < 2 * tol and rel_err(obj.cpu(), ref_obj) < 2 * tol
        # two images, 2 + 1 prompt sets: the embedding is repeated per image; the high-res skips must then come per prompt set
        emb2 = torch.cat([emb, rnd(1, 256, E, E, seed=607)])
        hr2 = [torch.cat([h, h * 0.5])[[0, 0, 1]] for h in hr]
        ref2, _, _, _ = O.mask_decoder_predict(sd, emb2[[0, 0, 1]], pe, sparse, dense.expand(n, -1, -1, -1), hr2)
        masks2, _, _, _ = dec(image_embeddings=d(emb2), image_pe=d(pe), sparse_prompt_embeddings=d(sparse), dense_prompt_embeddings=d(dense),
                              multimask_output=True, repeat_image=False, cell_nums=torch.tensor([2, 1]), high_res_features=[d(h) for h in hr2])
        assert rel_err(masks2.cpu(), ref2[:, 1:]) < 2 * tol
        with pytest.raises(RuntimeError):     # un-repeated skips of two images cannot broadcast over three prompt sets (the reference raises too)
            dec(image_embeddings=d(emb2), image_pe=d(pe), sparse_prompt_embeddings=d(sparse), dense_prompt_embeddings=d(dense),
                multimask_output=True, repeat_image=False, cell_nums=torch.tensor([2, 1]), high_res_features=[d(torch.cat([h, h])) for h in hr])
        with pytest.raises(RuntimeError):
            dec(image_embeddings=d(emb2), image_pe=d(pe), sparse_prompt_embeddings=d(sparse), dense_prompt_embeddings=d(dense),
                multimask_output=True, repeat_image=False, cell_nums=torch.tensor([1, 1]), high_res_features=[d(h) for h in hr2])
