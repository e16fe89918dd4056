"""hipGraph replay of the 3-D propagation's per-slice forward (graphs.GraphedPropagation) and the device-side key count of the
memory cross-attention it rests on (msam2_attention_kv64_dyn_fwd)."""
import math
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import medical_sam2_amd.synthetic as syn  # noqa: E402
import medical_sam2_amd.weights as wts  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _qkv(B, Lq, cap, seed):
    import medical_sam2_amd.ops as ops
    g = torch.Generator().manual_seed(seed)
    q = (torch.randn(B, 1, Lq, 256, generator=g) * 0.5).to(DEV).to(ops.OP16)
    k = (torch.randn(B, 1, cap, 256, generator=g) * 0.5).to(DEV).to(ops.OP16)
    v = torch.randn(B, 1, cap, 64, generator=g).to(DEV).to(ops.OP16)
    return q, k, v


@pytest.mark.parametrize("cap,n,splits", [(4096 + 64, 4096 + 12, 1), (4096 + 64, 4096 + 64, 4), (3 * 4096 + 128, 3 * 4096 + 36, 4),
                                          (3 * 4096 + 128, 3 * 4096 + 100, 8), (8192 + 64, 8192 + 4, 2)])
def test_device_key_count_matches_host_key_count(cap, n, splits):
    """attention over a bank padded to `cap` keys with the valid count n on the device == the host-count call on the first n keys,
    bit for bit (same split boundaries: they are derived from the count inside the kernel), for ragged tails and full banks; the
    padded rows hold large garbage that must never leak into the result."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import medical_sam2_amd.ops as ops
    q, k, v = _qkv(2, 1024, cap, n)
    k[:, :, n:] = 30000.0 if ops.OP16 == torch.float16 else 1e30
    v[:, :, n:] = -30000.0 if ops.OP16 == torch.float16 else -1e30
    cnt = torch.tensor([n], dtype=torch.int32, device=DEV)
    eff = ops.attention_effective_splits(cap, splits)
    assert ops.attention_effective_splits(n, splits) == eff
    ref = ops.attention_kv64(q, k[:, :, :n], v[:, :, :n], splits=splits)
    got = ops.attention_kv64(q, k, v, splits=splits, key_count=cnt)
    assert torch.isfinite(got.float()).all()
    assert torch.equal(got, ref)
    # the count is read at run time: same launch, other fill level
    cnt.fill_(n - 37)
    got2 = ops.attention_kv64(q, k, v, splits=splits, key_count=cnt)
    assert torch.equal(got2, ops.attention_kv64(q, k[:, :, :n - 37], v[:, :, :n - 37], splits=splits))


def test_device_key_count_empty_trailing_split():
    """a fill level far below the capacity leaves trailing splits without keys: they must report (max -inf, sum 0) and drop out of the
    merge -- result equal to the fp32 softmax over the valid keys"""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import medical_sam2_amd.ops as ops
    cap, n = 1024, 100
    q, k, v = _qkv(1, 256, cap, 5)
    cnt = torch.tensor([n], dtype=torch.int32, device=DEV)
    got = ops.attention_kv64(q, k, v, splits=4, key_count=cnt).float()
    s = torch.einsum("bhqd,bhkd->bhqk", q.float(), k[:, :, :n].float()) / math.sqrt(256)
    ref = torch.softmax(s, -1) @ v[:, :, :n].float()
    assert torch.isfinite(got).all()
    tol = 4e-3 if ops.OP16 == torch.float16 else 3e-2
    assert (got - ref).abs().max() < tol, float((got - ref).abs().max())


def _volume_case(T=14, n_obj=2, every=4, S=256):
    import medical_sam2_amd.build_sam as bs
    m = bs.build_sam2("sam2_hiera_t", device="cpu", hydra_overrides_extra=[f"++model.image_size={S}"])
    m.load_state_dict(wts.init_weights("hiera_t", 0), strict=True)
    m = m.to(DEV).eval()
    volume, boxes = syn.blob_volume(3, n_slices=T, size=S, n_objects=n_obj)

    def box_at(t):
        return torch.tensor([[float(x) for x in (boxes[o][t] or (S * 0.3, S * 0.3, S * 0.6, S * 0.6))] for o in range(n_obj)], device=DEV)
    prompts = {t: {"boxes": box_at(t)} for t in range(0, T, every)}
    return m, volume.to(DEV), prompts


def test_graphed_propagation_is_the_padded_eager_path_bit_for_bit():
    """segment_volume(graphs=True): every propagated slice equals the eager run of the same padded-bank launches bit for bit (masks,
    pointers, new memories), graphs are actually replayed, and the padded path agrees with the un-padded eager path (same keys, same
    math; only the split boundaries of the cross-attention may move by a tile)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import medical_sam2_amd.volume as vol
    m, volume, prompts = _volume_case(T=22)
    T = volume.shape[0]
    st_g, st_p = {}, {}
    with torch.no_grad():
        plain, s_plain = vol.segment_volume(m, volume, prompts, return_state=True, padded_bank=False)
        padded, s_pad = vol.segment_volume(m, volume, prompts, return_state=True, padded_bank=True, stats=st_p)
        graphed, s_gr = vol.segment_volume(m, volume, prompts, return_state=True, graphs=True, stats=st_g)
    assert st_p["captures"] == 0 and st_p["replays"] == 0
    import medical_sam2_amd.ops as ops
    tol = 0.15 if ops.OP16 == torch.float16 else 1.0
    n_prop = T - len(prompts)
    # prompts every 4th slice: three steady-state buckets (slice index mod 4), each run eagerly once, captured once, then replayed
    assert st_g["replays"] >= 4 and 1 <= st_g["captures"] <= st_g["buckets"] <= n_prop, st_g
    assert st_g["replays"] + st_g["eager_steps"] == n_prop
    for t in range(T):
        assert torch.equal(graphed[t], padded[t]), t
        # padded vs un-padded: the capacity can change the cross-attention's split count at these small banks (256 tokens per memory),
        # i.e. the order of the 16-bit P roundings -- the propagated-slice tolerance of the e2e tests applies (3 x 0.05 logits, fp16)
        assert (padded[t] > 0).eq(plain[t] > 0).float().mean() > 0.999, t
        assert (padded[t] - plain[t]).abs().max() < tol, (t, float((padded[t] - plain[t]).abs().max()))
    for t, o in s_gr["non_cond_frame_outputs"].items():
        p = s_pad["non_cond_frame_outputs"][t]
        assert torch.equal(o["obj_ptr"], p["obj_ptr"]) and torch.equal(o["maskmem_features"], p["maskmem_features"]), t


def test_graph_cache_reuse_and_weight_change():
    """graphs kept between volumes of the same shape are replayed from the first steady-state slice on; they bake in the kernel-ready
    weight copies, so writing a parameter drops them and the next volume runs (and is re-captured) with the new weights"""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import medical_sam2_amd.volume as vol
    m, volume, prompts = _volume_case(T=22, n_obj=1, every=4)
    cache, s1, s2, s3 = {}, {}, {}, {}
    with torch.no_grad():
        a = vol.segment_volume(m, volume, prompts, graphs=True, graph_cache=cache, stats=s1)
        a2 = vol.segment_volume(m, volume, prompts, graphs=True, graph_cache=cache, stats=s2)
        a3 = vol.segment_volume(m, volume, prompts, graphs=True, graph_cache=cache, stats=s3)
        # volume 1 captures the steady-state buckets, volume 2 the warm-up ones (second sight), volume 3 replays every slice
        assert s1["captures"] > 0 and s2["replays"] > s1["replays"] and s3["captures"] == 0 and s3["eager_steps"] == 0, (s1, s2, s3)
        assert all(torch.equal(a[t], a2[t]) and torch.equal(a[t], a3[t]) for t in a)
        m.sam_mask_decoder.output_hypernetworks_mlps[0].layers[0].weight.mul_(1.5)
        b = vol.segment_volume(m, volume, prompts, graphs=True, graph_cache=cache, stats=s3)
        c = vol.segment_volume(m, volume, prompts, padded_bank=False)
    assert s3["captures"] > 0
    t = max(a)
    assert not torch.equal(a[t], b[t])
    assert (b[t] - c[t]).abs().max() < 0.05


def test_video_predictor_with_hip_graphs():
    """SAM2VideoPredictor(use_hip_graphs=True): propagate_in_video replays graphs for the prompt-free frames (forward and reverse) and
    yields the masks of the eager predictor (same keys, same math; only the cross-attention's split boundaries may move by a tile)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import medical_sam2_amd.build_sam as bs
    S, T = 256, 20
    vol = torch.stack([syn.blob_image(20 + t, S)[0] for t in range(T)])

    def run(graphs):
        m = bs.build_sam2_video_predictor("sam2_hiera_t", device="cpu", hydra_overrides_extra=[f"++model.image_size={S}"])
        m.load_state_dict(wts.init_weights("hiera_t", 0), strict=True)
        m = m.to(DEV).eval()
        m.use_hip_graphs = graphs
        out = {}
        with torch.no_grad():
            st = m.val_init_state(vol)
            m.add_new_points(st, 5, 7, [[120.0, 130.0]], [1])
            m.add_new_bbox(st, 5, 9, [60.0, 70.0, 150.0, 160.0])
            for f, ids, masks in m.propagate_in_video(st):
                out[("f", f)] = masks.float().clone()
            for f, ids, masks in m.propagate_in_video(st, start_frame_idx=5, reverse=True):
                out[("r", f)] = masks.float().clone()
        return out, st.get("graphed_propagation")

    eager, none = run(False)
    graphed, props = run(True)
    assert none is None and props
    assert sum(p.replays for p in props.values()) >= 4
    assert sorted(eager) == sorted(graphed) and ("r", 0) in eager and ("f", T - 1) in eager
    for k in eager:
        a, b = eager[k], graphed[k]
        fin = a.abs() < 1000
        assert torch.equal(fin, b.abs() < 1000), k
        assert ((a > 0) == (b > 0)).float().mean() > 0.9995, k
        assert (a - b)[fin].abs().max() < 0.1, (k, float((a - b)[fin].abs().max()))
