#!/usr/bin/env python3
"""Benchmark of the MI355X hot path: BASELINE.json metric "slices/sec @1024^2 (Hiera-S ...)".

Workload at N=1 (BASELINE.json configs[1], the 2D `train_2d.py` SAM2 sub-sequence, func_2d/function.py:70-191, forward only):
  4 synthetic 1024x1024 slices, 16-bit MFMA operands (bf16 as configs[1] states, `--dtype f16` for the fp16 build; fp32 accumulate / residual streams) ->  forward_image -> _prepare_backbone_features -> memory_attention against a
  pre-filled 16-entry memory bank (4 sampled memories per slice, fixed indices instead of torch.multinomial) -> prompt encoder
  (one click per slice) -> mask decoder (+ high-res features) -> bilinear x4 -> _encode_new_memory.
One "step" = that sequence for the batch of 4 slices; value = slices / second with inputs resident in HBM.
The step is captured into one hipGraph (torch.cuda.graph: PyTorch provides the stream + memory pool) and replayed.

N>1, one rank per GPU.  Either the driver launches the ranks (torch.distributed.run sets RANK / WORLD_SIZE) or `python bench.py --gpus N`
does it itself: with WORLD_SIZE unset the parent process -- before it makes any GPU call -- starts `python -m torch.distributed.run
--nproc-per-node N ... bench.py <same flags>` as a child and only relays its exit code; rank 0 of the child prints the line.  (On a box
with fewer than N GPUs the ranks share device 0 over gloo: a rehearsal of the launch path, flagged as such on the line.)
  --mode 2d (default): every rank runs the same per-GPU workload on its own slices and its own memory bank -- the 2D path shares
      nothing between replicas -- so there is no data-path collective; RCCL is used for the barrier and the
      max-over-ranks time only.  scaling = "weak".
  --mode volume: BASELINE.json configs[3] -- ONE 512-slice 1024^2 volume, bbox prompt on every 2nd slice, one object, through
      `volume.segment_volume` on all N ranks (slice-sharded image encoder + conditioning pass, one RCCL exchange of memories / features,
      the propagation chain with the memory cross-attention's key range split over the ranks).  A step = one whole volume;
      scaling = "strong"; the per-phase seconds of the last volume are on the line.
The 2d line also carries that volume pass as the side object `volume_3d` (after the timed region; guarded by a watchdog: if the exchange
hangs, rank 0 still prints the line with the error recorded in it).

Extra objects on the JSON line: "roofline" (dominant kernel = the memory cross-attention, attn_kv64x2_kernel, MFMA-bound; timed with
HIP events on the launch stream), "roofline_gemm" (the GEMM family of the step: every distinct shape the step launches, timed the same
way; flop-weighted fraction of the MFMA peak), "roofline_hiera_attention" and "cpu_baseline" (the CPU oracle on a bounded sample,
rank 0, N=1 only).

`cpu_baseline.parity_slice0`: the oracle sample IS slice 0 of the timed step (same image, click and memory bank), so its mask is compared
with the HIP path's -- mask IoU and max / mean |delta logit| at the full 1024^2 size.
Side figure (never part of `value`, `--no-train` skips it): `train_iteration` = one whole training iteration of the same configuration
(frozen encoders forward, memory attention + mask decoder forward / backward / Adam, memory encoding) replayed as a hipGraph after the
timed region, on a deep copy of the model; a failure there is reported inside that object and does not affect the line.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0


def build_model(device):
    import medical_sam2_amd.build_sam as bs
    import medical_sam2_amd.weights as wts
    m = bs.build_sam2("sam2_hiera_s", device="cpu", hydra_overrides_extra=["++model.image_size=1024"])
    m.load_state_dict(wts.init_weights("hiera_s", 0), strict=True)
    return m.to(device).eval()


def make_inputs(device, batch, rank):
    import medical_sam2_amd.synthetic as syn
    imgs, pts, labels = syn.image_batch([rank * 1000 + i for i in range(batch)], 1024)
    g = torch.Generator().manual_seed(1234 + rank)
    # 16-entry memory bank (maskmem_features [1,64,64,64] + its position encoding per entry); synthetic values of the scale the
    # memory encoder produces
    bank_feats = torch.randn(16, 64, 64, 64, generator=g) * 0.5
    sampled = torch.tensor([[(3 * b + 5 * j) % 16 for j in range(batch)] for b in range(batch)])  # fixed "multinomial" draw
    return imgs.to(device), pts.to(device), labels.to(device), bank_feats.to(device), sampled


def step_2d(m, imgs, pts, labels, memory, memory_pos):
    """func_2d/function.py:70-191 restated against the drop-in modules (same calls, same tensor conventions)."""
    B = imgs.shape[0]
    backbone_out = m.forward_image(imgs)
    _, vision_feats, vision_pos_embeds, feat_sizes = m._prepare_backbone_features(backbone_out)
    vision_feats[-1] = m.memory_attention(curr=[vision_feats[-1]], curr_pos=[vision_pos_embeds[-1]], memory=memory,
                                          memory_pos=memory_pos, num_obj_ptr_tokens=0)
    feats = [f.permute(1, 2, 0).view(B, -1, *s) for f, s in zip(vision_feats[::-1], feat_sizes[::-1])][::-1]
    image_embed, high_res_feats = feats[-1], feats[:-1]
    se, de = m.sam_prompt_encoder(points=(pts, labels), boxes=None, masks=None, batch_size=B)
    low_res, iou, tokens, obj = m.sam_mask_decoder(image_embeddings=image_embed, image_pe=m.sam_prompt_encoder.get_dense_pe(),
                                                   sparse_prompt_embeddings=se, dense_prompt_embeddings=de, multimask_output=False,
                                                   repeat_image=False, cell_nums=None, high_res_features=high_res_feats)
    import medical_sam2_amd.ops as ops
    high_res = ops.bilinear_upsample(low_res.contiguous(), 1024, 1024)
    maskmem_features, maskmem_pos_enc = m._encode_new_memory(current_vision_feats=vision_feats, feat_sizes=feat_sizes,
                                                             pred_masks_high_res=high_res, is_mask_from_pts=True)
    return low_res, iou, maskmem_features


def assemble_memory(m, bank_feats, sampled):
    """func_2d/function.py:92-116 with fixed indices: memory [B*HW, B, 64] (+ position encoding)."""
    B, S = sampled.shape                                       # images, sampled memories per image
    pos = m.memory_encoder.position_encoding(bank_feats[:1])  # [1,64,64,64] view, same for every entry
    mem = bank_feats[sampled.to(bank_feats.device)]            # [B(img), S(samples), 64, 64, 64]
    memory = mem.flatten(3).permute(1, 3, 0, 2).reshape(-1, B, 64).contiguous()   # (sample, hw) x img x C
    memory_pos = pos.flatten(2).permute(2, 0, 1).repeat(S, B, 1).contiguous()
    return memory, memory_pos


def time_dominant_kernel(device, batch):
    """HIP-event timing (on the launch stream) of the dominant kernel -- the memory cross-attention (RoPEAttention with kv_in_dim 64,
    transformer.py:288-331) as the model runs it: attn_kv64x2_kernel, 256-wide rotated q / k rows, the value product contracted in the
    64-channel memory space (v_proj folded into out_proj) -- at the cross-attention shape of this workload: Lq = 4096,
    Lk = batch*4096 per slice.  The split pass is launched alone (negative split count, partials stay in a caller-owned workspace)
    so the figure is that kernel's own average duration, comparable with rocprofv3's kernel trace; the fwd+merge pair is timed too.
    `flops` is the ALGORITHMIC figure of SURVEY.md 8(d) (4 Lq Lk 256 per object: the reference's 256-wide formulation);
    `executed` = 2 Lq Lk (256 + 64) is what the kernel's MFMAs actually do after the fold.
    Returns (avg seconds per launch, algorithmic flops, executed flops, algorithmic bytes, splits, avg seconds of the fwd+merge pair)."""
    import medical_sam2_amd.ops as ops
    from medical_sam2_amd._lib import lib, check
    from medical_sam2_amd.modeling.common import attn_splits
    B, Lq, Lk, D = batch, 4096, batch * 4096, 256
    g = torch.Generator().manual_seed(5)
    q = (torch.randn(B, 1, Lq, D, generator=g)).to(ops.OP16).to(device)
    k = (torch.randn(B, 1, Lk, D, generator=g)).to(ops.OP16).to(device)
    v = (torch.randn(B, 1, Lk, 64, generator=g)).to(ops.OP16).to(device)
    splits = attn_splits(B, 1, Lq, Lk)
    ws = ops.attention_workspace(B, 1, Lq, 64, splits, device)
    stream = torch.cuda.current_stream().cuda_stream
    e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
    check(lib().msam2_event_create(ctypes.byref(e0)))
    check(lib().msam2_event_create(ctypes.byref(e1)))
    n = 20

    def timed(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        check(lib().msam2_event_record(e0, stream))
        for _ in range(n):
            fn()
        check(lib().msam2_event_record(e1, stream))
        ms = ctypes.c_float()
        check(lib().msam2_event_elapsed_ms(e0, e1, ctypes.byref(ms)))
        return ms.value / 1e3 / n

    defer = splits > 1
    run_kernel = lambda: ops.attention_kv64(q, k, v, splits=splits, workspace=ws, defer_merge=defer)
    run_pair = lambda: ops.attention_kv64(q, k, v, splits=splits, workspace=ws)
    # alternate the two measurements (the chip's clock moves with load) and average
    tp1, tk1, tp2, tk2 = timed(run_pair), timed(run_kernel), timed(run_pair), timed(run_kernel)
    t_kernel, t_pair = 0.5 * (tk1 + tk2), 0.5 * (tp1 + tp2)
    lib().msam2_event_destroy(e0)
    lib().msam2_event_destroy(e1)
    flops = 4.0 * B * Lq * Lk * D
    executed = 2.0 * B * Lq * Lk * (D + 64)
    bytes_ = 2.0 * B * (Lq * D + Lk * D + Lk * 64 + Lq * 64)
    return t_kernel, flops, executed, bytes_, splits, t_pair


def time_hiera_attention(device, batch):
    """north_star's second figure: the Hiera attention kernels of this workload timed alone (HIP events on the launch stream, graph-free:
    20 back-to-back launches), at the two shapes that carry the trunk's attention time:
      * global blocks 7 / 10 / 13: attn_g96x2_kernel<1,8> (round 2: attn_glds_kernel<96,128,4,3>), B x 4 heads x 4096 x 4096 x 96 -- MFMA-bound (AI ~ 2000);
      * stage-3 windowed blocks (x7): attn_win_kernel<96>, 25 windows of 14x14 per image x 4 heads, 196 x 196 x 96 -- HBM-bound stand-alone
        (AI ~ 98): priced on the algorithmic bytes q, k, v, o once in 16 bits.
    Returns the `roofline_hiera_attention` object."""
    import medical_sam2_amd.ops as ops
    from medical_sam2_amd._lib import lib, check
    g = torch.Generator().manual_seed(6)
    stream = torch.cuda.current_stream().cuda_stream
    e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
    check(lib().msam2_event_create(ctypes.byref(e0)))
    check(lib().msam2_event_create(ctypes.byref(e1)))

    def timed(fn, n=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        check(lib().msam2_event_record(e0, stream))
        for _ in range(n):
            fn()
        check(lib().msam2_event_record(e1, stream))
        ms = ctypes.c_float()
        check(lib().msam2_event_elapsed_ms(e0, e1, ctypes.byref(ms)))
        return ms.value / 1e3 / n

    B, heads, L, D = batch, 4, 4096, 96
    qkv = (torch.randn(B, L, 3, heads, D, generator=g) * 0.5).to(ops.OP16).to(device)
    q, k, v = (qkv[:, :, i].permute(0, 2, 1, 3) for i in range(3))
    out = torch.empty(B, L, heads, D, dtype=ops.OP16, device=device).permute(0, 2, 1, 3)
    t_glob = timed(lambda: ops.attention(q, k, v, out=out))
    fl_glob = 4.0 * B * heads * L * L * D
    T = B * 64 * 64
    qkv2 = (torch.randn(T, 3 * heads * D, generator=g) * 0.5).to(ops.OP16).to(device)
    bias = torch.zeros(3 * heads * D, device=device)
    t_win = timed(lambda: ops.window_attention(qkv2, B, 64, 64, heads, 14, bias))
    by_win = 2.0 * heads * D * 4 * T
    fl_win = 4.0 * B * 25 * heads * 196 * 196 * D
    lib().msam2_event_destroy(e0)
    lib().msam2_event_destroy(e1)
    return {"global": {"kernel": f"attn_g96x2_kernel<1,8> at B={B} heads=4 Lq=Lk=4096 D=96", "bound": "mfma", "avg_launch_us": t_glob * 1e6,
                       "achieved": fl_glob / t_glob / 1e12, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": fl_glob / t_glob / 1e12 / MFMA_BF16_PEAK_TFLOPS,
                       "flops_per_launch": fl_glob},
            "windowed": {"kernel": f"attn_win_kernel<96> at B={B} 25 windows x 4 heads, 196 x 196 x 96 (stage-3 blocks)", "bound": "hbm",
                         "avg_launch_us": t_win * 1e6, "achieved": by_win / t_win / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": by_win / t_win / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": by_win,
                         "mfma_TFLOPs": fl_win / t_win / 1e12, "note": "eager launches include ~4 us of launch gap per kernel at this size"}}


def pmc_traffic():
    """HBM-side bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (profiles/, collected with
    `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` on `tools/one_attn.py 4 1 4096 16384 kv64 <splits>` at this exact shape): counter units of
    1 KiB, FETCH_SIZE doubled on gfx950 (MI355X_MICROARCH.md, HBM/rocprofv3 section).  None when the files are absent."""
    import csv
    here = os.path.dirname(os.path.abspath(__file__))
    tot = 0.0
    for name, mult in (("FETCH_SIZE", 2.0), ("WRITE_SIZE", 1.0)):
        path = next((q for q in (os.path.join(here, "profiles", f"{r}_attnkv64_pmc_{name}.csv") for r in ("r04", "r03", "r02")) if os.path.exists(q)), None)
        if path is None:
            return None
        vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(path))
                if r.get("Counter_Name") == name and "attn_kv64x2_kernel" in r.get("Kernel_Name", "")]
        if not vals:
            return None
        tot += mult * 1024.0 * sum(vals) / len(vals)
    return tot


def train_iteration(m, imgs, pts, labels, memory, memory_pos, device, full=True):
    """Side figure, never part of `value`: one whole training iteration of the same configuration (SURVEY.md 8(d) config 2 "time
    forward (+backward when available)") -- frozen image / prompt encoders, forward + backward + Adam of memory attention and mask
    decoder, memory encoding of the new prediction (`training.train_step_2d`) -- replayed as a hipGraph.  Reported as
    {"ms": per iteration, "slices_per_s": ..}; None with the reason if anything in it fails (the headline line must not depend on it)."""
    try:
        import copy
        import medical_sam2_amd.training as T
        mt = copy.deepcopy(m)                              # the optimiser must not touch the benchmarked weights
        B = imgs.shape[0]
        g = torch.Generator().manual_seed(3)
        target = (torch.randn(B, 4, 256, 256, generator=g) > 0.5).float().to(device)
        om, od = T.DecoderAdam(mt.memory_attention, lr=1e-6), T.DecoderAdam(mt.sam_mask_decoder, lr=1e-4)
        oe = T.DecoderAdam(mt.image_encoder, lr=1e-6, weight_decay=0.0) if full else None
        step = lambda sync: T.train_step_2d(mt, om, od, imgs, pts, labels, memory, memory_pos, target, sync=sync, opt_enc=oe)
        step(True)                                          # eager: calibrates the loss scale, packs weights
        graph = T.GraphedStep(lambda: step(False), [o for o in (om, od, oe) if o is not None])   # replays advance Adam's device-side step count
        graph.replay()
        torch.cuda.synchronize()
        n = 10
        t0 = time.perf_counter()
        for _ in range(n):
            graph.replay()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        what = ("train_step_2d, every group of net.parameters() the 2-D loop trains (train_2d.py:43-47): image encoder (Hiera trunk + FPN neck) + memory "
                "attention + mask decoder forward / backward / Adam, + memory encoding; eval-mode dropout; hipGraph replay") if full else \
               "train_step_2d: frozen encoders fwd + memory attention & mask decoder fwd/bwd + Adam + memory encoding, hipGraph replay"
        return {"ms": dt * 1e3, "slices_per_s": B / dt, "what": what}
    except Exception as e:  # noqa: BLE001 -- a side figure: report, do not fail the benchmark line
        return {"ms": None, "error": f"{type(e).__name__}: {e}"[:300]}


def host_cpu_share() -> int:
    """CPUs this process may actually use: the cgroup quota when there is one (the GPU box gives 16 of its 256 hardware threads;
    running the oracle on all 256 is 3x SLOWER than on 16: tests/probes/cpu_threads_probe.py), else the affinity mask."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(sample=None):
    """The CPU oracle (fp32 torch restatement, "port") on ONE slice of the same workload (1 memory of the 4 -> scaled
    honestly: the sample is one slice with its full 4-memory bank), 1 warm-up + 2 timed repetitions, on the host's CPU share.
    sample = (img [1,3,S,S], pts, labels, memory [Nk,1,64], memory_pos [Nk,1,64], hip_low_res [1,1,256,256]) on the CPU: slice 0 of the
    timed step with the HIP path's mask logits for it -- the oracle then runs on exactly these inputs and the returned object also
    carries the parity of the two masks (the second half of BASELINE.json's metric: mask IoU vs reference)."""
    from oracle import sam2_oracle as O
    torch.set_num_threads(host_cpu_share())
    import medical_sam2_amd.synthetic as syn
    import medical_sam2_amd.weights as wts
    torch.set_grad_enabled(False)
    P = wts.init_weights("hiera_s", 0)
    cfg = O.model_config("hiera_s", 1024)
    hip_low = None
    if sample is not None:
        img, pts, labels, memory, memory_pos, hip_low = sample
    else:
        img, pts, labels = syn.image_batch([0], 1024)
        g = torch.Generator().manual_seed(1234)
        memory = torch.randn(4 * 4096, 1, 64, generator=g) * 0.5
        memory_pos = O.sine_pos_2d(64, 64, 64).flatten(1).t()[:, None, :].repeat(4, 1, 1)
    last = {}

    def one():
        bo = O.forward_image(P, cfg, img)
        feats, pos, sizes = O.prepare_backbone_features(bo)
        top = O.memory_attention(P, cfg, feats[-1], memory, pos[-1], memory_pos, 0)
        emb = top.permute(1, 2, 0).reshape(1, 256, 64, 64)
        hr = [f.permute(1, 2, 0).reshape(1, -1, *s) for f, s in zip(feats[:-1], sizes[:-1])]
        sp, de = O.prompt_encoder(P, cfg, (pts, labels), None, None)
        masks, iou, toks, obj = O.mask_decoder(P, cfg, emb, O.dense_pe(P, 64, 64), sp, de, False, hr)
        high = torch.nn.functional.interpolate(masks, size=(1024, 1024), mode="bilinear", align_corners=False)
        last["masks"] = masks
        return O.encode_new_memory(P, cfg, top, (64, 64), high, True)

    one()
    reps = 2
    t0 = time.perf_counter()
    for _ in range(reps):
        one()
    dt = (time.perf_counter() - t0) / reps
    res = {"value": 1.0 / dt, "unit": "slices/s", "cores": torch.get_num_threads(), "kind": "port",
           "sample": f"1 slice (of the 4-slice step) through oracle/sam2_oracle.py fp32, its 4x4096-token memory bank, "
                     f"1 warm-up + {reps} timed reps, {dt:.2f} s per slice"}
    if hip_low is not None:
        a, b = hip_low.float() > 0, last["masks"] > 0
        union = (a | b).sum().item()
        res["parity_slice0"] = {"mask_iou_vs_oracle": (a & b).sum().item() / union if union else 1.0,
                                "max_abs_dlogit": (hip_low.float() - last["masks"]).abs().max().item(),
                                "mean_abs_dlogit": (hip_low.float() - last["masks"]).abs().mean().item(),
                                "foreground_pixels": int(b.sum().item())}
    res["_oracle_low"] = last["masks"]          # popped by the caller (parity of the bf16 side run)
    return res


# ---------------------------------------------------------------------------------------------------------------------
# the GEMM family of the step (half of the step, no roofline object so far)
GEMM_ENTRY_POINTS = ("gemm", "gemm_rope", "gemm_pool2x2", "gemm_qkv_pool2x2", "gemm_tokens")


def time_gemm_family(m, imgs, pts, labels, memory, memory_pos, device):
    """Every GEMM launch of one eager step (all five entry points of ops: plain, +RoPE store, pooled shortcut, pooled qkv, token rows),
    grouped by (entry point, M, N, K, output type, residual, activation); one representative call per group is re-launched 20 times on
    its own live operands -- as ONE hipGraph replay, so that host launch time does not enter -- and timed with HIP events on the launch stream.  `achieved` = the step's GEMM flops (2 M N K per launch)
    over the sum of count x average launch time: the flop-weighted rate of the family against the dense MFMA peak."""
    import medical_sam2_amd.ops as ops
    from medical_sam2_amd._lib import lib, check
    groups = {}
    real = {n: getattr(ops, n) for n in GEMM_ENTRY_POINTS}

    def spy(name):
        fn = real[name]

        def wrapped(a, w, *args, **kw):
            out = fn(a, w, *args, **kw)
            o = out[0] if isinstance(out, tuple) else out
            key = (name, int(a.shape[0]), int(w.shape[0]), int(a.shape[1]), str(o.dtype).replace("torch.", ""),
                   kw.get("residual") is not None, int(kw.get("act", 0)))
            g = groups.setdefault(key, {"n": 0, "call": (fn, (a, w) + args, kw)})
            g["n"] += 1
            return out
        return wrapped

    for n in GEMM_ENTRY_POINTS:
        setattr(ops, n, spy(n))
    try:
        step_2d(m, imgs, pts, labels, memory, memory_pos)
    finally:
        for n in GEMM_ENTRY_POINTS:
            setattr(ops, n, real[n])
    stream = torch.cuda.current_stream().cuda_stream
    e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
    check(lib().msam2_event_create(ctypes.byref(e0)))
    check(lib().msam2_event_create(ctypes.byref(e1)))
    rows, reps = [], 20
    for key, g in groups.items():
        fn, a, kw = g["call"]
        kw = {k: v for k, v in kw.items() if k != "out"}        # never re-write a live buffer of the model (out= aliases)
        for _ in range(3):
            fn(*a, **kw)
        torch.cuda.synchronize()
        # The `reps` launches are replayed from a hipGraph: launched one by one from Python a call costs ~18 us of host time, which
        # is what the table showed for every shape under 18 us in round 3 (30 of the 50 shapes; in the step's graph they take 5-8 us)
        run = None
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                fn(*a, **kw)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                for _ in range(reps):
                    fn(*a, **kw)
            run = gr.replay
        except Exception:  # noqa: BLE001 -- fall back to eager launches
            torch.cuda.synchronize()
            run = lambda: [fn(*a, **kw) for _ in range(reps)]
        run()
        torch.cuda.synchronize()
        stream = torch.cuda.current_stream().cuda_stream
        check(lib().msam2_event_record(e0, stream))
        run()
        check(lib().msam2_event_record(e1, stream))
        ms = ctypes.c_float()
        check(lib().msam2_event_elapsed_ms(e0, e1, ctypes.byref(ms)))
        t = ms.value / 1e3 / reps
        name, M, N, K = key[:4]
        rows.append({"entry": name, "M": M, "N": N, "K": K, "out": key[4], "residual": key[5], "act": key[6], "launches_per_step": g["n"],
                     "avg_launch_us": t * 1e6, "TFLOPs": 2.0 * M * N * K / t / 1e12, "step_us": g["n"] * t * 1e6,
                     "step_flops": 2.0 * M * N * K * g["n"]})
    lib().msam2_event_destroy(e0)
    lib().msam2_event_destroy(e1)
    groups.clear()
    tot_t = sum(r["step_us"] for r in rows) * 1e-6
    tot_f = sum(r["step_flops"] for r in rows)
    rows.sort(key=lambda r: -r["step_us"])
    for r in rows:
        r["share_of_gemm_time"] = r["step_us"] * 1e-6 / tot_t
    ach = tot_f / tot_t / 1e12
    if os.environ.get("MSAM2_BENCH_GEMM_TABLE"):          # the whole table (every distinct shape of the step) for the profiles/ directory
        with open(os.environ["MSAM2_BENCH_GEMM_TABLE"], "w") as f:
            json.dump(rows, f, indent=1)
    return {"bound": "mfma", "achieved": ach, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / MFMA_BF16_PEAK_TFLOPS,
            "traffic": None, "what": "all GEMM launches of one step (flop-weighted): sum of 2MNK over sum of launches x stand-alone average launch time",
            "gemm_ms_per_step": tot_t * 1e3, "flops_per_step": tot_f, "launches_per_step": sum(r["launches_per_step"] for r in rows),
            "distinct_shapes": len(rows), "top_shapes": rows[:8],
            "ladder": "profiles/r04_gemm_wstat_ladder.txt: the same launch with pieces removed (a third epilogue + output stores, a third fixed cost, ~9 us of matrix work)",
            "pmc": "profiles/r03_gemm_16384x1536x384_pmc_*.csv, profiles/r03_gemm_16384x384x1536_pmc_*.csv (stage-3 fc1 / fc2 on the tiled kernel), "
                   "profiles/r03_gemm_qkv16384x1152x384_{wstat,tiled}_pmc_*.csv (the W-stationary kernel against the tiled one): HBM bytes, L2 "
                   "requests / hit rate, MFMA-busy, wait cycles; summary in profiles/README.md"}


# ---------------------------------------------------------------------------------------------------------------------
# BASELINE.json configs[3]: one 3-D volume through segment_volume on all ranks (strong scaling)
def make_volume(device, T, S=1024, seed=0):
    """Synthetic one-organ volume [T,3,S,S] generated ON the device (the 512-slice volume is 6.4 GB: half a minute of host arithmetic
    per rank through synthetic.blob_volume): the same model -- an ellipsoid "organ" of Gaussian intensity over noise, ImageNet
    normalisation, the per-slice bounding box of its iso-surface -- with the noise drawn by the device generator."""
    import math
    g = torch.Generator().manual_seed(5000 + seed)
    cx, cy, cz = (torch.rand(3, generator=g) * 0.5 + 0.25).tolist()
    rx, ry, rz = (torch.rand(3, generator=g) * 0.12 + 0.1).tolist()
    dg = torch.Generator(device=device).manual_seed(77 + seed)
    ys = torch.arange(S, dtype=torch.float32, device=device)[:, None]
    xs = torch.arange(S, dtype=torch.float32, device=device)[None, :]
    gains = torch.tensor([1.0, 0.92, 0.85], device=device)[:, None, None]
    mean = torch.tensor([0.485, 0.456, 0.406], device=device)[:, None, None]
    std = torch.tensor([0.229, 0.224, 0.225], device=device)[:, None, None]
    plane = ((xs / S - cx) / rx) ** 2 + ((ys / S - cy) / ry) ** 2
    vol = torch.empty(T, 3, S, S, device=device)
    boxes = []
    for t in range(T):
        dz = ((t + 0.5) / T - cz) / rz
        sl = 30.0 + 6.0 * torch.randn(S, S, generator=dg, device=device) + 150.0 * torch.exp(-1.5 * (plane + dz * dz))
        vol[t] = ((sl.clamp(0, 255)[None] * gains) / 255.0 - mean) / std
        if abs(dz) < 1.0:
            r = math.sqrt(1.0 - dz * dz)
            boxes.append(((cx - rx * r) * S, (cy - ry * r) * S, (cx + rx * r) * S, (cy + ry * r) * S))
        else:
            boxes.append((S * 0.3, S * 0.3, S * 0.6, S * 0.6))
    return vol, boxes


def volume_workload(device, T):
    volume, boxes = make_volume(device, T)
    prompts = {t: {"boxes": torch.tensor([[float(v) for v in boxes[t]]], device=device)} for t in range(0, T, 2)}
    return volume, prompts


LIVE_VOLUME_STATS: dict = {}       # the running pass's `stats` (phases done so far): what the watchdog reports when the pass hangs


def run_volume(m, device, T, world, sync_dev, steps, warmup):
    """`steps` timed passes of the whole T-slice volume through segment_volume (all ranks, strong scaling).  Returns (seconds for all
    steps = max over ranks, phase seconds of the last pass on this rank, how the chain used the ranks, foreground fraction)."""
    import medical_sam2_amd.parallel as par
    import medical_sam2_amd.volume as vol
    volume, prompts = volume_workload(device, T)
    small = {t: prompts[t] for t in (0, 2)}
    vol.segment_volume(m, volume[:4], small, fill_hole_area=8)       # code objects, weight packing, tables (and the exchange, once)
    for _ in range(warmup):
        vol.segment_volume(m, volume, prompts, fill_hole_area=8)
    torch.cuda.synchronize()
    par.barrier(sync_dev)
    torch.cuda.synchronize()
    st = {}
    t0 = time.perf_counter()
    for i in range(steps):
        st = {"time_phases": i == steps - 1}
        LIVE_VOLUME_STATS.clear()
        LIVE_VOLUME_STATS["stats"] = st
        masks = vol.segment_volume(m, volume, prompts, fill_hole_area=8, stats=st)
    torch.cuda.synchronize()
    par.barrier(sync_dev)
    torch.cuda.synchronize()
    dt = par.max_over_ranks(time.perf_counter() - t0, sync_dev)
    fg = sum(float((v > 0).float().mean()) for v in masks.values()) / len(masks)
    chain = "single rank" if world == 1 else "memory cross-attention key range split over the ranks (1 object < ranks), rest of the chain replicated"
    return dt, st.get("phase_s"), chain, fg, len(masks)


VOLUME_WORKLOAD = ("configs[3]: sam2_hiera_s 3D, ONE {T}-slice 1024x1024 synthetic volume, bbox prompt on every 2nd slice ({C} conditioning "
                   "slices), one object, segment_volume over all ranks: slice-sharded image encoder + conditioning pass, one exchange "
                   "(conditioning memories + non-conditioning features), propagation chain attending to up to {K} memory keys, hole filling")


def volume_side_object(m, device, T, world, sync_dev):
    dt, phases, chain, fg, n = run_volume(m, device, T, world, sync_dev, steps=1, warmup=0)
    return {"workload": VOLUME_WORKLOAD.format(T=T, C=T // 2, K=(T // 2 + 3) * 4096), "scaling": "strong", "n_gpus": world, "slices": n,
            "seconds_per_volume": dt, "slices_per_s": T / dt, "phase_s_rank0": phases, "chain": chain, "mean_foreground_fraction": fg}


# ---------------------------------------------------------------------------------------------------------------------
def other_dtype_side_object(other, steps, warmup, oracle_low):
    """The headline line runs on bf16 operands (BASELINE.json configs[1] states bf16; `--dtype`).  The library also builds on IEEE fp16
    operands (11-bit significand: 8x finer operand rounding at the same MFMA rate): the same benchmark step through
    the OTHER build in a CHILD process (the operand type is fixed when the library is loaded): its slices/s and the parity of its
    slice-0 mask against the oracle mask of `cpu_baseline`."""
    import subprocess
    import tempfile
    name = "libmsam2_hip_bf16.so" if other == "bf16" else "libmsam2_hip.so"
    so = os.path.join(ROOT, "medical-sam2_amd", name)
    if not os.path.exists(so):
        return {"error": f"{name} not built (__graft_entry__.build())"}
    with tempfile.TemporaryDirectory() as td:
        dump = os.path.join(td, "slice0.pt")
        env = dict(os.environ, MSAM2_LIB_PATH=so)
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
            env.pop(k, None)
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", str(steps), "--warmup", str(warmup), "--no-cpu-baseline",
               "--no-train", "--no-volume", "--no-bf16", "--no-rooflines", "--dump-slice0", dump]
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        if r.returncode != 0:
            return {"error": f"{other} child exited {r.returncode}: {r.stderr[-300:]}"}
        child = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
        res = {"dtype": child["dtype"], "value": child["value"], "unit": child["unit"], "ms_per_step": child["ms_per_step"],
               "library": f"{name}, same step, same graph capture, child process"}
        if oracle_low is not None and os.path.exists(dump):
            low = torch.load(dump).float()
            a, b = low > 0, oracle_low > 0
            union = (a | b).sum().item()
            res["parity_slice0"] = {"mask_iou_vs_oracle": (a & b).sum().item() / union if union else 1.0,
                                    "max_abs_dlogit": (low - oracle_low).abs().max().item(),
                                    "mean_abs_dlogit": (low - oracle_low).abs().mean().item()}
        return res


# ---------------------------------------------------------------------------------------------------------------------
def _free_port() -> int:
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n: int, argv) -> int:
    """`python bench.py --gpus N` outside torchrun: start the N ranks as a CHILD (`python -m torch.distributed.run`, one process per GPU,
    rendezvous on 127.0.0.1) and return its exit code.  This parent makes no GPU call before or after (torch.cuda.device_count() does not
    initialise the runtime on this image), so nothing that has touched the GPU is ever replaced or forked."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    have = torch.cuda.device_count()
    if have < n and env.get("MSAM2_BENCH_ONE_GPU") != "1":
        print(f"bench.py: {n} ranks requested, {have} GPU(s) visible -> rehearsal: all ranks share device 0 over gloo (not a scaling figure)",
              file=sys.stderr, flush=True)
        env["MSAM2_BENCH_ONE_GPU"], env["MSAM2_BENCH_BACKEND"] = "1", "gloo"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)


class Watchdog:
    """Side legs that talk to other ranks run under a deadline.  When it passes, the process has given up on a hung collective or a
    stuck kernel: rank 0 prints the line it has (the reason recorded under `key`), every rank dumps the stacks of all its threads and
    what it was doing (`note`: e.g. the phase of the volume pass) to stderr, and EXITS NON-ZERO (3) -- a hang must surface as a failed
    run, not as rc 0 with an error string inside an otherwise valid-looking line."""
    EXIT_CODE = 3

    def __init__(self, seconds, rank, line, key, note=None):
        import threading
        self.rank, self.line, self.key, self.seconds, self.note = rank, line, key, seconds, note
        self.timer = threading.Timer(seconds, self.fire)
        self.timer.daemon = True
        self.timer.start()

    def fire(self):
        import faulthandler
        what = self.note() if callable(self.note) else self.note
        sys.stderr.write(f"[bench watchdog] rank {self.rank}: '{self.key or 'shutdown'}' gave no result within {self.seconds} s; state: {what}\n")
        faulthandler.dump_traceback(file=sys.stderr, all_threads=True)
        sys.stderr.flush()
        if self.rank == 0 and self.line is not None:
            self.line[self.key] = {"error": f"no result within {self.seconds} s (stuck exchange or kernel); state: {what}; exit code {self.EXIT_CODE}"}
            print(json.dumps(self.line), flush=True)
        os._exit(self.EXIT_CODE)

    def cancel(self):
        self.timer.cancel()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 20; 3 volumes with --mode volume)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed steps (default 3; 1 volume with --mode volume)")
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--mode", choices=("2d", "volume"), default="2d")
    ap.add_argument("--slices", type=int, default=512, help="slices of the 3-D volume (--mode volume and the volume_3d side object)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--streams", type=int, default=int(os.environ.get("MSAM2_BENCH_STREAMS", "1")),
                    help="process the step's slices as this many concurrent sub-batches on separate HIP streams (same work)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train", action="store_true", help="skip the (untimed-in-value) training-iteration figure")
    ap.add_argument("--no-volume", action="store_true", help="skip the volume_3d side object of the 2d line")
    ap.add_argument("--dtype", choices=("bf16", "f16"), default="bf16",
                    help="operand type of the 16-bit MFMA operands = which build of the library is loaded: bf16 (BASELINE.json configs[1]; "
                         "libmsam2_hip_bf16.so) or f16 (libmsam2_hip.so).  MSAM2_LIB_PATH, when set, wins.")
    ap.add_argument("--no-bf16", "--no-other-dtype", dest="no_bf16", action="store_true",
                    help="skip the side object that runs the same step on the OTHER operand type's library (N=1 only)")
    ap.add_argument("--no-rooflines", action="store_true", help="skip the per-kernel roofline legs (used by the bf16 child run)")
    ap.add_argument("--dump-slice0", default=None, help="write the low-res mask logits of slice 0 to this file (bf16 child run)")
    args = ap.parse_args()
    if args.steps is None:
        args.steps = 20 if args.mode == "2d" else 3
    if args.warmup is None:
        args.warmup = 3 if args.mode == "2d" else 1

    # the operand type is a property of the loaded library: choose it before the package is imported (and before ranks are launched:
    # the children inherit the environment)
    if not os.environ.get("MSAM2_LIB_PATH"):
        os.environ["MSAM2_LIB_PATH"] = os.path.join(ROOT, "medical-sam2_amd", "libmsam2_hip_bf16.so" if args.dtype == "bf16" else "libmsam2_hip.so")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))        # before any GPU call in this process

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: the launcher's world size is used", file=sys.stderr, flush=True)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP library is the only compute path (no CPU fallback)")
    # rehearsal mode for a 1-GPU box: several ranks share device 0 and talk over gloo (RCCL refuses duplicate devices)
    one_gpu = os.environ.get("MSAM2_BENCH_ONE_GPU") == "1"
    dev_index = 0 if one_gpu else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    import medical_sam2_amd.parallel as par
    backend = os.environ.get("MSAM2_BENCH_BACKEND", "nccl")
    if world > 1:
        par.init_distributed(backend=backend, device=device)
    dist = torch.distributed if world > 1 else None
    sync_dev = device if backend == "nccl" else torch.device("cpu")
    rehearsal = {"rehearsal": f"{world} ranks share ONE GPU over {backend}: launch-path check, not a scaling figure"} if (one_gpu and world > 1) else {}

    torch.set_grad_enabled(False)
    m = build_model(device)
    import medical_sam2_amd.ops as ops
    dtype = "f16" if ops.OP16 == torch.float16 else "bf16"

    if args.mode == "volume":
        T = args.slices
        dt, phases, chain, fg, n = run_volume(m, device, T, world, sync_dev, args.steps, args.warmup)
        if rank == 0:
            line = {"metric": "slices/sec @1024^2 (Hiera-S)", "value": T * args.steps / dt, "unit": "slices/s", "n_gpus": world,
                    "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
                    "scaling": "strong", "vs_baseline": None, "dtype": dtype, "data": "synthetic",
                    "config": {"workload": VOLUME_WORKLOAD.format(T=T, C=T // 2, K=(T // 2 + 3) * 4096), "slices_per_step": n,
                               "step": "one whole volume", "backend": backend if world > 1 else None, "chain": chain,
                               "weights": "random name-keyed init", **rehearsal},
                    "phase_s_rank0_last_volume": phases, "mean_foreground_fraction": fg}
            print(json.dumps(line), flush=True)
        if dist is not None:
            par.barrier(sync_dev)
            dist.destroy_process_group()
        return

    imgs, pts, labels, bank_feats, sampled = make_inputs(device, args.batch, rank)
    memory, memory_pos = assemble_memory(m, bank_feats, sampled)

    if args.streams > 1:
        # independent slices => independent streams: the tail/epilogue of one sub-batch's kernels overlaps the head of the other's
        assert args.batch % args.streams == 0
        per = args.batch // args.streams
        side = [torch.cuda.Stream() for _ in range(args.streams)]
        parts = []
        for i in range(args.streams):
            sl = slice(i * per, (i + 1) * per)
            mem_i, pos_i = assemble_memory(m, bank_feats, sampled[sl])
            parts.append((imgs[sl].contiguous(), pts[sl].contiguous(), labels[sl].contiguous(), mem_i, pos_i))

        def run():
            cur = torch.cuda.current_stream()
            outs = []
            for st, (im, pt, lb, mem_i, pos_i) in zip(side, parts):
                st.wait_stream(cur)
                with torch.cuda.stream(st):
                    outs.append(step_2d(m, im, pt, lb, mem_i, pos_i))
            for st in side:
                cur.wait_stream(st)
            return outs
    else:
        run = lambda: step_2d(m, imgs, pts, labels, memory, memory_pos)
    out = run()  # first call: weight packing, table generation, kernel module load
    torch.cuda.synchronize()
    graph = None
    if not args.no_graph:
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            run()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out = run()
        run = graph.replay
    for _ in range(args.warmup):
        run()
    torch.cuda.synchronize()
    par.barrier(sync_dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    torch.cuda.synchronize()
    par.barrier(sync_dev)
    torch.cuda.synchronize()
    dt = par.max_over_ranks(time.perf_counter() - t0, sync_dev)
    low0 = (out[0][0] if isinstance(out, list) else out[0])[:1]
    if args.dump_slice0 and rank == 0:
        torch.save(low0.detach().float().cpu(), args.dump_slice0)

    line = None
    if rank == 0:
        slices = args.batch * args.steps * world
        line = {
            "metric": "slices/sec @1024^2 (Hiera-S)", "value": slices / dt, "unit": "slices/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": dtype, "data": "synthetic",
            "config": {"workload": "configs[1]: sam2_hiera_s 2D (train_2d SAM2 sub-sequence, forward), b=4 x 1024x1024 per GPU, "
                                   "16-entry memory bank with 4 sampled memories per slice, one click per slice; "
                                   "forward_image -> memory_attention -> prompt encoder -> mask decoder -> memory encoder",
                       "slices_per_step_per_gpu": args.batch, "hip_graph": graph is not None, "weights": "random name-keyed init",
                       "backend": backend if world > 1 else None, **rehearsal}}
        if not args.no_rooflines:
            k_s, k_flops, k_exec, k_bytes, splits, pair_s = time_dominant_kernel(device, args.batch)
            achieved = k_flops / k_s / 1e12
            line["roofline"] = {"bound": "mfma", "achieved": achieved, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                                "frac": achieved / MFMA_BF16_PEAK_TFLOPS, "traffic": pmc_traffic(),
                                "kernel": f"attn_kv64x2_kernel<4> (memory cross-attention, value product folded to the 64-channel memory space, 64 queries per wave; "
                                          f"split-KV pass, {splits} splits) at B={args.batch} Lq=4096 Lk={args.batch * 4096}",
                                "avg_launch_us": k_s * 1e6, "with_merge_us": pair_s * 1e6, "flops_per_launch": k_flops,
                                "flops_definition": "algorithmic, SURVEY 8(d): 4*Lq*Lk*256 per object (the reference's 256-wide SDPA)",
                                "executed_flops_per_launch": k_exec, "frac_on_executed_flops": k_exec / k_s / 1e12 / MFMA_BF16_PEAK_TFLOPS,
                                "algorithmic_bytes_per_launch": k_bytes,
                                "hbm_GBs_on_algorithmic_bytes": k_bytes / k_s / 1e9, "hbm_frac_of_peak": k_bytes / k_s / 1e9 / HBM_PEAK_GBS,
                                "note": "largest single kernel of the step; the largest kernel FAMILY is the GEMMs: roofline_gemm"}
            line["roofline_gemm"] = time_gemm_family(m, imgs, pts, labels, memory, memory_pos, device)
            line["roofline_hiera_attention"] = time_hiera_attention(device, args.batch)
        oracle_low = None
        if world == 1 and not args.no_cpu_baseline:
            c = lambda t: t.detach().float().cpu()
            sample = (c(imgs[:1]), c(pts[:1]), labels[:1].cpu(), c(memory[:, :1]), c(memory_pos[:, :1]), c(low0))
            line["cpu_baseline"] = cpu_baseline(sample)
            oracle_low = line["cpu_baseline"].pop("_oracle_low", None)
        if world == 1 and not args.no_train:
            line["train_iteration"] = train_iteration(m, imgs, pts, labels, memory, memory_pos, device, full=True)
            line["train_iteration_frozen_encoder"] = train_iteration(m, imgs, pts, labels, memory, memory_pos, device, full=False)
        if world == 1 and not args.no_bf16:
            other = "f16" if dtype == "bf16" else "bf16"
            try:
                line[other] = other_dtype_side_object(other, args.steps, args.warmup, oracle_low)
            except Exception as e:  # noqa: BLE001 -- a side figure
                line[other] = {"error": f"{type(e).__name__}: {e}"[:300]}
    if not args.no_volume:
        # all ranks: the 3-D path at configs[3]'s size (strong scaling), after the timed region; never part of `value`
        dog = Watchdog(420, rank, line, "volume_3d",
                       note=lambda: f"volume pass, phases completed so far: {LIVE_VOLUME_STATS.get('stats', {}).get('phases_done', 'none (warm-up pass)')}")
        try:
            side_obj = volume_side_object(m, device, args.slices, world, sync_dev)
        except Exception as e:  # noqa: BLE001 -- a side figure: report, do not fail the benchmark line
            side_obj = {"error": f"{type(e).__name__}: {e}"[:300]}
        dog.cancel()
        if rank == 0:
            line["volume_3d"] = side_obj
    if rank == 0:
        print(json.dumps(line), flush=True)
    if dist is not None:
        dog = Watchdog(120, rank, None, "", note="final barrier / destroy_process_group")
        par.barrier(sync_dev)
        dist.destroy_process_group()
        dog.cancel()


if __name__ == "__main__":
    main()
