"""Training steps on the HIP path for the parameter groups train_3d.py:34-54 optimises around the frozen image
and prompt encoders -- no autograd graph: explicit recomputing backward (`backward.py`), BCE-with-logits loss (func_3d/function.py:69,
`criterion_G`) and Adam (`train_3d.py:50-54`) as kernels of libmsam2_hip.so.

  decoder_finetune_step          mask decoder only (`sam_layers`, lr 1e-4)
  memory_decoder_finetune_step   memory attention (the bulk of `mem_layers`) + mask decoder, memory bank detached as
                                 func_2d/function.py:204-243 stores it
  train_step_2d                  one whole iteration of the 2-D flow: frozen encoders forward, the joint step, memory encoding of the new
                                 prediction (`backward.memory_encoder_backward` exists for the third group; it sees no gradient in the 2-D
                                 flow because the bank is detached)
  memory_bank_finetune_step      one level of BPTT through the memory bank (the `non_prompt_loss` path of func_3d/function.py:160-184): the
                                 current slice's loss also trains the memory encoder that produced the previous slice's memory
  data_parallel=True             gradients averaged over the ranks with one bucketed all-reduce (`parallel.allreduce_gradients`)

Scope: the loss is taken either on the decoder's low-resolution logits of all `num_mask_tokens` masks against one target per mask token,
or -- `mask_index` / `upsampled_mask_loss`, the reference's form -- on one mask up-sampled to the video resolution.  The reference's 3-D
loop back-propagates through the whole memory bank including the previous slices' decoders -- only the first level is built.  The IoU / object-score heads do not receive a gradient
from this loss.  16-bit operands: the loss gradient is scaled by a fixed power of two and un-scaled inside the Adam kernel.
Pinned by tests/test_backward_gpu.py (oracle + autograd + torch.optim) and tests/test_grads_golden.py (the reference's own `.grad`).
"""
from __future__ import annotations

import math
from typing import Dict

import torch

from . import backward as bwd
from . import ops
from . import parallel
from ._lib import check, lib
from .ops import F32, _p, _stream


def bce_with_logits(logits: torch.Tensor, target: torch.Tensor, pos_weight: float = 1.0):
    """(loss fp32 scalar tensor, dloss/dlogits fp32) for mean-reduced BCEWithLogitsLoss(pos_weight)."""
    x, y = logits.to(F32).contiguous(), target.to(F32).contiguous()
    assert x.shape == y.shape
    dx = torch.empty_like(x)
    loss = torch.zeros(1, dtype=F32, device=x.device)
    check(lib().msam2_bce_logits(_p(x), _p(y), _p(dx), _p(loss), x.numel(), float(pos_weight), _stream()))
    return loss, dx


def upsampled_mask_loss(masks: torch.Tensor, target_highres: torch.Tensor, mask_index: int = 0, pos_weight: float = 1.0):
    """The reference's training loss (func_3d/function.py:137-170 on `_get_orig_video_res_output`, sam2_video_predictor.py:724-744):
    mean BCE-with-logits between ONE mask of the decoder ([B, num_mask_tokens, h4, w4] logits -> token `mask_index`) bilinearly
    up-sampled to the target's resolution and target_highres [B, 1, S, S] in {0, 1}.
    Returns (loss 1-element tensor, d loss / d masks [B, num_mask_tokens, h4, w4] -- zero for the other tokens)."""
    Bn, nm, h4, w4 = masks.shape
    S_h, S_w = target_highres.shape[-2:]
    low = masks[:, mask_index].contiguous()                                      # [B, h4, w4]
    up = ops.bilinear_upsample(low.view(Bn, 1, h4, w4), S_h, S_w)
    loss, d_up = bce_with_logits(up.reshape(Bn, -1), target_highres.reshape(Bn, -1), pos_weight)
    d_low = torch.empty(Bn, h4, w4, dtype=F32, device=masks.device)
    check(lib().msam2_bilinear_upsample_bwd(_p(d_up), _p(d_low), Bn, h4, w4, S_h, S_w, _stream()))
    d_masks = torch.zeros_like(masks, dtype=F32)
    d_masks[:, mask_index] = d_low
    return loss, d_masks


class DecoderAdam:
    """torch.optim.Adam(params, lr, betas, eps) semantics for the parameters of one module (the mask decoder, the memory attention, ...:
    gradient names are relative to it), state kept as flat fp32 tensors."""

    def __init__(self, decoder, lr: float = 1e-4, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0):
        """weight_decay = 0: torch.optim.Adam as train_3d.py:50-54 builds it; > 0: torch.optim.AdamW's decoupled decay (train_2d.py:43-47)."""
        self.decoder, self.lr, self.betas, self.eps, self.weight_decay = decoder, lr, betas, eps, weight_decay
        self.state: Dict[str, tuple] = {}
        self._t_dev = None          # int32 [2] on the parameters' device: [step count t (advanced by a kernel inside every step), number of
                                    # non-finite gradient elements skipped so far]

    @property
    def t(self) -> int:
        """number of steps taken so far, eager and replayed (one device -> host read)"""
        return 0 if self._t_dev is None else int(self._t_dev[0].item())

    @property
    def skipped_elements(self) -> int:
        """gradient elements that were non-finite (an overflowed 16-bit backward operand: stale loss scale) and left their parameter
        untouched, summed over all steps so far, eager and replayed (one device -> host read).  Non-zero = re-calibrate."""
        return 0 if self._t_dev is None else int(self._t_dev[1].item())

    def mark_updated(self):
        """Bump the version counters of the parameters (no kernel).  `step` does it itself; a hipGraph REPLAY of a captured step
        updates the parameters through raw pointers without running this host code, so whoever replays calls it afterwards
        (`GraphedStep.replay` does) -- otherwise the kernel-ready 16-bit weight copies of `WeightCache` that EAGER code reads would
        go stale after the first post-capture rebuild."""
        ps = tuple(self.decoder.parameters())
        torch._C._autograd._unsafe_set_version_counter(ps, tuple(p._version + 1 for p in ps))

    @torch.no_grad()
    def step(self, grads: Dict[str, torch.Tensor], grad_scale: float = 1.0):
        """`grads` may carry a loss scale: they are multiplied by `grad_scale` (its inverse) inside the update kernel.
        The step count t lives on the device and is incremented by a kernel of the same call, so a captured graph advances the bias
        corrections 1 - beta^t on every replay exactly like torch.optim.Adam (tests/test_backward_gpu.py::test_adam_graph_replays)."""
        import ctypes
        params = dict(self.decoder.named_parameters())
        if self._t_dev is None:
            self._t_dev = torch.zeros(2, dtype=torch.int32, device=next(iter(params.values())).device)
        names = list(grads.keys())
        keep = []                                      # fp32 contiguous gradient copies stay alive until the launches are queued
        tabs = [[], [], [], []]
        numel = []
        for name in names:
            p, g = params[name], grads[name]
            assert p.dtype == F32 and p.is_contiguous() and g.shape == p.shape, name
            if name not in self.state:
                self.state[name] = (torch.zeros_like(p), torch.zeros_like(p))
            m, v = self.state[name]
            gc = g.to(F32).contiguous()
            keep.append(gc)
            for tab, t in zip(tabs, (p, gc, m, v)):
                tab.append(t.data_ptr())
            numel.append(p.numel())
        n = len(names)
        arr = [(ctypes.c_void_p * n)(*tab) for tab in tabs]
        check(lib().msam2_adam_step_multi(arr[0], arr[1], arr[2], arr[3], (ctypes.c_int64 * n)(*numel), n, self.lr, self.betas[0],
                                          self.betas[1], self.eps, 0, float(grad_scale), float(self.weight_decay), _p(self._t_dev),
                                          self._t_dev.data_ptr() + 4, _stream()))
        # the update went through raw pointers: bump the tensor versions (no kernel) so cached kernel-ready weights are rebuilt
        ps = tuple(params[name] for name in names)
        torch._C._autograd._unsafe_set_version_counter(ps, tuple(p._version + 1 for p in ps))


class GraphedStep:
    """A training step captured into a hipGraph together with what a replay needs on the host side: `replay()` relaunches the graph and
    then bumps the version counters of every parameter the captured optimisers update (`DecoderAdam.mark_updated`), so that eager code
    running between replays (validation, the predictor, `_encode_new_memory`) rebuilds its 16-bit weight copies instead of reading
    stale ones.  The step function must be capturable (sync=False: no host reads) and already calibrated by one eager call.

    Frozen loss scales.  The power-of-two scales of the 16-bit backward operands are calibrated on the eager step and baked
    into the graph; gradient magnitudes drift over training.  Every link records its scaled max|gradient| on the device inside the step
    (`backward_encoder.record_scaled_amax`) and Adam counts the non-finite elements it skipped: `drift()` reads both (one host
    synchronisation), `check_every=N` does so every N replays, and when anything left the safe band and an `eager_fn` (the same step with
    sync=True) was given, the calibrations are dropped, the step runs eagerly once (re-calibrating) and the graph is captured again."""

    def __init__(self, step_fn, optimizers, eager_fn=None, check_every: int = 0, model=None):
        """model (optional): the whole network -- its frozen modules' kernel-ready weights are pinned with the trained ones (below)."""
        self.optimizers, self.step_fn, self.eager_fn, self.check_every = list(optimizers), step_fn, eager_fn, int(check_every)
        self.model = model
        self.replays = self.recalibrations = 0
        self._skipped = 0
        self._capture()

    def _capture(self):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            self.step_fn()                              # warm-up on a side stream (allocator, lazy module loads)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = self.step_fn()
        # The graph holds raw pointers to the kernel-ready weight copies it read or refreshed (`refresh_casts` copies in place into tensors
        # that were allocated eagerly and are owned by the modules' WeightCache only).  Plain casts are refreshed in place by every
        # reader (`WeightCache.get`), so an eager forward between two replays keeps the pointers valid; the references held here keep
        # the blocks alive even if a cache is cleared or an entry replaced while this graph exists.
        from .modeling.common import cached_weight_tensors
        mods = ([self.model] if self.model is not None else []) + [o.decoder for o in self.optimizers]
        self._pinned = [t for mod in mods for t in cached_weight_tensors(mod)]
        # capture RECORDED the weight refreshes without running them, but the host side stamped the cache entries as fresh: bump the
        # versions so that an eager reader before the first replay converts the current weights instead of trusting those stamps
        for o in self.optimizers:
            o.mark_updated()

    def replay(self):
        self.graph.replay()
        for o in self.optimizers:
            o.mark_updated()
        self.replays += 1
        if self.check_every and self.replays % self.check_every == 0:
            self.check()
        return self.out

    def _scale_dicts(self):
        for o in self.optimizers:
            if getattr(o, "scale_monitor", None):
                yield o.scale_monitor
            for d in (getattr(o, "calibrated_block_scales", None) or {}).values():
                yield d

    def drift(self) -> dict:
        """{link: scaled max|gradient| outside the safe band since the last call} plus {"skipped_elements": n} when Adam skipped non-finite
        gradient elements since the last call; empty = the frozen scales still fit.  One host synchronisation."""
        from .backward_encoder import scale_drift
        out = {}
        for d in self._scale_dicts():
            out.update(scale_drift(d))
        skipped = sum(o.skipped_elements for o in self.optimizers)
        if skipped != self._skipped:
            out["skipped_elements"] = skipped - self._skipped
            self._skipped = skipped
        return out

    def check(self) -> dict:
        d = self.drift()
        if d and self.eager_fn is not None:
            self.recalibrate()
        return d

    def recalibrate(self):
        """drop every cached scale, run the step eagerly once (it calibrates afresh, with host reads) and capture it again"""
        for o in self.optimizers:
            if hasattr(o, "calibrated_loss_scales"):
                o.calibrated_loss_scales = {}
            if hasattr(o, "calibrated_block_scales"):
                o.calibrated_block_scales = None
            if hasattr(o, "scale_monitor"):
                o.scale_monitor = {}
        self.eager_fn()
        self._capture()
        self.recalibrations += 1


def _shared_pow2_scale(amax: torch.Tensor, data_parallel: bool, group=None) -> float:
    """Power-of-two loss scale from max|gradient| (a 1-element device tensor).  Data parallel: the maximum is taken over ALL ranks first
    (one MAX all-reduce), so every rank scales -- and later un-scales -- by the same factor; summing gradients that carry different
    scales would be wrong."""
    if data_parallel and parallel._is_dist(group):
        import torch.distributed as dist
        dist.all_reduce(amax, op=dist.ReduceOp.MAX, group=group)
    a = float(amax.item())
    return 2.0 ** (-3 - math.ceil(math.log2(a))) if a > 0 and math.isfinite(a) else 1.0


def _assert_same_on_all_ranks(value: float, what: str, group=None):
    if parallel._is_dist(group):
        import torch.distributed as dist
        vals = [None] * dist.get_world_size(group)
        dist.all_gather_object(vals, float(value), group=group)
        assert all(v == vals[0] for v in vals), f"{what} differs across the data-parallel ranks: {vals}"


@torch.no_grad()
def decoder_finetune_step(decoder, optimizer: DecoderAdam, src_tokens, pe_tokens, sparse, feat_s0, feat_s1, B: int, h: int, w: int,
                          target_masks: torch.Tensor, pos_weight: float = 1.0, sync: bool = True, data_parallel: bool = False,
                          mask_index: int = None):
    """One optimisation step of the mask decoder.  Inputs as for `MaskDecoder.predict_masks_tokens`; target_masks [B, nm, 4h, 4w] in
    {0, 1} (one target per mask token), or -- with mask_index -- [B, 1, S, S] at the video resolution for that one mask (the
    reference's loss, `upsampled_mask_loss`).  Returns the loss value before the update (a Python float; with sync=False the 1-element device tensor, so that the whole
    step -- ~3000 small launches -- can be captured in a hipGraph and replayed without host work).  data_parallel: one process per GPU,
    each on its own slices; the gradients are averaged with `parallel.allreduce_gradients` (one bucketed RCCL all-reduce) before Adam."""
    masks, _, _, _ = decoder.predict_masks_tokens(src_tokens, pe_tokens, sparse, feat_s0, feat_s1, B, h, w)
    if mask_index is None:
        loss, d_masks = bce_with_logits(masks, target_masks, pos_weight)
        n_loss = masks.numel()
    else:   # the reference's loss: one mask, up-sampled to the target's (video) resolution -- target_masks [B, 1, S, S]
        loss, d_masks = upsampled_mask_loss(masks, target_masks, mask_index, pos_weight)
        n_loss = target_masks.numel()
    # Loss scale: |dloss/dlogit| <= max(pos_weight, 1) / n is ~1e-6 at 1024^2 -- below the 16-bit operand's normal range (fp16: 6e-5).
    # A fixed power of two (data independent, so the step stays capturable) brings the largest entry to 2^-4; the backward is linear
    # in d_masks, the update kernel multiplies the gradients by the inverse.
    # (n_loss elements share the mean; the up-sampling adjoint sums ~16 of them per low-res pixel: still <= 2^0)
    scale = 2.0 ** (math.floor(math.log2(n_loss / max(float(pos_weight), 1.0))) - 4 - (0 if mask_index is None else 4))
    d_masks.mul_(scale)
    _, _, grads = bwd.mask_decoder_backward(decoder, src_tokens, pe_tokens, sparse, feat_s0, feat_s1, B, h, w, d_masks)
    inv_world = 1.0
    if data_parallel:                                   # each rank on its own slices: mean of the per-rank gradients over RCCL
        grads, inv_world = parallel.allreduce_gradients(grads)
    optimizer.step(grads, grad_scale=inv_world / scale)
    return float(loss.item()) if sync else loss


@torch.no_grad()
def memory_decoder_loss_grads(memory_attention, decoder, curr, curr_pos, memory, memory_pos, num_obj_ptr_tokens: int, pe_tokens, sparse,
                              feat_s0, feat_s1, B: int, h: int, w: int, target_masks: torch.Tensor, dense_tokens=None,
                              pos_weight: float = 1.0, mem_scale: float = None, aux: dict = None, mask_index: int = None,
                              data_parallel: bool = False, on_decoder_grads=None, monitor: dict = None):
    """Forward + backward of the memory-conditioned slice step (func_2d/function.py:70-191 / sam2_base.py:705-790 with a frozen image
    encoder and a detached memory bank, as func_2d/function.py:204-243 stores it): curr / curr_pos [L, B, C] current-slice features,
    memory / memory_pos [Nk, B, 64] the assembled bank -> memory attention -> (+ dense prompt embedding) -> mask decoder -> mean BCE with
    logits on its mask logits (all mask tokens against target_masks [B, nm, 4h, 4w], or with mask_index the reference's form: that one
    mask up-sampled to target_masks [B, 1, S, S]).  Returns (loss 1-element tensor, decoder loss scale, memory loss scale, decoder gradients,
    memory-attention gradients, dcurr); each group's gradients (and dcurr) carry its loss scale (see `decoder_finetune_step`).
    The gradient that leaves the decoder towards the memory attention is orders of magnitude smaller than the one that entered it
    (it has crossed two attention blocks and two transposed convolutions), again below the fp16 operand range, so it is re-scaled by a
    second power of two: `mem_scale` if given (a captured graph must pass the value calibrated on an eager step), else chosen from
    max|d_src| -- one host synchronisation; with data_parallel the maximum is taken over all ranks (one MAX all-reduce), because the
    ranks' gradients are summed afterwards and must carry the same scale.  monitor (a dict kept by the caller between steps): receives
    the device-side running max of the re-scaled gradient (`backward_encoder.record_scaled_amax`), what `GraphedStep.drift` reads."""
    L, _, C = curr.shape
    # train() mode: nn.Dropout / attention dropout of the memory attention, masks re-created by the backward from the same counter stream
    y, state = bwd.memory_attention_forward_saved(memory_attention, curr, curr_pos, memory, memory_pos, num_obj_ptr_tokens,
                                                  dropout=memory_attention.next_dropout())
    src = y.transpose(0, 1).reshape(B * L, C)
    if dense_tokens is not None:                                                  # [L, C] or [1, C] (no_mask_embed), broadcast over the batch
        d2 = dense_tokens.reshape(-1, C).to(F32)
        src = ops.add_cast(src.view(B, L, C), d2.view(1, -1, C).expand(B, L, C), 1.0, F32).view(B * L, C)
    else:
        src = src.contiguous()
    masks, _, _, _ = decoder.predict_masks_tokens(src, pe_tokens, sparse, feat_s0, feat_s1, B, h, w)
    if aux is not None:
        aux["masks"] = masks                                                     # [B, num_mask_tokens, 4h, 4w] logits of this forward
    if mask_index is None:
        loss, d_masks = bce_with_logits(masks, target_masks, pos_weight)
        n_loss, extra = masks.numel(), 0
    else:                                                                        # the reference's loss form (see upsampled_mask_loss)
        loss, d_masks = upsampled_mask_loss(masks, target_masks, mask_index, pos_weight)
        n_loss, extra = target_masks.numel(), 4
    scale = 2.0 ** (math.floor(math.log2(n_loss / max(float(pos_weight), 1.0))) - 4 - extra)
    d_masks.mul_(scale)
    d_src, _, g_dec = bwd.mask_decoder_backward(decoder, src, pe_tokens, sparse, feat_s0, feat_s1, B, h, w, d_masks, aux=aux)
    if on_decoder_grads is not None:
        on_decoder_grads(g_dec)                                                  # e.g. start their all-reduce under the memory attention's backward
    if aux is not None:
        aux["d_src"] = d_src                                                     # gradient entering the memory attention, carrying `scale`
    calibrating = mem_scale is None
    if calibrating:
        mem_scale = _shared_pow2_scale(d_src.abs().max().reshape(1), data_parallel)
    d_src = d_src * mem_scale
    if monitor is not None:
        from .backward_encoder import record_scaled_amax
        record_scaled_amax(monitor, "mem_scale", d_src, calibrating)
    dcurr, dmemory, dmemory_pos, g_mem = bwd.memory_attention_backward_saved(memory_attention, state, d_src.view(B, L, C).transpose(0, 1))
    if aux is not None:
        aux["dmemory"], aux["dmemory_pos"] = dmemory, dmemory_pos                # [Nk, B, 64] each, carrying scale * mem_scale
    return loss, scale, scale * mem_scale, g_dec, g_mem, dcurr


@torch.no_grad()
def memory_decoder_finetune_step(memory_attention, decoder, opt_mem: DecoderAdam, opt_dec: DecoderAdam, *args, sync: bool = True,
                                 data_parallel: bool = False, **kwargs):
    """One optimisation step of both parameter groups train_3d.py:34-54 builds around the frozen image encoder -- the mask decoder
    (`sam_layers`) and the memory attention (the bulk of `mem_layers`) -- on the loss of `memory_decoder_loss_grads` (same arguments).  The first call calibrates the memory group's loss scale (one host
    synchronisation) and stores it on `opt_mem`; later calls -- and a hipGraph captured after it -- reuse it."""
    cal = getattr(opt_mem, "calibrated_loss_scales", {})                         # one calibration per loss form, made on its first (eager) step
    if kwargs.get("mem_scale") is None:
        kwargs["mem_scale"] = cal.get(kwargs.get("mask_index"))
    calibrating = kwargs.get("mem_scale") is None
    pending = []
    hook = (lambda g: pending.append(parallel.allreduce_gradients_async(g))) if data_parallel else None   # decoder sums overlap the next backward
    loss, scale, scale_mem, g_dec, g_mem, _ = memory_decoder_loss_grads(memory_attention, decoder, *args, data_parallel=data_parallel,
                                                                       on_decoder_grads=hook, **kwargs)
    cal[kwargs.get("mask_index")] = scale_mem / scale
    opt_mem.calibrated_loss_scales = cal
    inv_world = 1.0
    if data_parallel:
        if calibrating:                                 # the calibration was collective (MAX over ranks): every rank must hold the same scale
            _assert_same_on_all_ranks(scale_mem / scale, "memory-attention loss scale")
        mem_pending = parallel.allreduce_gradients_async(g_mem)
        g_dec, inv_world = pending[0].wait()
        g_mem, _ = mem_pending.wait()
    opt_dec.step(g_dec, grad_scale=inv_world / scale)
    opt_mem.step(g_mem, grad_scale=inv_world / scale_mem)
    return float(loss.item()) if sync else loss


@torch.no_grad()
def train_step_2d(model, opt_mem: DecoderAdam, opt_dec: DecoderAdam, imgs, pts, labels, memory, memory_pos, target_masks, sync: bool = True,
                  mask_index: int = None, opt_enc: DecoderAdam = None, grads_out: dict = None, data_parallel: bool = False):
    """One training iteration of the 2-D flow (func_2d/function.py:70-259) on the HIP path: image encoder forward -> memory attention
    over the (detached) bank -> prompt encoder (no gradient: it runs under torch.no_grad() in the reference, func_2d/function.py:140-149)
    -> mask decoder -> BCE on the mask logits -> backward of decoder + memory attention (+ the image encoder) -> Adam -> the new memory
    is encoded from the predicted mask for the bank (forward only, stored detached as func_2d/function.py:204-243 does).
    opt_enc (a DecoderAdam over `model.image_encoder`): also differentiate and update the IMAGE ENCODER -- Hiera trunk + FPN neck, and
    through the neck the mask decoder's conv_s0 / conv_s1 -- as train_2d.py:43-47 does (AdamW over every net.parameters(), the encoder
    under grad at func_2d/function.py:70-72); None keeps it frozen (train_3d.py:34-37's choice of groups).
    imgs [B,3,S,S] normalised, pts [B,P,2] / labels [B,P] clicks, memory / memory_pos [Nk,B,64] (bench.assemble_memory layout),
    target_masks [B, num_mask_tokens, S/4, S/4], or with mask_index [B, 1, S, S] (the reference's loss on that up-sampled mask).
    grads_out (optional dict): receives the TRUE gradients of every group ("decoder", "memory_attention", "image_encoder") for audits.
    data_parallel: one process per GPU on its own images (BASELINE.json configs[4]); every group's gradients are summed over the ranks
    by `parallel.allreduce_gradients_async`, started as soon as the group's backward is done so that the transfer runs under the next
    group's backward (decoder under memory attention, memory attention under image encoder), and averaged inside Adam (grad_scale).
    Returns (loss, maskmem_features [B,64,S/16,S/16])."""
    from . import backward as bwd_mod
    from . import backward_encoder as be
    from .modeling.common import refresh_casts, to_bf16, tokens_of
    B = imgs.shape[0]
    refresh_casts(model)                               # the previous step's Adam left every 16-bit weight copy stale: one multi-tensor copy
    if opt_enc is not None:
        backbone_out, enc_state = be.image_encoder_forward_saved(model, imgs)
    else:
        backbone_out = model.forward_image(imgs)
    _, vision_feats, vision_pos_embeds, feat_sizes = model._prepare_backbone_features(backbone_out)
    h, w = feat_sizes[-1]
    se, _ = model.sam_prompt_encoder(points=(pts, labels), boxes=None, masks=None, batch_size=B)
    pe = tokens_of(model.sam_prompt_encoder.get_dense_pe().to(torch.float32))[: h * w]
    hr = [f.permute(1, 2, 0).view(B, -1, *s) for f, s in zip(vision_feats[:-1], feat_sizes[:-1])]
    f0, f1 = to_bf16(tokens_of(hr[0])), to_bf16(tokens_of(hr[1]))
    dense = model.sam_prompt_encoder.no_mask_embed.weight.detach().reshape(1, -1)
    aux: dict = {}
    cal = getattr(opt_mem, "calibrated_loss_scales", {})                         # one calibration per loss form
    pend: dict = {}
    mon = getattr(opt_mem, "scale_monitor", None)
    if mon is None:
        mon = opt_mem.scale_monitor = {}
    kwargs = dict(dense_tokens=dense, aux=aux, mem_scale=cal.get(mask_index), mask_index=mask_index, data_parallel=data_parallel, monitor=mon,
                  on_decoder_grads=(lambda g: pend.__setitem__("dec", parallel.allreduce_gradients_async(g))) if data_parallel else None)
    # weight-gradient outputs of this backward pass come from buffers zeroed once (backward.ZeroArena): no zeroing launch per GEMM
    with bwd_mod.zero_arena():
        loss, scale, scale_mem, g_dec, g_mem, dcurr = memory_decoder_loss_grads(
            model.memory_attention, model.sam_mask_decoder, vision_feats[-1], vision_pos_embeds[-1], memory, memory_pos, 0, pe, se.to(torch.float32),
            f0, f1, B, h, w, target_masks, **kwargs)
        cal[mask_index] = scale_mem / scale
        opt_mem.calibrated_loss_scales = cal
        inv_world = 1.0
        if data_parallel:
            pend["mem"] = parallel.allreduce_gradients_async(g_mem)
        if opt_enc is not None:
            # the encoder's three outputs received: level 0 / 1 through the decoder's up-scaling adds (loss scale `scale`), level 2 through
            # the memory attention's query stream (`scale_mem`); the encoder backward runs every block under its own cached scale
            C = dcurr.shape[-1]
            d_top = dcurr.transpose(0, 1).reshape(B * h * w, C).contiguous()
            enc_scales = getattr(opt_enc, "calibrated_block_scales", None)
            if enc_scales is None:
                enc_scales = opt_enc.calibrated_block_scales = {}
            g_all = be.image_encoder_backward(model, enc_state, [aux["d_feat_s0"], aux["d_feat_s1"], d_top], [scale, scale, scale_mem],
                                              enc_scales.setdefault(mask_index, {}))
            if data_parallel:
                g_all, inv_world = parallel.allreduce_gradients(g_all)
            g_enc = {k[len("image_encoder."):]: v for k, v in g_all.items() if k.startswith("image_encoder.")}
            g_convs = {k[len("sam_mask_decoder."):]: v * scale for k, v in g_all.items() if k.startswith("sam_mask_decoder.")}
        if data_parallel:
            g_dec, inv_world = pend["dec"].wait()
            g_mem, _ = pend["mem"].wait()
        if opt_enc is not None:
            g_dec = dict(g_dec)
            g_dec.update(g_convs)                                                    # conv_s0 / conv_s1 live in the decoder's group
            opt_enc.step(g_enc, grad_scale=inv_world)
            if grads_out is not None:
                grads_out["image_encoder"] = g_enc
        if grads_out is not None:
            grads_out["decoder"] = {k: v / scale for k, v in g_dec.items()}
            grads_out["memory_attention"] = {k: v / scale_mem for k, v in g_mem.items()}
        opt_dec.step(g_dec, grad_scale=inv_world / scale)
        opt_mem.step(g_mem, grad_scale=inv_world / scale_mem)
    low_res = aux["masks"][:, :1].contiguous()                                   # single-mask output token (multimask_output=False)
    high_res = ops.bilinear_upsample(low_res, model.image_size, model.image_size)
    maskmem_features, _ = model._encode_new_memory(current_vision_feats=vision_feats, feat_sizes=feat_sizes, pred_masks_high_res=high_res,
                                                   is_mask_from_pts=True)
    return (float(loss.item()) if sync else loss), maskmem_features


@torch.no_grad()
def memory_bank_loss_grads(model, curr, curr_pos, prev_pix_tokens, prev_mask_logits, prev_is_mask_from_pts: bool, memory_pos, pe_tokens, sparse,
                           feat_s0, feat_s1, B: int, h: int, w: int, target_masks: torch.Tensor, dense_tokens=None, pos_weight: float = 1.0,
                           mem_scale: float = None, prev_sam_token: torch.Tensor = None, mask_index: int = None):
    """One level of back-propagation through the memory bank -- the path of the 3-D loop's `non_prompt_loss` (func_3d/function.py:160-184)
    that trains the memory ENCODER: the previous slice's memory is encoded here from its pixel features and predicted mask
    (`_encode_new_memory`, sam2_base.py:665-703), the current (unprompted) slice attends to it, and the mask loss of the current slice
    flows back through decoder -> memory attention -> memory tokens -> memory encoder.  Truncated where the reference's graph goes on
    into the previous slice's decoder: the previous mask and both slices' image features are treated as constants.
    curr / curr_pos [L, B, C]; prev_pix_tokens [B*L, C]; prev_mask_logits fp32 [B, 1, 16h, 16w]; memory_pos [L, B, 64] (position +
    temporal encoding of that bank entry, a constant).  prev_sam_token (fp32 [B, C], optional): the previous slice's SAM output token; its
    object pointer `obj_ptr_proj(token)` then joins the bank as C / 64 extra tokens without position encoding (sam2_base.py:591-635 with
    add_tpos_enc_to_obj_ptrs = False, object present) and the pointer projection is trained too.
    Returns (loss, {group: loss scale}, {group: {parameter: gradient}}) for the groups "decoder", "memory_attention", "memory_encoder"
    (+ "obj_ptr_proj")."""
    enc = model.memory_encoder
    # (sam2_base.py:686-688: masks from clicks are binarised only in eval mode; binarisation has no gradient anyway)
    mode = 2 if (model.binarize_mask_from_pts_for_mem_enc and prev_is_mask_from_pts and not model.training) else 1
    sc, bi = float(model.sigmoid_scale_for_mem_enc), float(model.sigmoid_bias_for_mem_enc)
    L, _, C = curr.shape
    mem = enc.run(prev_pix_tokens, prev_mask_logits, mode, sc, bi, B, h, w)                  # fp32 [B*L, 64]
    memory = mem.view(B, L, -1).transpose(0, 1)                                              # [L, B, 64] like the bank's flattened entries
    n_ptr = 0
    if prev_sam_token is not None:
        from .modeling.common import to_bf16
        md = memory.shape[2]
        n_ptr = C // md
        tok16 = to_bf16(prev_sam_token.reshape(B, C).contiguous())
        ptr = model.obj_ptr_proj.run(tok16)                                                  # fp32 [B, C]
        memory = torch.cat([memory, ptr.view(B, n_ptr, md).transpose(0, 1)], dim=0)           # (bank assembly: data movement)
        memory_pos = torch.cat([memory_pos, torch.zeros(n_ptr, B, md, dtype=memory_pos.dtype, device=memory_pos.device)], dim=0)
    aux: dict = {}
    loss, scale, scale_mem, g_dec, g_mem, _ = memory_decoder_loss_grads(
        model.memory_attention, model.sam_mask_decoder, curr, curr_pos, memory, memory_pos, n_ptr, pe_tokens, sparse, feat_s0, feat_s1, B, h, w,
        target_masks, dense_tokens=dense_tokens, pos_weight=pos_weight, mem_scale=mem_scale, aux=aux, mask_index=mask_index)
    dmem = aux["dmemory"]                                                                    # [L + n_ptr, B, 64], scaled by scale_mem
    d_mem = dmem[:L].transpose(0, 1).reshape(B * L, -1).contiguous()
    _, g_enc = bwd.memory_encoder_backward(enc, prev_pix_tokens, prev_mask_logits, mode, sc, bi, B, h, w, d_mem)
    scales = {"decoder": scale, "memory_attention": scale_mem, "memory_encoder": scale_mem}
    grads = {"decoder": g_dec, "memory_attention": g_mem, "memory_encoder": g_enc}
    if n_ptr:
        d_ptr = dmem[L:].transpose(0, 1).reshape(B, C).contiguous()
        g_ptr: dict = {}
        bwd.mlp_layers_backward(model.obj_ptr_proj, tok16, d_ptr, "p", g_ptr)
        scales["obj_ptr_proj"], grads["obj_ptr_proj"] = scale_mem, {k[2:]: v for k, v in g_ptr.items()}
    return loss, scales, grads


@torch.no_grad()
def memory_bank_finetune_step(model, optimizers: Dict[str, DecoderAdam], *args, sync: bool = True, **kwargs):
    """Adam step of the three groups on `memory_bank_loss_grads` (same arguments).  `optimizers` maps "decoder" / "memory_attention" /
    "memory_encoder" to a DecoderAdam over `model.sam_mask_decoder` / `model.memory_attention` / `model.memory_encoder` (train_3d.py:50-54
    runs the first at lr 1e-4 and the memory groups at 1e-8); groups without an optimiser are left alone."""
    holder = optimizers.get("memory_attention")
    cal = getattr(holder, "calibrated_loss_scales", {}) if holder is not None else {}
    if kwargs.get("mem_scale") is None:
        kwargs["mem_scale"] = cal.get(("bank", kwargs.get("mask_index")))
    loss, scales, grads = memory_bank_loss_grads(model, *args, **kwargs)
    if holder is not None:
        cal[("bank", kwargs.get("mask_index"))] = scales["memory_attention"] / scales["decoder"]
        holder.calibrated_loss_scales = cal
    for grp, opt in optimizers.items():
        opt.step(grads[grp], grad_scale=1.0 / scales[grp])
    return float(loss.item()) if sync else loss
