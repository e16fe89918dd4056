#!/usr/bin/env python3
"""Regenerates include/msam2_hip.h from the `extern "C"` definitions in medical-sam2_amd/csrc (prototypes are taken
verbatim from the sources; the per-entry documentation below names the reference interface each entry replaces)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "medical-sam2_amd", "csrc")

DOC = {
    "msam2_version": "Library version (major*10000 + minor*100 + patch).",
    "msam2_operand_is_fp16": "16-bit operand type of this build: 1 = IEEE fp16 (default), 0 = bf16 (built with -DMSAM2_OPERAND_BF16).  All\n`*_is_16bit` flags below select between that type and fp32.",
    "msam2_last_error": "Message of the last failing call on this thread.  Errors never cross the ABI as exceptions: every entry returns\n0 on success, <0 on failure (reference behaviour: AT_ASSERTM -> RuntimeError, connected_components.cu:215-228; the\nPython wrapper re-raises as RuntimeError).",
    "msam2_patch_embed7x7s4": "PatchEmbed.forward (backbones/utils.py:84-95: Conv2d(3, E, k7, s4, p3), NHWC out) plus the position table added at hieradet.py:283\n(x = x + self._get_pos_embed(x.shape[1:3])) as ONE kernel: fp32 image in, fp32 tokens out, no im2col map.  w_perm: the conv weight in the\nkernel's reduction order, 16-bit [ceil(E/32)*32, 176], k' = (c*7 + ky)*8 + 1 + kx with a zero tap in front of each run of 7; pos: fp32 [(S/4)^2, E] or NULL.\nNeeds (S/4) % 32 == 0 and E <= 128; other sizes use msam2_im2col_patch7x7s4 + msam2_gemm.",
    "msam2_gemm_pool2x2": "Hiera's pooled shortcut `do_pool(self.proj(x_norm), self.pool)` (hieradet.py:141-145, 23-34) as one GEMM: C (fp32)\n[B*(H/2)*(W/2), N] = maxpool2x2(A W^T + bias) over the [B,H,W] token image A; the un-pooled map is never written.",
    "msam2_gemm_qkv_pool2x2": "Fused qkv projection of a q-pooling Hiera block (hieradet.py:61-70 with do_pool, 23-34): k/v columns to QKV in image order,\nQ2 = maxpool2x2 of the q columns; the un-pooled q is neither written nor read back.",
    "msam2_gemm_tokens": "Token-side linear layers of the two-way decoder (transformer.py:165-196, 239-263): M <= 32 rows, A in fp32 from the residual\nstream, columns < add_cols computed from A + A2 (queries + query_pe), add and 16-bit conversion fused into the operand load.",
    "msam2_gemm_rope": "Linear projection with the axial RoPE of RoPEAttention fused into the store (transformer.py:241-243 + 299-315,\nposition_encoding.py:200-216): C (16-bit) = rope(A W^T + bias) on columns < rope_cols (whole heads, adjacent channel pairs) of rows\nwhose position l = m % rows_per_batch is < n_rope, with table row l % n_pos of cos/sin [n_pos, head_dim/2] (rope_k_repeat).",
    "msam2_gemm": "C[M,N] = residual[m % res_mod] + colscale[n] * act(A[M,K] W[N,K]^T + bias[n]); A, W 16-bit (K contiguous), bias/colscale\nfp32, residual/C 16-bit or fp32.  act: 0 none, 1 exact-erf GELU, 2 ReLU, 3 sigmoid.\nReplaces every nn.Linear / 1x1 Conv2d / im2col'ed conv of the path: hieradet.py:61,79,141; sam2_utils.py:127-131;\ntransformer.py:241-243,261; memory_attention.py:96; image_encoder.py:112; mask_decoder.py:240-256;\nmemory_encoder.py:103-105,171-175; sam2_base.py:470-475.",
    "msam2_ln_mlp_residual_supported": "1 when the trunk should take msam2_ln_mlp_residual_fwd at this width (96 / 192: Hiera stages 1 and 2; 384 only with MSAM2_MLP_384=1 -- built and\ncallable, but slower than its three launches: DESIGN 3.5).",
    "msam2_mlp_fused_permute_w2": "Kernel-ready copy of the second MLP weight for msam2_ln_mlp_residual_fwd: the hidden index of every 32-block permuted to the k order\nin which the fc1 accumulator is consumed as an MFMA operand (dim 96 / 192: [dim, hidden]; dim 384: chunk-major [hidden / 32][dim][32] in the\nkernel's LDS image -- opaque to the caller, same size).",
    "msam2_ln_mlp_residual_fwd_dual": "msam2_ln_mlp_residual_fwd with the result also written in the 16-bit operand type: the last block of a Hiera stage, whose output feeds the\nFPN's lateral 1x1 convolution (image_encoder.py:95-110) -- no cast pass over the stage-1 / stage-2 feature maps (100 MB + 50 MB at 4 x 1024^2).",
    "msam2_ln_mlp_residual_fwd": "The MLP half of MultiScaleBlock.forward as ONE kernel (hieradet.py:166-167: x = x + self.mlp(self.norm2(x)); sam2_utils.py:108-132 with\nnn.GELU): LayerNorm, fc1, exact-erf GELU, fc2 and the residual add; the 4x hidden activation never leaves the registers.  Block-level\nfused entry for the two high-resolution stages (dim 96 / 192), where the three separate launches are bound by the hidden map's HBM\nround trip.",
    "msam2_layernorm_dual": "nn.LayerNorm on fp32 rows with two outputs from one pass: the fp32 rows (residual stream) and their 16-bit copy (operand of the next\nprojection) -- norm4 of the two-way block (transformer.py:190-196), whose output `keys` is both.",
    "msam2_layernorm": "Row LayerNorm (fp32 statistics) on [rows, C], optional GELU: nn.LayerNorm at hieradet.py:138,166,\nmemory_attention.py:60,73,94,162, transformer.py:173-194,116; LayerNorm2d (sam2_utils.py:137-149) on NHWC tokens.",
    "msam2_attention_workspace_bytes": "Scratch needed by msam2_attention_fwd when splits > 1 (fp32 partial O, running max, partial sum).",
    "msam2_attention_merge": "Second half of a split-KV attention call issued with a NEGATIVE split count (the split pass alone, partials left in the\nworkspace): combines the per-split (max, sum, O) triples into o.  Lets a host time / overlap the two kernels separately.",
    "msam2_attention_fwd": "softmax(Q K^T * scale) V, non-causal, 16-bit in/out, fp32 softmax/accumulate; head dim 64/96/128/256; q/k/v/o given by\nelement strides {batch, head, token}.  splits > 1 = split-KV (flash-decoding) with an in-library merge; splits < 0 = the split pass only\n(finish with msam2_attention_merge).\nReplaces F.scaled_dot_product_attention at hieradet.py:72-76 (global blocks) and transformer.py:318 (RoPEAttention,\nmemory attention self/cross).",
    "msam2_attention_kv64_fwd": "Memory cross-attention of RoPEAttention with kv_in_dim = 64 (transformer.py:288-331 as called at memory_attention.py:76-85) with\nthe value product contracted in the 64-channel memory space: O' = softmax(Q K^T * scale) M for 256-wide rotated q / k rows and the\n64-wide memory rows M.  Because the values carry no rotary encoding and softmax rows sum to one, P (M W_v^T + b_v) = O' W_v^T + b_v:\nthe caller folds v_proj into out_proj (one K = 64 GEMM).  Strides / splits / workspace / merge as msam2_attention_fwd with D = 64.",
    "msam2_attention_kv64_dyn_fwd": "msam2_attention_kv64_fwd with the key count read on the device: Lk is the CAPACITY (buffers, split count, workspace), *key_count_dev\n(int32, 1 .. Lk) the number of memory tokens attended to.  One launch shape serves every fill level of a padded memory bank, so the\nper-slice forward of the 3-D propagation (sam2_video_predictor.py:1302-1367 -> sam2_base.py:494-663, whose bank grows by one object\npointer per slice) can be captured into one hipGraph per bank bucket and replayed.",
    "msam2_attention_kv64_dyn_partial": "msam2_attention_kv64_partial with the device-side key count of msam2_attention_kv64_dyn_fwd (cross-GPU key split under a hipGraph).",
    "msam2_attention_effective_splits": "The split count msam2_attention_fwd / msam2_attention_kv64_fwd actually run with for a requested one (every split owns at least one\n32-key tile); what msam2_attention_kv64_partial expects as `splits`.",
    "msam2_attention_kv64_partial": "Splits [split_begin, split_begin + split_count) of a `splits`-way msam2_attention_kv64_fwd: the partial (max, sum, O') triples land in\nthe workspace slots the full call would use, nothing is merged.  Cross-GPU key split of the 3-D propagation chain (the reference has no\ncounterpart: sam2_base.py:494-663 attends to the whole bank on one device): every rank computes its share of the splits, the slots are\nall-gathered, msam2_attention_merge (D = 64) finishes -- bit-identical to one rank computing all splits.",
    "msam2_window_attention_fwd": "Windowed Hiera attention straight from un-partitioned qkv tokens: replaces window_partition -> SDPA ->\nwindow_unpartition (backbones/utils.py:16-62 + hieradet.py:138-158,72-76).  Zero-padded window tokens are unmasked keys\nwhose K/V rows are kpad/vpad (= qkv bias), exactly what the reference computes; q may come from a 2x2 max-pooled image\n(q-pool at stage changes, hieradet.py:65-69).",
    "msam2_attention_small_fwd": "Attention with head dim 16/32 (two-way decoder: transformer.py:239-263 via 165-196, 74-118): tokens->image,\nimage->tokens and token self-attention.  q/k/v/o: 16-bit [B, L, heads*D].",
    "msam2_add_cast": "out = a + alpha * b on a logical [D0,D1,C] volume with arbitrary outer strides (0 = broadcast) and dtype conversion:\nmemory_attention.py:139-147 (+0.1*pos, seq-first -> batch-first), 74-76 (memory + pos), transformer.py:175-190 (q + pe,\nk + pe), mask_decoder.py:231 (src + dense), sam2_base.py:642 (+ no_mem_embed), 571-580,626-635 (memory-bank assembly).",
    "msam2_maxpool2x2": "MaxPool2d(2,2) on NHWC tokens (do_pool, hieradet.py:23-34: q-pool and pooled shortcut).",
    "msam2_upsample2x_add": "FPN top-down step y += nearest2x(top) (image_encoder.py:113-124).",
    "msam2_rope_table": "cos/sin of compute_axial_cis (position_encoding.py:174-183) for a side x side grid.",
    "msam2_rope_inplace": "apply_rotary_enc (position_encoding.py:194-216) in place on 16-bit rows; rows >= n_rope of each batch are left\nuntouched (num_k_exclude_rope, transformer.py:308-315); positions wrap modulo n_pos (rope_k_repeat).",
    "msam2_bilinear_upsample": "F.interpolate(mode=\"bilinear\", align_corners=False) on fp32 planes (sam2_base.py:367-373).",
    "msam2_sine_pos_2d": "PositionEmbeddingSine.forward (position_encoding.py:78-112) as a token-major [h*w, C] table.",
    "msam2_fourier_pe_grid": "PromptEncoder.get_dense_pe (prompt_encoder.py:68-77; position_encoding.py:130-151) as [h*w, C].",
    "msam2_hiera_pos_embed": "Hiera._get_pos_embed (hieradet.py:269-277): bicubic resize of pos_embed + tiled pos_embed_window.",
    "msam2_aa_downsample": "F.interpolate(mode=\"bilinear\", antialias=True) by an integer factor (sam2_base.py:321-327, 421-427).",
    "msam2_transpose16": "16-bit matrix transpose (backward GEMMs: dX = dY W needs W^T rows, dW = dY^T X needs dY^T and X^T as K-contiguous operands\nof msam2_gemm).",
    "msam2_colsum": "Column sums into a zeroed fp32 vector: the bias gradient of nn.Linear (sam2_utils.py:127-131 under torch.autograd).",
    "msam2_act_bwd": "dpre = dy * act'(pre) as a 16-bit GEMM operand; act 1 = exact-erf GELU (hieradet.py:96, memory_encoder.py:95), 2 = ReLU\n(memory_attention.py:96, transformer.py MLP blocks).",
    "msam2_layernorm_bwd": "nn.LayerNorm backward (hieradet.py:101-102, memory_attention.py:43-45): dx, and dgamma / dbeta accumulated into zeroed fp32\nvectors; statistics are recomputed from x, the forward saves nothing.",
    "msam2_softmax_rows": "P (16-bit) = softmax(scale * S) row-wise from materialised fp32 scores: forward half of the materialised attention backward\n(F.scaled_dot_product_attention under autograd, transformer.py:318, hieradet.py:72-76).",
    "msam2_softmax_bwd_rows": "dS (16-bit) = scale * P * (dP - sum_k P dP) row-wise: the softmax Jacobian of the attention backward.",
    "msam2_convt2x2_gather": "Training-forward tail of ConvTranspose2d(k2,s2) (mask_decoder.py:244-247) without the fused norm / activation: z = shuffle(gemm) + bias\n+ skip in fp32 (the inference path's msam2_convt2x2_shuffle fuses LayerNorm2d + GELU and keeps nothing).",
    "msam2_convt2x2_scatter_grad": "Adjoint of the 2x2 pixel shuffle: the gradient of the ConvTranspose GEMM output as a 16-bit operand.",
    "msam2_bce_logits": "BCEWithLogitsLoss(pos_weight) value (accumulated into a zeroed scalar) and its gradient w.r.t. the logits, mean reduction\n(func_3d/function.py:69 criterion_G).",
    "msam2_attention_fwd_lse": "msam2_attention_fwd that also writes the log-sum-exp of every query row (log2 domain, fp32 [B, H, Lq]) for msam2_attention_bwd.",
    "msam2_attention_fwd_lse_dropout": "msam2_attention_fwd_lse with dropout on the attention probabilities INSIDE the flash kernel (F.scaled_dot_product_attention(dropout_p)\nas RoPEAttention calls it in train mode, transformer.py:317-318): counter-based mask, element offset + ((b*H + h)*Lq + q)*Lk + k of\nstream seed (+ *seed_dev, optional, see msam2_counter_bump); the softmax denominator and lse are those of the un-dropped probabilities.\nHead dim 96 / 128 / 256, Lq > 64.",
    "msam2_attention_bwd_dropout": "msam2_attention_bwd for a forward run by msam2_attention_fwd_lse_dropout with the same (p, seed, offset, seed_dev): every pass re-creates\nthe mask; O(L) memory -- the train-mode attention backward of the memory attention no longer materialises [Lq, Lk] tensors.",
    "msam2_attention_bwd_workspace_bytes": "Scratch needed by msam2_attention_bwd (16-bit copy of dO and the delta rows).",
    "msam2_attention_bwd": "Flash-style dQ / dK / dV of softmax(Q K^T * scale) V (torch.autograd of F.scaled_dot_product_attention at transformer.py:318 and\nhieradet.py:72-76 in the training loops func_3d/function.py:182-191, func_2d/function.py:246-259): head dim 64 / 96 / 128 / 256,\n16-bit q / k / v / o and lse (the outputs of msam2_attention_fwd_lse), fp32 dO in, fp32 gradients out, O(L) memory (no [Lq, Lk] tensor).",
    "msam2_dwconv7x7": "Plain depthwise 7x7 convolution (pad 3) on fp32 NHWC tokens, taps [49, C]; flip = 1 gives the input gradient of CXBlock.dwconv\n(memory_encoder.py:83-90) -- the forward uses the fused msam2_dwconv7x7_ln.",
    "msam2_dwconv7x7_wgrad": "Weight gradient of CXBlock.dwconv accumulated into a zeroed fp32 [49, C] buffer.",
    "msam2_col2im3x3s2": "Adjoint of msam2_im2col3x3s2: input gradient of the mask down-sampler's k3 s2 p1 convolutions (memory_encoder.py:38-47) from the\ncolumn gradient of their GEMM form.",
    "msam2_gemm_nt": "Input-gradient GEMM C[M,N] = residual + A[M,K] B[K,N] (+ bias) with B k-major: dX = dY W of nn.Linear under autograd (sam2_utils.py:127-131,\nmemory_attention.py:96, transformer.py:241-261, hieradet.py:61,79) with W [out, in] exactly as the forward stores it -- no transposed weight\ncopy.  K % 64 == 0 (the layer's out_features), N % 8 == 0.",
    "msam2_window_unpartition_cvt": "window_unpartition (backbones/utils.py:41-62) of fp32 windows [B*nW, heads, ws*ws, D] into a 16-bit token image in one pass: the\nwindowed attention's dq / dk / dv (fp32, window order) become the 16-bit operand of the fused-qkv gradient GEMMs.",
    "msam2_window_pad_colsum": "Sum of fp32 window rows [B*nW, heads, ws*ws, D] over the zero-padded window tokens (outside the [H, W] image), added into out [heads*D]:\nthe part of the windowed attention's dk / dv that belongs to the qkv bias (backbones/utils.py:28-31 pads AFTER the LayerNorm, so padded\ntokens carry k = v = bias).",
    "msam2_gemm_tt_acc": "msam2_gemm_tt that adds into C and a_colsum instead of overwriting them (torch.autograd's .grad accumulation; or outputs carved from a\nbuffer zeroed once per backward pass).",
    "msam2_gemm_tt": "Weight-gradient GEMM C[M,N] (fp32) = sum_k A[k][m] B[k][n] on k-major 16-bit operands: dW = dY^T X of nn.Linear under autograd\n(sam2_utils.py:127-131, memory_attention.py:96, transformer.py:241-261) straight from the token-major dY and X -- no transposed copies;\nthe token reduction is split over workgroups (fp32 atomics into the zeroed output).  a_colsum (optional, [M]) receives sum_k A[k][m]:\nthe bias gradient in the same pass over dY.",
    "msam2_bilinear_upsample_bwd": "Adjoint of msam2_bilinear_upsample: gradient of the video-resolution mask logits (sam2_video_predictor.py:724-744, the tensor the\ntraining loss of func_3d/function.py:137-170 is taken on) back to the decoder's low-resolution logits.",
    "msam2_maxpool2x2_bwd": "Backward of MaxPool2d(2, 2) on token-major maps (do_pool, hieradet.py:23-34, under autograd in the 2-D training loop,\nfunc_2d/function.py:70-72): dy is routed to the first maximum of each 2x2 window, every dx element is written.",
    "msam2_window_move": "window_partition (to_windows = 1) / window_unpartition (0) of backbones/utils.py:16-62 on a projected token image, in 16-byte chunks:\nimg [B, H, W, heads * D] (row stride ld_img elements) <-> win [B * nW, heads, ws * ws, D] contiguous; padded window tokens take `fill`\n[heads * D] (or 0) on the way in and are cropped on the way out.  The operands of the attention backward of the Hiera trunk under\nautograd (hieradet.py:138-158 in the 2-D training loop, func_2d/function.py:70-72); the forward's window kernel gathers by itself.",
    "msam2_sumpool2x2": "Adjoint of msam2_upsample2x_add (FPN nearest-2x top-down step, image_encoder.py:113-124): sums of the 2x2 blocks.",
    "msam2_hiera_pos_embed_bwd": "Adjoint of msam2_hiera_pos_embed (hieradet.py:269-277): gradient of the position-token table -> d pos_embed (transposed bicubic\nresize) and d pos_embed_window (sum over the tiling).  Two gather passes through a caller-owned workspace, no atomics.",
    "msam2_hiera_pos_embed_bwd_workspace_bytes": "Scratch needed by msam2_hiera_pos_embed_bwd (per-row partial sums).",
    "msam2_dropout": "Train-mode nn.Dropout / SDPA dropout_p (memory_attention.py:40-48,63,80,97-98; transformer.py:317-318): y = keep ? x / (1 - p) : 0\n(+ fp32 residual) with a counter-based mask -- element i of stream (seed, offset) -- so the backward re-creates the forward's mask by\ncalling it on the gradient with the same (seed, offset).  seed_dev (optional device uint64): added to `seed` on the device, see\nmsam2_counter_bump.",
    "msam2_counter_bump": "*counter += 1 on the device (uint64), the new value copied to *snapshot (optional): the per-forward sub-stream counter of the\ntrain-mode dropout (MemoryAttention in train(), memory_attention.py:40-48), advanced by a kernel of the step itself so that a hipGraph\nreplay of a training step draws fresh masks -- pass the snapshot as msam2_dropout's seed_dev.",
    "msam2_adam_step": "One torch.optim.Adam step (no weight decay / amsgrad) on a flat fp32 parameter (train_3d.py:50).",
    "msam2_adam_step_multi": "The same Adam step over `count` parameters (host arrays of device pointers and element counts), 24 per launch;\ngradients are multiplied by grad_scale first (1 / loss scale);\nstep_counter (device int32, optional): the step count lives on the device and is incremented by the call (a kernel), so a captured\nhipGraph advances the bias corrections on every replay; non-finite gradient entries are skipped (and counted into skipped_counter, device int32, optional);\nweight_decay > 0 gives torch.optim.AdamW's decoupled decay (train_2d.py:43-47), 0 plain Adam (train_3d.py:50-54).",
    "msam2_attention_small_bwd": "Backward of the two-way decoder's attention (transformer.py:239-263 under autograd; 8 heads of 16 / 32 channels) when one side has\n<= 32 tokens: dq / dk / dv (fp32, token-major) from 16-bit q / k / v and the fp32 upstream gradient, one workgroup per (batch, head).",
    "msam2_seg_counts": "Counts behind eval_seg (func_3d/utils.py:139-214, func_2d/utils.py:505-580): per threshold, batch element and class the\ninteger |pred>t & gt>t|, |pred>t|, |gt>t| in one pass; IoU / Dice follow on the host.",
    "msam2_non_overlap": "SAM2Base._apply_non_overlapping_constraints (sam2_base.py:812-830): keep the arg-max object per pixel, clamp the\nothers to <= -10.",
    "msam2_gate_rows": "masks[b] = value where object score <= 0 (NO_OBJ_SCORE fill, sam2_base.py:354-363).",
    "msam2_any_positive": "is_obj_appearing = any(mask > 0) per object (sam2_base.py:445-447).",
    "msam2_im2col_patch7x7s4": "PatchEmbed Conv2d(3,E,k7,s4,p3) (backbones/utils.py:84-95) lowered to im2col (+ msam2_gemm_bf16).",
    "msam2_im2col3x3s2": "im2col of the 64->256 k3/s2/p1 mask down-sampler conv (memory_encoder.py:41-49).",
    "msam2_conv3x3s2_ln_gelu": "One MaskDownSampler stage: Conv2d(k3,s2,p1) + LayerNorm2d + GELU (memory_encoder.py:37-54), with the scaled\nsigmoid / binarisation of the mask logits (sam2_base.py:686-696) fused into the first stage.",
    "msam2_dwconv7x7_ln": "CXBlock head: depth-wise 7x7 conv + LayerNorm2d (memory_encoder.py:99-101).",
    "msam2_convt2x2_shuffle_f32skip": "msam2_convt2x2_shuffle with the high-resolution skip features (mask_decoder.py:244-247: feat_s1 / feat_s0) read in fp32, the type the FPN\nreturns them in (C = 32 / 64): no 16-bit copy of the two largest feature maps in front of the decoder.",
    "msam2_convt2x2_shuffle": "ConvTranspose2d(k2,s2) tail of the mask decoder up-scaling: pixel shuffle of the GEMM output + bias + high-res\nskip feature, then LayerNorm2d + GELU or GELU (mask_decoder.py:244-247).",
    "msam2_token_mlp3": "The mask decoder's token heads in one launch (mask_decoder.py:249-266; MLP = sam2_utils.py:108-132): G independent\n3-layer ReLU MLPs of width 256 (4 hyper-networks, IoU head with sigmoid, object-score head), each on one token of every batch element.",
    "msam2_hyper_masks": "masks = hyper_in @ upscaled_embedding (mask_decoder.py:249-256).",
    "msam2_token_mlp3_packed": "msam2_token_mlp3 with a packed output: head g writes out[out_offset[g] + b * out_stride[g] + o] for o < out_dim[g], so that the hyper-network\nvectors, the IoU predictions and the object score land in contiguous tensors of their own (mask_decoder.py:249-266 slices them apart).",
    "msam2_prompt_points": "Point / box-corner prompt embeddings (prompt_encoder.py:79-114; position_encoding.py:153-158).",
    "msam2_prompt_points_padded": "msam2_prompt_points with the padding point of prompt_encoder.py:87-91 ((0, 0), label -1: points without a box) appended to every prompt\nset inside the kernel: xy [n_sets, P, 2], labels [n_sets, P] -> out [n_sets, P + n_pad, C] (replaces torch.zeros / torch.ones + two torch.cat).",
    "msam2_select_mask": "Mask selection without a host round trip: best-IoU multimask or dynamic multimask via stability\n(mask_decoder.py:147-168,269-317) and object-score gating (sam2_base.py:354-385).",
    "msam2_gather_rows": "Pick the SAM output token of the selected mask (sam2_base.py:375-383).",
    "msam2_obj_ptr_mix": "obj_ptr = lam*obj_ptr + (1-lam)*no_obj_ptr with the hard lam of fixed_no_obj_ptr (sam2_base.py:389-400).",
    "msam2_space_to_depth": "Non-overlapping k x k patches for the k2/s2 mask_downscaling convs (prompt_encoder.py:58-66) and the k4/s4\nmask_downsample conv (sam2_base.py:108,439).",
    "msam2_cc_workspace_bytes": "Scratch (union-find parents + area histogram) for msam2_cc_label.",
    "msam2_cc_label": "Drop-in for the reference's only native op, `_C.get_connected_componnets` (sam2_train/csrc/connected_components.cu:\n213-282; Python wrapper utils/misc.py:47-63): 8-connected labels (1 + smallest 2x2-block corner index of the component)\nand per-pixel component areas for uint8 masks [N,1,H,W], H and W even.  The caller allocates labels/counts/workspace.",
    "msam2_fill_holes_workspace_bytes": "Scratch for msam2_fill_holes.",
    "msam2_fill_components": "Small-component filling on mask scores: generalisation of msam2_fill_holes used by SAM2Transforms.postprocess_masks\n(utils/transforms.py:74-98).",
    "msam2_image_prep": "SAM2Transforms.__call__ (utils/transforms.py:22-37): uint8 HWC -> /255 -> bilinear resize -> normalise -> fp32 CHW.\nmean3/std3 are HOST pointers to 3 floats.",
    "msam2_fill_holes": "fill_holes_in_mask_scores (utils/misc.py:247-258): background components of area <= max_area get score 0.1.",
    "msam2_graph_begin": "hipGraph capture of everything enqueued on `stream` until msam2_graph_end (the per-slice forward is launch-bound in\nthe reference: ~750 dependent ATen kernels per slice, SURVEY.md section 0.9).",
    "msam2_graph_end": "Ends the capture and instantiates the executable graph.",
    "msam2_graph_launch": "Replays a captured graph on `stream`.",
    "msam2_graph_destroy": "Releases a graph.",
    "msam2_event_create": "HIP event helpers so callers can time kernels on the launch stream itself.",
    "msam2_event_record": None, "msam2_event_elapsed_ms": None, "msam2_event_destroy": None,
}


def main():
    decls = []
    for f in ["api.hip", "gemm.hip", "attention.hip", "attention_bwd.hip", "elementwise.hip", "conv.hip", "cc.hip", "backward.hip", "mlp_fused.hip"]:
        s = open(os.path.join(CSRC, f)).read()
        for m in re.finditer(r'extern "C" ([^{;]+?)\s*\{', s, re.S):
            decls.append(" ".join(m.group(1).split()))
    out = ['/* msam2_hip.h -- C ABI of libmsam2_hip.so: the MI355X (gfx950) hot path of Medical-SAM2.',
           ' *',
           ' * GENERATED by tools/gen_header.py from the extern "C" definitions in the .hip sources under medical-sam2_amd/csrc.',
           ' *',
           ' * Conventions (SURVEY.md section 8(b)): plain pointers and sizes only -- no torch / ATen types.  Every pointer is a',
           ' * device pointer unless stated; the CALLER owns all memory (inputs, outputs, workspaces): the library never allocates or',
           ' * frees device memory and keeps no mutable global state besides lazily loaded code objects.  All work is enqueued on the',
           ' * `stream` argument (a hipStream_t passed as void*, 0 = default stream) without host synchronisation, so every entry',
           ' * is hipGraph-capturable and re-entrant.  Return value: 0 = ok, <0 = error (see msam2_last_error).',
           ' * Tensors: "16-bit" = the operand type of the build (IEEE fp16 by default, bf16 with -DMSAM2_OPERAND_BF16, see',
           ' * msam2_operand_is_fp16), "fp32" = IEEE float; *_is_16bit flags select between the two.',
           ' * file:line citations refer to the reference tree (1275468127/Medical-SAM2 @ 2024_10_08).',
           ' */',
           '#ifndef MSAM2_HIP_H', '#define MSAM2_HIP_H', '', '#include <stddef.h>', '#include <stdint.h>', '',
           '#ifdef __cplusplus', 'extern "C" {', '#endif', '']
    for d in decls:
        name = re.search(r"(msam2_\w+)\(", d).group(1)
        doc = DOC.get(name, "")
        if doc:
            out.append("/* " + doc.replace("\n", "\n * ") + " */")
        elif name not in DOC:
            raise SystemExit(f"undocumented entry point {name}")
        out.append(d + ";")
        out.append("")
    out += ['#ifdef __cplusplus', '}', '#endif', '#endif /* MSAM2_HIP_H */', '']
    with open(os.path.join(ROOT, "include", "msam2_hip.h"), "w") as f:
        f.write("\n".join(out))
    print(f"wrote include/msam2_hip.h with {len(decls)} entry points")


if __name__ == "__main__":
    main()
