"""Drop-in SAM prompt encoder, two-way transformer and mask decoder (sam2_train/modeling/sam/{prompt_encoder,transformer,
mask_decoder}.py) on the MI355X kernels.  Token-side tensors are a handful of rows; image-side tensors are token-major
[B*h*w, C] (fp32 residual stream + bf16 operand copies)."""
from __future__ import annotations

from typing import List, Optional, Tuple, Type

import torch
import torch.nn as nn

from .. import ops
from .common import OP16, F32, WeightCache, nchw_view, to_bf16, tokens_of, v_f32, w_bf16
from .encoder import MLP
from .memory import Attention, LayerNorm2d
from .position import PositionEmbeddingRandom


class PromptEncoder(nn.Module):
    """prompt_encoder.py:15-190.  Dense embeddings are returned at `image_embedding_size` (the upstream behaviour that the
    fork's hard-coded 16x16 resize at 189-190 reproduces only for 256-pixel inputs; SURVEY.md shim S2)."""

    def __init__(self, embed_dim: int, image_embedding_size: Tuple[int, int], input_image_size: Tuple[int, int],
                 mask_in_chans: int, activation: Type[nn.Module] = nn.GELU) -> None:
        super().__init__()
        self.embed_dim = embed_dim
        self.input_image_size = input_image_size
        self.image_embedding_size = image_embedding_size
        self.pe_layer = PositionEmbeddingRandom(embed_dim // 2)
        self.num_point_embeddings: int = 4
        self.point_embeddings = nn.ModuleList([nn.Embedding(1, embed_dim) for _ in range(self.num_point_embeddings)])
        self.not_a_point_embed = nn.Embedding(1, embed_dim)
        self.mask_input_size = (4 * image_embedding_size[0], 4 * image_embedding_size[1])
        self.mask_downscaling = nn.Sequential(
            nn.Conv2d(1, mask_in_chans // 4, kernel_size=2, stride=2), LayerNorm2d(mask_in_chans // 4), activation(),
            nn.Conv2d(mask_in_chans // 4, mask_in_chans, kernel_size=2, stride=2), LayerNorm2d(mask_in_chans), activation(),
            nn.Conv2d(mask_in_chans, embed_dim, kernel_size=1))
        self.no_mask_embed = nn.Embedding(1, embed_dim)
        self._wc = WeightCache()

    def dense_pe_tokens(self) -> torch.Tensor:
        g = self.pe_layer.positional_encoding_gaussian_matrix
        return self._wc.get("dense_pe", [g], lambda: self.pe_layer.grid_tokens(self.image_embedding_size))

    def get_dense_pe(self) -> torch.Tensor:
        h, w = self.image_embedding_size
        return nchw_view(self.dense_pe_tokens(), 1, h, w)

    def _points(self, xy: torch.Tensor, labels: torch.Tensor, n_pad: int = 0) -> torch.Tensor:
        assert self.input_image_size[0] == self.input_image_size[1]
        wc = self._wc
        emb = wc.get("pemb", [m.weight for m in self.point_embeddings],
                     lambda: torch.cat([m.weight.detach() for m in self.point_embeddings], 0).float().contiguous())
        return ops.prompt_points(xy.to(F32), labels.to(torch.int32), self.pe_layer.positional_encoding_gaussian_matrix.to(F32), emb,
                                 v_f32(wc, "nap", self.not_a_point_embed.weight), float(self.input_image_size[0]), n_pad=n_pad)

    def _embed_masks_tokens(self, masks: torch.Tensor) -> torch.Tensor:
        """mask_downscaling (two k2/s2 convs + LN + GELU, then 1x1) -> fp32 tokens [n*h*w, embed_dim]."""
        n, _, H, W = masks.shape
        wc, md = self._wc, self.mask_downscaling
        f = lambda key, p: v_f32(wc, key, p)

        def patch_w(conv, key):
            def build():
                w = conv.weight.detach().permute(0, 2, 3, 1).reshape(conv.weight.shape[0], -1)
                ld = (w.shape[1] + 7) // 8 * 8
                out = torch.zeros(w.shape[0], ld, dtype=OP16, device=w.device)
                out[:, : w.shape[1]] = w.to(OP16)
                return out
            return wc.get(key, [conv.weight], build)

        h = ops.space_to_depth(masks.to(F32).contiguous().reshape(n * H * W, 1), n, H, W, 2)
        h = ops.gemm(h, patch_w(md[0], "w0"), f("b0", md[0].bias), out_dtype=F32)
        h = ops.layernorm(h, f("lw0", md[1].weight), f("lb0", md[1].bias), md[1].eps, act=ops.ACT_GELU)
        h = ops.space_to_depth(h, n, H // 2, W // 2, 2)
        h = ops.gemm(h, patch_w(md[3], "w1"), f("b1", md[3].bias), out_dtype=F32)
        h = ops.layernorm(h, f("lw1", md[4].weight), f("lb1", md[4].bias), md[4].eps, act=ops.ACT_GELU)
        return ops.gemm(h, w_bf16(wc, "w2", md[6].weight), f("b2", md[6].bias), out_dtype=F32)

    def _get_batch_size(self, points, boxes, masks) -> int:
        if points is not None:
            return points[0].shape[0]
        if boxes is not None:
            return boxes.shape[0]
        if masks is not None:
            return masks.shape[0]
        return 1

    def forward(self, points: Optional[Tuple[torch.Tensor, torch.Tensor]], boxes: Optional[torch.Tensor],
                masks: Optional[torch.Tensor], batch_size=-1) -> Tuple[torch.Tensor, torch.Tensor]:
        bs = self._get_batch_size(points, boxes, masks)
        dev = self.no_mask_embed.weight.device
        xy_parts, lab_parts = [], []
        n_pad = 0
        if points is not None:
            coords, labels = points
            xy_parts.append(coords.to(F32))
            lab_parts.append(labels.to(torch.int32))
            if boxes is None:  # padding point, label -1 (prompt_encoder.py:87-91): appended inside the kernel
                n_pad = 1
        if boxes is not None:  # box corners are points with labels 2 / 3 (prompt_encoder.py:103-112)
            xy_parts.append(boxes.to(F32).reshape(-1, 2, 2))
            lab_parts.append(torch.tensor([[2, 3]], dtype=torch.int32, device=dev).expand(bs, 2))
        if len(xy_parts) == 1:
            sparse = self._points(xy_parts[0].contiguous(), lab_parts[0].contiguous(), n_pad)
        elif xy_parts:
            sparse = self._points(torch.cat(xy_parts, 1).contiguous(), torch.cat(lab_parts, 1).contiguous())
        else:
            sparse = torch.empty((bs, 0, self.embed_dim), device=dev)
        h, w = self.image_embedding_size
        if masks is not None:
            dense = nchw_view(self._embed_masks_tokens(masks), bs, h, w)
        else:
            dense = self.no_mask_embed.weight.reshape(1, -1, 1, 1).expand(bs, -1, h, w)
        return sparse, dense


def _image_side_projections(wc: WeightCache, key: str, keys_b: torch.Tensor, key_pe: torch.Tensor, L: int, lins, with_pe):
    """All projections a two-way block takes of the IMAGE tokens as one GEMM.  lins: the nn.Linear modules, with_pe: which of them see
    keys + key_pe (the others see keys).  (keys + pe) W^T = keys W^T + pe W^T, and pe is the batch-shared, input-independent dense
    position encoding: pe W^T is a cached [L, N] constant added as a row-modulo residual in the GEMM's store -- one launch instead of one
    per projection, no `keys + pe` pass over the tokens (transformer.py:165-196 calls three separate nn.Linear on keys / keys + pe).
    Returns the 16-bit [B*L, sum(out_i)] result (column blocks in the order of `lins`)."""
    ws = [l.weight for l in lins]
    W = w_bf16(wc, key + "w", *ws)
    b = v_f32(wc, key + "b", *[l.bias for l in lins])

    def build_pe():
        R = ops.gemm(to_bf16(key_pe.contiguous()), W, None, out_dtype=F32)            # [L, N] fp32
        col = 0
        for l, use in zip(lins, with_pe):
            if not use:
                R[:, col: col + l.out_features] = 0.0
            col += l.out_features
        return R
    R = wc.get(key + "pe", ws + [key_pe], build_pe)
    return ops.gemm(keys_b, W, b, residual=R, res_mod=L)


class TwoWayAttentionBlock(nn.Module):
    """transformer.py:121-196."""

    def __init__(self, embedding_dim: int, num_heads: int, mlp_dim: int = 2048, activation: Type[nn.Module] = nn.ReLU,
                 attention_downsample_rate: int = 2, skip_first_layer_pe: bool = False) -> None:
        super().__init__()
        self.self_attn = Attention(embedding_dim, num_heads)
        self.norm1 = nn.LayerNorm(embedding_dim)
        self.cross_attn_token_to_image = Attention(embedding_dim, num_heads, downsample_rate=attention_downsample_rate)
        self.norm2 = nn.LayerNorm(embedding_dim)
        self.mlp = MLP(embedding_dim, mlp_dim, embedding_dim, num_layers=2, activation=activation)
        self.norm3 = nn.LayerNorm(embedding_dim)
        self.norm4 = nn.LayerNorm(embedding_dim)
        self.cross_attn_image_to_token = Attention(embedding_dim, num_heads, downsample_rate=attention_downsample_rate)
        self.skip_first_layer_pe = skip_first_layer_pe
        self._wc = WeightCache()

    def _ln(self, name, x):
        n = getattr(self, name)
        return ops.layernorm(x, v_f32(self._wc, name + "w", n.weight), v_f32(self._wc, name + "b", n.bias), n.eps, out_dtype=F32)

    def run(self, queries, keys, query_pe, key_pe, B, T, L, keys_b=None):
        """queries/query_pe fp32 [B*T, C]; keys fp32 [B*L, C]; key_pe fp32 [L, C] (shared over the batch); keys_b: the 16-bit copy of keys when
        the caller has it (the previous block's norm4 writes both).  Returns (queries, keys, 16-bit keys)."""
        C = queries.shape[1]
        q3 = lambda t, n: t.view(B, n, -1)
        sa, ca, ia, wc = self.self_attn, self.cross_attn_token_to_image, self.cross_attn_image_to_token, self._wc
        # token-side projections read the fp32 residual stream directly (ops.gemm_tokens: "+ query_pe" and the 16-bit conversion
        # happen in the operand load; q | k | v of one attention are one launch)
        fused = B * T <= 32 and C % 32 == 0 and sa.internal_dim % 32 == 0 and ia.internal_dim % 32 == 0
        if fused:
            Ci = sa.internal_dim
            qkv = ops.gemm_tokens(queries, w_bf16(wc, "sqkv", sa.q_proj.weight, sa.k_proj.weight, sa.v_proj.weight),
                                  v_f32(wc, "sqkvb", sa.q_proj.bias, sa.k_proj.bias, sa.v_proj.bias),
                                  addend=None if self.skip_first_layer_pe else query_pe, add_cols=2 * Ci).view(B, T, 3 * Ci)
            o = sa.core(qkv[:, :, :Ci], qkv[:, :, Ci:2 * Ci], qkv[:, :, 2 * Ci:])
            queries = sa.out(o, None if self.skip_first_layer_pe else queries)
        elif self.skip_first_layer_pe:
            qb = to_bf16(queries)
            queries = sa.out(sa.core(q3(sa.proj("q", qb), T), q3(sa.proj("k", qb), T), q3(sa.proj("v", qb), T)), None)
        else:
            qb = ops.add_cast(queries.view(1, B * T, C), query_pe.view(1, B * T, C), 1.0, OP16)[0]
            queries = sa.out(sa.core(q3(sa.proj("q", qb), T), q3(sa.proj("k", qb), T), q3(sa.proj("v", to_bf16(queries)), T)), queries)
        queries = self._ln("norm1", queries)
        # tokens -> image.  The three image-side projections of the block (k, v of this attention and q of the image -> token attention
        # further down: `keys` does not change in between) are one GEMM
        if keys_b is None:
            keys_b = to_bf16(keys)
        Cc, Ci2 = ca.internal_dim, ia.internal_dim
        kvq = _image_side_projections(wc, "ikvq", keys_b, key_pe, L, [ca.k_proj, ca.v_proj, ia.q_proj], [True, False, True])
        k_img, v_img, q_img = kvq[:, :Cc], kvq[:, Cc:2 * Cc], kvq[:, 2 * Cc:2 * Cc + Ci2]
        if fused:
            qp = ops.gemm_tokens(queries, w_bf16(wc, "cqw", ca.q_proj.weight), v_f32(wc, "cqb", ca.q_proj.bias), addend=query_pe,
                                 add_cols=ca.internal_dim)
        else:
            qp = ca.proj("q", ops.add_cast(queries.view(1, B * T, C), query_pe.view(1, B * T, C), 1.0, OP16)[0])
        o = ca.core(q3(qp, T), q3(k_img, L), q3(v_img, L))
        queries = self._ln("norm2", ca.out(o, queries))
        if fused and self.mlp.num_layers == 2:
            l1, l2 = self.mlp.layers
            hmid = ops.gemm_tokens(queries, w_bf16(wc, "m1w", l1.weight), v_f32(wc, "m1b", l1.bias), act=self.mlp._act_code)
            queries = self._ln("norm3", ops.gemm(hmid, w_bf16(wc, "m2w", l2.weight), v_f32(wc, "m2b", l2.bias), residual=queries, out_dtype=F32))
        else:
            queries = self._ln("norm3", self.mlp.run(to_bf16(queries), residual=queries, out_dtype=F32))
        # image -> tokens
        if fused:
            Cd = ia.internal_dim
            kv = ops.gemm_tokens(queries, w_bf16(wc, "ikv", ia.k_proj.weight, ia.v_proj.weight), v_f32(wc, "ikvb", ia.k_proj.bias, ia.v_proj.bias),
                                 addend=query_pe, add_cols=Cd).view(B, T, 2 * Cd)
            o = ia.core(q3(q_img, L), kv[:, :, :Cd], kv[:, :, Cd:])
        else:
            qb = ops.add_cast(queries.view(1, B * T, C), query_pe.view(1, B * T, C), 1.0, OP16)[0]
            o = ia.core(q3(q_img, L), q3(ia.proj("k", qb), T), q3(ia.proj("v", to_bf16(queries)), T))
        n4 = self.norm4          # fp32 rows (the residual stream) and their 16-bit copy (the next projections' operand) from one launch
        keys, keys16 = ops.layernorm_dual(ia.out(o, keys), v_f32(wc, "norm4w", n4.weight), v_f32(wc, "norm4b", n4.bias), n4.eps)
        return queries, keys, keys16

    def forward(self, queries, keys, query_pe, key_pe):
        B, T, C = queries.shape
        L = keys.shape[1]
        assert key_pe.shape[0] == 1 or B == 1, "key_pe is the batch-shared dense position encoding"
        q, k, _ = self.run(queries.to(F32).reshape(B * T, C).contiguous(), keys.to(F32).reshape(B * L, C).contiguous(),
                           query_pe.to(F32).reshape(B * T, C).contiguous(), key_pe.to(F32).reshape(-1, C)[:L].contiguous(), B, T, L)
        return q.view(B, T, C), k.view(B, L, C)


class TwoWayTransformer(nn.Module):
    """transformer.py:28-118."""

    def __init__(self, depth: int, embedding_dim: int, num_heads: int, mlp_dim: int, activation: Type[nn.Module] = nn.ReLU,
                 attention_downsample_rate: int = 2) -> None:
        super().__init__()
        self.depth, self.embedding_dim, self.num_heads, self.mlp_dim = depth, embedding_dim, num_heads, mlp_dim
        self.layers = nn.ModuleList([
            TwoWayAttentionBlock(embedding_dim=embedding_dim, num_heads=num_heads, mlp_dim=mlp_dim, activation=activation,
                                 attention_downsample_rate=attention_downsample_rate, skip_first_layer_pe=(i == 0))
            for i in range(depth)])
        self.final_attn_token_to_image = Attention(embedding_dim, num_heads, downsample_rate=attention_downsample_rate)
        self.norm_final_attn = nn.LayerNorm(embedding_dim)
        self._wc = WeightCache()

    def run(self, keys, key_pe, tokens, B, T, L):
        """keys fp32 [B*L, C] tokens of the image embedding; key_pe fp32 [L, C]; tokens fp32 [B*T, C]."""
        C = tokens.shape[1]
        queries, qpe = tokens, tokens
        keys16 = None
        for layer in self.layers:
            queries, keys, keys16 = layer.run(queries, keys, qpe, key_pe, B, T, L, keys16)
        if keys16 is None:
            keys16 = to_bf16(keys)
        self._keys16 = keys16          # the up-scaling GEMM behind the transformer takes the same operand (MaskDecoder.predict_masks_tokens)
        fa = self.final_attn_token_to_image
        kv_img = _image_side_projections(self._wc, "fkv", keys16, key_pe, L, [fa.k_proj, fa.v_proj], [True, False])
        Cf = fa.internal_dim
        q3 = lambda t, n: t.view(B, n, -1)
        if B * T <= 32 and fa.internal_dim % 32 == 0:
            qp = ops.gemm_tokens(queries, w_bf16(self._wc, "fqw", fa.q_proj.weight), v_f32(self._wc, "fqb", fa.q_proj.bias), addend=qpe,
                                 add_cols=fa.internal_dim)
        else:
            qp = fa.proj("q", ops.add_cast(queries.view(1, B * T, C), qpe.view(1, B * T, C), 1.0, OP16)[0])
        o = fa.core(q3(qp, T), q3(kv_img[:, :Cf], L), q3(kv_img[:, Cf:], L))
        n = self.norm_final_attn
        queries = ops.layernorm(fa.out(o, queries), v_f32(self._wc, "nw", n.weight), v_f32(self._wc, "nb", n.bias), n.eps, out_dtype=F32)
        return queries, keys

    def forward(self, image_embedding, image_pe, point_embedding):
        B, C, h, w = image_embedding.shape
        T = point_embedding.shape[1]
        keys = tokens_of(image_embedding.to(F32))
        pe = tokens_of(image_pe.to(F32))[: h * w]
        q, k = self.run(keys, pe, point_embedding.to(F32).reshape(B * T, C).contiguous(), B, T, h * w)
        return q.view(B, T, C), k.view(B, h * w, C)


class MaskDecoder(nn.Module):
    """mask_decoder.py:15-317.  `cell_nums` (the fork's func_2d call site passes it by keyword, func_2d/function.py:159-168) repeats every
    image embedding for its prompt sets (215-231); None is the upstream one-prompt-set-per-image path (SURVEY.md shim S3)."""

    def __init__(self, *, transformer_dim: int, transformer: nn.Module, num_multimask_outputs: int = 3,
                 activation: Type[nn.Module] = nn.GELU, iou_head_depth: int = 3, iou_head_hidden_dim: int = 256,
                 use_high_res_features: bool = False, iou_prediction_use_sigmoid=False, dynamic_multimask_via_stability=False,
                 dynamic_multimask_stability_delta=0.05, dynamic_multimask_stability_thresh=0.98, pred_obj_scores: bool = False,
                 pred_obj_scores_mlp: bool = False, use_multimask_token_for_obj_ptr: bool = False) -> None:
        super().__init__()
        assert num_multimask_outputs == 3 and use_high_res_features and pred_obj_scores and pred_obj_scores_mlp, \
            "HIP path implements the SAM2 YAML decoder (3 multimasks, high-res skips, object-score MLP)"
        self.transformer_dim, self.transformer = transformer_dim, transformer
        self.num_multimask_outputs = num_multimask_outputs
        self.iou_token = nn.Embedding(1, transformer_dim)
        self.num_mask_tokens = num_multimask_outputs + 1
        self.mask_tokens = nn.Embedding(self.num_mask_tokens, transformer_dim)
        self.pred_obj_scores = pred_obj_scores
        self.obj_score_token = nn.Embedding(1, transformer_dim)
        self.use_multimask_token_for_obj_ptr = use_multimask_token_for_obj_ptr
        self.output_upscaling = nn.Sequential(
            nn.ConvTranspose2d(transformer_dim, transformer_dim // 4, kernel_size=2, stride=2), LayerNorm2d(transformer_dim // 4),
            activation(), nn.ConvTranspose2d(transformer_dim // 4, transformer_dim // 8, kernel_size=2, stride=2), activation())
        self.use_high_res_features = use_high_res_features
        self.conv_s0 = nn.Conv2d(transformer_dim, transformer_dim // 8, kernel_size=1, stride=1)
        self.conv_s1 = nn.Conv2d(transformer_dim, transformer_dim // 4, kernel_size=1, stride=1)
        self.output_hypernetworks_mlps = nn.ModuleList(
            [MLP(transformer_dim, transformer_dim, transformer_dim // 8, 3) for _ in range(self.num_mask_tokens)])
        self.iou_prediction_head = MLP(transformer_dim, iou_head_hidden_dim, self.num_mask_tokens, iou_head_depth,
                                       sigmoid_output=iou_prediction_use_sigmoid)
        self.pred_obj_score_head = MLP(transformer_dim, transformer_dim, 1, 3)
        self.dynamic_multimask_via_stability = dynamic_multimask_via_stability
        self.dynamic_multimask_stability_delta = dynamic_multimask_stability_delta
        self.dynamic_multimask_stability_thresh = dynamic_multimask_stability_thresh
        self._wc = WeightCache()

    def conv_s(self, which: int, tokens_bf16: torch.Tensor, out_dtype=OP16) -> torch.Tensor:
        """conv_s0 / conv_s1 (sam2_base.py:470-475) on token-major bf16 features."""
        c = self.conv_s0 if which == 0 else self.conv_s1
        return ops.gemm(tokens_bf16, w_bf16(self._wc, f"cs{which}w", c.weight), v_f32(self._wc, f"cs{which}b", c.bias), out_dtype=out_dtype)

    def predict_masks_tokens(self, src_tokens: torch.Tensor, pe_tokens: torch.Tensor, sparse: torch.Tensor, feat_s0: torch.Tensor,
                             feat_s1: torch.Tensor, B: int, h: int, w: int):
        """src_tokens fp32 [B*h*w, C] (= image embedding + dense prompt); pe_tokens fp32 [h*w, C]; sparse fp32 [B,P,C];
        feat_s0 [B*16hw, C/8], feat_s1 [B*4hw, C/4] token-major, 16-bit or fp32.  Returns (masks [B,4,4h,4w] fp32, iou [B,4],
        mask_tokens_out [B,4,C], object_score_logits [B,1])."""
        wc = self._wc
        C = self.transformer_dim
        out_tok = wc.get("otok", [self.obj_score_token.weight, self.iou_token.weight, self.mask_tokens.weight],
                         lambda: torch.cat([self.obj_score_token.weight, self.iou_token.weight, self.mask_tokens.weight], 0).detach().float())
        T = out_tok.shape[0] + sparse.shape[1]
        tokens = torch.cat([out_tok.unsqueeze(0).expand(B, -1, -1), sparse], 1)          # data movement: the query token list (one launch)
        hs, keys = self.transformer.run(src_tokens, pe_tokens, tokens.view(B * T, C), B, T, h * w)
        hs = hs.view(B, T, C)
        up = self.output_upscaling
        dc1_w = wc.get("dc1", [up[0].weight], lambda: up[0].weight.detach().permute(2, 3, 1, 0).reshape(-1, C).to(OP16).contiguous())
        keys16 = getattr(self.transformer, "_keys16", None)
        if keys16 is None or keys16.shape != keys.shape:
            keys16 = to_bf16(keys)
        self.transformer._keys16 = None
        g = ops.gemm(keys16, dc1_w)
        u = ops.convt2x2_shuffle(g, v_f32(wc, "dc1b", up[0].bias), feat_s1, v_f32(wc, "lnw", up[1].weight), v_f32(wc, "lnb", up[1].bias), B, h, w)
        dc2_w = wc.get("dc2", [up[3].weight], lambda: up[3].weight.detach().permute(2, 3, 1, 0).reshape(-1, C // 4).to(OP16).contiguous())
        g = ops.gemm(u, dc2_w)
        u = ops.convt2x2_shuffle(g, v_f32(wc, "dc2b", up[3].bias), feat_s0, None, None, B, 2 * h, 2 * w)  # [B*16hw, C/8] bf16
        heads = list(self.output_hypernetworks_mlps) + [self.iou_prediction_head, self.pred_obj_score_head]
        fusable = C == 256 and self.pred_obj_scores and all(
            m.num_layers == 3 and m._act_code == ops.ACT_RELU and m.layers[0].in_features == C and m.layers[0].out_features == C
            and m.layers[1].out_features == C and m.layers[2].out_features <= C for m in heads)
        if fusable:
            # the 4 hyper-network MLPs, the IoU head and the object-score head: one launch (ops.token_mlp3)
            G = len(heads)
            params = [t for m in heads for l in m.layers for t in (l.weight, l.bias)]

            def pack():
                dev = hs.device
                w1 = torch.stack([m.layers[0].weight.detach() for m in heads]).to(OP16).contiguous()
                w2 = torch.stack([m.layers[1].weight.detach() for m in heads]).to(OP16).contiguous()
                w3 = torch.zeros(G, C, C, dtype=OP16, device=dev)
                b3 = torch.zeros(G, C, dtype=F32, device=dev)
                for gi, m in enumerate(heads):
                    n = m.layers[2].out_features
                    w3[gi, :n] = m.layers[2].weight.detach().to(OP16)
                    b3[gi, :n] = m.layers[2].bias.detach().float()
                b1 = torch.stack([m.layers[0].bias.detach() for m in heads]).float().contiguous()
                b2 = torch.stack([m.layers[1].bias.detach() for m in heads]).float().contiguous()
                return w1, b1, w2, b2, w3, b3

            def consts():   # shape-only index tensors: built once (host -> device copies are not allowed inside a hipGraph capture,
                dev = hs.device   # and the weight pack above re-runs whenever an optimiser step has touched the weights)
                nm_ = self.num_mask_tokens
                return (torch.tensor([2 + i for i in range(nm_)] + [1, 0], dtype=torch.int32, device=dev),
                        torch.tensor([m.layers[2].out_features for m in heads], dtype=torch.int32, device=dev),
                        torch.tensor([int(m.sigmoid_output) for m in heads], dtype=torch.int32, device=dev))
            tok, od, sg = wc.get("heads_const", [], consts)
            w1, b1, w2, b2, w3, b3 = wc.get("heads", params, pack)
            nm = self.num_mask_tokens
            n_hyper, n_iou, n_obj = C // 8, heads[nm].layers[2].out_features, heads[nm + 1].layers[2].out_features

            def layout():   # packed result: hyper [B, nm, C/8] | iou [B, n_iou] | obj [B, n_obj], each contiguous (no slicing copies)
                dev = hs.device
                off = [g * n_hyper for g in range(nm)] + [B * nm * n_hyper, B * nm * n_hyper + B * n_iou]
                ld = [nm * n_hyper] * nm + [n_iou, n_obj]
                return torch.tensor(off, dtype=torch.int32, device=dev), torch.tensor(ld, dtype=torch.int32, device=dev)
            off, ld = wc.get(f"heads_layout{B}", [], layout)
            y = ops.token_mlp3(hs, tok, w1, b1, w2, b2, w3, b3, od, sg, packed=(off, ld, B * (nm * n_hyper + n_iou + n_obj)))
            hyper = y[: B * nm * n_hyper].view(B, nm, n_hyper)
            iou = y[B * nm * n_hyper: B * (nm * n_hyper + n_iou)].view(B, n_iou)
            obj = y[B * (nm * n_hyper + n_iou):].view(B, n_obj)
        else:
            hyper = torch.empty(B, self.num_mask_tokens, C // 8, dtype=F32, device=u.device)
            for i, m in enumerate(self.output_hypernetworks_mlps):
                hyper[:, i] = m.run(to_bf16(hs[:, 2 + i].contiguous()))
            iou = self.iou_prediction_head.run(to_bf16(hs[:, 1].contiguous()))
            obj = self.pred_obj_score_head.run(to_bf16(hs[:, 0].contiguous()))
        masks = ops.hyper_masks(hyper, u, B, 16 * h * w).view(B, self.num_mask_tokens, 4 * h, 4 * w)
        return masks, iou, hs[:, 2:2 + self.num_mask_tokens], obj

    def predict_masks(self, image_embeddings, image_pe, sparse_prompt_embeddings, dense_prompt_embeddings, repeat_image,
                      cell_nums=None, high_res_features=None):
        # mask_decoder.py:215-231: when there are more prompt sets than image embeddings, `cell_nums[i]` prompt sets belong to image i
        # (torch.repeat_interleave of the embedding; the position encoding and the dense embedding broadcast).  The high-res skips are
        # added un-repeated in the reference (244-247), which only broadcasts for ONE image: mirrored, anything else raises like torch.
        n_sets = sparse_prompt_embeddings.shape[0]
        if image_embeddings.shape[0] != n_sets:
            if cell_nums is not None:
                cn = torch.as_tensor(cell_nums, device=image_embeddings.device).reshape(-1).long()
                if cn.numel() != image_embeddings.shape[0] or int(cn.sum()) != n_sets:
                    raise RuntimeError(f"cell_nums {cn.tolist()} does not split {n_sets} prompt sets over {image_embeddings.shape[0]} images")
                image_embeddings = torch.repeat_interleave(image_embeddings, cn, dim=0)          # data movement
            elif image_embeddings.shape[0] == 1:
                image_embeddings = image_embeddings.expand(n_sets, -1, -1, -1)
            else:
                raise RuntimeError(f"The size of tensor a ({image_embeddings.shape[0]}) must match the size of tensor b ({n_sets}) at non-singleton dimension 0")
            hr = []
            for f in high_res_features:
                if f.shape[0] == 1:
                    f = f.expand(n_sets, -1, -1, -1)
                elif f.shape[0] != n_sets:
                    raise RuntimeError(f"The size of tensor a ({n_sets}) must match the size of tensor b ({f.shape[0]}) at non-singleton dimension 0")
                hr.append(f)
            high_res_features = hr
        B, C, h, w = image_embeddings.shape
        assert sparse_prompt_embeddings.shape[0] == B, "one prompt set per (repeated) image embedding"
        from .. import autograd as ag
        if ag.active(self):
            # train() + grad mode (func_2d/function.py:140-170 calls this module directly): differentiable glue + the decoder's Function
            src = tokens_of(image_embeddings.to(F32)).view(B, h * w, C) + tokens_of(dense_prompt_embeddings.to(F32).expand(B, C, h, w)).view(B, h * w, C)
            f0, f1 = high_res_features
            return ag.mask_decoder(self, src.reshape(B * h * w, C), tokens_of(image_pe.to(F32))[: h * w].detach(), sparse_prompt_embeddings.to(F32),
                                   tokens_of(f0), tokens_of(f1), B, h, w)
        emb = image_embeddings.to(F32)
        dense = dense_prompt_embeddings.to(F32).expand(B, C, h, w)
        # image embedding + dense prompt embedding -> token-major fp32 (the spatially constant no-mask embedding is read as a
        # broadcast vector inside the kernel)
        e3 = tokens_of(emb).view(B, h * w, C)
        if dense.stride(2) == 0 and dense.stride(3) == 0:
            d3 = dense[:, :, 0, 0].unsqueeze(1).expand(B, h * w, C)            # strides (0 | C, 0, 1): broadcast inside the kernel
        else:
            d3 = tokens_of(dense).view(B, h * w, C)
        src = ops.add_cast(e3, d3, 1.0, F32).view(B * h * w, C)
        pe = tokens_of(image_pe.to(F32))[: h * w]
        f0, f1 = high_res_features
        # the skip features stay fp32 token-major (the up-scaling tail reads them as they are) when their widths are the kernel's 32 / 64
        skip = lambda f: tokens_of(f) if (f.dtype == F32 and f.shape[1] in (32, 64)) else to_bf16(tokens_of(f))
        return self.predict_masks_tokens(src, pe, sparse_prompt_embeddings.to(F32), skip(f0), skip(f1), B, h, w)

    def forward(self, image_embeddings, image_pe, sparse_prompt_embeddings, dense_prompt_embeddings, multimask_output: bool,
                repeat_image: bool, cell_nums=None, high_res_features: Optional[List[torch.Tensor]] = None):
        masks, iou_pred, mask_tokens_out, object_score_logits = self.predict_masks(
            image_embeddings, image_pe, sparse_prompt_embeddings, dense_prompt_embeddings, repeat_image, cell_nums, high_res_features)
        if multimask_output:
            masks, iou_pred = masks[:, 1:, :, :], iou_pred[:, 1:]
        elif self.dynamic_multimask_via_stability and not self.training:
            nb = object_score_logits.shape[0]                       # no object gating at this level: a cached vector of ones
            obj_pos = self._wc.get(f"ones{nb}", [], lambda: torch.ones(nb, dtype=F32, device=object_score_logits.device))
            low, sel, iou_sel = ops.select_mask(masks.contiguous(), iou_pred.contiguous(), obj_pos, False, True,
                                                self.dynamic_multimask_stability_delta, self.dynamic_multimask_stability_thresh)
            masks, iou_pred = low, iou_sel
        else:
            masks, iou_pred = masks[:, 0:1, :, :], iou_pred[:, 0:1]
        if multimask_output and self.use_multimask_token_for_obj_ptr:
            sam_tokens_out = mask_tokens_out[:, 1:]
        else:
            sam_tokens_out = mask_tokens_out[:, 0:1]
        return masks, iou_pred, sam_tokens_out, object_score_logits
