"""Drop-in image encoder: Hiera trunk + FPN neck, same class names, constructor arguments, state-dict keys and forward
signatures as the reference (sam2_train/modeling/backbones/{hieradet,utils,image_encoder}.py), every forward routed
through the MI355X C-ABI kernels.

Internal layout: token-major fp32 residual stream [B*H*W, C]; bf16 operands for the MFMA GEMMs / attention.  Feature
maps are handed out as NCHW *views* of that memory, so the callers' flatten/permute/view chains stay copy-free.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.nn as nn

from .. import ops
from .common import OP16, F32, WeightCache, nchw_view, to_bf16, tokens_of, v_f32, w_bf16
from .position import PositionEmbeddingSine


import os as _os
_POOL_GEMM = _os.environ.get("MSAM2_NO_POOL_GEMM") is None   # experiment switches
_QPOOL_GEMM = _os.environ.get("MSAM2_NO_QPOOL_GEMM") is None
_FUSED_MLP = _os.environ.get("MSAM2_NO_FUSED_MLP") is None
_FUSED_PATCH = _os.environ.get("MSAM2_NO_FUSED_PATCH") is None


class PatchEmbed(nn.Module):
    """backbones/utils.py:65-95 (Conv2d k7 s4 p3 -> NHWC)."""

    def __init__(self, kernel_size=(7, 7), stride=(4, 4), padding=(3, 3), in_chans: int = 3, embed_dim: int = 768):
        super().__init__()
        assert tuple(kernel_size) == (7, 7) and tuple(stride) == (4, 4) and tuple(padding) == (3, 3) and in_chans == 3, \
            "the HIP path implements the 7x7/4/3 RGB patch embedding used by every SAM2 config"
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=kernel_size, stride=stride, padding=padding)
        self._wc = WeightCache()

    def _weight(self):
        def build():
            w = torch.zeros(self.proj.weight.shape[0], 160, dtype=OP16, device=self.proj.weight.device)
            w[:, :147] = self.proj.weight.detach().reshape(-1, 147).to(OP16)
            return w
        return self._wc.get("w", [self.proj.weight], build)

    def _weight_perm(self):
        """the conv weight in the one-kernel patch embedding's reduction order: [ceil32(E), 176], k' = (c * 7 + ky) * 8 + 1 + kx, zero at
        k' % 8 == 0 and in the padding (ops.patch_embed: 8 consecutive k' = 8 consecutive pixels of one image row, the first of them the
        16-byte-aligned pixel in front of the window)"""
        def build():
            w = self.proj.weight.detach()
            E = w.shape[0]
            wp = torch.zeros((E + 31) // 32 * 32, 22, 8, dtype=OP16, device=w.device)
            wp[:E, :21, 1:] = w.reshape(E, 21, 7).to(OP16)
            return wp.reshape(-1, 176).contiguous()
        return self._wc.get("wp", [self.proj.weight], build)

    def tokens(self, x: torch.Tensor, pos: Optional[torch.Tensor] = None) -> torch.Tensor:
        """[B,3,S,S] -> fp32 tokens [B*(S/4)^2, E] (+ position table broadcast over the batch)."""
        x = x.to(F32).contiguous()
        E = self.proj.weight.shape[0]
        if _FUSED_PATCH and ops.patch_embed_supported(x.shape[-1], E) and x.shape[-1] == x.shape[-2] and (pos is None or pos.shape[0] == (x.shape[-1] // 4) ** 2):
            return ops.patch_embed(x, self._weight_perm(), v_f32(self._wc, "b", self.proj.bias), None if pos is None else pos.to(F32).contiguous())
        cols = ops.im2col_patch(x)
        return ops.gemm(cols, self._weight(), v_f32(self._wc, "b", self.proj.bias), residual=pos,
                        res_mod=pos.shape[0] if pos is not None else 0, out_dtype=F32)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        B, _, S, _ = x.shape
        return self.tokens(x).reshape(B, S // 4, S // 4, -1)


class MLP(nn.Module):
    """sam2_utils.py:108-132 (Linear-act-Linear...)."""

    def __init__(self, input_dim: int, hidden_dim: int, output_dim: int, num_layers: int, activation: nn.Module = nn.ReLU,
                 sigmoid_output: bool = False):
        super().__init__()
        self.num_layers = num_layers
        h = [hidden_dim] * (num_layers - 1)
        self.layers = nn.ModuleList(nn.Linear(n, k) for n, k in zip([input_dim] + h, h + [output_dim]))
        self.sigmoid_output = sigmoid_output
        self.act = activation()
        self._act_code = ops.ACT_GELU if isinstance(self.act, nn.GELU) else ops.ACT_RELU
        self._wc = WeightCache()

    def run(self, x_bf16: torch.Tensor, residual: Optional[torch.Tensor] = None, out_dtype=F32) -> torch.Tensor:
        """x [rows, in] bf16 -> [rows, out]; hidden activations bf16, last layer (+ residual) in out_dtype."""
        h = x_bf16
        for i, layer in enumerate(self.layers):
            last = i == self.num_layers - 1
            last_act = ops.ACT_SIGMOID if self.sigmoid_output else ops.ACT_NONE
            h = ops.gemm(h, w_bf16(self._wc, f"w{i}", layer.weight), v_f32(self._wc, f"b{i}", layer.bias),
                         act=last_act if last else self._act_code, residual=residual if last else None,
                         out_dtype=out_dtype if last else OP16)
        return h

    def run_tokens(self, x: torch.Tensor) -> torch.Tensor:
        """fp32 [rows, in] (a few rows: decoder tokens) -> fp32 [rows, out].  A 3-layer ReLU MLP of width 256 runs as ONE launch
        (ops.token_mlp3, fp32 activations); anything else goes through the GEMM path."""
        L = self.layers
        if (self.num_layers == 3 and self._act_code == ops.ACT_RELU and x.shape[0] <= 4096 and L[0].in_features == 256
                and L[0].out_features == 256 and L[1].out_features == 256 and L[2].out_features <= 256):
            def pack():
                dev = x.device
                n = L[2].out_features
                w3 = torch.zeros(1, 256, 256, dtype=OP16, device=dev)
                b3 = torch.zeros(1, 256, dtype=F32, device=dev)
                w3[0, :n] = L[2].weight.detach().to(OP16)
                b3[0, :n] = L[2].bias.detach().float()
                return (L[0].weight.detach().to(OP16)[None].contiguous(), L[0].bias.detach().float()[None].contiguous(),
                        L[1].weight.detach().to(OP16)[None].contiguous(), L[1].bias.detach().float()[None].contiguous(), w3, b3)

            def consts():   # built once: host -> device copies are not capturable, the weight pack may re-run inside a capture
                dev = x.device
                return (torch.zeros(1, dtype=torch.int32, device=dev), torch.tensor([L[2].out_features], dtype=torch.int32, device=dev),
                        torch.tensor([int(self.sigmoid_output)], dtype=torch.int32, device=dev))
            tok, od, sg = self._wc.get("tok3_const", [], consts)
            w1, b1, w2, b2, w3, b3 = self._wc.get("tok3", [t for l in L for t in (l.weight, l.bias)], pack)
            y = ops.token_mlp3(x.to(F32).contiguous().view(x.shape[0], 1, 256), tok, w1, b1, w2, b2, w3, b3, od, sg)
            return y[:, 0, : L[2].out_features].contiguous()
        return self.run(to_bf16(x.contiguous()))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        shp = x.shape
        return self.run(to_bf16(x.reshape(-1, shp[-1]).contiguous())).reshape(*shp[:-1], -1)


class MultiScaleAttention(nn.Module):
    """hieradet.py:37-83; parameters only -- the fused block below drives the kernels."""

    def __init__(self, dim: int, dim_out: int, num_heads: int, q_pool: nn.Module = None):
        super().__init__()
        self.dim, self.dim_out, self.num_heads = dim, dim_out, num_heads
        self.q_pool = q_pool
        self.qkv = nn.Linear(dim, dim_out * 3)
        self.proj = nn.Linear(dim_out, dim_out)


class MultiScaleBlock(nn.Module):
    """hieradet.py:86-168."""

    def __init__(self, dim: int, dim_out: int, num_heads: int, mlp_ratio: float = 4.0, drop_path: float = 0.0,
                 norm_layer="LayerNorm", q_stride: Tuple[int, int] = None, act_layer: nn.Module = nn.GELU, window_size: int = 0):
        super().__init__()
        assert drop_path == 0.0, "stochastic depth is a training-time option outside the forward hot path"
        self.dim, self.dim_out, self.window_size, self.q_stride = dim, dim_out, window_size, q_stride
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.pool = nn.MaxPool2d(kernel_size=q_stride, stride=q_stride, ceil_mode=False) if q_stride else None
        self.attn = MultiScaleAttention(dim, dim_out, num_heads=num_heads, q_pool=self.pool)
        self.norm2 = nn.LayerNorm(dim_out, eps=1e-6)
        self.mlp = MLP(dim_out, int(dim_out * mlp_ratio), dim_out, num_layers=2, activation=act_layer)
        if dim != dim_out:
            self.proj = nn.Linear(dim, dim_out)
        self._wc = WeightCache()

    def _packed_attn_weights(self):
        """qkv / proj weights laid out for the attention kernels.  Head dims the kernels are built for (64/96/128/256) are used
        as they are; others (56 of Hiera-B+) are zero-padded per head to the next one (a zero q/k channel adds nothing to the
        scores, a zero v channel yields an output channel that the padded proj column ignores)."""
        a = self.attn
        heads, dim_out = a.num_heads, self.dim_out
        D = dim_out // heads
        Dp = D if D in (64, 96, 128, 256) else next(d for d in (64, 96, 128, 256) if d >= D)
        wc = self._wc

        def qkv_w():
            w = a.qkv.weight.detach().reshape(3, heads, D, -1)
            out = torch.zeros(3, heads, Dp, w.shape[-1], dtype=w.dtype, device=w.device)
            out[:, :, :D] = w
            return out.reshape(3 * heads * Dp, -1).to(OP16).contiguous()

        def qkv_b():
            b = a.qkv.bias.detach().reshape(3, heads, D)
            out = torch.zeros(3, heads, Dp, dtype=F32, device=b.device)
            out[:, :, :D] = b
            return out.reshape(-1).contiguous()

        def proj_w():
            w = a.proj.weight.detach().reshape(dim_out, heads, D)
            out = torch.zeros(dim_out, heads, Dp, dtype=w.dtype, device=w.device)
            out[:, :, :D] = w
            return out.reshape(dim_out, heads * Dp).to(OP16).contiguous()

        if Dp == D:
            return D, Dp, w_bf16(wc, "qkvw", a.qkv.weight), v_f32(wc, "qkvb", a.qkv.bias), w_bf16(wc, "ow", a.proj.weight)
        return (D, Dp, wc.get("qkvw_p", [a.qkv.weight], qkv_w), wc.get("qkvb_p", [a.qkv.bias], qkv_b),
                wc.get("ow_p", [a.proj.weight], proj_w))

    def run(self, t: torch.Tensor, B: int, H: int, W: int, emit16: bool = False) -> Tuple[torch.Tensor, int, int]:
        """fp32 tokens [B*H*W, dim] -> (fp32 tokens [B*H'*W', dim_out], H', W').  emit16 (a stage's last block): `self.out16` also holds the
        result in the 16-bit operand type when the block's last kernel could write it in the same store (else None)."""
        self.out16 = None
        wc, a = self._wc, self.attn
        heads, dim_out = a.num_heads, self.dim_out
        D, Dp, qkv_w, qkv_b, proj_w = self._packed_attn_weights()
        width = heads * Dp                      # per-token width of each of q, k, v in the (possibly padded) layout
        scale = 1.0 / (D ** 0.5)
        xn = ops.layernorm(t, v_f32(wc, "n1w", self.norm1.weight), v_f32(wc, "n1b", self.norm1.bias), 1e-6)
        pool = self.q_stride is not None
        if self.dim != dim_out:
            if _POOL_GEMM and pool and B * H * W >= 256 and H % 2 == 0 and W % 2 == 0:   # projection + 2x2 max-pool in one GEMM
                shortcut = ops.gemm_pool2x2(xn, w_bf16(wc, "pw", self.proj.weight), v_f32(wc, "pb", self.proj.bias), B, H, W)
            else:
                shortcut = ops.gemm(xn, w_bf16(wc, "pw", self.proj.weight), v_f32(wc, "pb", self.proj.bias), out_dtype=F32)
                if pool:
                    shortcut = ops.maxpool2x2(shortcut, B, H, W)
        else:
            shortcut = t
        if _POOL_GEMM and _QPOOL_GEMM and pool and B * H * W >= 256 and H % 2 == 0 and W % 2 == 0 and width % 32 == 0:
            qkv, qp = ops.gemm_qkv_pool2x2(xn, qkv_w, qkv_b, B, H, W, width)   # k | v in image order + pooled q, one GEMM
        else:
            qkv = ops.gemm(xn, qkv_w, qkv_b)  # 16-bit [T, 3*width]
            qp = ops.maxpool2x2(qkv[:, :width], B, H, W) if pool else None
        Hq, Wq = (H // 2, W // 2) if pool else (H, W)
        if self.window_size > 0:
            o = ops.window_attention(qkv, B, H, W, heads, self.window_size, qkv_b, q_pooled=qp, scale=scale)
        else:
            v5 = qkv.view(B, H * W, 3, heads, Dp)
            q = (qp.view(B, Hq * Wq, heads, Dp) if pool else v5[:, :, 0]).permute(0, 2, 1, 3)
            o = ops.attention(q, v5[:, :, 1].permute(0, 2, 1, 3), v5[:, :, 2].permute(0, 2, 1, 3), scale=scale)
            o = o.permute(0, 2, 1, 3).reshape(B * Hq * Wq, width)
        t = ops.gemm(o, proj_w, v_f32(wc, "ob", a.proj.bias), residual=shortcut, out_dtype=F32)
        mlp = self.mlp
        if (_FUSED_MLP and ops.ln_mlp_residual_supported(dim_out) and mlp.num_layers == 2 and mlp._act_code == ops.ACT_GELU
                and mlp.layers[0].out_features == 4 * dim_out and Hq * Wq >= 1024):
            # the two high-resolution stages: LayerNorm + fc1 + GELU + fc2 + residual as one kernel, the hidden map stays in registers
            # (the gate looks at the tokens PER SLICE, never at the batch: a slice's features must not depend on its batch mates)
            w2p = mlp._wc.get("w1p", [mlp.layers[1].weight], lambda: ops.mlp_fused_permute_w2(mlp.layers[1].weight.detach().to(OP16).contiguous()))
            t = ops.ln_mlp_residual(t, v_f32(wc, "n2w", self.norm2.weight), v_f32(wc, "n2b", self.norm2.bias), 1e-6,
                                    w_bf16(mlp._wc, "w0", mlp.layers[0].weight), v_f32(mlp._wc, "b0", mlp.layers[0].bias), w2p,
                                    v_f32(mlp._wc, "b1", mlp.layers[1].bias), also16=emit16)
            if emit16:
                t, self.out16 = t
            return t, Hq, Wq
        xn2 = ops.layernorm(t, v_f32(wc, "n2w", self.norm2.weight), v_f32(wc, "n2b", self.norm2.bias), 1e-6)
        t = self.mlp.run(xn2, residual=t, out_dtype=F32)
        return t, Hq, Wq

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        B, H, W, C = x.shape
        t, h, w = self.run(x.to(F32).reshape(B * H * W, C).contiguous(), B, H, W)
        return t.reshape(B, h, w, -1)


class Hiera(nn.Module):
    """hieradet.py:171-295."""

    def __init__(self, embed_dim: int = 96, num_heads: int = 1, drop_path_rate: float = 0.0, q_pool: int = 3,
                 q_stride: Tuple[int, int] = (2, 2), stages: Tuple[int, ...] = (2, 3, 16, 3), dim_mul: float = 2.0,
                 head_mul: float = 2.0, window_pos_embed_bkg_spatial_size: Tuple[int, int] = (14, 14),
                 window_spec: Tuple[int, ...] = (8, 4, 14, 7), global_att_blocks: Tuple[int, ...] = (12, 16, 20),
                 return_interm_layers=True):
        super().__init__()
        assert len(stages) == len(window_spec) and drop_path_rate == 0.0
        self.window_spec = window_spec
        depth = sum(stages)
        self.q_stride = q_stride
        self.stage_ends = [sum(stages[:i]) - 1 for i in range(1, len(stages) + 1)]
        self.q_pool_blocks = [x + 1 for x in self.stage_ends[:-1]][:q_pool]
        self.return_interm_layers = return_interm_layers
        self.patch_embed = PatchEmbed(embed_dim=embed_dim)
        self.global_att_blocks = global_att_blocks
        self.window_pos_embed_bkg_spatial_size = window_pos_embed_bkg_spatial_size
        self.pos_embed = nn.Parameter(torch.zeros(1, embed_dim, *window_pos_embed_bkg_spatial_size))
        self.pos_embed_window = nn.Parameter(torch.zeros(1, embed_dim, window_spec[0], window_spec[0]))
        cur_stage = 1
        self.blocks = nn.ModuleList()
        for i in range(depth):
            dim_out = embed_dim
            window_size = window_spec[cur_stage - 1]
            if global_att_blocks is not None and i in global_att_blocks:
                window_size = 0
            if i - 1 in self.stage_ends:
                dim_out = int(embed_dim * dim_mul)
                num_heads = int(num_heads * head_mul)
                cur_stage += 1
            self.blocks.append(MultiScaleBlock(dim=embed_dim, dim_out=dim_out, num_heads=num_heads,
                                               q_stride=q_stride if i in self.q_pool_blocks else None, window_size=window_size))
            embed_dim = dim_out
        self.channel_list = ([self.blocks[i].dim_out for i in self.stage_ends[::-1]] if return_interm_layers
                             else [self.blocks[-1].dim_out])
        self._wc = WeightCache()

    def _pos_tokens(self, h: int, w: int) -> torch.Tensor:
        """input-independent: rebuilt only when the two parameters change (hieradet.py:269-277)."""
        return self._wc.get(f"pos{h}x{w}", [self.pos_embed, self.pos_embed_window],
                            lambda: ops.hiera_pos_embed(self.pos_embed.detach().float(), self.pos_embed_window.detach().float(), h, w))

    def _get_pos_embed(self, hw: Tuple[int, int]) -> torch.Tensor:
        h, w = hw
        return self._pos_tokens(h, w).reshape(1, h, w, -1)

    def forward(self, x: torch.Tensor) -> List[torch.Tensor]:
        B, _, S, _ = x.shape
        h = w = S // 4
        t = self.patch_embed.tokens(x, self._pos_tokens(h, w))
        outputs = []
        for i, blk in enumerate(self.blocks):
            is_out = i == self.stage_ends[-1] or (i in self.stage_ends and self.return_interm_layers)
            t, h, w = blk.run(t, B, h, w, emit16=is_out)
            if is_out:
                v = nchw_view(t, B, h, w)
                # the 16-bit token-major copy the block's last kernel wrote beside the fp32 rows (stages 1 / 2): FpnNeck takes it as its lateral
                # GEMM operand instead of casting the map again (a private attribute of this tensor object; any other consumer ignores it)
                v._op16_tokens, blk.out16 = blk.out16, None
                outputs.append(v)
        return outputs


class FpnNeck(nn.Module):
    """image_encoder.py:45-133 (1x1 lateral convs, nearest-2x top-down sum on selected levels, sine position maps)."""

    def __init__(self, position_encoding: nn.Module, d_model: int, backbone_channel_list: List[int], kernel_size: int = 1,
                 stride: int = 1, padding: int = 0, fpn_interp_model: str = "bilinear", fuse_type: str = "sum",
                 fpn_top_down_levels: Optional[List[int]] = None):
        super().__init__()
        assert kernel_size == 1 and stride == 1 and padding == 0, "lateral convs are 1x1 in every SAM2 config"
        assert fpn_interp_model == "nearest" and fuse_type == "sum", "HIP path implements the YAML's nearest/sum top-down fusion"
        self.position_encoding = position_encoding
        self.convs = nn.ModuleList()
        self.backbone_channel_list = backbone_channel_list
        for dim in backbone_channel_list:
            cur = nn.Sequential()
            cur.add_module("conv", nn.Conv2d(in_channels=dim, out_channels=d_model, kernel_size=1))
            self.convs.append(cur)
        self.fpn_interp_model, self.fuse_type = fpn_interp_model, fuse_type
        self.fpn_top_down_levels = list(range(len(self.convs)) if fpn_top_down_levels is None else fpn_top_down_levels)
        self._wc = WeightCache()

    def forward(self, xs: List[torch.Tensor], post_convs=None):
        """post_convs (private, used by SAM2Base.forward_image): {level: 1x1 nn.Conv2d applied to that level's output}.  On a level
        that takes no part in the top-down pathway the lateral conv and the following 1x1 conv are two linear maps in a row, so
        they run as ONE GEMM with the composed weight W_post W_lat / bias W_post b_lat + b_post (composed in fp32): the
        256-channel level-0 map (268 MB fp32 at 4 x 1024^2) is then never written, converted or read back."""
        out, pos = [None] * len(self.convs), [None] * len(self.convs)
        assert len(xs) == len(self.convs)
        prev = None
        n = len(self.convs) - 1
        td = self.fpn_top_down_levels
        def operand(x):            # the stage output as the lateral GEMM's 16-bit operand
            x16 = getattr(x, "_op16_tokens", None)
            ok = x16 is not None and x16.dtype == OP16 and x16.shape == (x.shape[0] * x.shape[2] * x.shape[3], x.shape[1])
            return x16 if ok else to_bf16(tokens_of(x))
        for i in range(n, -1, -1):
            B, C, H, W = xs[i].shape
            conv = self.convs[n - i].conv
            pc = post_convs.get(i) if post_convs else None
            if pc is not None and i not in td and (i == 0 or (i - 1) not in td):
                w = self._wc.get(f"cw{i}", [conv.weight, pc.weight], lambda: (
                    pc.weight.detach().float().reshape(pc.weight.shape[0], -1) @ conv.weight.detach().float().reshape(conv.weight.shape[0], -1)
                ).to(OP16).contiguous())
                b = self._wc.get(f"cb{i}", [conv.bias, pc.weight, pc.bias], lambda: (
                    pc.weight.detach().float().reshape(pc.weight.shape[0], -1) @ conv.bias.detach().float() + pc.bias.detach().float()).contiguous())
                lat = ops.gemm(operand(xs[i]), w, b, out_dtype=F32)
                prev = None
            else:
                lat = ops.gemm(operand(xs[i]), w_bf16(self._wc, f"w{i}", conv.weight), v_f32(self._wc, f"b{i}", conv.bias),
                               out_dtype=F32)
                if i in td and prev is not None:
                    ops.upsample2x_add_(lat, prev, B, H, W)
                prev = lat
                if pc is not None:
                    lat = ops.gemm(to_bf16(lat), w_bf16(self._wc, f"pw{i}", pc.weight), v_f32(self._wc, f"pb{i}", pc.bias), out_dtype=F32)
            out[i] = nchw_view(lat, B, H, W)
            pos[i] = self.position_encoding(out[i]).to(out[i].dtype)
        return out, pos


class ImageEncoder(nn.Module):
    """image_encoder.py:14-42."""

    def __init__(self, trunk: nn.Module, neck: nn.Module, scalp: int = 0):
        super().__init__()
        self.trunk, self.neck, self.scalp = trunk, neck, scalp
        assert self.trunk.channel_list == self.neck.backbone_channel_list, \
            f"Channel dims of trunk and neck do not match. Trunk: {self.trunk.channel_list}, neck: {self.neck.backbone_channel_list}"

    def forward(self, sample: torch.Tensor, post_convs=None):
        features, pos = self.neck(self.trunk(sample), post_convs) if post_convs else self.neck(self.trunk(sample))
        if self.scalp > 0:
            features, pos = features[: -self.scalp], pos[: -self.scalp]
        return {"vision_features": features[-1], "vision_pos_enc": pos, "backbone_fpn": features}
