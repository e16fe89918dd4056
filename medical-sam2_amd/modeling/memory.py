"""Drop-in memory attention and memory encoder (sam2_train/modeling/{memory_attention,memory_encoder}.py and the
RoPEAttention of sam/transformer.py:266-331), routed through the MI355X kernels.

Memory attention runs batch-first internally: fp32 residual stream [B*L, 256], fused q|k|v projection for the self
attention, per-layer K/V projection of the 64-channel memory bank, in-place axial RoPE on the bf16 q/k rows, and the
D=256 single-head flash kernel with split-KV sized to fill the 256 CUs.
"""
from __future__ import annotations

import copy
import math
from typing import Optional, Tuple

import torch
import torch.nn as nn

from .. import ops
from .common import OP16, F32, WeightCache, attn_splits, nchw_view, to_bf16, tokens_of, v_f32, w_bf16
from .encoder import MLP  # noqa: F401  (re-export for the registry)


import os as _os
_FUSED_ROPE = _os.environ.get("MSAM2_NO_FUSED_ROPE") is None   # experiment switch: rotate in a separate in-place kernel instead
_FOLD_V = _os.environ.get("MSAM2_NO_VALUE_FOLD") is None       # experiment switch: 256-wide values through v_proj instead
_BATCH_K_PROJ = _os.environ.get("MSAM2_BATCHED_KPROJ") is not None   # experiment switch (OFF: measured slower, see _project_all_keys)


class LayerNorm2d(nn.Module):
    """sam2_utils.py:137-149 -- parameters only; the kernels normalise NHWC tokens over the channel dim."""

    def __init__(self, num_channels: int, eps: float = 1e-6) -> None:
        super().__init__()
        self.weight = nn.Parameter(torch.ones(num_channels))
        self.bias = nn.Parameter(torch.zeros(num_channels))
        self.eps = eps

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        B, C, H, W = x.shape
        y = ops.layernorm(tokens_of(x), self.weight.detach().float(), self.bias.detach().float(), self.eps, out_dtype=F32)
        return nchw_view(y, B, H, W)


class Attention(nn.Module):
    """sam/transformer.py:199-263.  `forward(q,k,v)` takes [B, L, C] tensors like the reference."""

    def __init__(self, embedding_dim: int, num_heads: int, downsample_rate: int = 1, dropout: float = 0.0, kv_in_dim: int = None):
        super().__init__()
        self.embedding_dim = embedding_dim
        self.kv_in_dim = kv_in_dim if kv_in_dim is not None else embedding_dim
        self.internal_dim = embedding_dim // downsample_rate
        self.num_heads = num_heads
        assert self.internal_dim % num_heads == 0, "num_heads must divide embedding_dim."
        self.q_proj = nn.Linear(embedding_dim, self.internal_dim)
        self.k_proj = nn.Linear(self.kv_in_dim, self.internal_dim)
        self.v_proj = nn.Linear(self.kv_in_dim, self.internal_dim)
        self.out_proj = nn.Linear(self.internal_dim, embedding_dim)
        self.dropout_p = dropout
        self._wc = WeightCache()

    # -- kernel-ready pieces -------------------------------------------------------------------------------------
    def proj(self, which: str, x_bf16: torch.Tensor) -> torch.Tensor:
        lin = getattr(self, which + "_proj")
        return ops.gemm(x_bf16, w_bf16(self._wc, which + "w", lin.weight), v_f32(self._wc, which + "b", lin.bias))

    def out(self, o_bf16: torch.Tensor, residual: Optional[torch.Tensor], out_dtype=F32) -> torch.Tensor:
        return ops.gemm(o_bf16, w_bf16(self._wc, "ow", self.out_proj.weight), v_f32(self._wc, "ob", self.out_proj.bias),
                        residual=residual, out_dtype=out_dtype)

    def core(self, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
        """projected bf16 [B, L, internal] -> attention output [B*Lq, internal] bf16."""
        B, Lq, C = q.shape
        D = C // self.num_heads
        if D in (16, 32):
            return ops.attention_small(q, k, v, self.num_heads).reshape(B * Lq, C)
        H = self.num_heads
        sp = lambda t: t.view(B, t.shape[1], H, D).permute(0, 2, 1, 3)
        o = ops.attention(sp(q), sp(k), sp(v), splits=attn_splits(B, H, Lq, k.shape[1]))
        return o.permute(0, 2, 1, 3).reshape(B * Lq, C)

    def forward(self, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
        assert not (self.training and self.dropout_p > 0.0), "attention dropout (train mode) is outside the forward hot path"
        B, Lq, _ = q.shape
        f = lambda t, n: self.proj(n, to_bf16(t.reshape(-1, t.shape[-1]).contiguous())).view(B, t.shape[1], -1)
        o = self.core(f(q, "q"), f(k, "k"), f(v, "v"))
        return self.out(o, None).view(B, Lq, -1)


class RoPEAttention(Attention):
    """sam/transformer.py:266-331."""

    def __init__(self, *args, rope_theta=10000.0, rope_k_repeat=False, feat_sizes=(32, 32), **kwargs):
        super().__init__(*args, **kwargs)
        self.rope_theta = rope_theta
        self.rope_k_repeat = rope_k_repeat
        self.feat_sizes = feat_sizes
        self._tables = {}

    def table(self, n_q: int, device) -> Tuple[torch.Tensor, torch.Tensor]:
        """axial cos/sin table for a sqrt(n_q) x sqrt(n_q) query grid (transformer.py:299-305), generated on device once."""
        side = int(round(math.sqrt(n_q)))
        assert side * side == n_q, "RoPE attention expects a square token grid"
        key = (side, str(device))
        if key not in self._tables:
            self._tables[key] = ops.rope_table(side, self.internal_dim // self.num_heads, float(self.rope_theta), device)
        return self._tables[key]

    def proj_rope(self, which: str, x_bf16: torch.Tensor, B: int, rows_per_batch: int, n_rope: int, tab, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """q/k projection + RoPE -> bf16 [B, rows_per_batch, internal]; the rotation rides in the GEMM's store when the problem
        qualifies (ops.gemm_rope), otherwise it is applied in place after the projection.  out: write into this [B*rows, internal] view."""
        lin = getattr(self, which + "_proj")
        w, b = w_bf16(self._wc, which + "w", lin.weight), v_f32(self._wc, which + "b", lin.bias)
        M = x_bf16.shape[0]
        if _FUSED_ROPE and M >= 256 and rows_per_batch >= 128 and tab[0].shape[0] >= 128 and n_rope > 0 and M * self.internal_dim * 4 < 2 ** 31:
            return ops.gemm_rope(x_bf16, w, b, tab, rope_cols=self.internal_dim, head_dim=self.internal_dim // self.num_heads,
                                 rows_per_batch=rows_per_batch, n_rope=n_rope, out=out).view(B, rows_per_batch, -1)
        y = ops.gemm(x_bf16, w, b, out=out).view(B, rows_per_batch, -1)
        ops.rope_(y, n_rope, tab)
        return y

    def proj_rope_key_range(self, mem_k: torch.Tensor, n_ptr_tokens: int, tab, valid_keys: int, eff: int, share: Tuple[int, int]) -> torch.Tensor:
        """The rotated key projection for ONE RANK of a cross-GPU key split (parallel.KVSplit): only the keys of this rank's splits
        [share[0], share[1]) of an `eff`-way split over `valid_keys` keys are projected -- rounded out to whole position-table periods
        (memory entries of n_pos tokens), so the rotation of a projected row is the one the full call applies to it.  The other rows of the
        returned [B, Nk, internal] buffer are uninitialised: this rank's attention partials never read them.  (under
        the key split every rank used to project and rotate the whole bank -- ~1 ms of the replicated 2.7 ms per propagated slice at 1.06 M keys.)"""
        B, Nk, _ = mem_k.shape
        n_pos = tab[0].shape[0]
        tiles_total = -(-valid_keys // 32)                       # the kernels' split -> key-tile map (attention.hip: 32-key tiles)
        tiles_per = -(-tiles_total // eff)
        k0 = min(share[0] * tiles_per * 32, valid_keys)
        k1 = min(share[1] * tiles_per * 32, valid_keys)
        r0, r1 = (k0 // n_pos) * n_pos, min(-(-k1 // n_pos) * n_pos, Nk)
        kk = torch.empty(B, Nk, self.internal_dim, dtype=OP16, device=mem_k.device)
        if r1 > r0:
            n_rope = max(0, min(Nk - n_ptr_tokens, r1) - r0)     # rows past the spatial memories (object pointers) are not rotated
            for b in range(B):
                self.proj_rope("k", mem_k[b, r0:r1], 1, r1 - r0, n_rope, tab, out=kk[b, r0:r1])
        return kk

    def folds_values(self) -> bool:
        """True when the value projection can be folded out of the attention: P (M W_v^T + b_v) = (P M) W_v^T + b_v because the
        values carry no rotary encoding (transformer.py:318 sees v untouched) and softmax rows sum to one.  Built for the memory
        cross-attention's shape (kv_in_dim 64 -> one 256-wide head): the attention then runs on the 64-channel memory rows
        themselves (ops.attention_kv64) and v_proj rides in the out-projection (out_folded)."""
        return _FOLD_V and self.kv_in_dim == 64 and self.internal_dim == 256 and self.num_heads == 1

    def out_folded(self, o64: torch.Tensor, residual: Optional[torch.Tensor], out_dtype=F32) -> torch.Tensor:
        """out_proj(o64 W_v^T + b_v) as ONE K = 64 GEMM: weight W_o W_v and bias W_o b_v + b_o composed in fp32."""
        ps = [self.out_proj.weight, self.out_proj.bias, self.v_proj.weight, self.v_proj.bias]
        w = self._wc.get("ovw", ps, lambda: (self.out_proj.weight.detach().float() @ self.v_proj.weight.detach().float()).to(OP16).contiguous())
        b = self._wc.get("ovb", ps, lambda: (self.out_proj.weight.detach().float() @ self.v_proj.bias.detach().float()
                                             + self.out_proj.bias.detach().float()).contiguous())
        return ops.gemm(o64, w, b, residual=residual, out_dtype=out_dtype)

    def core_folded(self, q: torch.Tensor, k: torch.Tensor, mem_v: torch.Tensor, key_count: Optional[torch.Tensor] = None) -> torch.Tensor:
        """rotated q [B, Lq, 256], rotated k [B, Lk, 256], un-projected values mem_v [B, Lk, 64] -> 16-bit [B*Lq, 64].  key_count: int32
        device scalar, the number of leading keys attended to when the bank is padded to a capacity Lk (ops.attention_kv64)."""
        from .. import parallel
        B, Lq, C = q.shape
        Lk = k.shape[1]
        q4, k4, v4 = (q.view(B, Lq, 1, C).permute(0, 2, 1, 3), k.view(B, Lk, 1, C).permute(0, 2, 1, 3),
                      mem_v.reshape(B, Lk, 1, 64).permute(0, 2, 1, 3))
        splits = attn_splits(B, 1, Lq, Lk)
        kvs = parallel.current_kv_split()
        eff = ops.attention_effective_splits(Lk, splits) if kvs is not None else 1
        if eff > 1:
            # cross-GPU key split (parallel.KVSplit): this rank's share of the SAME partials one rank would compute, all-gather of the
            # (max, sum, O') slots, then the library's merge -- bit-identical to the single-rank call below
            ws = ops.attention_workspace(B, 1, Lq, 64, eff, q.device)
            s0, s1 = kvs.share(eff)
            ops.attention_kv64_partial(q4, k4, v4, splits=eff, split_begin=s0, split_count=s1 - s0, workspace=ws, key_count=key_count)
            kvs.exchange(ws, eff, B * Lq)
            o = torch.empty(B, Lq, 1, 64, dtype=OP16, device=q.device).permute(0, 2, 1, 3)
            ops.attention_merge(o, Lk, eff, ws)
        else:
            o = ops.attention_kv64(q4, k4, v4, splits=splits, key_count=key_count)
        return o.permute(0, 2, 1, 3).reshape(B * Lq, 64)

    def forward(self, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, num_k_exclude_rope: int = 0) -> torch.Tensor:
        assert not (self.training and self.dropout_p > 0.0), "attention dropout (train mode) is outside the forward hot path"
        B, Lq, _ = q.shape
        Lk = k.shape[1]
        if Lq != Lk:
            assert self.rope_k_repeat
        flat = lambda t: to_bf16(t.reshape(-1, t.shape[-1]).contiguous())
        tab = self.table(Lq, q.device)
        qp = self.proj_rope("q", flat(q), B, Lq, Lq, tab)
        kp = self.proj_rope("k", flat(k), B, Lk, Lk - num_k_exclude_rope, tab)
        vp = self.proj("v", flat(v)).view(B, Lk, -1)
        return self.out(self.core(qp, kp, vp), None).view(B, Lq, -1)


class MemoryAttentionLayer(nn.Module):
    """memory_attention.py:17-99 (pre-LN: RoPE self-attn, RoPE cross-attn to the memory bank, ReLU FFN)."""

    def __init__(self, activation: str, cross_attention: nn.Module, d_model: int, dim_feedforward: int, dropout: float,
                 pos_enc_at_attn: bool, pos_enc_at_cross_attn_keys: bool, pos_enc_at_cross_attn_queries: bool,
                 self_attention: nn.Module):
        super().__init__()
        assert activation == "relu" and not pos_enc_at_attn and pos_enc_at_cross_attn_keys and not pos_enc_at_cross_attn_queries, \
            "HIP path implements the YAML's layer wiring (relu; pos enc only on cross-attn keys)"
        self.d_model, self.dim_feedforward, self.dropout_value = d_model, dim_feedforward, dropout
        self.self_attn = self_attention
        self.cross_attn_image = cross_attention
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.dropout = nn.Dropout(dropout)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1, self.norm2, self.norm3 = nn.LayerNorm(d_model), nn.LayerNorm(d_model), nn.LayerNorm(d_model)
        self.dropout1, self.dropout2, self.dropout3 = nn.Dropout(dropout), nn.Dropout(dropout), nn.Dropout(dropout)
        self.activation_str = activation
        self.pos_enc_at_attn = pos_enc_at_attn
        self.pos_enc_at_cross_attn_queries = pos_enc_at_cross_attn_queries
        self.pos_enc_at_cross_attn_keys = pos_enc_at_cross_attn_keys
        self._wc = WeightCache()

    def _ln(self, name: str, x: torch.Tensor) -> torch.Tensor:
        n = getattr(self, name)
        return ops.layernorm(x, v_f32(self._wc, name + "w", n.weight), v_f32(self._wc, name + "b", n.bias), n.eps)

    def run(self, x: torch.Tensor, mem_k: torch.Tensor, mem_v: torch.Tensor, B: int, L: int, n_ptr_tokens: int,
            key_count: Optional[torch.Tensor] = None, kk: Optional[torch.Tensor] = None) -> torch.Tensor:
        """x fp32 [B*L, C]; mem_k (= memory + pos) / mem_v (= memory) bf16 [B, Nk, 64]; key_count: see RoPEAttention.core_folded.
        kk: this layer's rotated key projection [B, Nk, 256] (row stride free) when the caller projected all layers' keys in one GEMM
        (MemoryAttention.forward)."""
        wc, sa, ca = self._wc, self.self_attn, self.cross_attn_image
        C = self.d_model
        Nk = mem_k.shape[1]
        tab = sa.table(L, x.device)
        # self attention: fused q|k|v projection, RoPE on q and k rows in place
        t = self._ln("norm1", x)
        w_qkv = w_bf16(wc, "sqkv", sa.q_proj.weight, sa.k_proj.weight, sa.v_proj.weight)
        b_qkv = v_f32(wc, "sqkvb", sa.q_proj.bias, sa.k_proj.bias, sa.v_proj.bias)
        fused_rope = _FUSED_ROPE and B * L >= 256 and L >= 128 and sa.num_heads == 1 and B * L * 3 * C * 4 < 2 ** 31
        if fused_rope:   # q | k columns rotated in the GEMM's store, v columns untouched
            qkv = ops.gemm_rope(t, w_qkv, b_qkv, tab, rope_cols=2 * C, head_dim=C, rows_per_batch=L, n_rope=L).view(B, L, 3 * C)
        else:
            qkv = ops.gemm(t, w_qkv, b_qkv).view(B, L, 3 * C)
        q, k, v = qkv[:, :, :C], qkv[:, :, C:2 * C], qkv[:, :, 2 * C:]
        if not fused_rope:
            ops.rope_(q, L, tab)
            ops.rope_(k, L, tab)
        x = sa.out(sa.core(q, k, v), x)
        # cross attention to the memory bank (keys carry the position encoding, values do not)
        t = self._ln("norm2", x)
        q = ca.proj_rope("q", t, B, L, L, tab)
        from .. import parallel
        kvs = parallel.current_kv_split()
        eff = ops.attention_effective_splits(Nk, attn_splits(B, 1, L, Nk)) if (kvs is not None and ca.folds_values()) else 1
        valid = Nk if key_count is None else (kvs.host_key_count if kvs is not None else None)
        if kk is not None:
            pass                                            # projected for all layers at once by the caller
        elif eff > 1 and valid is not None and (Nk - n_ptr_tokens) % tab[0].shape[0] == 0:
            # cross-GPU key split: this rank projects + rotates only the keys its attention partials read
            kk = ca.proj_rope_key_range(mem_k, n_ptr_tokens, tab, int(valid), eff, kvs.share(eff))
        else:
            kk = ca.proj_rope("k", mem_k.reshape(B * Nk, -1), B, Nk, Nk - n_ptr_tokens, tab)
        if ca.folds_values():
            # O' = softmax(q k^T) M on the 64-channel memory rows; v_proj is folded into the out-projection (5/8 of the MFMA work)
            x = ca.out_folded(ca.core_folded(q, kk, mem_v, key_count), x)
        else:
            if key_count is not None:
                raise RuntimeError("a padded memory bank (key_count) needs the value-folded cross-attention (MSAM2_NO_VALUE_FOLD is set)")
            vv = ca.proj("v", mem_v.reshape(B * Nk, -1)).view(B, Nk, C)
            x = ca.out(ca.core(q, kk, vv), x)
        # FFN
        t = self._ln("norm3", x)
        h = ops.gemm(t, w_bf16(wc, "f1", self.linear1.weight), v_f32(wc, "f1b", self.linear1.bias), act=ops.ACT_RELU)
        return ops.gemm(h, w_bf16(wc, "f2", self.linear2.weight), v_f32(wc, "f2b", self.linear2.bias), residual=x, out_dtype=F32)

    def forward(self, tgt, memory, pos: Optional[torch.Tensor] = None, query_pos: Optional[torch.Tensor] = None,
                num_k_exclude_rope: int = 0) -> torch.Tensor:
        assert not self.training or self.dropout_value == 0.0, \
            "train-mode dropout runs through MemoryAttention.forward / training.* (backward.memory_attention_forward_saved), not a bare layer"
        B, L, C = tgt.shape
        mem_k = ops.add_cast(memory, pos, 1.0, OP16)
        mem_v = ops.add_cast(memory, None, 1.0, OP16)
        x = ops.add_cast(tgt, None, 1.0, F32).reshape(B * L, C)
        return self.run(x, mem_k, mem_v, B, L, num_k_exclude_rope).view(B, L, C)


class MemoryAttention(nn.Module):
    """memory_attention.py:102-169; seq-first [L, B, C] in and out, like the reference."""

    def __init__(self, d_model: int, pos_enc_at_input: bool, layer: nn.Module, num_layers: int, batch_first: bool = True):
        super().__init__()
        assert batch_first
        self.d_model = d_model
        self.layers = nn.ModuleList([copy.deepcopy(layer) for _ in range(num_layers)])
        self.num_layers = num_layers
        self.norm = nn.LayerNorm(d_model)
        self.pos_enc_at_input = pos_enc_at_input
        self.batch_first = batch_first
        self._wc = WeightCache()
        self.dropout_seed = 0            # stream of the train-mode dropout masks; every train-mode forward draws a fresh sub-stream
        self._dropout_calls = 0          # host-visible count of eager train-mode forwards (reset it together with dropout_seed to replay a stream)
        self._drop_ctr = None            # the count the kernels read: int64 [1] on the device, advanced by a kernel of every forward
        self._drop_ctr_host = -1

    def next_dropout(self):
        """(p, seed) of the next train-mode forward, or None in eval mode / with dropout 0 (memory_attention.py:40-48: nn.Dropout(0.1)
        in every layer, transformer.py:317-318: dropout_p on the attention probabilities while self.training).
        The seed is (dropout_seed << 32) + the number of train-mode forwards so far, and that number lives ON THE DEVICE
        (`ops.DeviceSeed`): it is advanced by a kernel of this call and snapshotted for this forward (and its backward), so a training
        step captured into a hipGraph draws fresh masks on every replay instead of re-applying the captured ones."""
        p = float(self.layers[0].dropout_value) if len(self.layers) else 0.0
        if not self.training or p <= 0.0:
            return None
        dev = self.norm.weight.device
        if self._drop_ctr is None or self._drop_ctr.device != dev or self._drop_ctr_host != self._dropout_calls:
            # first use, a moved module, or a caller re-positioned the stream through `_dropout_calls`: (re)load the device counter
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("the first train-mode forward of a MemoryAttention must run eagerly (it creates the device-side dropout counter)")
            self._drop_ctr = torch.full((1,), int(self._dropout_calls), dtype=torch.int64, device=dev)
        snap = torch.empty(1, dtype=torch.int64, device=dev)
        ops.counter_bump(self._drop_ctr, snap)
        self._dropout_calls += 1
        self._drop_ctr_host = self._dropout_calls
        return p, ops.DeviceSeed(int(self.dropout_seed) << 32, snap)

    def _project_all_keys(self, mem_k: torch.Tensor, B: int, L: int, n_ptr_tokens: int) -> Optional[torch.Tensor]:
        """The rotated key projections of ALL layers' cross-attentions as ONE GEMM: the four layers project the same 64-channel bank rows
        (memory + position) with their own k_proj, i.e. one [layers * 256, 64] weight, and each layer then reads its 256 columns as a
        strided view.  MEASURED AND SWITCHED OFF (round 3, MSAM2_BATCHED_KPROJ=1 enables it): one launch instead of four, but the step
        got slower (6.67 -> 6.75 ms, twice, alternating runs) and the 64-slice volume much slower (294 -> 259 slices/s) -- the attention's
        K tiles then come from rows 2 KB apart in a buffer four times the size, which costs more than three launches save.  Not under the cross-GPU key split (each rank projects only its own key range there) and not when the
        fused output would pass the 32-bit offsets of the RoPE store (banks beyond ~500 k keys keep the per-layer launches)."""
        from .. import parallel
        n = len(self.layers)
        Nk = mem_k.shape[1]
        cas = [l.cross_attn_image for l in self.layers]
        if not (_BATCH_K_PROJ and n > 1 and parallel.current_kv_split() is None and all(isinstance(c, RoPEAttention) and c.folds_values() for c in cas)):
            return None
        C = cas[0].internal_dim
        tab = cas[0].table(L, mem_k.device)
        M = B * Nk
        if not (_FUSED_ROPE and M >= 256 and Nk >= 128 and tab[0].shape[0] >= 128 and Nk - n_ptr_tokens > 0 and M * n * C * 4 < 2 ** 31
                and len({(c.rope_theta, c.num_heads) for c in cas}) == 1):
            return None
        w = w_bf16(self._wc, "kall", *[c.k_proj.weight for c in cas])
        b = v_f32(self._wc, "kallb", *[c.k_proj.bias for c in cas])
        out = ops.gemm_rope(mem_k.reshape(M, -1), w, b, tab, rope_cols=n * C, head_dim=C // cas[0].num_heads, rows_per_batch=Nk,
                            n_rope=Nk - n_ptr_tokens)
        return out.view(B, Nk, n, C)

    def forward(self, curr: torch.Tensor, memory: torch.Tensor, curr_pos: Optional[torch.Tensor] = None,
                memory_pos: Optional[torch.Tensor] = None, num_obj_ptr_tokens: int = 0, key_count: Optional[torch.Tensor] = None):
        """key_count (not in the reference's signature): int32 device scalar, number of valid leading rows of a `memory` padded to a
        fixed capacity; num_obj_ptr_tokens then counts the padded tail too (it only marks where RoPE stops)."""
        if isinstance(curr, list):
            assert isinstance(curr_pos, list) and len(curr) == len(curr_pos) == 1
            curr, curr_pos = curr[0], curr_pos[0]
        assert curr.shape[1] == memory.shape[1], "Batch size must be the same for curr and memory"
        L, B, C = curr.shape
        from .. import autograd as ag
        if ag.active(self) and key_count is None:
            # train() + grad mode (the reference's training loops): the grad-carrying form, dropout drawn inside
            pos = curr_pos if (curr_pos is not None and self.pos_enc_at_input) else torch.zeros_like(curr)
            return ag.memory_attention(self, curr, pos, memory, memory_pos, num_obj_ptr_tokens)
        drop = self.next_dropout()
        if drop is not None:
            # train mode: the dropout-carrying path of the training steps (backward.memory_attention_forward_saved), forward half only
            from .. import backward as bwd
            assert key_count is None, "padded memory banks are an inference-path feature"
            zero = torch.zeros_like(curr) if curr_pos is None or not self.pos_enc_at_input else curr_pos
            return bwd.memory_attention_forward_saved(self, curr, zero, memory, memory_pos, num_obj_ptr_tokens, dropout=drop)[0]
        # seq-first -> batch-first happens inside the add/cast kernels (strided reads), no separate transpose
        use_pos = self.pos_enc_at_input and curr_pos is not None
        x = ops.add_cast(curr.transpose(0, 1), curr_pos.transpose(0, 1) if use_pos else None, 0.1, F32).reshape(B * L, C)
        mem_bf = memory.transpose(0, 1)
        mem_k = ops.add_cast(mem_bf, memory_pos.transpose(0, 1), 1.0, OP16)
        mem_v = ops.add_cast(mem_bf, None, 1.0, OP16)
        kk_all = self._project_all_keys(mem_k, B, L, num_obj_ptr_tokens)
        for i, layer in enumerate(self.layers):
            x = layer.run(x, mem_k, mem_v, B, L, num_obj_ptr_tokens, key_count, None if kk_all is None else kk_all[:, :, i])
        y = ops.layernorm(x, v_f32(self._wc, "nw", self.norm.weight), v_f32(self._wc, "nb", self.norm.bias), self.norm.eps,
                          out_dtype=F32)
        return y.view(B, L, C).transpose(0, 1)


# ---------------------------------------------------------------------------------------------------------------------
class MaskDownSampler(nn.Module):
    """memory_encoder.py:17-58 (conv k3 s2 p1 + LayerNorm2d + GELU, x4, then 1x1 to embed_dim)."""

    def __init__(self, embed_dim=256, kernel_size=4, stride=4, padding=0, total_stride=16, activation=nn.GELU):
        super().__init__()
        assert (kernel_size, stride, padding, total_stride) == (3, 2, 1, 16) and activation is nn.GELU, \
            "HIP path implements the YAML's k3/s2/p1 mask down-sampler"
        num_layers = int(math.log2(total_stride) // math.log2(stride))
        self.encoder = nn.Sequential()
        cin = 1
        for _ in range(num_layers):
            cout = cin * (stride ** 2)
            self.encoder.append(nn.Conv2d(cin, cout, kernel_size=kernel_size, stride=stride, padding=padding))
            self.encoder.append(LayerNorm2d(cout))
            self.encoder.append(activation())
            cin = cout
        self.encoder.append(nn.Conv2d(cin, embed_dim, kernel_size=1))
        self._wc = WeightCache()

    def run(self, mask: torch.Tensor, mode: int, scale: float, bias: float) -> torch.Tensor:
        """fp32 mask [n,1,S,S] (raw logits when mode != 0) -> fp32 tokens [n*(S/16)^2, embed_dim]."""
        n, _, S, _ = mask.shape
        wc, enc = self._wc, self.encoder
        f = lambda key, p: v_f32(wc, key, p)
        h = mask.to(F32).contiguous().reshape(n * S * S, 1)
        conv, ln = enc[0], enc[1]
        # stage 1 (1 -> 4 channels, reads the full-resolution mask once): direct kernel with the mask transform fused in
        h = ops.conv3x3s2_ln_gelu(h, n, S, S, wc.get("cw0", [conv.weight], lambda: conv.weight.detach().float().contiguous()),
                                  f("cb0", conv.bias), f("lw0", ln.weight), f("lb0", ln.bias), mode, scale, bias)
        side = S // 2
        # stages 2-4 (4->16, 16->64, 64->256): im2col + MFMA GEMM, LayerNorm2d + GELU on the fp32 result
        for j in range(1, 4):
            conv, ln = enc[3 * j], enc[3 * j + 1]
            cols = ops.im2col3x3s2(h, n, side, side)

            def pack(c=conv, ld=cols.shape[1]):
                w = c.weight.detach().permute(0, 2, 3, 1).reshape(c.weight.shape[0], -1)
                out = torch.zeros(w.shape[0], ld, dtype=OP16, device=w.device)
                out[:, : w.shape[1]] = w.to(OP16)
                return out
            g = ops.gemm(cols, wc.get(f"cw{j}", [conv.weight], pack), f(f"cb{j}", conv.bias), out_dtype=F32)
            h = ops.layernorm(g, f(f"lw{j}", ln.weight), f(f"lb{j}", ln.bias), ln.eps, act=ops.ACT_GELU)
            side //= 2
        return ops.gemm(h, w_bf16(wc, "pw", enc[12].weight), f("pb", enc[12].bias), out_dtype=F32)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        n, _, S, _ = x.shape
        return nchw_view(self.run(x, 0, 0.0, 0.0), n, S // 16, S // 16)


class CXBlock(nn.Module):
    """memory_encoder.py:62-117 (ConvNeXt block: dw 7x7 -> LN -> 1x1 -> GELU -> 1x1 -> layer scale -> residual)."""

    def __init__(self, dim, kernel_size=7, padding=3, drop_path=0.0, layer_scale_init_value=1e-6, use_dwconv=True):
        super().__init__()
        assert kernel_size == 7 and padding == 3 and use_dwconv and drop_path == 0.0 and layer_scale_init_value > 0
        self.dwconv = nn.Conv2d(dim, dim, kernel_size=7, padding=3, groups=dim)
        self.norm = LayerNorm2d(dim, eps=1e-6)
        self.pwconv1 = nn.Linear(dim, 4 * dim)
        self.act = nn.GELU()
        self.pwconv2 = nn.Linear(4 * dim, dim)
        self.gamma = nn.Parameter(layer_scale_init_value * torch.ones(dim), requires_grad=True)
        self._wc = WeightCache()

    def run(self, x: torch.Tensor, n: int, H: int, W: int) -> torch.Tensor:
        """fp32 tokens [n*H*W, C] -> same."""
        wc = self._wc
        C = x.shape[1]
        wt = wc.get("dw", [self.dwconv.weight], lambda: self.dwconv.weight.detach().reshape(C, 49).t().float().contiguous())
        h = ops.dwconv7x7_ln(x, n, H, W, wt, v_f32(wc, "dwb", self.dwconv.bias), v_f32(wc, "lw", self.norm.weight),
                             v_f32(wc, "lb", self.norm.bias))
        h = ops.gemm(h, w_bf16(wc, "w1", self.pwconv1.weight), v_f32(wc, "b1", self.pwconv1.bias), act=ops.ACT_GELU)
        return ops.gemm(h, w_bf16(wc, "w2", self.pwconv2.weight), v_f32(wc, "b2", self.pwconv2.bias),
                        colscale=v_f32(wc, "g", self.gamma), residual=x, out_dtype=F32)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        n, C, H, W = x.shape
        return nchw_view(self.run(tokens_of(x.to(F32)), n, H, W), n, H, W)


class Fuser(nn.Module):
    """memory_encoder.py:120-135."""

    def __init__(self, layer, num_layers, dim=None, input_projection=False):
        super().__init__()
        assert not input_projection, "SAM2 configs use no input projection in the fuser"
        self.proj = nn.Identity()
        self.layers = nn.ModuleList([copy.deepcopy(layer) for _ in range(num_layers)])

    def run(self, x, n, H, W):
        for layer in self.layers:
            x = layer.run(x, n, H, W)
        return x

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        n, C, H, W = x.shape
        return nchw_view(self.run(tokens_of(x.to(F32)), n, H, W), n, H, W)


class MemoryEncoder(nn.Module):
    """memory_encoder.py:138-181."""

    def __init__(self, out_dim, mask_downsampler, fuser, position_encoding, in_dim=256):
        super().__init__()
        self.mask_downsampler = mask_downsampler
        self.pix_feat_proj = nn.Conv2d(in_dim, in_dim, kernel_size=1)
        self.fuser = fuser
        self.position_encoding = position_encoding
        self.out_proj = nn.Identity()
        if out_dim != in_dim:
            self.out_proj = nn.Conv2d(in_dim, out_dim, kernel_size=1)
        self._wc = WeightCache()

    def run(self, pix_tokens: torch.Tensor, mask: torch.Tensor, mode: int, scale: float, bias: float, n: int, H: int, W: int):
        """pix_tokens [n*H*W, C] (fp32 or bf16), mask fp32 [n,1,16H,16W] -> (fp32 tokens [n*H*W, out_dim])."""
        wc = self._wc
        m = self.mask_downsampler.run(mask, mode, scale, bias)
        x = ops.gemm(to_bf16(pix_tokens), w_bf16(wc, "pw", self.pix_feat_proj.weight), v_f32(wc, "pb", self.pix_feat_proj.bias),
                     residual=m, out_dtype=F32)
        x = self.fuser.run(x, n, H, W)
        if isinstance(self.out_proj, nn.Identity):
            return x
        return ops.gemm(to_bf16(x), w_bf16(wc, "ow", self.out_proj.weight), v_f32(wc, "ob", self.out_proj.bias), out_dtype=F32)

    def forward(self, pix_feat: torch.Tensor, masks: torch.Tensor, skip_mask_sigmoid: bool = False):
        n, C, H, W = pix_feat.shape
        y = self.run(tokens_of(pix_feat), masks, 0 if skip_mask_sigmoid else 1, 1.0, 0.0, n, H, W)
        x = nchw_view(y, n, H, W)
        pos = self.position_encoding(x).to(x.dtype)
        return {"vision_features": x, "vision_pos_enc": [pos]}
