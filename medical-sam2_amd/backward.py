"""Explicit, recomputing backward pass on the HIP path: nn.LayerNorm, nn.Linear, the Linear-act-Linear MLP
(sam2_utils.py:108-132; hieradet.py:96-106,160-166; memory_attention.py:43-47,96), attention (flash-style `msam2_attention_bwd` for head
dims 64-256, the fused `msam2_attention_small_bwd` for the decoder's 16 / 32-wide heads, a materialised GEMM-composed fallback otherwise),
and on top of them the whole `MemoryAttention`, `MaskDecoder` (two-way transformer, transposed convolutions, hyper-network MLPs) and
`MemoryEncoder` modules.  Input gradients run on the forward GEMM kernels (`ops.gemm(dy, W^T)`), weight and bias gradients on
`msam2_gemm_tt` (k-major operands: no transposed copies), the rest on the kernels of csrc/backward.hip / csrc/attention_bwd.hip.
The forward saves nothing: each module backward re-runs its forward keeping the intermediates it needs.

Operands are 16-bit (ops.OP16) like the forward: with the default fp16 build callers must keep gradients in fp16 range (loss scaling:
see training.py); the bf16 build has fp32's range.  Parity: tests/test_backward_gpu.py against torch.autograd on the fp32 oracle
primitives, tests/test_grads_golden.py against the reference's own `.grad`.
"""
from __future__ import annotations

import os

from typing import Optional, Tuple

import torch

from . import ops
from ._lib import check, lib
from .ops import F32, OP16, _is_bf16, _p, _req, _stream


def transpose16(x: torch.Tensor, pad_rows_to: int = 1) -> torch.Tensor:
    """16-bit [R, C] (row-major, possibly strided rows) -> contiguous [C, R'] with R' = R rounded up to a multiple of `pad_rows_to`
    (extra columns zero: as a GEMM operand they add nothing to the reduction, and the row length stays 16-byte aligned)."""
    _req(x.dim() == 2 and x.dtype == OP16 and x.stride(1) == 1, "transpose16: 16-bit row-major matrix")
    R, C = x.shape
    Rp = -(-R // pad_rows_to) * pad_rows_to
    out = torch.empty(C, Rp, dtype=OP16, device=x.device) if Rp == R else torch.zeros(C, Rp, dtype=OP16, device=x.device)
    check(lib().msam2_transpose16(_p(x), x.stride(0), _p(out), out.stride(0), R, C, _stream()))
    return out


def colsum(x: torch.Tensor) -> torch.Tensor:
    """fp32 [C] = column sums of a [R, C] matrix (16-bit or fp32)."""
    _req(x.dim() == 2 and x.stride(1) == 1, "colsum: row-major matrix")
    out = torch.zeros(x.shape[1], dtype=F32, device=x.device)
    check(lib().msam2_colsum(_p(x), _is_bf16(x), x.stride(0), _p(out), x.shape[0], x.shape[1], _stream()))
    return out


def act_backward(pre: torch.Tensor, dy: torch.Tensor, act: int) -> torch.Tensor:
    """16-bit dy * act'(pre); act = ops.ACT_GELU | ops.ACT_RELU."""
    _req(pre.shape == dy.shape and pre.is_contiguous() and dy.is_contiguous(), "act_backward: contiguous tensors of one shape")
    out = torch.empty(pre.shape, dtype=OP16, device=pre.device)
    check(lib().msam2_act_bwd(_p(pre), _is_bf16(pre), _p(dy), _is_bf16(dy), _p(out), pre.numel(), act, _stream()))
    return out


def layernorm_backward(x: torch.Tensor, gamma: torch.Tensor, dy: torch.Tensor, eps: float,
                       add: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """x fp32 [rows, C] (the LayerNorm input), dy [rows, C] (16-bit or fp32) -> (dx fp32, dgamma fp32 [C], dbeta fp32 [C]).
    add (fp32 [rows, C]): gradient of a residual path around the norm, added into dx in the same pass (pre-LN blocks: dx = d_res + LN'(dy))."""
    _req(x.dim() == 2 and x.dtype == F32 and x.stride(1) == 1 and dy.shape == x.shape and dy.stride(1) == 1, "layernorm_backward shapes")
    _req(add is None or (add.dtype == F32 and add.shape == x.shape and add.stride(1) == 1), "layernorm_backward: add must be fp32 [rows, C]")
    rows, C = x.shape
    dx = torch.empty(rows, C, dtype=F32, device=x.device)
    dgb = torch.zeros(2, C, dtype=F32, device=x.device)
    check(lib().msam2_layernorm_bwd(_p(x), x.stride(0), _p(dy), _is_bf16(dy), dy.stride(0), _p(gamma), _p(dx), dx.stride(0), _p(dgb[0]), _p(dgb[1]),
                                    rows, C, eps, _p(add), add.stride(0) if add is not None else 0, _stream()))
    return dx, dgb[0], dgb[1]


def _op16(t: torch.Tensor) -> torch.Tensor:
    return t if t.dtype == OP16 else ops.add_cast(t.reshape(1, t.shape[0], t.shape[1]), None, 1.0, OP16)[0]


def linear_backward(x: torch.Tensor, w: torch.Tensor, dy: torch.Tensor, need_dx: bool = True, dx_dtype: torch.dtype = F32):
    """y = x W^T + b with x 16-bit [M, K], W 16-bit [N, K]; dy [M, N] (any float type).  Returns (dx [M, K] or None, dW fp32 [N, K],
    db fp32 [N])."""
    dy16 = _op16(dy)
    _req(w.shape[0] % 8 == 0, "linear_backward: out_features must be a multiple of 8 (GEMM reduction length of dX)")
    dx = gemm_nt(dy16, w, out_dtype=dx_dtype) if need_dx else None                       # [M, N] @ W [N, K], W as the forward stores it
    dw, db = gemm_tt(dy16, x, a_colsum=True)
    return dx, dw, db


def gemm_nt(a: torch.Tensor, b: torch.Tensor, out_dtype: torch.dtype = F32, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    """[M, N] = residual + a [M, K] @ b [K, N] with b k-major (16-bit operands, rows possibly strided): the input gradient dX = dY W of a
    linear layer straight from its weight [out, in] (`msam2_gemm_nt`) -- no transposed weight copy.  Reductions that are not a multiple of
    64 (the 96-wide layers of Hiera's first stage) go through the transposed copy and the forward GEMM."""
    _req(a.dim() == 2 and b.dim() == 2 and a.shape[1] == b.shape[0] and a.dtype == OP16 and b.dtype == OP16, "gemm_nt: 16-bit [M, K], [K, N]")
    _req(a.stride(1) == 1 and b.stride(1) == 1, "gemm_nt: row-major operands")
    M, K = a.shape
    N = b.shape[1]
    if K % 64 or N % 8 or a.stride(0) % 8 or b.stride(0) % 8 or a.data_ptr() % 16 or b.data_ptr() % 16:
        return ops.gemm(a, transpose16(b), residual=residual, out_dtype=out_dtype)
    out = torch.empty(M, N, dtype=out_dtype, device=a.device)
    if residual is not None:
        _req(residual.dtype == F32 and residual.shape == (M, N) and residual.stride(1) == 1, "gemm_nt: residual must be fp32 [M, N]")
    check(lib().msam2_gemm_nt(_p(a), a.stride(0), _p(b), b.stride(0), None, _p(residual), residual.stride(0) if residual is not None else 0,
                              _p(out), out.stride(0), _is_bf16(out), M, N, K, _stream()))
    return out


def gemm_tt(a: torch.Tensor, b: torch.Tensor, a_colsum: bool = False):
    """fp32 [Ma, Nb] = a^T b for 16-bit a [K, Ma], b [K, Nb] (both k-major, rows possibly strided): dW = dY^T X without transposed
    copies of the token-major operands (`msam2_gemm_tt`; the reduction over the K tokens is split over workgroups).  a_colsum=True also
    returns the column sums of a (fp32 [Ma]: the bias gradient when a = dY) from the same pass."""
    _req(a.dim() == 2 and b.dim() == 2 and a.shape[0] == b.shape[0] and a.dtype == OP16 and b.dtype == OP16, "gemm_tt: 16-bit [K, M], [K, N]")
    _req(a.stride(1) == 1 and b.stride(1) == 1, "gemm_tt: row-major operands")
    arena = _ARENA[0]
    if arena is not None:
        # outputs carved from a buffer zeroed once for the whole backward pass: the library adds into them, no zeroing launch per call
        out = arena.take(a.shape[1] * b.shape[1], a.device).view(a.shape[1], b.shape[1])
        cs = arena.take(a.shape[1], a.device) if a_colsum else None
        check(lib().msam2_gemm_tt_acc(_p(a), a.stride(0), _p(b), b.stride(0), _p(out), out.stride(0), _p(cs), a.shape[1], b.shape[1], a.shape[0],
                                      _stream()))
        return (out, cs) if a_colsum else out
    out = torch.empty(a.shape[1], b.shape[1], dtype=F32, device=a.device)
    cs = torch.empty(a.shape[1], dtype=F32, device=a.device) if a_colsum else None     # (zeroed by the library)
    check(lib().msam2_gemm_tt(_p(a), a.stride(0), _p(b), b.stride(0), _p(out), out.stride(0), _p(cs), a.shape[1], b.shape[1], a.shape[0], _stream()))
    return (out, cs) if a_colsum else out


class ZeroArena:
    """fp32 outputs of the weight-gradient GEMMs carved from large buffers that are zeroed ONCE (one fill per 64 MB instead of one
    zeroing launch per GEMM).  `with zero_arena():` around a backward pass; the gradients are views that keep their buffer alive."""
    CHUNK = 16 << 20                                   # elements (64 MB)

    def __init__(self):
        self.buf, self.off = None, 0

    def take(self, numel: int, device) -> torch.Tensor:
        if self.buf is None or self.off + numel > self.buf.numel() or self.buf.device != device:
            self.buf, self.off = torch.zeros(max(numel, self.CHUNK), dtype=F32, device=device), 0
        v = self.buf[self.off:self.off + numel]
        self.off += -(-numel // 64) * 64               # 256-byte aligned slices
        return v


_ARENA = [None]


class zero_arena:
    def __enter__(self):
        import os
        self._prev, _ARENA[0] = _ARENA[0], (None if os.environ.get("MSAM2_NO_ARENA") else ZeroArena())
        return _ARENA[0]

    def __exit__(self, *exc):
        _ARENA[0] = self._prev
        return False


def mlp_backward(x: torch.Tensor, w1: torch.Tensor, b1: torch.Tensor, w2: torch.Tensor, b2: Optional[torch.Tensor], dy: torch.Tensor,
                 act: int):
    """y = W2 act(W1 x + b1) + b2 (two-layer MLP of hieradet / memory attention / two-way blocks).  The hidden activations are
    recomputed (one extra forward GEMM pair) instead of being saved by the forward.  Returns (dx fp32, dW1, db1, dW2, db2)."""
    h = ops.gemm(x, w1, b1, act=act)
    # act'(pre): ReLU's is the sign of h itself; GELU's needs the pre-activations -- recomputed into 16 bits (the W-stationary kernel takes
    # the shape; an fp32 [tokens, hidden] map is 100 MB per stage-3 block).  The hidden gradient likewise stays in 16 bits: it is
    # multiplied by act' and rounded to the GEMM operand type right behind it.
    pre = h if act == ops.ACT_RELU else ops.gemm(x, w1, b1)
    dh, dw2, db2 = linear_backward(h, w2, dy, dx_dtype=OP16)
    dpre = act_backward(pre, dh, act)
    dx, dw1, db1 = linear_backward(x, w1, dpre)
    return dx, dw1, db1, dw2, db2


def attention_forward_lse(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, scale: Optional[float] = None, dropout: Optional[tuple] = None,
                          out: Optional[torch.Tensor] = None):
    """Forward attention on [B,H,L,D] 16-bit views that keeps what the flash-style backward needs: (o 16-bit [B,H,Lq,D] view of a
    [B,Lq,H,D] buffer, lse fp32 [B,H,Lq] in the log2 domain).  Hand both to `attention_backward(..., o_lse=...)`.
    dropout = (p, seed, offset): dropout on the probabilities inside the kernel (pass the same triple to `attention_backward`)."""
    from .modeling.common import attn_splits
    B, H, Lq, D = q.shape
    lse = torch.empty(B, H, Lq, dtype=F32, device=q.device)
    o = ops.attention(q, k, v, scale=scale if scale is not None else D ** -0.5, splits=attn_splits(B, H, Lq, k.shape[2]), lse=lse, dropout=dropout,
                      out=out)                                   # out: a 16-bit [B,H,Lq,D] view to write into (any strides ops.attention takes)
    return o, lse


def flash_dropout_supported(D: int, Lq: int) -> bool:
    """the flash kernels carry the dropout mask generator for head dims 96 / 128 / 256 and more than 64 queries"""
    return D in (96, 128, 256) and Lq > 64 and not os.environ.get("MSAM2_MATERIALISED_DROPOUT") and not os.environ.get("MSAM2_MATERIALISED_BWD")


def attention_backward(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, do: torch.Tensor, scale: Optional[float] = None, o_lse=None,
                       dropout: Optional[tuple] = None, out: Optional[tuple] = None):
    """Gradients of o = softmax(q k^T * scale) v for 16-bit q [B,H,Lq,D], k/v [B,H,Lk,D], upstream do [B,H,Lq,D] (any float type).
    Returns (dq, dk, dv) in fp32, shaped like q / k / v.
    Head dims 64 / 96 / 128 / 256: FLASH-STYLE (`msam2_attention_bwd`, csrc/attention_bwd.hip) -- the forward is re-run for O and the log-sum-exp rows,
    then three recomputing passes (dQ, dK, dV) keep every score tile in registers; O(L) memory.
    Other head dims (D % 8 == 0), or MSAM2_MATERIALISED_BWD=1: MATERIALISED form -- per (batch, head) the [Lq, Lk] scores live in
    HBM (fp32 S, 16-bit P / dS), the five products run on the forward GEMM kernel and the softmax and its Jacobian on two row
    kernels (reduction dims zero-padded to multiples of 8).
    o_lse: the (o, lse) pair of `attention_forward_lse` on the same q, k, v when the caller has already run it (saves the re-run).
    dropout = (p, seed, offset): the forward ran with dropout on the probabilities (`attention_forward_lse(..., dropout=...)`); flash path only."""
    B, H, Lq, D = q.shape
    Lk = k.shape[2]
    _req(q.dtype == OP16 and k.dtype == OP16 and v.dtype == OP16, "attention_backward: 16-bit q, k, v")
    _req(D % 8 == 0, "attention_backward: head dim must be a multiple of 8")
    scale = scale if scale is not None else D ** -0.5
    if out is not None:
        # fp32 [B,H,L,D] VIEWS to write into (e.g. the column thirds of a fused-qkv gradient buffer): any batch / head / token strides
        # that are multiples of 4 elements, channels contiguous
        dq, dk, dv = out
        _req(all(t.dtype == F32 and t.stride(3) == 1 and all(st % 4 == 0 for st in t.stride()[:3]) for t in out) and
             dq.shape == q.shape and dk.shape == k.shape and dv.shape == v.shape, "attention_backward: out must be fp32 views shaped like q, k, v")
    else:
        dq = torch.empty(B, H, Lq, D, dtype=F32, device=q.device)
        dk = torch.empty(B, H, Lk, D, dtype=F32, device=q.device)
        dv = torch.empty(B, H, Lk, D, dtype=F32, device=q.device)
    if D in (64, 96, 128, 256) and not os.environ.get("MSAM2_MATERIALISED_BWD"):
        q, k, v = (t if t.stride(3) == 1 and all(st % 8 == 0 for st in t.stride()[:3]) else t.contiguous() for t in (q, k, v))
        o, lse = o_lse if o_lse is not None else attention_forward_lse(q, k, v, scale, dropout=dropout)
        g = do.to(F32)
        g = g if g.stride(3) == 1 and all(st % 4 == 0 for st in g.stride()[:3]) else g.contiguous()
        nbytes = lib().msam2_attention_bwd_workspace_bytes(B, H, Lq, D)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=q.device)
        if dropout is not None and dropout[0] > 0:
            pd, seed, offset = dropout
            seed_dev = None
            if isinstance(seed, ops.DeviceSeed):
                seed, seed_dev = seed.base, seed.dev
            check(lib().msam2_attention_bwd_dropout(_p(q), ops._strides3(q), _p(k), ops._strides3(k), _p(v), ops._strides3(v), _p(o), ops._strides3(o),
                                                    _p(lse), _p(g), ops._strides3(g), _p(dq), ops._strides3(dq), _p(dk), ops._strides3(dk), _p(dv),
                                                    ops._strides3(dv), _p(ws), nbytes, B, H, Lq, Lk, D, float(scale), float(pd),
                                                    int(seed) & (2 ** 64 - 1), int(offset) & (2 ** 64 - 1), _p(seed_dev), _stream()))
            return dq, dk, dv
        check(lib().msam2_attention_bwd(_p(q), ops._strides3(q), _p(k), ops._strides3(k), _p(v), ops._strides3(v), _p(o), ops._strides3(o),
                                        _p(lse), _p(g), ops._strides3(g), _p(dq), ops._strides3(dq), _p(dk), ops._strides3(dk), _p(dv),
                                        ops._strides3(dv), _p(ws), nbytes, B, H, Lq, Lk, D, float(scale), _stream()))
        return dq, dk, dv
    _req(dropout is None or dropout[0] == 0, "attention_backward: dropout needs the flash path (attention_dropout_backward is the materialised form)")
    # P / dS are GEMM operands with the keys as reduction dim: rows padded to a multiple of 8 keys with zeros (memory banks hold
    # 4 tokens per object pointer, so Lk is only a multiple of 4)
    Lkp = -(-Lk // 8) * 8
    Pbuf = torch.zeros(Lq, Lkp, dtype=OP16, device=q.device)
    dSbuf = torch.zeros(Lq, Lkp, dtype=OP16, device=q.device)
    P, dS = Pbuf[:, :Lk], dSbuf[:, :Lk]
    for b in range(B):
        for h in range(H):
            qm, km, vm = (t[b, h] if t[b, h].stride(1) == 1 and t[b, h].stride(0) % 8 == 0 else t[b, h].contiguous() for t in (q, k, v))
            dom = _op16(do[b, h].contiguous())
            S = ops.gemm(qm, km, out_dtype=F32)                                   # [Lq, Lk]
            check(lib().msam2_softmax_rows(_p(S), S.stride(0), _p(P), P.stride(0), Lq, Lk, scale, _stream()))
            ops.gemm(transpose16(P, 8), transpose16(dom, 8), out=dv[b, h])       # dV = P^T dO        (reduction over Lq, padded)
            dP = ops.gemm(dom, vm, out_dtype=F32)                                 # dO V^T
            check(lib().msam2_softmax_bwd_rows(_p(P), P.stride(0), _p(dP), dP.stride(0), _p(dS), dS.stride(0), Lq, Lk, scale, _stream()))
            ops.gemm(dSbuf, transpose16(km, 8), out=dq[b, h])                    # dQ = dS K          (reduction over Lk, padded)
            ops.gemm(transpose16(dS, 8), transpose16(qm, 8), out=dk[b, h])       # dK = dS^T Q        (reduction over Lq, padded)
    return dq, dk, dv


DROP_SITES = {"sa_attn": 0, "drop1": 1, "ca_attn": 2, "drop2": 3, "ffn": 4, "drop3": 5}


def drop_offset(layer_idx: int, site: str) -> int:
    """start of the counter stream of one dropout site of one memory-attention layer (2^40 elements apart: the largest site, the
    cross-attention probabilities, has B * L * Nk < 2^40 elements)"""
    return (layer_idx * 8 + DROP_SITES[site]) << 40


def _softmax_probs(qm: torch.Tensor, km: torch.Tensor, scale: float):
    """16-bit P = softmax(q k^T * scale) of one (batch, head): [Lq, Lk] view of a buffer whose rows are zero-padded to a multiple of 8"""
    Lq, Lk = qm.shape[0], km.shape[0]
    Lkp = -(-Lk // 8) * 8
    S = ops.gemm(qm, km, out_dtype=F32)
    Pbuf = torch.zeros(Lq, Lkp, dtype=OP16, device=qm.device)
    check(lib().msam2_softmax_rows(_p(S), S.stride(0), _p(Pbuf), Pbuf.stride(0), Lq, Lk, scale, _stream()))
    return Pbuf, Lkp


def attention_dropout_forward(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, p: float, seed: int, offset: int) -> torch.Tensor:
    """Single-head attention with dropout on the probabilities (F.scaled_dot_product_attention(dropout_p=p), transformer.py:317-318
    in train mode): q [B, Lq, C], k / v [B, Lk, C] 16-bit -> 16-bit [B*Lq, C].  Materialised per batch element (scores fp32, P 16-bit
    in HBM): the flash kernels carry no mask generator yet.  Mask stream: element offset + (b * Lq + i) * Lk + j."""
    B, Lq, C = q.shape
    Lk = k.shape[1]
    scale = C ** -0.5
    out = torch.empty(B * Lq, C, dtype=OP16, device=q.device)
    for b in range(B):
        Pbuf, Lkp = _softmax_probs(q[b], k[b], scale)
        Pd = torch.zeros(Lq, Lkp, dtype=OP16, device=q.device)
        ops.dropout(Pbuf[:, :Lk], p, seed, offset + b * Lq * Lk, out=Pd[:, :Lk])
        ops.gemm(Pd, transpose16(v[b], 8), out=out[b * Lq:(b + 1) * Lq])
    return out


def attention_dropout_backward(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, do: torch.Tensor, p: float, seed: int, offset: int):
    """Backward of `attention_dropout_forward` (P recomputed, the mask re-created from the same stream): do [B*Lq, C] any float type.
    Returns fp32 (dq [B, Lq, C], dk [B, Lk, C], dv [B, Lk, C])."""
    B, Lq, C = q.shape
    Lk = k.shape[1]
    scale = C ** -0.5
    dq = torch.empty(B, Lq, C, dtype=F32, device=q.device)
    dk = torch.empty(B, Lk, C, dtype=F32, device=q.device)
    dv = torch.empty(B, Lk, C, dtype=F32, device=q.device)
    do16 = _op16(do.reshape(B * Lq, C)).view(B, Lq, C)
    for b in range(B):
        off = offset + b * Lq * Lk
        Pbuf, Lkp = _softmax_probs(q[b], k[b], scale)
        P = Pbuf[:, :Lk]
        Pd = ops.dropout(P, p, seed, off)                                          # [Lq, Lk] contiguous
        ops.gemm(transpose16(Pd, 8), transpose16(do16[b], 8), out=dv[b])          # dV = Pd^T dO
        dPd = ops.gemm(do16[b], v[b], out_dtype=F32)                               # dO V^T  [Lq, Lk]
        dP = ops.dropout(dPd, p, seed, off)                                        # the same mask and 1 / (1 - p)
        dSbuf = torch.zeros(Lq, Lkp, dtype=OP16, device=q.device)
        dS = dSbuf[:, :Lk]
        check(lib().msam2_softmax_bwd_rows(_p(P), P.stride(0), _p(dP), dP.stride(0), _p(dS), dS.stride(0), Lq, Lk, scale, _stream()))
        ops.gemm(dSbuf, transpose16(k[b], 8), out=dq[b])                          # dQ = dS K
        ops.gemm(transpose16(dS, 8), transpose16(q[b], 8), out=dk[b])             # dK = dS^T Q
    return dq, dk, dv


def _rope_adjoint_(t: torch.Tensor, n_rope: int, table) -> torch.Tensor:
    """adjoint of the axial RoPE rotation (a rotation by the negative angle), in place on a 16-bit [B, N, D] gradient"""
    cs, sn = table
    return ops.rope_(t, n_rope, (cs, -sn))


def _memory_attention_layer_forward_saved(layer, x: torch.Tensor, mem_k: torch.Tensor, mem_v: torch.Tensor, B: int, L: int,
                                          n_ptr_tokens: int, drop: Optional[dict] = None):
    """`MemoryAttentionLayer.run` (memory_attention.py:17-99) with every intermediate the backward needs kept: returns
    (y fp32 [B*L, C], context for `_memory_attention_layer_backward_saved`).
    drop = {"p", "seed", "layer"}: TRAIN mode -- nn.Dropout(p) on the three residual branches (dropout1 / 2 / 3, memory_attention.py:63,
    80, 98), inside the FFN (97) and on both attentions' probabilities (transformer.py:317-318), masks from the counter streams
    `drop_offset(layer, site)` of `seed`; None: eval mode (dropout is the identity)."""
    from .modeling.common import v_f32, w_bf16
    sa, ca, wc = layer.self_attn, layer.cross_attn_image, layer._wc
    C, Nk = layer.d_model, mem_k.shape[1]
    assert sa.num_heads == 1 and ca.num_heads == 1, "memory attention runs single-head (sam2_hiera_*.yaml)"
    tab = sa.table(L, x.device)
    ln = lambda name, t: ops.layernorm(t, v_f32(wc, name + "w", getattr(layer, name).weight), v_f32(wc, name + "b", getattr(layer, name).bias),
                                       getattr(layer, name).eps)
    W = lambda key, *ws: w_bf16(wc, key, *ws)
    Bv = lambda key, *bs: v_f32(wc, key, *bs)
    # ---- forward with intermediates
    t1 = ln("norm1", x)
    w_qkv, b_qkv = W("sqkv", sa.q_proj.weight, sa.k_proj.weight, sa.v_proj.weight), Bv("sqkvb", sa.q_proj.bias, sa.k_proj.bias, sa.v_proj.bias)
    qkv = ops.gemm(t1, w_qkv, b_qkv).view(B, L, 3 * C)
    q1, k1, v1 = qkv[:, :, :C], qkv[:, :, C:2 * C], qkv[:, :, 2 * C:]
    ops.rope_(q1, L, tab)
    ops.rope_(k1, L, tab)
    u4 = lambda t: t.unsqueeze(1)                                                 # [B, N, C] -> [B, 1, N, C]
    flash = C in (64, 96, 128, 256) and not os.environ.get("MSAM2_MATERIALISED_BWD")
    dp, dseed, dl = (drop["p"], drop["seed"], drop["layer"]) if drop else (0.0, 0, 0)
    fdrop = bool(drop) and flash and flash_dropout_supported(C, L)     # dropout on the probabilities inside the flash kernels (round 3)
    if drop:
        if fdrop:
            ol1 = attention_forward_lse(u4(q1), u4(k1), u4(v1), dropout=(dp, dseed, drop_offset(dl, "sa_attn")))
            a1 = ol1[0].permute(0, 2, 1, 3).reshape(B * L, C)
        else:
            ol1 = None
            a1 = attention_dropout_forward(q1, k1, v1, dp, dseed, drop_offset(dl, "sa_attn"))
        x1 = ops.dropout(sa.out(a1, None), dp, dseed, drop_offset(dl, "drop1"), residual=x)
    else:
        ol1 = attention_forward_lse(u4(q1), u4(k1), u4(v1)) if flash else None
        a1 = ol1[0].permute(0, 2, 1, 3).reshape(B * L, C) if flash else sa.core(q1, k1, v1)   # 16-bit [B*L, C]
        x1 = sa.out(a1, x)
    t2 = ln("norm2", x1)
    wq, wk, wv = W("qw", ca.q_proj.weight), W("kw", ca.k_proj.weight), W("vw", ca.v_proj.weight)
    q2 = ops.gemm(t2, wq, Bv("qb", ca.q_proj.bias)).view(B, L, C)
    ops.rope_(q2, L, tab)
    mk2, mv2 = mem_k.reshape(B * Nk, -1), mem_v.reshape(B * Nk, -1)
    kk = ops.gemm(mk2, wk, Bv("kb", ca.k_proj.bias)).view(B, Nk, C)
    ops.rope_(kk, Nk - n_ptr_tokens, tab)
    vv = ops.gemm(mv2, wv, Bv("vb", ca.v_proj.bias)).view(B, Nk, C)
    if drop:
        if fdrop:
            ol2 = attention_forward_lse(u4(q2), u4(kk), u4(vv), dropout=(dp, dseed, drop_offset(dl, "ca_attn")))
            a2 = ol2[0].permute(0, 2, 1, 3).reshape(B * L, C)
        else:
            ol2 = None
            a2 = attention_dropout_forward(q2, kk, vv, dp, dseed, drop_offset(dl, "ca_attn"))
        x2 = ops.dropout(ca.out(a2, None), dp, dseed, drop_offset(dl, "drop2"), residual=x1)
    else:
        ol2 = attention_forward_lse(u4(q2), u4(kk), u4(vv)) if flash else None
        a2 = ol2[0].permute(0, 2, 1, 3).reshape(B * L, C) if flash else ca.core(q2, kk, vv)
        x2 = ca.out(a2, x1)
    t3 = ln("norm3", x2)
    w1, w2 = W("f1", layer.linear1.weight), W("f2", layer.linear2.weight)
    hid = ops.gemm(t3, w1, Bv("f1b", layer.linear1.bias), act=ops.ACT_RELU)
    if drop:
        hid = ops.dropout(hid, dp, dseed, drop_offset(dl, "ffn"))
        y = ops.dropout(ops.gemm(hid, w2, Bv("f2b", layer.linear2.bias), out_dtype=F32), dp, dseed, drop_offset(dl, "drop3"), residual=x2)
    else:
        y = ops.gemm(hid, w2, Bv("f2b", layer.linear2.bias), residual=x2, out_dtype=F32)
    ctx = dict(drop=drop, x=x, t1=t1, w_qkv=w_qkv, q1=q1, k1=k1, v1=v1, ol1=ol1, a1=a1, x1=x1, t2=t2, wq=wq, wk=wk, wv=wv, q2=q2, mk2=mk2, mv2=mv2,
               kk=kk, vv=vv, ol2=ol2, a2=a2, x2=x2, t3=t3, w1=w1, w2=w2, tab=tab, B=B, L=L, Nk=Nk, C=C, n_ptr=n_ptr_tokens)
    return y, ctx


def _memory_attention_layer_backward_saved(layer, ctx: dict, dy: torch.Tensor):
    """Backward half of `memory_attention_layer_backward` on the context of `_memory_attention_layer_forward_saved`."""
    from .modeling.common import v_f32, w_bf16
    sa, ca, wc = layer.self_attn, layer.cross_attn_image, layer._wc
    W = lambda key, *ws: w_bf16(wc, key, *ws)
    Bv = lambda key, *bs: v_f32(wc, key, *bs)
    u4 = lambda t: t.unsqueeze(1)
    x, t1, w_qkv, q1, k1, v1, ol1, a1, x1, t2 = (ctx[k] for k in ("x", "t1", "w_qkv", "q1", "k1", "v1", "ol1", "a1", "x1", "t2"))
    wq, wk, wv, q2, mk2, mv2, kk, vv, ol2, a2, x2, t3 = (ctx[k] for k in ("wq", "wk", "wv", "q2", "mk2", "mv2", "kk", "vv", "ol2", "a2", "x2", "t3"))
    w1, w2, tab, B, L, Nk, C, n_ptr_tokens = (ctx[k] for k in ("w1", "w2", "tab", "B", "L", "Nk", "C", "n_ptr"))
    g = {}
    drop = ctx.get("drop")
    dp, dseed, dl = (drop["p"], drop["seed"], drop["layer"]) if drop else (0.0, 0, 0)
    if drop:
        # FFN with its two dropouts: y = x2 + drop3(linear2(drop_ffn(relu(linear1(t3)))))
        b1v = Bv("f1b", layer.linear1.bias)
        pre = ops.gemm(t3, w1, b1v, act=ops.ACT_RELU)                                # ReLU'(pre) is the sign of relu(pre)
        hid_d = ops.dropout(pre, dp, dseed, drop_offset(dl, "ffn"))
        d_ffn = ops.dropout(dy.to(F32).contiguous(), dp, dseed, drop_offset(dl, "drop3"))
        dhid_d, g["linear2.weight"], g["linear2.bias"] = linear_backward(hid_d, w2, d_ffn)
        dpre = act_backward(pre, ops.dropout(dhid_d, dp, dseed, drop_offset(dl, "ffn")), ops.ACT_RELU)
        dt3, g["linear1.weight"], g["linear1.bias"] = linear_backward(t3, w1, dpre)
    else:
        dt3, g["linear1.weight"], g["linear1.bias"], g["linear2.weight"], g["linear2.bias"] = mlp_backward(
            t3, w1, Bv("f1b", layer.linear1.bias), w2, Bv("f2b", layer.linear2.bias), dy, ops.ACT_RELU)
    dx2, g["norm3.weight"], g["norm3.bias"] = layernorm_backward(x2, layer.norm3.weight.detach().float(), dt3, layer.norm3.eps,
                                                                 add=dy)          # residual branch + LayerNorm branch
    # cross attention
    d_ca = ops.dropout(dx2, dp, dseed, drop_offset(dl, "drop2")) if drop else dx2
    da2, g["cross_attn_image.out_proj.weight"], g["cross_attn_image.out_proj.bias"] = linear_backward(a2, W("ow", ca.out_proj.weight), d_ca)
    if drop and ol2 is not None:
        dq2, dkk, dvv = attention_backward(u4(q2), u4(kk), u4(vv), u4(da2.view(B, L, C)), o_lse=ol2, dropout=(dp, dseed, drop_offset(dl, "ca_attn")))
    elif drop:
        dq2, dkk, dvv = attention_dropout_backward(q2, kk, vv, da2, dp, dseed, drop_offset(dl, "ca_attn"))
    else:
        dq2, dkk, dvv = attention_backward(u4(q2), u4(kk), u4(vv), u4(da2.view(B, L, C)), o_lse=ol2)
    dq2, dkk = _op16(dq2.view(B * L, C)).view(B, L, C), _op16(dkk.view(B * Nk, C)).view(B, Nk, C)
    _rope_adjoint_(dq2, L, tab)
    _rope_adjoint_(dkk, Nk - n_ptr_tokens, tab)
    dt2, g["cross_attn_image.q_proj.weight"], g["cross_attn_image.q_proj.bias"] = linear_backward(t2, wq, dq2.view(B * L, C))
    dmk, g["cross_attn_image.k_proj.weight"], g["cross_attn_image.k_proj.bias"] = linear_backward(mk2, wk, dkk.view(B * Nk, C))
    dmv, g["cross_attn_image.v_proj.weight"], g["cross_attn_image.v_proj.bias"] = linear_backward(mv2, wv, dvv.view(B * Nk, C))
    dx1, g["norm2.weight"], g["norm2.bias"] = layernorm_backward(x1, layer.norm2.weight.detach().float(), dt2, layer.norm2.eps, add=dx2)
    # self attention
    d_sa = ops.dropout(dx1, dp, dseed, drop_offset(dl, "drop1")) if drop else dx1
    da1, g["self_attn.out_proj.weight"], g["self_attn.out_proj.bias"] = linear_backward(a1, W("ow_s", sa.out_proj.weight), d_sa)
    if drop and ol1 is not None:
        dq1, dk1, dv1 = attention_backward(u4(q1), u4(k1), u4(v1), u4(da1.view(B, L, C)), o_lse=ol1, dropout=(dp, dseed, drop_offset(dl, "sa_attn")))
    elif drop:
        dq1, dk1, dv1 = attention_dropout_backward(q1, k1, v1, da1, dp, dseed, drop_offset(dl, "sa_attn"))
    else:
        dq1, dk1, dv1 = attention_backward(u4(q1), u4(k1), u4(v1), u4(da1.view(B, L, C)), o_lse=ol1)
    dq16, dk16, dv16 = (_op16(t.view(B * L, C)).view(B, L, C) for t in (dq1, dk1, dv1))
    _rope_adjoint_(dq16, L, tab)
    _rope_adjoint_(dk16, L, tab)
    dqkv = torch.cat([dq16, dk16, dv16], dim=2)                                  # data movement only: the fused projection's gradient rows
    dt1, dw_qkv, db_qkv = linear_backward(t1, w_qkv, dqkv.view(B * L, 3 * C))
    for i, nm in enumerate(("q", "k", "v")):
        g[f"self_attn.{nm}_proj.weight"], g[f"self_attn.{nm}_proj.bias"] = dw_qkv[i * C:(i + 1) * C], db_qkv[i * C:(i + 1) * C]
    dx0, g["norm1.weight"], g["norm1.bias"] = layernorm_backward(x, layer.norm1.weight.detach().float(), dt1, layer.norm1.eps, add=dx1)
    return dx0, dmk.view(B, Nk, -1), dmv.view(B, Nk, -1), g




def memory_attention_layer_backward(layer, x: torch.Tensor, mem_k: torch.Tensor, mem_v: torch.Tensor, B: int, L: int, n_ptr_tokens: int,
                                    dy: torch.Tensor):
    """Backward of `MemoryAttentionLayer.run` (memory_attention.py:17-99 in eval mode: pre-LN RoPE self-attention, RoPE cross-attention
    to the memory bank, ReLU FFN, three residuals).  The forward is recomputed here with its intermediates kept (the inference forward
    saves nothing).  x fp32 [B*L, C]; mem_k / mem_v 16-bit [B, Nk, 64]; dy fp32 [B*L, C].
    Returns (dx fp32 [B*L, C], dmem_k fp32 [B, Nk, 64], dmem_v fp32 [B, Nk, 64], {parameter name: fp32 gradient})."""
    _, ctx = _memory_attention_layer_forward_saved(layer, x, mem_k, mem_v, B, L, n_ptr_tokens)
    return _memory_attention_layer_backward_saved(layer, ctx, dy)


def memory_attention_forward_saved(module, curr: torch.Tensor, curr_pos: torch.Tensor, memory: torch.Tensor, memory_pos: torch.Tensor,
                                   num_obj_ptr_tokens: int, dropout: Optional[Tuple[float, int]] = None):
    """`MemoryAttention.forward` (memory_attention.py:119-169; seq-first [L, B, C] tensors) keeping every layer's intermediates:
    returns (y fp32 [L, B, C], state for `memory_attention_backward_saved`).  dropout = (p, seed): train mode (see
    `_memory_attention_layer_forward_saved`); the backward re-creates every mask from the same seed."""
    from .modeling.common import v_f32
    L, B, C = curr.shape
    x = ops.add_cast(curr.transpose(0, 1), curr_pos.transpose(0, 1), 0.1, F32).reshape(B * L, C)
    mem_bf = memory.transpose(0, 1)
    mem_k = ops.add_cast(mem_bf, memory_pos.transpose(0, 1), 1.0, OP16)
    mem_v = ops.add_cast(mem_bf, None, 1.0, OP16)
    ctxs = []
    for li, layer in enumerate(module.layers):
        seed = dropout[1] if (dropout and isinstance(dropout[1], ops.DeviceSeed)) else (int(dropout[1]) if dropout else 0)
        drop = {"p": float(dropout[0]), "seed": seed, "layer": li} if dropout and dropout[0] > 0 else None
        x, ctx = _memory_attention_layer_forward_saved(layer, x, mem_k, mem_v, B, L, num_obj_ptr_tokens, drop=drop)
        ctxs.append(ctx)
    y = ops.layernorm(x, v_f32(module._wc, "nw", module.norm.weight), v_f32(module._wc, "nb", module.norm.bias), module.norm.eps, out_dtype=F32)
    return y.view(B, L, C).transpose(0, 1), dict(ctxs=ctxs, x_last=x, B=B, L=L, C=C)


def memory_attention_backward_saved(module, state: dict, dy: torch.Tensor):
    """Backward half of `memory_attention_backward` on the state of `memory_attention_forward_saved` (consumed: the per-layer
    intermediates are released as the layers are walked back)."""
    B, L, C, ctxs = state["B"], state["L"], state["C"], state["ctxs"]
    grads = {}
    d = ops.add_cast(dy.transpose(0, 1), None, 1.0, F32).reshape(B * L, C)
    d, grads["norm.weight"], grads["norm.bias"] = layernorm_backward(state["x_last"], module.norm.weight.detach().float(), d, module.norm.eps)
    dmk = dmv = None
    for i in range(len(module.layers) - 1, -1, -1):
        d, gk, gv, g = _memory_attention_layer_backward_saved(module.layers[i], ctxs[i], d)
        ctxs[i] = None
        dmk = gk if dmk is None else dmk + gk                                     # (tensor adds on small gradient slabs: plumbing)
        dmv = gv if dmv is None else dmv + gv
        grads.update({f"layers.{i}.{k}": v for k, v in g.items()})
    dcurr = d.view(B, L, C).transpose(0, 1)
    return dcurr, (dmk + dmv).transpose(0, 1), dmk.transpose(0, 1), grads


def memory_attention_backward(module, curr: torch.Tensor, curr_pos: torch.Tensor, memory: torch.Tensor, memory_pos: torch.Tensor,
                              num_obj_ptr_tokens: int, dy: torch.Tensor):
    """Backward of `MemoryAttention.forward` (memory_attention.py:119-169; seq-first [L, B, C] tensors like the forward): x = curr +
    0.1 curr_pos, the layers of `memory_attention_layer_backward` over keys memory + memory_pos / values memory, final LayerNorm.
    Returns (dcurr [L,B,C], dmemory [Nk,B,64], dmemory_pos [Nk,B,64], {"layers.i.<param>" | "norm.weight|bias": fp32 gradient})."""
    _, state = memory_attention_forward_saved(module, curr, curr_pos, memory, memory_pos, num_obj_ptr_tokens)
    return memory_attention_backward_saved(module, state, dy)


# ---------------------------------------------------------------------------------------------------------------------
# Two-way transformer of the mask decoder (sam/transformer.py:28-263): recomputing forward + backward
# ---------------------------------------------------------------------------------------------------------------------
def _add16(a: torch.Tensor, b: Optional[torch.Tensor]) -> torch.Tensor:
    """16-bit (a + b) for fp32 [rows, C] a and b ([rows, C] or [L, C] broadcast over the leading batch of a)"""
    rows, C = a.shape
    if b is None:
        return ops.add_cast(a.view(1, rows, C), None, 1.0, OP16)[0]
    if b.shape[0] == rows:
        return ops.add_cast(a.view(1, rows, C), b.view(1, rows, C), 1.0, OP16)[0]
    L = b.shape[0]
    return ops.add_cast(a.view(rows // L, L, C), b.view(1, L, C), 1.0, OP16).view(rows, C)


def _attn_fwd(att, q_in16, k_in16, v_in16, B):
    """Attention.forward pieces (transformer.py:239-263) on 16-bit token-major inputs; returns (projected q, k, v, attention output)"""
    from .modeling.common import v_f32, w_bf16
    wc = att._wc
    qp = ops.gemm(q_in16, w_bf16(wc, "qw", att.q_proj.weight), v_f32(wc, "qb", att.q_proj.bias))
    kp = ops.gemm(k_in16, w_bf16(wc, "kw", att.k_proj.weight), v_f32(wc, "kb", att.k_proj.bias))
    vp = ops.gemm(v_in16, w_bf16(wc, "vw", att.v_proj.weight), v_f32(wc, "vb", att.v_proj.bias))
    a = att.core(qp.view(B, -1, qp.shape[1]), kp.view(B, -1, kp.shape[1]), vp.view(B, -1, vp.shape[1]))   # 16-bit [B*Lq, Ci]
    return qp, kp, vp, a


def attention_module_backward(att, q_in16, k_in16, v_in16, B: int, d_out: torch.Tensor, prefix: str, grads: dict):
    """Backward of `Attention.forward` (q/k/v projections, H heads, output projection).  q_in16 [B*Lq, C], k_in16 / v_in16 [B*Lk, Ck]
    16-bit; d_out fp32 [B*Lq, C].  Adds the 8 parameter gradients under `prefix` to `grads`; returns (dq_in, dk_in, dv_in) fp32."""
    from .modeling.common import w_bf16
    wc, H = att._wc, att.num_heads
    qp, kp, vp, a = _attn_fwd(att, q_in16, k_in16, v_in16, B)
    Ci = qp.shape[1]
    D = Ci // H
    amax = None
    if d_out.shape[0] <= 1024:                                                    # token-side attention: see _unit_max
        d_out, amax = _unit_max(d_out.to(F32))
    da, grads[prefix + ".out_proj.weight"], grads[prefix + ".out_proj.bias"] = linear_backward(a, w_bf16(wc, "ow", att.out_proj.weight), d_out)
    Lq, Lk = qp.shape[0] // B, kp.shape[0] // B
    if D in (16, 32) and min(Lq, Lk) <= 32:
        # decoder attention: one side is a handful of tokens -> one fused kernel for all (batch, head) pairs, token-major in and out
        dq = torch.empty(B * Lq, Ci, dtype=F32, device=qp.device)
        dk = torch.empty(B * Lk, Ci, dtype=F32, device=qp.device)
        dv = torch.empty(B * Lk, Ci, dtype=F32, device=qp.device)
        dac = da.contiguous()
        check(lib().msam2_attention_small_bwd(_p(qp), Lq * qp.stride(0), qp.stride(0), _p(kp), Lk * kp.stride(0), kp.stride(0), _p(vp),
                                              Lk * vp.stride(0), vp.stride(0), _p(dac), _p(dq), _p(dk), _p(dv), B, H, Lq, Lk, D,
                                              D ** -0.5, _stream()))
        rows = lambda t: t
    else:
        heads = lambda t: t.view(B, -1, H, D).permute(0, 2, 1, 3)                 # [B, H, L, D] views of the token-major rows
        dq, dk, dv = attention_backward(heads(qp), heads(kp), heads(vp), heads(da))
        rows = lambda t: t.permute(0, 2, 1, 3).reshape(-1, Ci)                    # back to token-major (data movement)
    outs = []
    for nm, x16, g in (("q", q_in16, dq), ("k", k_in16, dk), ("v", v_in16, dv)):
        dx, grads[f"{prefix}.{nm}_proj.weight"], grads[f"{prefix}.{nm}_proj.bias"] = linear_backward(
            x16, w_bf16(wc, nm + "w", getattr(att, nm + "_proj").weight), rows(g))
        outs.append(dx)
    if amax is not None:
        # back from the unit-max normalisation: one multi-tensor launch for the 8 parameter gradients + 3 input gradients
        torch._foreach_mul_([grads[f"{prefix}.{k}.{wb}"] for k in ("out_proj", "q_proj", "k_proj", "v_proj") for wb in ("weight", "bias")] + outs, amax)
    return outs


def two_way_transformer_backward(tw, keys: torch.Tensor, key_pe: torch.Tensor, tokens: torch.Tensor, B: int, T: int, L: int,
                                 d_queries: torch.Tensor, d_keys: torch.Tensor):
    """Backward of `TwoWayTransformer.run` (transformer.py:74-118 with its TwoWayAttentionBlocks 165-196): keys fp32 [B*L, C] (image
    embedding + dense prompt), key_pe fp32 [L, C] (constant), tokens fp32 [B*T, C] (output tokens + prompt embeddings; also the
    query position encoding), upstream gradients d_queries [B*T, C], d_keys [B*L, C].
    Returns (d_keys_in fp32 [B*L, C], d_tokens fp32 [B*T, C], {"layers.i.<...>" | "final_attn_token_to_image.<...>" |
    "norm_final_attn.<...>": gradient})."""
    from .modeling.common import v_f32, w_bf16
    grads: dict = {}
    qpe = tokens

    def ln_fwd(mod, x):
        return ops.layernorm(x, mod.weight.detach().float(), mod.bias.detach().float(), mod.eps, out_dtype=F32)

    def ln_bwd(mod, x, dy, name):
        dx, grads[name + ".weight"], grads[name + ".bias"] = layernorm_backward(x, mod.weight.detach().float(), dy, mod.eps)
        return dx

    # ---- forward, keeping the inputs of every LayerNorm and attention
    saved = []
    Q, K = tokens, keys
    for i, blk in enumerate(tw.layers):
        s = {"Q0": Q, "K0": K}
        sa, ca, ia = blk.self_attn, blk.cross_attn_token_to_image, blk.cross_attn_image_to_token
        if blk.skip_first_layer_pe:
            s["sa_in"] = (_add16(Q, None),) * 3
            s["Q0p"] = sa.out(_attn_fwd(sa, *s["sa_in"], B)[3], None)
        else:
            qk = _add16(Q, qpe)
            s["sa_in"] = (qk, qk, _add16(Q, None))
            s["Q0p"] = sa.out(_attn_fwd(sa, *s["sa_in"], B)[3], Q)
        s["Q1"] = ln_fwd(blk.norm1, s["Q0p"])
        s["ca_in"] = (_add16(s["Q1"], qpe), _add16(K, key_pe), _add16(K, None))
        s["Q1p"] = ca.out(_attn_fwd(ca, *s["ca_in"], B)[3], s["Q1"])
        s["Q2"] = ln_fwd(blk.norm2, s["Q1p"])
        s["Q2_16"] = _add16(s["Q2"], None)
        s["Q2p"] = blk.mlp.run(s["Q2_16"], residual=s["Q2"], out_dtype=F32)
        s["Q3"] = ln_fwd(blk.norm3, s["Q2p"])
        s["ia_in"] = (s["ca_in"][1], _add16(s["Q3"], qpe), _add16(s["Q3"], None))
        s["K1"] = ia.out(_attn_fwd(ia, *s["ia_in"], B)[3], K)
        Q, K = s["Q3"], ln_fwd(blk.norm4, s["K1"])
        saved.append(s)
    fa = tw.final_attn_token_to_image
    fa_in = (_add16(Q, qpe), _add16(K, key_pe), _add16(K, None))
    Qf = fa.out(_attn_fwd(fa, *fa_in, B)[3], Q)
    # ---- backward
    d_tok = torch.zeros_like(tokens)                                              # accumulates every use of the query position encoding
    dQf = ln_bwd(tw.norm_final_attn, Qf, d_queries, "norm_final_attn")
    dq_in, dk_in, dv_in = attention_module_backward(fa, *fa_in, B, dQf, "final_attn_token_to_image", grads)
    dQ = dQf + dq_in
    d_tok += dq_in
    dK = d_keys + dk_in + dv_in
    for i in range(len(tw.layers) - 1, -1, -1):
        blk, s, pre = tw.layers[i], saved[i], f"layers.{i}"
        sa, ca, ia = blk.self_attn, blk.cross_attn_token_to_image, blk.cross_attn_image_to_token
        # keys: K_out = LN4(K1), K1 = K0 + ia(K0 + kpe, Q3 + qpe, Q3)
        dK1 = ln_bwd(blk.norm4, s["K1"], dK, pre + ".norm4")
        dq_in, dk_in, dv_in = attention_module_backward(ia, *s["ia_in"], B, dK1, pre + ".cross_attn_image_to_token", grads)
        dK0 = dK1 + dq_in
        dQ3 = dQ + dk_in + dv_in
        d_tok += dk_in
        # Q3 = LN3(Q2 + MLP(Q2))
        dQ2p = ln_bwd(blk.norm3, s["Q2p"], dQ3, pre + ".norm3")
        l1, l2 = blk.mlp.layers
        res = mlp_backward(s["Q2_16"], w_bf16(blk.mlp._wc, "w0", l1.weight), v_f32(blk.mlp._wc, "b0", l1.bias),
                           w_bf16(blk.mlp._wc, "w1", l2.weight), v_f32(blk.mlp._wc, "b1", l2.bias), dQ2p, blk.mlp._act_code)
        dx = res[0]
        for nm, gten in zip(("mlp.layers.0.weight", "mlp.layers.0.bias", "mlp.layers.1.weight", "mlp.layers.1.bias"), res[1:]):
            grads[f"{pre}.{nm}"] = gten
        dQ2 = dQ2p + dx
        # Q2 = LN2(Q1 + ca(Q1 + qpe, K0 + kpe, K0))
        dQ1p = ln_bwd(blk.norm2, s["Q1p"], dQ2, pre + ".norm2")
        dq_in, dk_in, dv_in = attention_module_backward(ca, *s["ca_in"], B, dQ1p, pre + ".cross_attn_token_to_image", grads)
        dQ1 = dQ1p + dq_in
        d_tok += dq_in
        dK0 = dK0 + dk_in + dv_in
        # Q1 = LN1(Q0p), Q0p = sa(...) [+ Q0]
        dQ0p = ln_bwd(blk.norm1, s["Q0p"], dQ1, pre + ".norm1")
        dq_in, dk_in, dv_in = attention_module_backward(sa, *s["sa_in"], B, dQ0p, pre + ".self_attn", grads)
        if blk.skip_first_layer_pe:
            dQ = dq_in + dk_in + dv_in
        else:
            dQ = dQ0p + dq_in + dk_in + dv_in
            d_tok += dq_in + dk_in
        dK = dK0
    return dK, d_tok + dQ, grads


# ---------------------------------------------------------------------------------------------------------------------
# Mask decoder (sam/mask_decoder.py:170-267): recomputing forward + backward of the mask logits w.r.t. every parameter
# ---------------------------------------------------------------------------------------------------------------------
def _unit_max(dy: torch.Tensor):
    """(dy / max|dy|, max|dy| as a device scalar) -- no host synchronisation, capturable.  The token-side links of the decoder (the four
    hyper-network MLPs, the token attention) carry branches whose gradients can be orders of magnitude below the step's global maximum
    the loss scale was chosen for (a mask token that only one slice of a chain uses; the object-pointer path); a few layers further
    down such a branch would sit in fp16's subnormals as a 16-bit GEMM operand.  These links are a few rows: they re-normalise their
    own upstream gradient and multiply the results back."""
    a = dy.abs().amax().clamp_min(1e-30)
    return dy / a, a


def mlp_layers_backward(mlp, x16: torch.Tensor, dy: torch.Tensor, prefix: str, grads: dict) -> torch.Tensor:
    """Backward of an n-layer `MLP` (sam2_utils.py:108-132, no output sigmoid) by recomputation: x16 16-bit [M, in], dy [M, out].
    Adds `prefix.layers.i.weight|bias` to grads, returns dx fp32."""
    from .modeling.common import v_f32, w_bf16
    assert not mlp.sigmoid_output, "sigmoid-output heads (IoU) are not differentiated here"
    wc, n = mlp._wc, mlp.num_layers
    Ws = [w_bf16(wc, f"w{i}", l.weight) for i, l in enumerate(mlp.layers)]
    Bs = [v_f32(wc, f"b{i}", l.bias) for i, l in enumerate(mlp.layers)]
    hs, pres = [x16], []
    for i in range(n - 1):
        pres.append(ops.gemm(hs[-1], Ws[i], Bs[i], out_dtype=F32))
        hs.append(ops.gemm(hs[-1], Ws[i], Bs[i], act=mlp._act_code))
    d, a = _unit_max(dy.to(F32))
    for i in range(n - 1, -1, -1):
        d, gw, gb = linear_backward(hs[i], Ws[i], d)
        grads[f"{prefix}.layers.{i}.weight"], grads[f"{prefix}.layers.{i}.bias"] = gw, gb
        if i > 0:
            d = act_backward(pres[i - 1], d, mlp._act_code)
    d = d if d.dtype == F32 else d.to(F32)
    torch._foreach_mul_([grads[f"{prefix}.layers.{i}.{wb}"] for i in range(n) for wb in ("weight", "bias")] + [d], a)   # one launch
    return d


def _convt_gather(g, bias, skip, B, h, w):
    C = g.shape[1] // 4
    z = torch.empty(B * 4 * h * w, C, dtype=F32, device=g.device)
    check(lib().msam2_convt2x2_gather(_p(g), _p(bias), _p(skip), _p(z), B, h, w, C, _stream()))
    return z


def _convt_scatter_grad(dz, B, h, w):
    C = dz.shape[1]
    dg = torch.empty(B * h * w, 4 * C, dtype=OP16, device=dz.device)
    check(lib().msam2_convt2x2_scatter_grad(_p(dz), _is_bf16(dz), _p(dg), B, h, w, C, _stream()))
    return dg


def mask_decoder_backward(dec, src_tokens: torch.Tensor, pe_tokens: torch.Tensor, sparse: torch.Tensor, feat_s0: torch.Tensor,
                          feat_s1: torch.Tensor, B: int, h: int, w: int, d_masks: torch.Tensor, aux: Optional[dict] = None,
                          d_mask_tokens: Optional[torch.Tensor] = None):
    """Backward of `MaskDecoder.predict_masks_tokens` for a loss on the 4 mask logit maps: src_tokens fp32 [B*h*w, C] (image embedding +
    dense prompt), pe_tokens fp32 [h*w, C], sparse fp32 [B, P, C] prompt embeddings, feat_s0 / feat_s1 16-bit token-major high-res
    features, d_masks fp32 [B, 4, 4h, 4w].  The IoU and object-score heads do not see the mask loss and get no gradient.
    Returns (d_src_tokens fp32 [B*h*w, C], d_sparse fp32 [B, P, C], {parameter name relative to the decoder: gradient}).
    aux (optional dict) receives "d_feat_s0" / "d_feat_s1": the gradients of the two high-resolution feature maps (token-major, like the
    inputs) -- what the image-encoder backward continues from (mask_decoder.py:244-247: both enter by a plain add).
    d_mask_tokens (optional fp32 [B, num_mask_tokens, C]): an upstream gradient on `mask_tokens_out` as well -- the path of the object
    pointer (sam2_base.py:376-388: obj_ptr_proj of the selected SAM output token), added to the hyper-network path's token gradient."""
    from .modeling.common import to_bf16, v_f32
    wc, C, L = dec._wc, dec.transformer_dim, h * w
    nm = dec.num_mask_tokens
    out_tok = torch.cat([dec.obj_score_token.weight, dec.iou_token.weight, dec.mask_tokens.weight], 0).detach().float()
    n_out = out_tok.shape[0]
    T = n_out + sparse.shape[1]
    tokens = torch.empty(B, T, C, dtype=F32, device=src_tokens.device)
    tokens[:, :n_out] = out_tok
    tokens[:, n_out:] = sparse
    tokens = tokens.view(B * T, C)
    # ---- forward with intermediates
    hs, keys = dec.transformer.run(src_tokens, pe_tokens, tokens, B, T, L)
    hs = hs.view(B, T, C)
    up = dec.output_upscaling
    dc1_w = wc.get("dc1", [up[0].weight], lambda: up[0].weight.detach().permute(2, 3, 1, 0).reshape(-1, C).to(OP16).contiguous())
    dc2_w = wc.get("dc2", [up[3].weight], lambda: up[3].weight.detach().permute(2, 3, 1, 0).reshape(-1, C // 4).to(OP16).contiguous())
    b1, b2 = v_f32(wc, "dc1b", up[0].bias), v_f32(wc, "dc2b", up[3].bias)
    lnw, lnb = v_f32(wc, "lnw", up[1].weight), v_f32(wc, "lnb", up[1].bias)
    keys16 = to_bf16(keys)
    g1 = ops.gemm(keys16, dc1_w)
    z1 = _convt_gather(g1, b1, feat_s1, B, h, w)                                  # [B*4L, C/4] pre-LayerNorm
    y1 = ops.layernorm(z1, lnw, lnb, 1e-6, out_dtype=F32)                         # pre-GELU
    u1 = ops.layernorm(z1, lnw, lnb, 1e-6, act=ops.ACT_GELU)                      # 16-bit
    g2 = ops.gemm(u1, dc2_w)
    z2 = _convt_gather(g2, b2, feat_s0, B, 2 * h, 2 * w)                          # [B*16L, C/8] pre-GELU
    u2 = ops.convt2x2_shuffle(g2, b2, feat_s0, None, None, B, 2 * h, 2 * w)       # 16-bit GELU(z2)
    Pn, Cu = 16 * L, C // 8
    hyper16 = torch.empty(B, nm, Cu, dtype=OP16, device=u2.device)
    tok16 = [to_bf16(hs[:, 2 + i].contiguous()) for i in range(nm)]
    for i, m in enumerate(dec.output_hypernetworks_mlps):
        hyper16[:, i] = m.run(tok16[i], out_dtype=OP16)
    # ---- backward
    grads: dict = {}
    dm16 = _op16(d_masks.reshape(B * nm, Pn)).view(B, nm, Pn)
    d_hyper = torch.empty(B, nm, Cu, dtype=F32, device=u2.device)
    d_u2 = torch.empty(B * Pn, Cu, dtype=F32, device=u2.device)
    for b in range(B):
        u2b = u2[b * Pn:(b + 1) * Pn]
        ops.gemm(dm16[b], transpose16(u2b), out=d_hyper[b])                      # [nm, P] @ [P, Cu]
        ops.gemm(transpose16(dm16[b], 8), transpose16(hyper16[b], 8), out=d_u2[b * Pn:(b + 1) * Pn])   # [P, nm] @ [nm, Cu]
    dz2 = act_backward(z2, d_u2, ops.ACT_GELU)
    if aux is not None:
        aux["d_feat_s0"] = dz2                                                       # z2 = shuffle(g2) + b2 + feat_s0
    grads["output_upscaling.3.bias"] = colsum(dz2)
    d_u1, dw2, _ = linear_backward(u1, dc2_w, _convt_scatter_grad(dz2, B, 2 * h, 2 * w))
    grads["output_upscaling.3.weight"] = dw2.view(2, 2, C // 8, C // 4).permute(3, 2, 0, 1).contiguous()
    dy1 = act_backward(y1, d_u1, ops.ACT_GELU)
    dz1, grads["output_upscaling.1.weight"], grads["output_upscaling.1.bias"] = layernorm_backward(z1, lnw, dy1, 1e-6)
    if aux is not None:
        aux["d_feat_s1"] = dz1                                                       # z1 = shuffle(g1) + b1 + feat_s1
    grads["output_upscaling.0.bias"] = colsum(dz1)
    d_keys, dw1, _ = linear_backward(keys16, dc1_w, _convt_scatter_grad(dz1, B, h, w))
    grads["output_upscaling.0.weight"] = dw1.view(2, 2, C // 4, C).permute(3, 2, 0, 1).contiguous()
    d_hs = torch.zeros(B, T, C, dtype=F32, device=u2.device)
    if d_mask_tokens is not None:
        d_hs[:, 2:2 + nm] = d_mask_tokens
    for i, m in enumerate(dec.output_hypernetworks_mlps):
        d_hs[:, 2 + i] += mlp_layers_backward(m, tok16[i], d_hyper[:, i].contiguous(), f"output_hypernetworks_mlps.{i}", grads)
    d_src, d_tok, g_tw = two_way_transformer_backward(dec.transformer, src_tokens, pe_tokens, tokens, B, T, L, d_hs.view(B * T, C), d_keys)
    grads.update({"transformer." + k: v for k, v in g_tw.items()})
    d_tok = d_tok.view(B, T, C)
    d_out_tok = colsum(d_tok[:, :n_out].reshape(B, n_out * C).contiguous()).view(n_out, C)    # the learned output tokens are shared over the batch
    grads["obj_score_token.weight"], grads["iou_token.weight"], grads["mask_tokens.weight"] = d_out_tok[0:1], d_out_tok[1:2], d_out_tok[2:]
    return d_src, d_tok[:, n_out:], grads


# ---------------------------------------------------------------------------------------------------------------------
# Memory encoder (memory_encoder.py:17-181; part of train_3d.py's `mem_layers` parameter group): recomputing forward + backward
# ---------------------------------------------------------------------------------------------------------------------
def dwconv7x7(x: torch.Tensor, taps: torch.Tensor, bias: Optional[torch.Tensor], n: int, H: int, W: int, flip: bool = False) -> torch.Tensor:
    """fp32 NHWC tokens [n*H*W, C] -> depthwise 7x7 (pad 3) with taps [49, C]; flip=True: the adjoint (input gradient)."""
    C = x.shape[1]
    _req(x.dtype == F32 and x.is_contiguous() and taps.shape == (49, C), "dwconv7x7: fp32 contiguous tokens, taps [49, C]")
    y = torch.empty_like(x)
    check(lib().msam2_dwconv7x7(_p(x), _p(taps), _p(bias), _p(y), n, H, W, C, 1 if flip else 0, _stream()))
    return y


def cxblock_backward(blk, x: torch.Tensor, n: int, H: int, W: int, dy: torch.Tensor):
    """Backward of `CXBlock.run` (memory_encoder.py:62-117: y = x + gamma * pwconv2(gelu(pwconv1(LN(dwconv(x)))))), x / dy fp32
    [n*H*W, C].  The layer-scale gamma starts at 1e-6, far below the 16-bit operand range for the branch gradient, so the branch is
    differentiated in units of gs ~ max|gamma| (a host scalar) and rescaled in fp32 at the end.  The two
    broadcast products with gamma / the branch output are torch elementwise ops.  Returns (dx fp32, {parameter name: gradient})."""
    from .modeling.common import v_f32, w_bf16
    wc, C = blk._wc, x.shape[1]
    taps = wc.get("dw", [blk.dwconv.weight], lambda: blk.dwconv.weight.detach().reshape(C, 49).t().float().contiguous())
    lw, lb = v_f32(wc, "lw", blk.norm.weight), v_f32(wc, "lb", blk.norm.bias)
    w1, b1 = w_bf16(wc, "w1", blk.pwconv1.weight), v_f32(wc, "b1", blk.pwconv1.bias)
    w2, b2 = w_bf16(wc, "w2", blk.pwconv2.weight), v_f32(wc, "b2", blk.pwconv2.bias)
    # any positive gs is exact (the branch is linear); it only has to keep gamma / gs near 1.  Refreshed from the weights (one host
    # sync) whenever the stream is not being captured, reused as is inside a hipGraph capture
    if not torch.cuda.is_current_stream_capturing() or not hasattr(blk, "_gamma_scale"):
        blk._gamma_scale = max(float(blk.gamma.detach().abs().max()), 1e-30)
    gs = blk._gamma_scale
    gn = wc.get("gnorm", [blk.gamma], lambda: (blk.gamma.detach().float() * (1.0 / gs)).contiguous())
    # ---- forward with intermediates
    h0 = dwconv7x7(x, taps, v_f32(wc, "dwb", blk.dwconv.bias), n, H, W)
    t = ops.layernorm(h0, lw, lb, blk.norm.eps)
    z2 = ops.gemm(ops.gemm(t, w1, b1, act=ops.ACT_GELU), w2, b2, out_dtype=F32)          # branch output before the layer scale
    # ---- backward (branch in units of gs)
    g = {"gamma": colsum((dy * z2).contiguous())}
    dt, dw1, db1, dw2, db2 = mlp_backward(t, w1, b1, w2, b2, dy * gn, ops.ACT_GELU)
    dh0, dlw, dlb = layernorm_backward(h0, lw, dt, blk.norm.eps)
    dtaps = torch.zeros(49, C, dtype=F32, device=x.device)
    check(lib().msam2_dwconv7x7_wgrad(_p(x), _p(dh0), _p(dtaps), n, H, W, C, _stream()))
    g["dwconv.weight"], g["dwconv.bias"] = dtaps.t().reshape(C, 1, 7, 7) * gs, colsum(dh0) * gs
    g["norm.weight"], g["norm.bias"] = dlw * gs, dlb * gs
    g["pwconv1.weight"], g["pwconv1.bias"], g["pwconv2.weight"], g["pwconv2.bias"] = dw1 * gs, db1 * gs, dw2 * gs, db2 * gs
    dxb = dwconv7x7(dh0, taps, None, n, H, W, flip=True)
    dx = ops.add_cast(dy.view(1, -1, C), dxb.view(1, -1, C), gs, F32).view(-1, C)
    return dx, g


def memory_encoder_backward(enc, pix_tokens: torch.Tensor, mask: torch.Tensor, mode: int, scale: float, bias: float, n: int, H: int, W: int,
                            dy: torch.Tensor, need_dmask: bool = False):
    """Backward of `MemoryEncoder.run` (memory_encoder.py:138-181 with MaskDownSampler 17-58 and the Fuser's CXBlocks 62-135): pix_tokens
    [n*H*W, C], mask fp32 [n,1,16H,16W] with the mask transform `mode` of the forward (0 raw, 1 sigmoid * scale + bias, 2 binarised),
    dy fp32 [n*H*W, out_dim].  The k3 s2 p1 convolutions are differentiated in their im2col GEMM form (`msam2_col2im3x3s2` is the
    adjoint of the patch gather); the mask transform itself (one elementwise op on the input mask) is done by torch.
    Returns (d pix_tokens fp32 [n*H*W, C], {parameter name relative to the memory encoder: gradient}); with need_dmask a third value,
    d mask fp32 [n,1,16H,16W] through the first convolution and the sigmoid transform (zero for the binarised mode 2) -- what
    back-propagation through time continues with into the slice that predicted the mask (training_3d)."""
    from .modeling.common import to_bf16, v_f32, w_bf16
    wc, ds = enc._wc, enc.mask_downsampler
    dwc, E = ds._wc, ds.encoder
    S = mask.shape[-1]
    mt = mask.to(F32)
    if mode == 1:
        mt = torch.sigmoid(mt) * scale + bias
    elif mode == 2:
        mt = (mt > 0).to(F32) * scale + bias
    # ---- forward with intermediates: mask down-sampler in GEMM form
    # (the patch gather moves 4-channel groups: the 1-channel mask rides in channel 0 of a zero-padded 4-channel image)
    h = torch.zeros(n * S * S, 4, dtype=OP16, device=mask.device)
    h[:, 0] = mt.reshape(-1).to(OP16)
    side, stages = S, []
    for j in range(4):
        conv, ln = E[3 * j], E[3 * j + 1]
        cout, cin = conv.weight.shape[0], conv.weight.shape[1]
        cin_p = h.shape[1]
        cols = ops.im2col3x3s2(h, n, side, side)
        np_ = -(-cout // 8) * 8                                                   # GEMM rows of W padded to a multiple of 8 (stage 1: 4)

        def pack(c=conv, ld=cols.shape[1], rows=np_, cp=cin_p):
            w = torch.zeros(c.weight.shape[0], 3, 3, cp, dtype=F32, device=c.weight.device)
            w[..., : c.weight.shape[1]] = c.weight.detach().permute(0, 2, 3, 1).float()
            out = torch.zeros(rows, ld, dtype=OP16, device=w.device)
            out[: w.shape[0], : 9 * cp] = w.reshape(w.shape[0], -1).to(OP16)
            return out

        def packb(c=conv, rows=np_):
            out = torch.zeros(rows, dtype=F32, device=c.bias.device)
            out[: c.bias.shape[0]] = c.bias.detach().float()
            return out
        wpk, bpk = dwc.get(f"bw_cw{j}", [conv.weight], pack), dwc.get(f"bw_cb{j}", [conv.bias], packb)
        gfull = ops.gemm(cols, wpk, bpk, out_dtype=F32)
        gj = gfull if np_ == cout else gfull[:, :cout].contiguous()
        lw, lb = v_f32(dwc, f"lw{j}", ln.weight), v_f32(dwc, f"lb{j}", ln.bias)
        yln = ops.layernorm(gj, lw, lb, ln.eps, out_dtype=F32)
        h = ops.layernorm(gj, lw, lb, ln.eps, act=ops.ACT_GELU)
        stages.append((cols, wpk, gj, yln, lw, cout, cin, side, np_, cin_p))
        side //= 2
    pw = w_bf16(dwc, "pw", E[12].weight)
    m_out = ops.gemm(h, pw, v_f32(dwc, "pb", E[12].bias), out_dtype=F32)
    pix16 = to_bf16(pix_tokens)
    ppw = w_bf16(wc, "pw", enc.pix_feat_proj.weight)
    x = ops.gemm(pix16, ppw, v_f32(wc, "pb", enc.pix_feat_proj.bias), residual=m_out, out_dtype=F32)
    xs = []
    for layer in enc.fuser.layers:
        xs.append(x)
        x = layer.run(x, n, H, W)
    # ---- backward
    g: dict = {}
    if isinstance(enc.out_proj, torch.nn.Identity):
        dx = dy.to(F32)
    else:
        ow = w_bf16(wc, "ow", enc.out_proj.weight)
        dx, dw, db = linear_backward(to_bf16(x), ow, dy)
        g["out_proj.weight"], g["out_proj.bias"] = dw.view_as(enc.out_proj.weight), db
    for j in range(len(enc.fuser.layers) - 1, -1, -1):
        dx, gl = cxblock_backward(enc.fuser.layers[j], xs[j], n, H, W, dx)
        g.update({f"fuser.layers.{j}.{k}": v for k, v in gl.items()})
    dpix, dw, db = linear_backward(pix16, ppw, dx)
    g["pix_feat_proj.weight"], g["pix_feat_proj.bias"] = dw.view_as(enc.pix_feat_proj.weight), db
    pre = "mask_downsampler.encoder"
    dh, dw, db = linear_backward(h, pw, dx)                                        # the down-sampled mask enters as a residual: dm = dx
    g[f"{pre}.12.weight"], g[f"{pre}.12.bias"] = dw.view_as(E[12].weight), db
    for j in range(3, -1, -1):
        cols, wpk, gj, yln, lw, cout, cin, side_in, np_, cin_p = stages[j]
        dyln = act_backward(yln, dh, ops.ACT_GELU)
        dg, g[f"{pre}.{3 * j + 1}.weight"], g[f"{pre}.{3 * j + 1}.bias"] = layernorm_backward(gj, lw, dyln, E[3 * j + 1].eps)
        if np_ != cout:
            dgp = torch.zeros(dg.shape[0], np_, dtype=F32, device=dg.device)
            dgp[:, :cout] = dg
            dg = dgp
        dcols, dwp, dbp = linear_backward(cols, wpk, dg, need_dx=j > 0 or need_dmask)
        g[f"{pre}.{3 * j}.weight"] = dwp[:cout, : 9 * cin_p].reshape(cout, 3, 3, cin_p)[..., :cin].permute(0, 3, 1, 2).contiguous()
        g[f"{pre}.{3 * j}.bias"] = dbp[:cout]
        if j > 0:
            dh = torch.empty(n * side_in * side_in, cin, dtype=F32, device=dy.device)
            check(lib().msam2_col2im3x3s2(_p(dcols), dcols.stride(0), _p(dh), n, side_in, side_in, cin, _stream()))
    if not need_dmask:
        return dpix, g
    # first convolution's input: the 1-channel mask rides in channel 0 of the zero-padded 4-channel image
    dimg = torch.empty(n * S * S, 4, dtype=F32, device=dy.device)
    check(lib().msam2_col2im3x3s2(_p(dcols), dcols.stride(0), _p(dimg), n, S, S, 4, _stream()))
    dmt = dimg[:, 0].reshape(n, 1, S, S)
    if mode == 1:
        sg = torch.sigmoid(mask.to(F32))
        dmask = dmt * (scale * sg * (1.0 - sg))
    elif mode == 2:
        dmask = torch.zeros_like(dmt)
    else:
        dmask = dmt.contiguous()
    return dpix, g, dmask
