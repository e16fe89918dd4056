"""torch-tensor front end of the C-ABI (include/msam2_hip.h): every function extracts raw device pointers, sizes and
strides and calls libmsam2_hip.so on torch's current HIP stream.  PyTorch provides memory and streams only.

Layout conventions: activations are token-major ("NHWC") ``[rows, C]`` with C contiguous; bf16 for MFMA operands,
fp32 for residual streams / logits.  Weights follow nn.Linear (``[out, in]``).
"""
from __future__ import annotations

import ctypes
import math
from typing import Optional, Tuple

import torch

from ._lib import check, lib

# 16-bit MFMA operand dtype of the loaded library: fp16 by default, bf16 when built with -DMSAM2_OPERAND_BF16
OP16 = torch.float16 if lib().msam2_operand_is_fp16() else torch.bfloat16

F32 = torch.float32
ACT_NONE, ACT_GELU, ACT_RELU, ACT_SIGMOID = 0, 1, 2, 3


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _is_bf16(t: torch.Tensor) -> int:
    """1 for the library's 16-bit operand dtype (OP16), 0 for fp32."""
    if t.dtype == OP16:
        return 1
    if t.dtype == F32:
        return 0
    raise TypeError(f"expected {OP16} or fp32 tensor, got {t.dtype}")


def _req(cond: bool, msg: str):
    if not cond:
        raise ValueError(msg)


# ---------------------------------------------------------------------------------------------------------------------
def gemm(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, *, act: int = ACT_NONE,
         colscale: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None, res_mod: int = 0,
         out_dtype: torch.dtype = OP16, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[M,N] = residual + colscale * act(a[M,K] @ w[N,K]^T + bias).  a, w bf16; bias/colscale fp32."""
    _req(a.dim() == 2 and w.dim() == 2 and a.shape[1] == w.shape[1], f"gemm shapes {tuple(a.shape)} x {tuple(w.shape)}")
    _req(a.dtype == OP16 and w.dtype == OP16, "gemm operands must be 16-bit (ops.OP16)")
    _req(a.stride(1) == 1 and w.stride(1) == 1, "gemm operands must be K-contiguous")
    M, K = a.shape
    N = w.shape[0]
    if out is None:
        out = torch.empty(M, N, dtype=out_dtype, device=a.device)
    _req(out.stride(1) == 1 and out.shape == (M, N), "gemm out must be [M,N] row-major")
    if residual is not None:
        _req(residual.dim() == 2 and residual.stride(1) == 1 and residual.shape[1] == N, "gemm residual must be [*,N]")
    check(lib().msam2_gemm(_p(a), a.stride(0), _p(w), w.stride(0), _p(bias), _p(colscale), _p(residual),
                                residual.stride(0) if residual is not None else 0,
                                _is_bf16(residual) if residual is not None else 0, res_mod, _p(out), out.stride(0),
                                _is_bf16(out), M, N, K, act, _stream()))
    return out


def gemm_tokens(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, *, addend: Optional[torch.Tensor] = None,
                add_cols: int = 0, act: int = ACT_NONE, residual: Optional[torch.Tensor] = None, out_dtype: torch.dtype = OP16) -> torch.Tensor:
    """out[M,N] = residual + act((a [+ addend on columns < add_cols]) @ w^T + bias) for M <= 32 fp32 token rows (a, addend fp32
    [M,K] row-major, w 16-bit [N,K]); add_cols must be a multiple of 32 (or >= N for all columns)."""
    _req(a.dim() == 2 and a.dtype == F32 and a.stride(1) == 1 and a.shape[0] <= 32, "gemm_tokens: a must be fp32 [M<=32, K]")
    _req(w.dtype == OP16 and w.stride(1) == 1 and w.shape[1] == a.shape[1], "gemm_tokens: w must be 16-bit [N, K]")
    if addend is not None:
        _req(addend.dtype == F32 and addend.shape == a.shape and addend.stride(1) == 1 and addend.stride(0) == a.stride(0),
             "gemm_tokens: addend must match a")
    M, K = a.shape
    N = w.shape[0]
    out = torch.empty(M, N, dtype=out_dtype, device=a.device)
    if residual is not None:
        _req(residual.dtype == F32 and residual.shape == (M, N) and residual.stride(1) == 1, "gemm_tokens: residual must be fp32 [M,N]")
    check(lib().msam2_gemm_tokens(_p(a), a.stride(0), _p(addend), add_cols if addend is not None else 0, _p(w), w.stride(0), _p(bias),
                                  _p(residual), residual.stride(0) if residual is not None else 0, _p(out), out.stride(0), _is_bf16(out),
                                  M, N, K, act, _stream()))
    return out


def gemm_pool2x2(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], B: int, H: int, W: int) -> torch.Tensor:
    """fp32 [B*(H/2)*(W/2), N] = maxpool2x2(a @ w^T + bias) over the [B,H,W] token image a [B*H*W, K] (16-bit, K-contiguous)."""
    _req(a.dim() == 2 and w.dim() == 2 and a.shape[1] == w.shape[1] and a.shape[0] == B * H * W, "gemm_pool2x2 shapes")
    _req(a.dtype == OP16 and w.dtype == OP16 and a.stride(1) == 1 and w.stride(1) == 1, "gemm_pool2x2 operands: 16-bit, K-contiguous")
    N, K = w.shape
    out = torch.empty(B * (H // 2) * (W // 2), N, dtype=F32, device=a.device)
    check(lib().msam2_gemm_pool2x2(_p(a), a.stride(0), _p(w), w.stride(0), _p(bias), _p(out), out.stride(0), B, H, W, N, K, _stream()))
    return out


def gemm_qkv_pool2x2(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], B: int, H: int, W: int, q_cols: int):
    """Fused qkv projection of a q-pooling block: returns (qkv 16-bit [B*H*W, N] whose columns >= q_cols hold k | v in image order --
    the q columns are left unwritten --, q_pooled 16-bit [B*(H/2)*(W/2), q_cols] = maxpool2x2 of the q columns)."""
    _req(a.dim() == 2 and w.dim() == 2 and a.shape[1] == w.shape[1] and a.shape[0] == B * H * W, "gemm_qkv_pool2x2 shapes")
    _req(a.dtype == OP16 and w.dtype == OP16 and a.stride(1) == 1 and w.stride(1) == 1, "gemm_qkv_pool2x2 operands: 16-bit, K-contiguous")
    N, K = w.shape
    qkv = torch.empty(B * H * W, N, dtype=OP16, device=a.device)
    q2 = torch.empty(B * (H // 2) * (W // 2), q_cols, dtype=OP16, device=a.device)
    check(lib().msam2_gemm_qkv_pool2x2(_p(a), a.stride(0), _p(w), w.stride(0), _p(bias), _p(qkv), qkv.stride(0), _p(q2), q2.stride(0),
                                       B, H, W, N, K, q_cols, _stream()))
    return qkv, q2


def gemm_rope(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], table: Tuple[torch.Tensor, torch.Tensor], *,
              rope_cols: int, head_dim: int, rows_per_batch: int, n_rope: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """16-bit out[M,N] = rope(a @ w^T + bias): columns < rope_cols (whole heads of head_dim channels) of rows whose position
    m % rows_per_batch is < n_rope are rotated with table row (m % rows_per_batch) % n_pos -- the q/k projections of RoPEAttention
    with the rotation applied to the fp32 accumulator in the store."""
    _req(a.dim() == 2 and w.dim() == 2 and a.shape[1] == w.shape[1], f"gemm_rope shapes {tuple(a.shape)} x {tuple(w.shape)}")
    _req(a.dtype == OP16 and w.dtype == OP16 and a.stride(1) == 1 and w.stride(1) == 1, "gemm_rope operands: 16-bit, K-contiguous")
    cs, sn = table
    _req(cs.dtype == F32 and cs.is_contiguous() and sn.is_contiguous() and cs.shape == sn.shape and cs.shape[1] * 2 == head_dim,
         "gemm_rope: table must be cos/sin fp32 [n_pos, head_dim/2]")
    M, K = a.shape
    N = w.shape[0]
    if out is None:
        out = torch.empty(M, N, dtype=OP16, device=a.device)
    _req(out.dtype == OP16 and out.shape == (M, N) and out.stride(1) == 1, "gemm_rope: out must be 16-bit [M, N] row-major")
    check(lib().msam2_gemm_rope(_p(a), a.stride(0), _p(w), w.stride(0), _p(bias), _p(out), out.stride(0), M, N, K, _p(cs), _p(sn),
                                rope_cols, head_dim, rows_per_batch, n_rope, cs.shape[0], _stream()))
    return out


def layernorm(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, eps: float, *, act: int = ACT_NONE,
              out_dtype: torch.dtype = OP16) -> torch.Tensor:
    """Row LayerNorm over the last dim of a [rows, C] tensor (C contiguous)."""
    C = x.shape[-1]
    x2 = x.reshape(-1, C)
    _req(x2.stride(1) == 1, "layernorm input rows must be contiguous")
    y = torch.empty(x2.shape, dtype=out_dtype, device=x.device)
    check(lib().msam2_layernorm(_p(x2), _is_bf16(x2), x2.stride(0), _p(weight), _p(bias), _p(y), _is_bf16(y), y.stride(0),
                                x2.shape[0], C, eps, act, _stream()))
    return y.reshape(x.shape)


def layernorm_dual(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, eps: float):
    """fp32 [rows, C] -> (LayerNorm rows in fp32, the same rows in the 16-bit operand type) from one launch"""
    C = x.shape[-1]
    x2 = x.reshape(-1, C)
    _req(x2.dtype == F32 and x2.stride(1) == 1 and C % 4 == 0, "layernorm_dual: fp32 rows, C % 4 == 0")
    y = torch.empty(x2.shape, dtype=F32, device=x.device)
    y16 = torch.empty(x2.shape, dtype=OP16, device=x.device)
    check(lib().msam2_layernorm_dual(_p(x2), x2.stride(0), _p(weight), _p(bias), _p(y), y.stride(0), _p(y16), y16.stride(0), x2.shape[0], C, eps,
                                     _stream()))
    return y.reshape(x.shape), y16.reshape(x.shape)


def ln_mlp_residual_supported(dim: int) -> bool:
    return bool(lib().msam2_ln_mlp_residual_supported(dim))


def mlp_fused_permute_w2(w2: torch.Tensor) -> torch.Tensor:
    """kernel-ready copy of fc2's weight [dim, hidden] (16-bit) for ln_mlp_residual"""
    _req(w2.dim() == 2 and w2.dtype == OP16 and w2.is_contiguous() and w2.shape[1] % 32 == 0, "mlp_fused_permute_w2: 16-bit [dim, hidden]")
    out = torch.empty_like(w2)
    check(lib().msam2_mlp_fused_permute_w2(_p(w2), _p(out), w2.shape[0], w2.shape[1], _stream()))
    return out


def ln_mlp_residual(x: torch.Tensor, ln_w: torch.Tensor, ln_b: torch.Tensor, eps: float, w1: torch.Tensor, b1: torch.Tensor, w2p: torch.Tensor,
                    b2: torch.Tensor, also16: bool = False):
    """fp32 [T, dim] = x + fc2(GELU(fc1(LayerNorm(x)))) in one kernel (dim 96 / 192, and 384 -- opt-in in the trunk: ln_mlp_residual_supported; w2p from
    mlp_fused_permute_w2, whose layout depends on dim).
    also16: returns (fp32 rows, the same rows in the 16-bit operand type) -- written by the same store."""
    T, dim = x.shape
    _req(x.dtype == F32 and x.is_contiguous() and w1.dtype == OP16 and w2p.dtype == OP16 and w1.shape == (4 * dim, dim) and w2p.shape == (dim, 4 * dim)
         and w1.is_contiguous() and w2p.is_contiguous(), "ln_mlp_residual: x fp32 [T, dim], w1 [4 dim, dim], w2p [dim, 4 dim] 16-bit contiguous")
    out = torch.empty_like(x)
    if also16:
        out16 = torch.empty(x.shape, dtype=OP16, device=x.device)
        check(lib().msam2_ln_mlp_residual_fwd_dual(_p(x), T, dim, _p(ln_w), _p(ln_b), float(eps), _p(w1), _p(b1), _p(w2p), _p(b2), _p(out), _p(out16),
                                                   _stream()))
        return out, out16
    check(lib().msam2_ln_mlp_residual_fwd(_p(x), T, dim, _p(ln_w), _p(ln_b), float(eps), _p(w1), _p(b1), _p(w2p), _p(b2), _p(out), _stream()))
    return out


def _strides3(t: torch.Tensor) -> "ctypes.Array":
    return (ctypes.c_int64 * 3)(t.stride(0), t.stride(1), t.stride(2))


def attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, *, splits: int = 1,
              out: Optional[torch.Tensor] = None, scale: Optional[float] = None, workspace: Optional[torch.Tensor] = None,
              defer_merge: bool = False, lse: Optional[torch.Tensor] = None, dropout: Optional[tuple] = None) -> torch.Tensor:
    """softmax(q k^T / sqrt(D)) v.  q [B,H,Lq,D], k/v [B,H,Lk,D] as (possibly strided) bf16 views with D contiguous.
    Returns [B,H,Lq,D] view of a [B,Lq,H,D] buffer (heads recombined for the following out-projection).
    defer_merge (needs a caller-owned `workspace`): run the split-KV pass only; finish with attention_merge().
    lse (fp32 [B,H,Lq] contiguous): also return the log-sum-exp rows (log2 domain) the backward needs; runs with >= 2 splits.
    dropout = (p, seed, offset) with lse: train-mode dropout on the attention probabilities inside the flash kernel (seed: int or
    DeviceSeed; probability (b,h,q,k) = element offset + ((b*H+h)*Lq+q)*Lk+k of the stream); `backward.attention_backward(...,
    dropout=...)` re-creates the mask."""
    B, H, Lq, D = q.shape
    Lk = k.shape[2]
    for t in (q, k, v):
        _req(t.dtype == OP16 and t.stride(3) == 1, "attention tensors must be 16-bit (ops.OP16) with contiguous head dim")
    if out is None:
        out = torch.empty(B, Lq, H, D, dtype=OP16, device=q.device).permute(0, 2, 1, 3)
    if lse is not None:
        _req(lse.dtype == F32 and lse.is_contiguous() and lse.numel() == B * H * Lq and not defer_merge, "attention: lse must be fp32 [B,H,Lq]")
        splits = max(splits, 2)
    ws_bytes = lib().msam2_attention_workspace_bytes(B, H, Lq, D, splits)
    ws = workspace if workspace is not None else (torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=q.device) if splits > 1 else None)
    _req(not defer_merge or (workspace is not None and splits > 1), "defer_merge needs splits > 1 and a caller-owned workspace")
    _req(ws is None or ws.numel() * ws.element_size() >= ws_bytes, "attention workspace too small")
    sc = scale if scale is not None else 1.0 / math.sqrt(D)
    if dropout is not None and dropout[0] > 0:
        _req(lse is not None, "attention: dropout is a training feature (pass lse)")
        pd, seed, offset = dropout
        seed_dev = None
        if isinstance(seed, DeviceSeed):
            seed, seed_dev = seed.base, seed.dev
        check(lib().msam2_attention_fwd_lse_dropout(_p(q), _strides3(q), _p(k), _strides3(k), _p(v), _strides3(v), _p(out), _strides3(out),
                                                    B, H, Lq, Lk, D, sc, splits, _p(ws), ws_bytes if ws is not None else 0, _p(lse), float(pd),
                                                    int(seed) & (2 ** 64 - 1), int(offset) & (2 ** 64 - 1), _p(seed_dev), _stream()))
        return out
    if lse is not None:
        check(lib().msam2_attention_fwd_lse(_p(q), _strides3(q), _p(k), _strides3(k), _p(v), _strides3(v), _p(out), _strides3(out),
                                            B, H, Lq, Lk, D, sc, splits, _p(ws), ws_bytes if ws is not None else 0, _p(lse), _stream()))
        return out
    check(lib().msam2_attention_fwd(_p(q), _strides3(q), _p(k), _strides3(k), _p(v), _strides3(v), _p(out), _strides3(out),
                                    B, H, Lq, Lk, D, sc, -splits if defer_merge else splits, _p(ws), ws_bytes if ws is not None else 0, _stream()))
    return out


def attention_kv64(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, *, splits: int = 1, scale: Optional[float] = None,
                   workspace: Optional[torch.Tensor] = None, defer_merge: bool = False,
                   key_count: Optional[torch.Tensor] = None) -> torch.Tensor:
    """softmax(q k^T / sqrt(256)) v for q [B,H,Lq,256], k [B,H,Lk,256] and 64-wide value rows v [B,H,Lk,64] (the memory bank itself:
    the memory cross-attention with v_proj folded into out_proj).  Returns the [B,H,Lq,64] view of a [B,Lq,H,64] buffer.
    defer_merge: the split pass only (finish with attention_merge on the returned view).
    key_count: int32 device scalar in [1, Lk]: only the first `key_count` keys are attended to, Lk being the capacity the launch is
    shaped for (a hipGraph captured for a padded memory bank serves every fill level)."""
    B, H, Lq, D = q.shape
    Lk = k.shape[2]
    _req(D == 256 and k.shape[3] == 256 and v.shape[3] == 64 and v.shape[2] == Lk, "attention_kv64: q/k rows of 256, v rows of 64")
    _req(key_count is None or (key_count.dtype == torch.int32 and key_count.numel() == 1 and key_count.device == q.device),
         "attention_kv64: key_count is one int32 on the tensors' device")
    for t in (q, k, v):
        _req(t.dtype == OP16 and t.stride(3) == 1, "attention tensors must be 16-bit (ops.OP16) with contiguous head dim")
    out = torch.empty(B, Lq, H, 64, dtype=OP16, device=q.device).permute(0, 2, 1, 3)
    ws_bytes = lib().msam2_attention_workspace_bytes(B, H, Lq, 64, splits)
    ws = workspace if workspace is not None else (torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=q.device) if splits > 1 else None)
    _req(not defer_merge or (workspace is not None and splits > 1), "defer_merge needs splits > 1 and a caller-owned workspace")
    _req(ws is None or ws.numel() * ws.element_size() >= ws_bytes, "attention workspace too small")
    sc = scale if scale is not None else 1.0 / math.sqrt(D)
    if key_count is not None:
        check(lib().msam2_attention_kv64_dyn_fwd(_p(q), _strides3(q), _p(k), _strides3(k), _p(v), _strides3(v), _p(out), _strides3(out),
                                                 B, H, Lq, Lk, _p(key_count), sc, -splits if defer_merge else splits, _p(ws),
                                                 ws_bytes if ws is not None else 0, _stream()))
        return out
    check(lib().msam2_attention_kv64_fwd(_p(q), _strides3(q), _p(k), _strides3(k), _p(v), _strides3(v), _p(out), _strides3(out),
                                         B, H, Lq, Lk, sc, -splits if defer_merge else splits, _p(ws), ws_bytes if ws is not None else 0, _stream()))
    return out


def attention_effective_splits(Lk: int, splits: int) -> int:
    return int(lib().msam2_attention_effective_splits(Lk, splits))


def attention_kv64_partial(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, *, splits: int, split_begin: int, split_count: int,
                           workspace: torch.Tensor, scale: Optional[float] = None, key_count: Optional[torch.Tensor] = None) -> None:
    """Splits [split_begin, split_begin + split_count) of a `splits`-way attention_kv64 (effective count) into `workspace`; finish with
    attention_merge on a [B,H,Lq,64] output view once every slot is filled (parallel.KVSplit all-gathers the other ranks' slots)."""
    B, H, Lq, D = q.shape
    Lk = k.shape[2]
    for t in (q, k, v):
        _req(t.dtype == OP16 and t.stride(3) == 1, "attention tensors must be 16-bit (ops.OP16) with contiguous head dim")
    sc = scale if scale is not None else 1.0 / math.sqrt(D)
    if key_count is not None:
        _req(key_count.dtype == torch.int32 and key_count.numel() == 1, "attention_kv64_partial: key_count is one int32 on the device")
        check(lib().msam2_attention_kv64_dyn_partial(_p(q), _strides3(q), _p(k), _strides3(k), _p(v), _strides3(v), B, H, Lq, Lk, _p(key_count),
                                                     sc, splits, split_begin, split_count, _p(workspace),
                                                     workspace.numel() * workspace.element_size(), _stream()))
        return
    check(lib().msam2_attention_kv64_partial(_p(q), _strides3(q), _p(k), _strides3(k), _p(v), _strides3(v), B, H, Lq, Lk, sc, splits,
                                             split_begin, split_count, _p(workspace), workspace.numel() * workspace.element_size(), _stream()))


def attention_workspace(B: int, H: int, Lq: int, D: int, splits: int, device) -> torch.Tensor:
    return torch.empty(max(lib().msam2_attention_workspace_bytes(B, H, Lq, D, splits), 1), dtype=torch.uint8, device=device)


def attention_merge(out: torch.Tensor, Lk: int, splits: int, workspace: torch.Tensor) -> torch.Tensor:
    """Finish attention(..., defer_merge=True): out is the [B,H,Lq,D] view that call returned."""
    B, H, Lq, D = out.shape
    check(lib().msam2_attention_merge(_p(out), _strides3(out), B, H, Lq, Lk, D, splits, _p(workspace),
                                      workspace.numel() * workspace.element_size(), _stream()))
    return out


def window_attention(qkv: torch.Tensor, B: int, H: int, W: int, heads: int, ws: int, qkv_bias: torch.Tensor,
                     q_pooled: Optional[torch.Tensor] = None, scale: Optional[float] = None) -> torch.Tensor:
    """Hiera windowed MHA straight from the fused qkv tokens [B*H*W, 3*heads*D] (bf16); if q_pooled is given
    ([B*(H/2)*(W/2), heads*D]) queries come from it with window ws/2.  Returns o [B*Hq*Wq, heads*D] bf16."""
    dim_out = qkv.shape[1] // 3
    D = dim_out // heads
    _req(qkv.dtype == OP16 and qkv.stride(1) == 1, "qkv must be 16-bit (ops.OP16) row-major")
    if q_pooled is None:
        qt, q_ts, hq, wq, ws_q = qkv, qkv.stride(0), H, W, ws
    else:
        qt, q_ts, hq, wq, ws_q = q_pooled, q_pooled.stride(0), H // 2, W // 2, ws // 2
    o = torch.empty(B * hq * wq, dim_out, dtype=OP16, device=qkv.device)
    kpad = qkv_bias[dim_out:2 * dim_out]
    vpad = qkv_bias[2 * dim_out:]
    kptr = qkv.data_ptr() + dim_out * 2
    vptr = qkv.data_ptr() + 2 * dim_out * 2
    check(lib().msam2_window_attention_fwd(_p(qt), q_ts, D, hq, wq, ws_q, kptr, vptr, qkv.stride(0), D, H, W, ws, _p(kpad),
                                           _p(vpad), _p(o), o.stride(0), D, B, heads, D, scale if scale is not None else 1.0 / math.sqrt(D), _stream()))
    return o


def attention_small(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, heads: int) -> torch.Tensor:
    """Decoder attention with head dim 16/32: q [B,Lq,C], k/v [B,Lk,C] bf16 (C = heads*D contiguous) -> [B,Lq,C] bf16."""
    B, Lq, C = q.shape
    Lk = k.shape[1]
    D = C // heads
    for t in (q, k, v):
        _req(t.dtype == OP16 and t.stride(2) == 1, "attention_small tensors must be 16-bit (ops.OP16), channel-contiguous")
    o = torch.empty(B, Lq, C, dtype=OP16, device=q.device)
    check(lib().msam2_attention_small_fwd(_p(q), q.stride(0), q.stride(1), _p(k), k.stride(0), k.stride(1), _p(v), v.stride(0),
                                          v.stride(1), _p(o), o.stride(0), o.stride(1), B, heads, Lq, Lk, D,
                                          1.0 / math.sqrt(D), _stream()))
    return o


def add_cast(a: torch.Tensor, b: Optional[torch.Tensor] = None, alpha: float = 1.0, out_dtype: torch.dtype = OP16,
             out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out = a + alpha*b over a logical [D0, D1, C] volume (C contiguous in a and b; outer dims may be strided or, for b,
    broadcast with stride 0).  Result is contiguous (or written into the contiguous `out`)."""
    _req(a.dim() == 3 and (a.stride(2) == 1 or a.shape[2] == 1), "add_cast: a must be [D0,D1,C] with contiguous C")
    D0, D1, C = a.shape
    bs0 = bs1 = 0
    if b is not None:
        b = b.expand(D0, D1, C)
        _req(b.stride(2) == 1 or C == 1, "add_cast: b must have contiguous C")
        bs0, bs1 = b.stride(0), b.stride(1)
    if out is None:
        out = torch.empty(D0, D1, C, dtype=out_dtype, device=a.device)
    _req(out.is_contiguous() and out.numel() == D0 * D1 * C, "add_cast: out must be a contiguous [D0,D1,C] buffer")
    check(lib().msam2_add_cast(_p(a), _is_bf16(a), a.stride(0), a.stride(1), _p(b), _is_bf16(b) if b is not None else 0, bs0,
                               bs1, alpha, _p(out), _is_bf16(out), D0, D1, C, _stream()))
    return out


class DeviceSeed:
    """Stream id of a dropout call whose low part lives on the device: effective seed = base + *dev (int64 [1] tensor, written by
    `counter_bump`).  A hipGraph replay of a training step then draws new masks, because the counter is advanced by a kernel of the
    step instead of being baked into the captured arguments."""
    __slots__ = ("base", "dev")

    def __init__(self, base: int, dev: torch.Tensor):
        assert dev.dtype == torch.int64 and dev.numel() == 1 and dev.is_cuda
        self.base, self.dev = int(base), dev


def counter_bump(counter: torch.Tensor, snapshot: Optional[torch.Tensor] = None) -> None:
    """counter[0] += 1 on the device (int64 [1]); snapshot[0] = the new value."""
    _req(counter.dtype == torch.int64 and counter.numel() == 1 and (snapshot is None or (snapshot.dtype == torch.int64 and snapshot.numel() == 1)),
         "counter_bump: int64 [1] tensors")
    check(lib().msam2_counter_bump(_p(counter), _p(snapshot), _stream()))


def dropout(x: torch.Tensor, p: float, seed, offset: int, residual: Optional[torch.Tensor] = None,
            out_dtype: Optional[torch.dtype] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """y = residual + (keep ? x / (1 - p) : 0) on a [rows, cols] map (row-major, rows may be strided); the mask is element
    (offset + r * cols + c) of the counter-based stream `seed` (an int, or a `DeviceSeed`), so calling it again on a gradient with the
    same (seed, offset) is the backward.  residual: fp32 [rows, cols]."""
    seed_dev = None
    if isinstance(seed, DeviceSeed):
        seed, seed_dev = seed.base, seed.dev
    _req(x.dim() == 2 and x.stride(1) == 1, "dropout: [rows, cols] row-major")
    rows, cols = x.shape
    y = out if out is not None else torch.empty(rows, cols, dtype=out_dtype or x.dtype, device=x.device)
    _req(y.shape == x.shape and y.stride(1) == 1, "dropout: out must be [rows, cols] row-major (rows may be strided)")
    if residual is not None:
        _req(residual.dtype == F32 and residual.shape == x.shape and residual.stride(1) == 1, "dropout: residual must be fp32 [rows, cols]")
    check(lib().msam2_dropout(_p(x), _is_bf16(x), x.stride(0), _p(residual), residual.stride(0) if residual is not None else 0, _p(y), _is_bf16(y),
                              y.stride(0), rows, cols, float(p), int(seed) & (2 ** 64 - 1), int(offset) & (2 ** 64 - 1), _p(seed_dev), _stream()))
    return y


def add_cast_into(out: torch.Tensor, a: torch.Tensor, b: Optional[torch.Tensor] = None, alpha: float = 1.0) -> torch.Tensor:
    """Strided gather/copy(+add) of a [D0,D1,C] view into a contiguous slice of a larger buffer (memory-bank assembly)."""
    return add_cast(a, b, alpha, out=out)


def gate_no_obj_(x: torch.Tensor, score: torch.Tensor, value: float) -> torch.Tensor:
    """x[b] = value where score[b] <= 0 (fp32, contiguous, in place)."""
    _req(x.dtype == F32 and x.is_contiguous(), "gate_no_obj: fp32 contiguous")
    B = x.shape[0]
    check(lib().msam2_gate_rows(_p(x), _p(score), value, B, x.numel() // B, _stream()))
    return x


def any_positive(x: torch.Tensor) -> torch.Tensor:
    """[B, ...] fp32 -> [B, 1] fp32 (1.0 where any element of the row is > 0)."""
    _req(x.dtype == F32 and x.is_contiguous(), "any_positive: fp32 contiguous")
    B = x.shape[0]
    out = torch.empty(B, 1, dtype=F32, device=x.device)
    check(lib().msam2_any_positive(_p(x), _p(out), B, x.numel() // B, _stream()))
    return out


def maxpool2x2(x: torch.Tensor, B: int, H: int, W: int, out_dtype: Optional[torch.dtype] = None) -> torch.Tensor:
    """x: [B*H*W, C] token-major view (row stride may exceed C) -> [B*(H/2)*(W/2), C]."""
    C = x.shape[1]
    _req(x.stride(1) == 1, "maxpool2x2: channel dim must be contiguous")
    y = torch.empty(B * (H // 2) * (W // 2), C, dtype=out_dtype or x.dtype, device=x.device)
    check(lib().msam2_maxpool2x2(_p(x), _is_bf16(x), x.stride(0), _p(y), _is_bf16(y), y.stride(0), B, H, W, C, _stream()))
    return y


def upsample2x_add_(y: torch.Tensor, top: torch.Tensor, B: int, H: int, W: int) -> torch.Tensor:
    """y[B,H,W,C] += nearest2x(top[B,H/2,W/2,C]) in place (fp32, contiguous)."""
    _req(y.dtype == F32 and top.dtype == F32 and y.is_contiguous() and top.is_contiguous(), "upsample2x_add: fp32 contiguous")
    check(lib().msam2_upsample2x_add(_p(y), _p(top), B, H, W, y.shape[-1], _stream()))
    return y


def rope_table(side: int, D: int, theta: float, device) -> Tuple[torch.Tensor, torch.Tensor]:
    cs = torch.empty(side * side, D // 2, dtype=F32, device=device)
    sn = torch.empty_like(cs)
    check(lib().msam2_rope_table(_p(cs), _p(sn), side, D, theta, _stream()))
    return cs, sn


def rope_(x: torch.Tensor, n_rope: int, table: Tuple[torch.Tensor, torch.Tensor]) -> torch.Tensor:
    """Rotate rows l < n_rope of every batch of x [B, L, D] (bf16 view, D contiguous) in place."""
    B, L, D = x.shape
    _req(x.dtype == OP16 and x.stride(2) == 1, "rope: 16-bit with contiguous D")
    cs, sn = table
    check(lib().msam2_rope_inplace(_p(x), x.stride(0), x.stride(1), B, L, n_rope, cs.shape[0], D, _p(cs), _p(sn), _stream()))
    return x


def bilinear_upsample(x: torch.Tensor, H: int, W: int) -> torch.Tensor:
    """fp32 [..., h, w] -> [..., H, W], align_corners=False."""
    _req(x.dtype == F32 and x.is_contiguous(), "bilinear: fp32 contiguous")
    h, w = x.shape[-2:]
    planes = x.numel() // (h * w)
    y = torch.empty(*x.shape[:-2], H, W, dtype=F32, device=x.device)
    check(lib().msam2_bilinear_upsample(_p(x), _p(y), planes, h, w, H, W, _stream()))
    return y


def sine_pos_2d(h: int, w: int, C: int, device, temperature: float = 10000.0) -> torch.Tensor:
    out = torch.empty(h * w, C, dtype=F32, device=device)
    check(lib().msam2_sine_pos_2d(_p(out), h, w, C, temperature, _stream()))
    return out


def fourier_pe_grid(gauss: torch.Tensor, h: int, w: int) -> torch.Tensor:
    C = 2 * gauss.shape[1]
    out = torch.empty(h * w, C, dtype=F32, device=gauss.device)
    check(lib().msam2_fourier_pe_grid(_p(out), _p(gauss.contiguous()), h, w, C, _stream()))
    return out


def hiera_pos_embed(pos_embed: torch.Tensor, pos_embed_window: torch.Tensor, h: int, w: int) -> torch.Tensor:
    _, C, bh, bw = pos_embed.shape
    out = torch.empty(h * w, C, dtype=F32, device=pos_embed.device)
    check(lib().msam2_hiera_pos_embed(_p(out), _p(pos_embed.contiguous()), _p(pos_embed_window.contiguous()), C, bh, bw, h, w,
                                      pos_embed_window.shape[-1], _stream()))
    return out


def im2col_patch(img: torch.Tensor) -> torch.Tensor:
    """img fp32 [B,3,S,S] -> bf16 [B*(S/4)^2, 160]."""
    _req(img.dtype == F32 and img.is_contiguous() and img.shape[1] == 3 and img.shape[2] == img.shape[3], "im2col_patch: [B,3,S,S] fp32")
    B, _, S, _ = img.shape
    out = torch.empty(B * (S // 4) ** 2, 160, dtype=OP16, device=img.device)
    check(lib().msam2_im2col_patch7x7s4(_p(img), _p(out), B, S, _stream()))
    return out


def patch_embed_supported(S: int, E: int) -> bool:
    """the one-kernel patch embedding (msam2_patch_embed7x7s4) takes this problem: whole 32-token wave blocks per token row, E <= 128"""
    So = S // 4
    return S % 4 == 0 and So % 32 == 0 and E <= 128


def patch_embed(img: torch.Tensor, w_perm: torch.Tensor, bias: torch.Tensor, pos: Optional[torch.Tensor] = None) -> torch.Tensor:
    """img fp32 [B,3,S,S] -> fp32 tokens [B*(S/4)^2, E] = conv7x7/s4/p3 + bias (+ pos [(S/4)^2, E], broadcast over the batch) in ONE kernel, no
    im2col map.  w_perm: 16-bit [ceil(E/32)*32, 176] in the kernel's reduction order, zero tap first (modeling.encoder.PatchEmbed._weight_perm)."""
    _req(img.dtype == F32 and img.is_contiguous() and img.shape[1] == 3 and img.shape[2] == img.shape[3], "patch_embed: [B,3,S,S] fp32")
    B, _, S, _ = img.shape
    E = bias.shape[0]
    _req(patch_embed_supported(S, E), "patch_embed: unsupported size (use im2col_patch + gemm)")
    _req(w_perm.dtype == OP16 and w_perm.is_contiguous() and tuple(w_perm.shape) == ((E + 31) // 32 * 32, 176), "patch_embed: w_perm [ceil32(E), 176] 16-bit")
    _req(bias.dtype == F32 and bias.is_contiguous(), "patch_embed: fp32 bias")
    if pos is not None:
        _req(pos.dtype == F32 and pos.is_contiguous() and tuple(pos.shape) == ((S // 4) ** 2, E), "patch_embed: pos fp32 [(S/4)^2, E]")
    out = torch.empty(B * (S // 4) ** 2, E, dtype=F32, device=img.device)
    check(lib().msam2_patch_embed7x7s4(_p(img), _p(w_perm), _p(bias), _p(pos), _p(out), B, S, E, _stream()))
    return out


def im2col3x3s2(x: torch.Tensor, B: int, H: int, W: int) -> torch.Tensor:
    """bf16 NHWC -> [B*(H/2)*(W/2), ld] patches, columns (ky,kx,c), ld = 9*C rounded up to a multiple of 8 (zero filled)."""
    C = x.shape[-1]
    _req(x.dtype == OP16 and x.is_contiguous(), "im2col3x3s2: 16-bit contiguous NHWC")
    ld = (9 * C + 7) // 8 * 8
    out = torch.empty(B * (H // 2) * (W // 2), ld, dtype=OP16, device=x.device)
    check(lib().msam2_im2col3x3s2(_p(x), _p(out), B, H, W, C, ld, _stream()))
    return out


def conv3x3s2_ln_gelu(x: torch.Tensor, B: int, H: int, W: int, weight, bias, ln_w, ln_b, mask_mode: int = 0,
                      mask_scale: float = 0.0, mask_bias: float = 0.0) -> torch.Tensor:
    cout, cin = weight.shape[0], weight.shape[1]
    y = torch.empty(B * (H // 2) * (W // 2), cout, dtype=OP16, device=x.device)
    check(lib().msam2_conv3x3s2_ln_gelu(_p(x), _is_bf16(x), _p(weight), _p(bias), _p(ln_w), _p(ln_b), _p(y), B, H, W, cin, cout,
                                        mask_mode, mask_scale, mask_bias, _stream()))
    return y


def dwconv7x7_ln(x: torch.Tensor, B: int, H: int, W: int, w_tap_major, bias, ln_w, ln_b) -> torch.Tensor:
    _req(x.dtype == F32 and x.is_contiguous(), "dwconv7x7_ln: fp32 contiguous NHWC")
    C = x.shape[-1]
    y = torch.empty(B * H * W, C, dtype=OP16, device=x.device)
    check(lib().msam2_dwconv7x7_ln(_p(x), _p(w_tap_major), _p(bias), _p(ln_w), _p(ln_b), _p(y), B, H, W, C, _stream()))
    return y


def convt2x2_shuffle(g: torch.Tensor, bias, skip: torch.Tensor, ln_w, ln_b, B: int, h: int, w: int) -> torch.Tensor:
    """skip: the high-resolution features, 16-bit or (C = 32 / 64) fp32 as the FPN returns them -- no 16-bit copy pass in front"""
    C = g.shape[1] // 4
    _req(g.dtype == OP16 and skip.dtype in (OP16, F32) and g.is_contiguous() and skip.is_contiguous(), "convt2x2_shuffle: contiguous 16-bit g, 16-bit / fp32 skip")
    y = torch.empty(B * 4 * h * w, C, dtype=OP16, device=g.device)
    if skip.dtype == F32:
        _req(C in (32, 64), "convt2x2_shuffle: fp32 skip features need C = 32 / 64")
        check(lib().msam2_convt2x2_shuffle_f32skip(_p(g), _p(bias), _p(skip), _p(ln_w), _p(ln_b), _p(y), B, h, w, C, _stream()))
    else:
        check(lib().msam2_convt2x2_shuffle(_p(g), _p(bias), _p(skip), _p(ln_w), _p(ln_b), _p(y), B, h, w, C, _stream()))
    return y


def hyper_masks(hyper: torch.Tensor, up: torch.Tensor, n: int, P: int) -> torch.Tensor:
    K, C = hyper.shape[1], hyper.shape[2]
    masks = torch.empty(n, K, P, dtype=F32, device=up.device)
    check(lib().msam2_hyper_masks(_p(hyper.contiguous()), _p(up), _p(masks), n, K, P, C, _stream()))
    return masks


def prompt_points(xy: torch.Tensor, labels: torch.Tensor, gauss, point_emb, not_a_point, image_size: float, n_pad: int = 0) -> torch.Tensor:
    """xy fp32 [n,P,2], labels int32 [n,P] -> fp32 [n,P + n_pad,C]; the last n_pad points of every set are the padding point ((0,0), label -1)."""
    n, P = labels.shape
    C = point_emb.shape[1]
    out = torch.empty(n, P + n_pad, C, dtype=F32, device=xy.device)
    check(lib().msam2_prompt_points_padded(_p(xy.contiguous()), _p(labels.contiguous()), _p(gauss), _p(point_emb), _p(not_a_point), _p(out),
                                           n, P, n_pad, C, float(image_size), _stream()))
    return out


def select_mask(masks: torch.Tensor, ious: torch.Tensor, obj: torch.Tensor, multimask: bool, dynamic: bool, delta: float,
                thresh: float):
    n, K, h, w = masks.shape
    low = torch.empty(n, 1, h, w, dtype=F32, device=masks.device)
    sel = torch.empty(n, dtype=torch.int32, device=masks.device)
    iou_sel = torch.empty(n, 1, dtype=F32, device=masks.device)
    check(lib().msam2_select_mask(_p(masks), _p(ious), _p(obj), _p(low), _p(sel), _p(iou_sel), n, h * w, int(multimask), int(dynamic),
                                  delta, thresh, _stream()))
    return low, sel, iou_sel


def gather_rows(x: torch.Tensor, sel: Optional[torch.Tensor], offset: int = 0) -> torch.Tensor:
    n, T, C = x.shape
    y = torch.empty(n, C, dtype=F32, device=x.device)
    check(lib().msam2_gather_rows(_p(x.contiguous()), _p(sel), _p(y), n, T, C, offset, _stream()))
    return y


def obj_ptr_mix_(ptr: torch.Tensor, obj: torch.Tensor, no_obj_ptr: torch.Tensor) -> torch.Tensor:
    check(lib().msam2_obj_ptr_mix(_p(ptr), _p(obj), _p(no_obj_ptr), ptr.shape[0], ptr.shape[1], _stream()))
    return ptr


def connected_components(mask_u8: torch.Tensor):
    """Drop-in for ``sam2_train._C.get_connected_componnets`` (connected_components.cu:213-282)."""
    if not mask_u8.is_cuda:
        raise RuntimeError("inputs must be a CUDA tensor")
    if mask_u8.dim() != 4 or mask_u8.shape[1] != 1:
        raise RuntimeError("inputs must be [N, 1, H, W] shape")
    if mask_u8.dtype != torch.uint8:
        raise RuntimeError("inputs must be a uint8 type")
    N, _, H, W = mask_u8.shape
    if H % 2:
        raise RuntimeError("height must be a even number")
    if W % 2:
        raise RuntimeError("width must be a even number")
    m = mask_u8.contiguous()
    labels = torch.empty(N, 1, H, W, dtype=torch.int32, device=m.device)
    counts = torch.empty_like(labels)
    nb = lib().msam2_cc_workspace_bytes(N, H, W)
    ws = torch.empty(nb, dtype=torch.uint8, device=m.device)
    check(lib().msam2_cc_label(_p(m), _p(labels), _p(counts), N, H, W, _p(ws), nb, _stream()))
    return [labels, counts]


def fill_holes_(mask: torch.Tensor, max_area: int) -> torch.Tensor:
    """fill_holes_in_mask_scores (utils/misc.py:247-258), in place on an fp32 [N,1,H,W] tensor."""
    _req(mask.dtype == F32 and mask.is_contiguous(), "fill_holes: fp32 contiguous")
    N, _, H, W = mask.shape
    nb = lib().msam2_fill_holes_workspace_bytes(N, H, W)
    ws = torch.empty(nb, dtype=torch.uint8, device=mask.device)
    check(lib().msam2_fill_holes(_p(mask), N, H, W, max_area, _p(ws), nb, _stream()))
    return mask


# ---------------------------------------------------------------------------------------------------------------------
class HipGraph:
    """Capture everything enqueued on the current stream inside the ``with`` block; ``replay()`` relaunches it."""

    def __init__(self):
        self._exec = ctypes.c_void_p()
        self._stream = None

    def __enter__(self):
        self._stream = _stream()
        check(lib().msam2_graph_begin(self._stream))
        return self

    def __exit__(self, et, ev, tb):
        rc = lib().msam2_graph_end(self._stream, ctypes.byref(self._exec))
        if et is None:
            check(rc)
        return False

    def replay(self):
        check(lib().msam2_graph_launch(self._exec, _stream()))

    def __del__(self):
        try:
            if self._exec:
                lib().msam2_graph_destroy(self._exec)
        except Exception:
            pass


def space_to_depth(x: torch.Tensor, B: int, H: int, W: int, k: int) -> torch.Tensor:
    """NHWC [B*H*W, C] -> bf16 [B*(H/k)*(W/k), ld] patches with columns (ky,kx,c), ld = k*k*C rounded up to 8."""
    C = x.shape[-1]
    ld = (k * k * C + 7) // 8 * 8
    out = torch.empty(B * (H // k) * (W // k), ld, dtype=OP16, device=x.device)
    check(lib().msam2_space_to_depth(_p(x), _is_bf16(x), _p(out), B, H, W, C, k, ld, _stream()))
    return out


def aa_downsample(x: torch.Tensor, factor: int, in_scale: float = 1.0, in_bias: float = 0.0) -> torch.Tensor:
    """fp32 [..., H, W] -> [..., H/f, W/f] anti-aliased bilinear (of x*in_scale + in_bias)."""
    _req(x.dtype == F32 and x.is_contiguous(), "aa_downsample: fp32 contiguous")
    H, W = x.shape[-2:]
    y = torch.empty(*x.shape[:-2], H // factor, W // factor, dtype=F32, device=x.device)
    check(lib().msam2_aa_downsample(_p(x), _p(y), x.numel() // (H * W), H, W, factor, in_scale, in_bias, _stream()))
    return y


def fill_components_(mask: torch.Tensor, max_area: int, threshold: float, above: bool, fill_value: float) -> torch.Tensor:
    """In place on fp32 [N,1,H,W]: components of (mask > threshold) if `above` else (mask <= threshold) with area <= max_area
    are set to fill_value (SAM2Transforms.postprocess_masks, utils/transforms.py:74-98)."""
    _req(mask.dtype == F32 and mask.is_contiguous(), "fill_components: fp32 contiguous")
    N, _, H, W = mask.shape
    nb = lib().msam2_fill_holes_workspace_bytes(N, H, W)
    ws = torch.empty(nb, dtype=torch.uint8, device=mask.device)
    check(lib().msam2_fill_components(_p(mask), N, H, W, int(max_area), float(threshold), int(above), float(fill_value), _p(ws), nb, _stream()))
    return mask


def image_prep(img_u8_hwc: torch.Tensor, size: int, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)) -> torch.Tensor:
    """uint8 [H,W,3] on device -> fp32 [3,size,size], /255, bilinear resize, ImageNet normalisation."""
    _req(img_u8_hwc.dtype == torch.uint8 and img_u8_hwc.dim() == 3 and img_u8_hwc.shape[2] == 3 and img_u8_hwc.is_contiguous(),
         "image_prep: uint8 [H,W,3] contiguous")
    H, W, _ = img_u8_hwc.shape
    out = torch.empty(3, size, size, dtype=F32, device=img_u8_hwc.device)
    m = (ctypes.c_float * 3)(*mean)
    s = (ctypes.c_float * 3)(*std)
    check(lib().msam2_image_prep(_p(img_u8_hwc), _p(out), H, W, size, m, s, _stream()))
    return out


def non_overlap(masks: torch.Tensor) -> torch.Tensor:
    """[n,1,H,W] fp32 -> same shape: per pixel the arg-max object keeps its score, the others are clamped to <= -10."""
    m = masks.to(F32).contiguous()
    n = m.shape[0]
    out = torch.empty_like(m)
    check(lib().msam2_non_overlap(_p(m), _p(out), n, m.numel() // n, _stream()))
    return out


def token_mlp3(hs: torch.Tensor, tok: torch.Tensor, w1, b1, w2, b2, w3, b3, out_dim: torch.Tensor, sigmoid: torch.Tensor,
               packed=None):
    """G three-layer ReLU MLPs of width 256 on tokens tok[g] of hs fp32 [B, T, 256] -> fp32 [B, G, 256] (first out_dim[g] valid).
    packed = (out_offset, out_stride, total): int32 [G] device tables + the element count of the flat fp32 result, head g of batch element b
    at [out_offset[g] + b * out_stride[g] : + out_dim[g]] -- heads of different widths as contiguous tensors without slicing copies."""
    B, T, C = hs.shape
    G = tok.shape[0]
    _req(hs.dtype == F32 and hs.stride(2) == 1 and C == 256, "token_mlp3: hs must be fp32 [B,T,256] with contiguous channels")
    _req(w1.dtype == OP16 and w1.shape == (G, C, C) and w1.is_contiguous() and w2.shape == (G, C, C) and w3.shape == (G, C, C), "token_mlp3 weights")
    if packed is not None:
        off, ld, total = packed
        _req(off.dtype == torch.int32 and ld.dtype == torch.int32 and off.numel() == G and ld.numel() == G, "token_mlp3: packed tables are int32 [G]")
        out = torch.empty(total, dtype=F32, device=hs.device)
        check(lib().msam2_token_mlp3_packed(_p(hs), hs.stride(0), hs.stride(1), _p(tok), _p(w1), _p(b1), _p(w2), _p(b2), _p(w3), _p(b3),
                                            _p(out_dim), _p(sigmoid), _p(out), _p(off), _p(ld), G, B, C, _stream()))
        return out
    out = torch.empty(B, G, C, dtype=F32, device=hs.device)
    check(lib().msam2_token_mlp3(_p(hs), hs.stride(0), hs.stride(1), _p(tok), _p(w1), _p(b1), _p(w2), _p(b2), _p(w3), _p(b3),
                                 _p(out_dim), _p(sigmoid), _p(out), G, B, C, _stream()))
    return out
