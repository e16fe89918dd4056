"""ctypes loader of the C-ABI library (include/msam2_hip.h).  There is no fallback: if libmsam2_hip.so is missing or a
symbol is absent the import of any product module fails loudly."""
from __future__ import annotations

import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MSAM2_LIB_PATH") or os.path.join(_HERE, "libmsam2_hip.so")   # override: experiments with alternative builds

c_p = ctypes.c_void_p
c_i = ctypes.c_int
c_l = ctypes.c_int64
c_f = ctypes.c_float
c_z = ctypes.c_size_t

# name -> (restype, argtypes); mirrors include/msam2_hip.h one to one (tests/test_abi.py checks both directions)
SIGNATURES = {
    "msam2_version": (c_i, []),
    "msam2_last_error": (ctypes.c_char_p, []),
    "msam2_operand_is_fp16": (c_i, []),
    "msam2_gemm": (c_i, [c_p, c_l, c_p, c_l, c_p, c_p, c_p, c_l, c_i, c_l, c_p, c_l, c_i, c_l, c_l, c_l, c_i, c_p]),
    "msam2_gemm_pool2x2": (c_i, [c_p, c_l, c_p, c_l, c_p, c_p, c_l, c_l, c_l, c_l, c_l, c_l, c_p]),
    "msam2_gemm_qkv_pool2x2": (c_i, [c_p, c_l, c_p, c_l, c_p, c_p, c_l, c_p, c_l, c_l, c_l, c_l, c_l, c_l, c_l, c_p]),
    "msam2_gemm_tokens": (c_i, [c_p, c_l, c_p, c_l, c_p, c_l, c_p, c_p, c_l, c_p, c_l, c_i, c_l, c_l, c_l, c_i, c_p]),
    "msam2_gemm_rope": (c_i, [c_p, c_l, c_p, c_l, c_p, c_p, c_l, c_l, c_l, c_l, c_p, c_p, c_l, c_l, c_l, c_l, c_l, c_p]),
    "msam2_ln_mlp_residual_supported": (c_i, [c_l]),
    "msam2_mlp_fused_permute_w2": (c_i, [c_p, c_p, c_l, c_l, c_p]),
    "msam2_ln_mlp_residual_fwd": (c_i, [c_p, c_l, c_l, c_p, c_p, c_f, c_p, c_p, c_p, c_p, c_p, c_p]),
    "msam2_ln_mlp_residual_fwd_dual": (c_i, [c_p, c_l, c_l, c_p, c_p, c_f, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    "msam2_layernorm": (c_i, [c_p, c_i, c_l, c_p, c_p, c_p, c_i, c_l, c_l, c_l, c_f, c_i, c_p]),
    "msam2_layernorm_dual": (c_i, [c_p, c_l, c_p, c_p, c_p, c_l, c_p, c_l, c_l, c_l, c_f, c_p]),
    "msam2_attention_workspace_bytes": (c_z, [c_l, c_l, c_l, c_l, c_i]),
    "msam2_attention_fwd": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_l, c_l, c_l, c_l, c_l, c_f, c_i, c_p, c_z, c_p]),
    "msam2_attention_fwd_lse": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_l, c_l, c_l, c_l, c_l, c_f, c_i, c_p, c_z, c_p, c_p]),
    "msam2_attention_fwd_lse_dropout": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_l, c_l, c_l, c_l, c_l, c_f, c_i, c_p, c_z, c_p, c_f,
                                              ctypes.c_uint64, ctypes.c_uint64, c_p, c_p]),
    "msam2_attention_kv64_fwd": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_l, c_l, c_l, c_l, c_f, c_i, c_p, c_z, c_p]),
    "msam2_attention_kv64_partial": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_l, c_l, c_l, c_l, c_f, c_i, c_i, c_i, c_p, c_z, c_p]),
    "msam2_attention_kv64_dyn_fwd": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_l, c_l, c_l, c_l, c_p, c_f, c_i, c_p, c_z, c_p]),
    "msam2_attention_kv64_dyn_partial": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_l, c_l, c_l, c_l, c_p, c_f, c_i, c_i, c_i, c_p, c_z, c_p]),
    "msam2_attention_effective_splits": (c_i, [c_l, c_i]),
    "msam2_attention_merge": (c_i, [c_p, c_p, c_l, c_l, c_l, c_l, c_l, c_i, c_p, c_z, c_p]),
    "msam2_window_attention_fwd": (c_i, [c_p, c_l, c_l, c_l, c_l, c_l, c_p, c_p, c_l, c_l, c_l, c_l, c_l, c_p, c_p, c_p, c_l, c_l,
                                         c_l, c_l, c_l, c_f, c_p]),
    "msam2_attention_small_fwd": (c_i, [c_p, c_l, c_l, c_p, c_l, c_l, c_p, c_l, c_l, c_p, c_l, c_l, c_l, c_l, c_l, c_l, c_l, c_f,
                                        c_p]),
    "msam2_add_cast": (c_i, [c_p, c_i, c_l, c_l, c_p, c_i, c_l, c_l, c_f, c_p, c_i, c_l, c_l, c_l, c_p]),
    "msam2_maxpool2x2": (c_i, [c_p, c_i, c_l, c_p, c_i, c_l, c_l, c_l, c_l, c_l, c_p]),
    "msam2_upsample2x_add": (c_i, [c_p, c_p, c_l, c_l, c_l, c_l, c_p]),
    "msam2_rope_table": (c_i, [c_p, c_p, c_l, c_l, c_f, c_p]),
    "msam2_rope_inplace": (c_i, [c_p, c_l, c_l, c_l, c_l, c_l, c_l, c_l, c_p, c_p, c_p]),
    "msam2_bilinear_upsample": (c_i, [c_p, c_p, c_l, c_l, c_l, c_l, c_l, c_p]),
    "msam2_sine_pos_2d": (c_i, [c_p, c_l, c_l, c_l, c_f, c_p]),
    "msam2_fourier_pe_grid": (c_i, [c_p, c_p, c_l, c_l, c_l, c_p]),
    "msam2_hiera_pos_embed": (c_i, [c_p, c_p, c_p, c_l, c_l, c_l, c_l, c_l, c_l, c_p]),
    "msam2_im2col_patch7x7s4": (c_i, [c_p, c_p, c_l, c_l, c_p]),
    "msam2_patch_embed7x7s4": (c_i, [c_p, c_p, c_p, c_p, c_p, c_l, c_l, c_l, c_p]),
    "msam2_im2col3x3s2": (c_i, [c_p, c_p, c_l, c_l, c_l, c_l, c_l, c_p]),
    "msam2_conv3x3s2_ln_gelu": (c_i, [c_p, c_i, c_p, c_p, c_p, c_p, c_p, c_l, c_l, c_l, c_l, c_l, c_i, c_f, c_f, c_p]),
    "msam2_dwconv7x7_ln": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_l, c_l, c_l, c_l, c_p]),
    "msam2_convt2x2_shuffle": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_l, c_l, c_l, c_l, c_p]),
    "msam2_convt2x2_shuffle_f32skip": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_l, c_l, c_l, c_l, c_p]),
    "msam2_space_to_depth": (c_i, [c_p, c_i, c_p, c_l, c_l, c_l, c_l, c_l, c_l, c_p]),
    "msam2_aa_downsample": (c_i, [c_p, c_p, c_l, c_l, c_l, c_l, c_f, c_f, c_p]),
    "msam2_gate_rows": (c_i, [c_p, c_p, c_f, c_l, c_l, c_p]),
    "msam2_any_positive": (c_i, [c_p, c_p, c_l, c_l, c_p]),
    "msam2_transpose16": (c_i, [c_p, c_l, c_p, c_l, c_l, c_l, c_p]),
    "msam2_colsum": (c_i, [c_p, c_i, c_l, c_p, c_l, c_l, c_p]),
    "msam2_act_bwd": (c_i, [c_p, c_i, c_p, c_i, c_p, c_l, c_i, c_p]),
    "msam2_layernorm_bwd": (c_i, [c_p, c_l, c_p, c_i, c_l, c_p, c_p, c_l, c_p, c_p, c_l, c_l, c_f, c_p, c_l, c_p]),
    "msam2_softmax_rows": (c_i, [c_p, c_l, c_p, c_l, c_l, c_l, c_f, c_p]),
    "msam2_softmax_bwd_rows": (c_i, [c_p, c_l, c_p, c_l, c_p, c_l, c_l, c_l, c_f, c_p]),
    "msam2_convt2x2_gather": (c_i, [c_p, c_p, c_p, c_p, c_l, c_l, c_l, c_l, c_p]),
    "msam2_convt2x2_scatter_grad": (c_i, [c_p, c_i, c_p, c_l, c_l, c_l, c_l, c_p]),
    "msam2_bce_logits": (c_i, [c_p, c_p, c_p, c_p, c_l, c_f, c_p]),
    "msam2_attention_bwd_workspace_bytes": (c_z, [c_l, c_l, c_l, c_l]),
    "msam2_attention_bwd": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_z,
                                  c_l, c_l, c_l, c_l, c_l, c_f, c_p]),
    "msam2_attention_bwd_dropout": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_z,
                                          c_l, c_l, c_l, c_l, c_l, c_f, c_f, ctypes.c_uint64, ctypes.c_uint64, c_p, c_p]),
    "msam2_dwconv7x7": (c_i, [c_p, c_p, c_p, c_p, c_l, c_l, c_l, c_l, c_i, c_p]),
    "msam2_dwconv7x7_wgrad": (c_i, [c_p, c_p, c_p, c_l, c_l, c_l, c_l, c_p]),
    "msam2_col2im3x3s2": (c_i, [c_p, c_l, c_p, c_l, c_l, c_l, c_l, c_p]),
    "msam2_gemm_nt": (c_i, [c_p, c_l, c_p, c_l, c_p, c_p, c_l, c_p, c_l, c_i, c_l, c_l, c_l, c_p]),
    "msam2_gemm_tt": (c_i, [c_p, c_l, c_p, c_l, c_p, c_l, c_p, c_l, c_l, c_l, c_p]),
    "msam2_window_unpartition_cvt": (c_i, [c_p, c_l, c_p, c_l, c_l, c_l, c_l, c_l, c_l, c_p]),
    "msam2_window_pad_colsum": (c_i, [c_p, c_p, c_l, c_l, c_l, c_l, c_l, c_l, c_p]),
    "msam2_gemm_tt_acc": (c_i, [c_p, c_l, c_p, c_l, c_p, c_l, c_p, c_l, c_l, c_l, c_p]),
    "msam2_bilinear_upsample_bwd": (c_i, [c_p, c_p, c_l, c_l, c_l, c_l, c_l, c_p]),
    "msam2_maxpool2x2_bwd": (c_i, [c_p, c_i, c_l, c_p, c_l, c_p, c_l, c_l, c_l, c_l, c_l, c_p]),
    "msam2_window_move": (c_i, [c_p, c_l, c_p, c_p, c_l, c_l, c_l, c_l, c_l, c_l, c_i, c_i, c_p]),
    "msam2_sumpool2x2": (c_i, [c_p, c_p, c_l, c_l, c_l, c_l, c_p]),
    "msam2_hiera_pos_embed_bwd": (c_i, [c_p, c_p, c_p, c_l, c_l, c_l, c_l, c_l, c_l, c_p, c_z, c_p]),
    "msam2_hiera_pos_embed_bwd_workspace_bytes": (c_z, [c_l, c_l, c_l, c_l]),
    "msam2_dropout": (c_i, [c_p, c_i, c_l, c_p, c_l, c_p, c_i, c_l, c_l, c_l, c_f, ctypes.c_uint64, ctypes.c_uint64, c_p, c_p]),
    "msam2_counter_bump": (c_i, [c_p, c_p, c_p]),
    "msam2_adam_step": (c_i, [c_p, c_p, c_p, c_p, c_l, c_f, c_f, c_f, c_f, c_l, c_p]),
    "msam2_adam_step_multi": (c_i, [c_p, c_p, c_p, c_p, c_p, c_l, c_f, c_f, c_f, c_f, c_l, c_f, c_f, c_p, c_p, c_p]),
    "msam2_attention_small_bwd": (c_i, [c_p, c_l, c_l, c_p, c_l, c_l, c_p, c_l, c_l, c_p, c_p, c_p, c_p, c_l, c_l, c_l, c_l, c_l, c_f, c_p]),
    "msam2_seg_counts": (c_i, [c_p, c_p, c_p, c_l, c_l, c_l, c_p, c_p]),
    "msam2_non_overlap": (c_i, [c_p, c_p, c_l, c_l, c_p]),
    "msam2_token_mlp3": (c_i, [c_p, c_l, c_l, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_l, c_l, c_l, c_p]),
    "msam2_hyper_masks": (c_i, [c_p, c_p, c_p, c_l, c_l, c_l, c_l, c_p]),
    "msam2_prompt_points": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_l, c_l, c_f, c_p]),
    "msam2_prompt_points_padded": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_l, c_l, c_l, c_l, c_f, c_p]),
    "msam2_token_mlp3_packed": (c_i, [c_p, c_l, c_l, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_l, c_l, c_l, c_p]),
    "msam2_select_mask": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_l, c_l, c_i, c_i, c_f, c_f, c_p]),
    "msam2_gather_rows": (c_i, [c_p, c_p, c_p, c_l, c_l, c_l, c_l, c_p]),
    "msam2_obj_ptr_mix": (c_i, [c_p, c_p, c_p, c_l, c_l, c_p]),
    "msam2_cc_workspace_bytes": (c_z, [c_l, c_l, c_l]),
    "msam2_cc_label": (c_i, [c_p, c_p, c_p, c_l, c_l, c_l, c_p, c_z, c_p]),
    "msam2_fill_holes_workspace_bytes": (c_z, [c_l, c_l, c_l]),
    "msam2_fill_holes": (c_i, [c_p, c_l, c_l, c_l, c_i, c_p, c_z, c_p]),
    "msam2_fill_components": (c_i, [c_p, c_l, c_l, c_l, c_i, c_f, c_i, c_f, c_p, c_z, c_p]),
    "msam2_image_prep": (c_i, [c_p, c_p, c_l, c_l, c_l, ctypes.POINTER(c_f), ctypes.POINTER(c_f), c_p]),
    "msam2_graph_begin": (c_i, [c_p]),
    "msam2_graph_end": (c_i, [c_p, ctypes.POINTER(c_p)]),
    "msam2_graph_launch": (c_i, [c_p, c_p]),
    "msam2_graph_destroy": (c_i, [c_p]),
    "msam2_event_create": (c_i, [ctypes.POINTER(c_p)]),
    "msam2_event_record": (c_i, [c_p, c_p]),
    "msam2_event_elapsed_ms": (c_i, [c_p, c_p, ctypes.POINTER(c_f)]),
    "msam2_event_destroy": (c_i, [c_p]),
}


def build(force: bool = False) -> str:
    """Compile the library for gfx950 (hipcc cross-compiles; no GPU needed)."""
    if force and os.path.exists(LIB_PATH):
        os.remove(LIB_PATH)
    subprocess.check_call(["make", "-s", "-C", os.path.join(_HERE, "csrc")])
    return LIB_PATH


_lib = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the MI355X HIP library is the only compute path of this package. "
                "Build it with `python -c 'import __graft_entry__ as g; g.build()'` (or `make -C medical-sam2_amd/csrc`).")
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


class Msam2Error(RuntimeError):
    pass


def check(rc: int) -> None:
    if rc != 0:
        raise Msam2Error(lib().msam2_last_error().decode())
