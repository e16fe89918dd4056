"""ctypes binding of oracle/cc_oracle.c (CPU ORACLE, test infrastructure only)."""
import ctypes
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libcc_oracle.so")


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def _lib():
    if not os.path.exists(_SO):
        build()
    lib = ctypes.CDLL(_SO)
    lib.cc_oracle_label.restype = ctypes.c_int
    return lib


def connected_components(mask_u8: torch.Tensor):
    """mask [N,1,H,W] uint8 -> (labels int32 [N,1,H,W], counts int32 [N,1,H,W]); contract of
    sam2_train/csrc/connected_components.cu:213-282."""
    m = np.ascontiguousarray(mask_u8.to(torch.uint8).cpu().numpy())
    N, C, H, W = m.shape
    assert C == 1
    labels = np.zeros((N, 1, H, W), np.int32)
    counts = np.zeros((N, 1, H, W), np.int32)
    scratch = np.zeros(H * W, np.int32)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    rc = _lib().cc_oracle_label(p(m), p(labels), p(counts), p(scratch), N, H, W)
    if rc != 0:
        raise RuntimeError("height and width must be even")
    return torch.from_numpy(labels), torch.from_numpy(counts)
