"""CPU checks of the drop-in boundary: the C-ABI library builds for gfx950, loads without a GPU, and exports exactly the
symbols include/msam2_hip.h declares (no compute calls here)."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    txt = open(os.path.join(ROOT, "include", "msam2_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(msam2_\w+)\s*\(", txt)))


def test_header_is_plain_c():
    subprocess.check_call(["gcc", "-fsyntax-only", "-x", "c", "-Wall", "-Werror", os.path.join(ROOT, "include", "msam2_hip.h")])
    txt = open(os.path.join(ROOT, "include", "msam2_hip.h")).read()
    assert "torch" not in re.sub(r"/\*.*?\*/", "", txt, flags=re.S).lower()
    assert "at::" not in txt


def test_library_exports_every_declared_symbol():
    from medical_sam2_amd import _lib
    _lib.build()
    L = ctypes.CDLL(_lib.LIB_PATH)
    syms = _header_symbols()
    assert len(syms) >= 40
    for s in syms:
        assert hasattr(L, s), f"{s} declared in include/msam2_hip.h but not exported"
    # the python binding table and the header agree, both directions
    assert sorted(_lib.SIGNATURES.keys()) == syms
    assert _lib.lib().msam2_version() >= 100


def test_header_matches_sources():
    before = open(os.path.join(ROOT, "include", "msam2_hip.h")).read()
    subprocess.check_call(["python", os.path.join(ROOT, "tools", "gen_header.py")], stdout=subprocess.DEVNULL)
    assert open(os.path.join(ROOT, "include", "msam2_hip.h")).read() == before, "include/msam2_hip.h is stale: run tools/gen_header.py"


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    from medical_sam2_amd import _lib
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    monkeypatch.setattr(_lib, "_lib", None)
    with pytest.raises(RuntimeError, match="only compute path"):
        _lib.lib()


def test_product_package_never_imports_oracle():
    pkg = os.path.join(ROOT, "medical-sam2_amd")
    for dp, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith(".py"):
                src = open(os.path.join(dp, fn)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f"{fn} imports the oracle"


def test_argument_errors_cross_the_abi_as_codes_not_exceptions():
    """Reference behaviour: AT_ASSERTM -> RuntimeError (connected_components.cu:215-228).  Here every entry validates its arguments
    on the host BEFORE touching the device and returns < 0 with a message in msam2_last_error(); checked without a GPU on entries of
    every kernel family (the wrappers in ops.py turn the code into RuntimeError)."""
    from medical_sam2_amd import _lib
    L = _lib.lib()
    L.msam2_last_error.restype = ctypes.c_char_p
    i64x3 = (ctypes.c_int64 * 3)(256, 256, 256)
    buf = ctypes.create_string_buffer(4096)              # a valid, 16-byte alignable host address: never dereferenced by the checks
    ptr = (ctypes.addressof(buf) + 15) & ~15
    cases = {
        "gemm: K": lambda: L.msam2_gemm(ptr, 12, ptr, 12, None, None, None, 0, 0, 0, ptr, 16, 0, 4, 4, 12, 0, None),
        "cc_label: height": lambda: L.msam2_cc_label(ptr, ptr, ptr, 1, 3, 4, ptr, 1 << 20, None),
        "attention: head dim": lambda: L.msam2_attention_fwd(ptr, i64x3, ptr, i64x3, ptr, i64x3, ptr, i64x3, 1, 1, 32, 32, 80, 0.1, 1, None, 0, None),
        "attention_bwd: head dim": lambda: L.msam2_attention_bwd(ptr, i64x3, ptr, i64x3, ptr, i64x3, ptr, i64x3, ptr, ptr, i64x3, ptr, i64x3, ptr,
                                                               i64x3, ptr, i64x3, ptr, 1 << 30, 1, 1, 32, 32, 80, 0.1, None),
        "attention_small_bwd: one side": lambda: L.msam2_attention_small_bwd(ptr, 0, 0, ptr, 0, 0, ptr, 0, 0, ptr, ptr, ptr, ptr, 1, 8, 64, 64, 16, 0.25, None),
        "gemm_tt: M, N": lambda: L.msam2_gemm_tt(ptr, 12, ptr, 16, ptr, 16, None, 12, 16, 64, None),
        "layernorm_bwd: C": lambda: L.msam2_layernorm_bwd(ptr, 2048, ptr, 0, 2048, ptr, ptr, 2048, ptr, ptr, 4, 2048, 1e-6, None, 0, None),
        "col2im3x3s2": lambda: L.msam2_col2im3x3s2(ptr, 8, ptr, 1, 3, 4, 4, None),
        "adam_step_multi": lambda: L.msam2_adam_step_multi(None, None, None, None, None, 0, 1e-4, 0.9, 0.999, 1e-8, 1, 1.0, 0.0, None, None, None),
    }
    for what, call in cases.items():
        rc = call()
        msg = L.msam2_last_error().decode()
        assert rc < 0, (what, rc)
        assert what.split(":")[0] in msg, (what, msg)
