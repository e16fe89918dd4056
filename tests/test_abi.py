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
