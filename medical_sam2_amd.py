"""Import shim: `import medical_sam2_amd` loads the package that lives in `medical-sam2_amd/`.

The package directory name carries a hyphen (repo layout contract), which is not a Python identifier, so this
one-file module replaces itself in ``sys.modules`` with the real package loaded from that directory.
"""
import importlib.util
import os
import sys

_root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "medical-sam2_amd")
_spec = importlib.util.spec_from_file_location(
    "medical_sam2_amd", os.path.join(_root, "__init__.py"), submodule_search_locations=[_root]
)
_mod = importlib.util.module_from_spec(_spec)
sys.modules["medical_sam2_amd"] = _mod
_spec.loader.exec_module(_mod)
