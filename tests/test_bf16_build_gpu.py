"""The bf16-operand build of the library (libmsam2_hip_bf16.so, -DMSAM2_OPERAND_BF16; BASELINE.json configs[1] states bf16) through
the same parity tests as the default fp16 build: a child pytest process with MSAM2_LIB_PATH pointing at it (the operand type is a
property of the loaded library, so it cannot be switched inside one process).  The e2e tests pick the bf16 tolerances themselves
(tests/test_e2e_gpu.py: features 2e-2, logits max 0.3 / mean 0.08, IoU 0.97 per slice / 0.99 pooled -- inside the reference's own
fp32-vs-bf16-autocast disagreement)."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "medical-sam2_amd", "libmsam2_hip_bf16.so")


def test_bf16_build_passes_the_parity_suite():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    assert os.path.exists(LIB), "libmsam2_hip_bf16.so is built by __graft_entry__.build()"
    env = dict(os.environ, MSAM2_LIB_PATH=LIB)
    probe = subprocess.run([sys.executable, "-c", "import medical_sam2_amd.ops as o, torch; print(o.OP16)"], cwd=ROOT, env=env,
                           capture_output=True, text=True, timeout=300)
    assert "bfloat16" in probe.stdout, probe.stdout + probe.stderr
    sel = ("chain_hiera_s_256 or chain_hiera_t_256 or chain_hiera_bplus_256 or modules_vs_reference or long_chain or "
           "attention_vs_oracle or attention_kv64 or window_attention or gemm_exact_integers or gemm_epilogue_modes or layernorm")
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_e2e_gpu.py", "tests/test_kernels_gpu.py", "-x", "-q", "-k", sel,
                        "-p", "no:cacheprovider"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    tail = "\n".join((r.stdout + r.stderr).splitlines()[-25:])
    assert r.returncode == 0, tail
    assert " passed" in r.stdout and "failed" not in r.stdout.splitlines()[-1], tail
