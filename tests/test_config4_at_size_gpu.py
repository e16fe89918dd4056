"""BASELINE.json configs[4] AT SIZE on one GPU: sam2_hiera_b+, 1024 x 1024, bf16 operands
(libmsam2_hip_bf16.so), the per-GPU share of the data-parallel batch (b = 4 slices), the whole train_2d iteration -- image encoder
(24-block Hiera-B+ trunk, FPN neck), memory attention and mask decoder forward / backward / Adam, memory encoding.  Its 8-GPU gradient
all-reduce is covered by tests/test_dp_training_gpu.py (2 ranks) and tests/test_rccl_gpu.py (RCCL).

Checked: every loss / parameter finite, the three trained groups move, the loss goes down over four iterations, and the image-encoder
gradient -- 250 tensors through decoder -> memory attention -> FPN -> 24 blocks -- is spot-checked by a central finite difference of the
loss ALONG the gradient itself (predicted change 2 eps |g|^2; a wrong scale or a wrong direction shows up as a ratio away from 1).

The operand type is a property of the loaded library, so the fp16 parent process runs this file again in a child with MSAM2_LIB_PATH."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "medical-sam2_amd", "libmsam2_hip_bf16.so")
sys.path.insert(0, ROOT)


def _run_at_size():
    import bench
    import medical_sam2_amd.build_sam as bs
    import medical_sam2_amd.training as T
    import medical_sam2_amd.weights as wts
    dev = torch.device("cuda", 0)
    m = bs.build_sam2("sam2_hiera_b+", device="cpu", hydra_overrides_extra=["++model.image_size=1024"])
    m.load_state_dict(wts.init_weights("hiera_b+", 3), strict=True)      # seed 3: real masks at these random weights (chain_hiera_bplus fixture)
    m = m.to(dev).eval()
    B = 4
    imgs, pts, labels, bank, sampled = bench.make_inputs(dev, B, 0)
    memory, memory_pos = bench.assemble_memory(m, bank, sampled)
    g = torch.Generator().manual_seed(3)
    target = (torch.randn(B, 4, 256, 256, generator=g) > 0.5).float().to(dev)
    args = (imgs, pts, labels, memory, memory_pos, target)
    with torch.no_grad():
        # ---- gradients at the initial point (lr 0), then the finite difference along the encoder gradient
        zero = [T.DecoderAdam(mod, lr=0.0) for mod in (m.memory_attention, m.sam_mask_decoder, m.image_encoder)]
        got: dict = {}
        loss0, _ = T.train_step_2d(m, zero[0], zero[1], *args, opt_enc=zero[2], grads_out=got)
        g_enc = got["image_encoder"]
        assert len(g_enc) > 240 and all(torch.isfinite(v).all() for v in g_enc.values())
        assert all(torch.isfinite(v).all() for grp in ("decoder", "memory_attention") for v in got[grp].values())
        params = dict(m.image_encoder.named_parameters())
        g2 = sum(float(v.double().pow(2).sum()) for v in g_enc.values())
        eps = 0.02 * loss0 / (2.0 * g2)                   # predicted central difference: 2 eps |g|^2 = 2 % of the loss
        step = lambda sgn: [params[k].add_(g_enc[k].to(params[k].dtype), alpha=sgn * eps) for k in g_enc]

        def loss_only():
            # parameters moved through raw tensor ops bump their version counters, so the kernel-ready 16-bit copies are rebuilt
            return T.train_step_2d(m, zero[0], zero[1], *args, opt_enc=zero[2])[0]
        step(+1.0)
        lp = loss_only()
        step(-2.0)
        lm = loss_only()
        step(+1.0)
        ratio = (lp - lm) / (2.0 * eps * g2)
        print(f"configs[4] at size: loss {loss0:.5f}; finite difference along the encoder gradient: measured {lp - lm:.6f} vs predicted "
              f"{2 * eps * g2:.6f} (ratio {ratio:.3f}), |g_enc| {g2 ** 0.5:.4e}")
        assert 0.7 < ratio < 1.3, ratio
        # ---- four real iterations
        before = {k: v.detach().clone() for k, v in m.state_dict().items()}
        opts = [T.DecoderAdam(m.memory_attention, lr=1e-5), T.DecoderAdam(m.sam_mask_decoder, lr=1e-4), T.DecoderAdam(m.image_encoder, lr=1e-5)]
        losses = [T.train_step_2d(m, opts[0], opts[1], *args, opt_enc=opts[2])[0] for _ in range(4)]
        moved = {k.split(".")[0] for k, v in m.state_dict().items() if not torch.equal(v, before[k])}
    print("losses:", losses)
    assert moved == {"memory_attention", "sam_mask_decoder", "image_encoder"}, moved
    assert all(map(lambda x: x == x and abs(x) < 1e6, losses)) and min(losses[1:]) < losses[0], losses
    assert all(torch.isfinite(v).all() for v in m.state_dict().values())
    assert all(o.skipped_elements == 0 for o in opts)


def test_configs4_hiera_bplus_train_iteration_at_1024_b4_bf16():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import medical_sam2_amd.ops as ops
    if ops.OP16 == torch.bfloat16:
        _run_at_size()
        return
    assert os.path.exists(LIB), "libmsam2_hip_bf16.so is built by __graft_entry__.build()"
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-s", "-p", "no:cacheprovider"], cwd=ROOT,
                       env=dict(os.environ, MSAM2_LIB_PATH=LIB), capture_output=True, text=True, timeout=1200)
    tail = "\n".join((r.stdout + r.stderr).splitlines()[-25:])
    print(tail)
    assert r.returncode == 0 and " passed" in r.stdout, tail
