"""BASELINE.json configs[4] AT SIZE on one GPU: sam2_hiera_b+, 1024 x 1024, bf16 operands
(libmsam2_hip_bf16.so), the per-GPU share of the data-parallel batch (b = 4 slices), the whole train_2d iteration -- image encoder
(24-block Hiera-B+ trunk, FPN neck), memory attention and mask decoder forward / backward / Adam, memory encoding.  Its 8-GPU gradient
all-reduce is covered by tests/test_dp_training_gpu.py (2 ranks) and tests/test_rccl_gpu.py (RCCL).

Checked (VERDICT r3 item 1a -- every clause is something the arithmetic guarantees, none depends on where four Adam steps at some
learning rate happen to land on a random-weight model):
  * every loss / gradient / parameter finite, no gradient element skipped;
  * per trained group (mask decoder, memory attention, image encoder): a central finite difference of the loss ALONG the group's own
    gradient (predicted change 2 eps |g|^2 = 2 % of the loss; a wrong scale or direction shows up as a ratio away from 1), and a plain
    gradient step p <- p - eta g of the same size must LOWER the loss by about the predicted 1 % (bar 0.5-1.5 %);
  * the optimiser: `DecoderAdam` steps on all three groups equal `torch.optim.Adam` run on the same parameters with the step's own
    gradients (two consecutive iterations, so the moment estimates and the bias correction are exercised), and exactly the three
    groups move.

The operand type is a property of the loaded library, so the fp16 parent process runs this file again in a child with MSAM2_LIB_PATH."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "medical-sam2_amd", "libmsam2_hip_bf16.so")
# central finite difference along a group's gradient / first-order prediction.  Measured (bf16, round 4): decoder 0.990, memory attention
# 0.848 and 0.968 on two builds whose forwards differ in the last bits (the decoder's input gradient is ill-conditioned at these random
# weights, DESIGN 7.2: it moves by 8.5 % per 0.05 % of its input), image encoder 0.914-0.977.  A wrong sign, a missing loss scale (a
# power of two) or a dropped term is far outside (0.75, 1.15).
FD_LO, FD_HI = 0.75, 1.15
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


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
    groups = {"decoder": m.sam_mask_decoder, "memory_attention": m.memory_attention, "image_encoder": m.image_encoder}
    with torch.no_grad():
        # ---- gradients at the initial point (lr 0)
        zero = [T.DecoderAdam(mod, lr=0.0) for mod in (m.memory_attention, m.sam_mask_decoder, m.image_encoder)]
        got: dict = {}
        loss0, _ = T.train_step_2d(m, zero[0], zero[1], *args, opt_enc=zero[2], grads_out=got)
        assert len(got["image_encoder"]) > 240
        assert all(torch.isfinite(v).all() for grp in groups for v in got[grp].values())

        def loss_only():
            # parameters moved through raw tensor ops bump their version counters, so the kernel-ready 16-bit copies are rebuilt
            return T.train_step_2d(m, zero[0], zero[1], *args, opt_enc=zero[2])[0]
        # (the forward is bit-reproducible; the mean over the 1 M loss terms is summed with fp32 atomics: a few ulps from run to run)
        assert abs(loss_only() - loss0) <= 2e-5 * abs(loss0), "the forward + loss is run-to-run reproducible at fixed parameters"
        for grp, mod in groups.items():
            g = got[grp]
            params = dict(mod.named_parameters())
            g2 = sum(float(v.double().pow(2).sum()) for v in g.values())
            eps = 0.02 * loss0 / (2.0 * g2)               # predicted central difference: 2 eps |g|^2 = 2 % of the loss
            step = lambda sgn: [params[k].add_(g[k].to(params[k].dtype), alpha=sgn * eps) for k in g]
            step(+1.0)
            lp = loss_only()
            step(-2.0)
            lm = loss_only()                              # = one plain gradient step of size eta = eps from the initial point
            step(+1.0)
            ratio = (lp - lm) / (2.0 * eps * g2)
            drop = (loss0 - lm) / loss0
            print(f"configs[4] at size, {grp}: loss {loss0:.5f}; finite difference along the gradient: measured {lp - lm:.6f} vs predicted "
                  f"{2 * eps * g2:.6f} (ratio {ratio:.3f}); plain gradient step lowers the loss by {100 * drop:.3f} % (predicted 1 %), "
                  f"|g| {g2 ** 0.5:.4e}")
            assert FD_LO < ratio < FD_HI, (grp, ratio)
            assert 0.004 < drop < 0.016, (grp, drop)       # predicted 1 %; measured 0.94-1.05 %
        # ---- two real iterations: DecoderAdam against torch.optim.Adam on the step's own gradients
        lrs = {"memory_attention": 1e-5, "decoder": 1e-4, "image_encoder": 1e-5}
        opts = {grp: T.DecoderAdam(groups[grp], lr=lrs[grp]) for grp in groups}
        twins = {grp: {k: torch.nn.Parameter(p.detach().clone()) for k, p in groups[grp].named_parameters()} for grp in groups}
        t_opts = {grp: torch.optim.Adam(list(twins[grp].values()), lr=lrs[grp], betas=(0.9, 0.999), eps=1e-8) for grp in groups}
        before = {k: v.detach().clone() for k, v in m.state_dict().items()}
        losses = []
        for it in range(2):
            got = {}
            losses.append(T.train_step_2d(m, opts["memory_attention"], opts["decoder"], *args, opt_enc=opts["image_encoder"], grads_out=got)[0])
            for grp in groups:
                for k, p in twins[grp].items():
                    p.grad = got[grp][k].to(p.dtype).clone() if k in got[grp] else None
                t_opts[grp].step()
                live = dict(groups[grp].named_parameters())
                worst = max(float(((live[k].detach() - p.detach()).abs() / p.detach().abs().clamp_min(1.0)).max()) for k, p in twins[grp].items())
                print(f"iteration {it}, {grp}: max |DecoderAdam - torch.optim.Adam| per element, in units of max(1, |p|): {worst:.3e} (lr {lrs[grp]:g})")
                # |update| <= lr per element and step.  The two optimisers agree to the fp32 rounding of the updated parameter (one or two
                # ulps of p: 1.2e-7 for |p| in [1, 2)) plus 1 % of lr for the corners where v-hat is of the order of eps^2
                assert worst <= 0.01 * lrs[grp] + 2.5e-7, (it, grp, worst)
        moved = {k.split(".")[0] for k, v in m.state_dict().items() if not torch.equal(v, before[k])}
    print("losses:", [loss0] + losses)
    assert moved == {"memory_attention", "sam_mask_decoder", "image_encoder"}, moved
    assert all(map(lambda x: x == x and abs(x) < 1e6, losses)), losses
    assert all(torch.isfinite(v).all() for v in m.state_dict().values())
    assert all(o.skipped_elements == 0 for o in opts.values())


def _run_encoder_gradients_at_size():
    """configs[4]'s model AT SIZE against the oracle ("the at-size configs[4] test has no oracle comparison at all"):
    hiera_b+, one 1024 x 1024 slice -- the three FPN outputs and the gradient of a random linear functional of them with respect to all
    ~250 image-encoder parameters and the two folded high-resolution convs, against torch.autograd through the fp32 oracle
    (oracle.forward_image).  Same statement as tests/test_backward_encoder_gpu.py::test_image_encoder_backward_vs_autograd, which runs at
    256 x 256: here the windows are the 1024^2 ones (64 x 64 tokens in stage 3, 14 x 14 windows padded to 70) and the global blocks see
    4096 tokens."""
    import medical_sam2_amd.backward_encoder as be
    import medical_sam2_amd.build_sam as bs
    import medical_sam2_amd.synthetic as syn
    import medical_sam2_amd.weights as wts
    from oracle import sam2_oracle as O
    from helpers import btol
    dev = torch.device("cuda", 0)
    S = 1024
    W = wts.init_weights("hiera_b+", 3)
    m = bs.build_sam2("sam2_hiera_b+", device="cpu", hydra_overrides_extra=[f"++model.image_size={S}"])
    m.load_state_dict(W, strict=True)
    m = m.to(dev).eval()
    cfg = O.model_config("hiera_b+", S)
    img, _, _ = syn.image_batch([10], S)
    train = lambda k: k.startswith("image_encoder.") or k.startswith("sam_mask_decoder.conv_s")
    rel = lambda a, b: float((a.detach().double().cpu() - b.detach().double().cpu()).norm() / b.detach().double().cpu().norm().clamp_min(1e-30))
    with torch.enable_grad():
        P = {k: (v.clone().requires_grad_(True) if train(k) else v) for k, v in W.items()}
        bo = O.forward_image(P, cfg, img)
        dys = [torch.randn(*f.shape, generator=torch.Generator().manual_seed(60 + l)) * 0.05 for l, f in enumerate(bo["backbone_fpn"])]
        sum((f * d).sum() for f, d in zip(bo["backbone_fpn"], dys)).backward()
    with torch.no_grad():
        out, st = be.image_encoder_forward_saved(m, img.to(dev))
        fe = [rel(out["backbone_fpn"][l], bo["backbone_fpn"][l]) for l in range(3)]
        d_fpn = [d.permute(0, 2, 3, 1).reshape(-1, d.shape[1]).contiguous().to(dev) for d in dys]
        grads = be.image_encoder_backward(m, st, d_fpn)
    missing = {k for k in P if train(k) and P[k].grad is not None and P[k].grad.abs().sum() > 0} - set(grads)
    assert not missing, sorted(missing)[:10]
    num = sum((grads[k].double().cpu() - P[k].grad.double()).pow(2).sum().item() for k in grads)
    den = sum(P[k].grad.double().pow(2).sum().item() for k in grads)
    errs = sorted(((rel(grads[k], P[k].grad), k) for k in grads), reverse=True)
    print(f"configs[4] at size, one slice vs oracle autograd: FPN features {fe}; {len(grads)} parameter gradients: overall relative L2 "
          f"{(num / den) ** 0.5:.4f}, worst {errs[:4]}")
    assert all(e < btol(3e-3) for e in fe), fe
    assert len(grads) > 240 and (num / den) ** 0.5 < btol(2e-2) and errs[0][0] < btol(8e-2, 2.5), errs[:6]


def _in_bf16_child(fn, test_name):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import medical_sam2_amd.ops as ops
    if ops.OP16 == torch.bfloat16:
        fn()
        return
    assert os.path.exists(LIB), "libmsam2_hip_bf16.so is built by __graft_entry__.build()"
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-s", "-k", test_name, "-p", "no:cacheprovider"], cwd=ROOT,
                       env=dict(os.environ, MSAM2_LIB_PATH=LIB), capture_output=True, text=True, timeout=1200)
    keep = [l for l in (r.stdout + r.stderr).splitlines() if "configs[4] at size" in l or "iteration " in l or "losses:" in l]
    tail = "\n".join(keep + (r.stdout + r.stderr).splitlines()[-25:])
    print(tail)
    assert r.returncode == 0 and " passed" in r.stdout, tail


def test_configs4_hiera_bplus_train_iteration_at_1024_b4_bf16():
    _in_bf16_child(_run_at_size, "test_configs4_hiera_bplus_train_iteration_at_1024_b4_bf16")


def test_configs4_hiera_bplus_encoder_gradients_at_1024_vs_oracle_autograd_bf16():
    _in_bf16_child(_run_encoder_gradients_at_size, "test_configs4_hiera_bplus_encoder_gradients_at_1024_vs_oracle_autograd_bf16")
