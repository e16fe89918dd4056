"""Probe: d(mask) of memory_encoder_backward(need_dmask=True) and of the bilinear up-sampling adjoint against oracle autograd."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import sam2_oracle as O
import medical_sam2_amd.weights as wts
import medical_sam2_amd.build_sam as bs
import medical_sam2_amd.backward as bwd
import medical_sam2_amd.ops as ops
from medical_sam2_amd._lib import check, lib
from medical_sam2_amd.ops import _p, _stream

rnd = lambda *shape, seed=0, scale=1.0: torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale
B, E = 2, 16
sd = wts.init_weights("hiera_t", 0)
cfg = O.model_config("hiera_t", 256)
P = {k: v.clone().float() for k, v in sd.items()}
pix, low, dyo = rnd(B, 256, E, E, seed=180), rnd(B, 1, 4 * E, 4 * E, seed=181, scale=4.0).requires_grad_(True), rnd(B, 64, E, E, seed=182)
with torch.enable_grad():
    mask = torch.nn.functional.interpolate(low, size=(16 * E, 16 * E), mode="bilinear", align_corners=False)
    mask.retain_grad()
    y, _ = O.memory_encoder(P, cfg, pix, torch.sigmoid(mask) * 20.0 - 10.0)
    y.backward(dyo)
ref_dmask, ref_dlow = mask.grad, low.grad
m = bs.build_sam2("sam2_hiera_t", device="cpu", hydra_overrides_extra=["++model.image_size=256"])
m.load_state_dict(sd, strict=True)
enc = m.memory_encoder.cuda().eval()
d = lambda t: t.detach().cuda()
tm = lambda t: d(t).permute(0, 2, 3, 1).reshape(-1, t.shape[1]).contiguous()
with torch.no_grad():
    _, g, dmask = bwd.memory_encoder_backward(enc, tm(pix), d(mask), 1, 20.0, -10.0, B, E, E, tm(dyo), need_dmask=True)
    dlow = torch.empty(B, 4 * E, 4 * E, dtype=torch.float32, device="cuda")
    check(lib().msam2_bilinear_upsample_bwd(_p(dmask.contiguous()), _p(dlow), B, 4 * E, 4 * E, 16 * E, 16 * E, _stream()))
    dlow_ref_in = torch.empty_like(dlow)
    check(lib().msam2_bilinear_upsample_bwd(_p(d(ref_dmask).contiguous()), _p(dlow_ref_in), B, 4 * E, 4 * E, 16 * E, 16 * E, _stream()))
rel = lambda a, b: float((a.cpu().double() - b.double()).norm() / b.double().norm())
print("d mask      rel err", rel(dmask, ref_dmask), "norms", float(dmask.norm()), float(ref_dmask.norm()))
print("d low (hip dmask -> hip adjoint) rel err", rel(dlow.view_as(ref_dlow), ref_dlow))
print("d low (ref dmask -> hip adjoint) rel err", rel(dlow_ref_in.view_as(ref_dlow), ref_dlow))
