"""Throughput of the decoder fine-tuning step (forward + BCE + mask-decoder backward + Adam) at the benchmark's shape: 4 slices at 1024^2
(64x64 image embedding, 256x256 low-res masks), one click per slice.  Eager launches."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import medical_sam2_amd.build_sam as bs, medical_sam2_amd.weights as wts, medical_sam2_amd.training as T, medical_sam2_amd.ops as ops
torch.set_grad_enabled(False)
m = bs.build_sam2("sam2_hiera_s", device="cpu", hydra_overrides_extra=["++model.image_size=1024"])
m.load_state_dict(wts.init_weights("hiera_s", 0), strict=True)
dec = m.sam_mask_decoder.cuda().eval()
B, E, C = 4, 64, 256
g = torch.Generator().manual_seed(0)
r = lambda *s: torch.randn(*s, generator=g).cuda()
src, pe, sparse = r(B * E * E, C), r(E * E, C), r(B, 2, C)
f0, f1 = r(B * 16 * E * E, 32).to(ops.OP16), r(B * 4 * E * E, 64).to(ops.OP16)
target = (r(B, 4, 4 * E, 4 * E) > 0.5).float()
opt = T.DecoderAdam(dec, lr=1e-4)
fwd = lambda: dec.predict_masks_tokens(src, pe, sparse, f0, f1, B, E, E)
step = lambda: T.decoder_finetune_step(dec, opt, src, pe, sparse, f0, f1, B, E, E, target)
for name, fn, n in (("decoder forward", fwd, 20), ("fine-tuning step (fwd + loss + bwd + Adam)", step, 5)):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"{name}: {dt * 1e3:.2f} ms  ({B / dt:.1f} slices/s)")

# the same step captured in a hipGraph (no host work per replay)
st = torch.cuda.Stream()
st.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(st):
    T.decoder_finetune_step(dec, opt, src, pe, sparse, f0, f1, B, E, E, target, sync=False)
torch.cuda.current_stream().wait_stream(st)
torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    loss_t = T.decoder_finetune_step(dec, opt, src, pe, sparse, f0, f1, B, E, E, target, sync=False)
graph.replay(); torch.cuda.synchronize()
l0 = loss_t.item()
t0 = time.perf_counter()
for _ in range(10): graph.replay()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 10
print(f"fine-tuning step, hipGraph replay: {dt * 1e3:.2f} ms  ({B / dt:.1f} slices/s); loss {l0:.5f} -> {loss_t.item():.5f} after 10 more steps")
