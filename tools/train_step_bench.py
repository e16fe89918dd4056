"""Throughput of the memory-conditioned fine-tuning step (memory attention + mask decoder: forward, BCE, backward of both, Adam on both
parameter groups) at the benchmark's shape: 4 slices at 1024^2 (64x64 embedding), 4 x 4096 memory tokens per slice.  Eager and as a hipGraph."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import medical_sam2_amd.build_sam as bs, medical_sam2_amd.weights as wts, medical_sam2_amd.training as T, medical_sam2_amd.ops as ops
torch.set_grad_enabled(False)
m = bs.build_sam2("sam2_hiera_s", device="cpu", hydra_overrides_extra=["++model.image_size=1024"])
m.load_state_dict(wts.init_weights("hiera_s", 0), strict=True)
mod, dec = m.memory_attention.cuda().eval(), m.sam_mask_decoder.cuda().eval()
B, E, C, Nk = 4, 64, 256, 16384
L = E * E
g = torch.Generator().manual_seed(0)
r = lambda *s: torch.randn(*s, generator=g).cuda()
curr, pos = r(L, B, C), r(L, B, C)
mem, mpos = r(Nk, B, 64) * 0.5, r(Nk, B, 64)
pe, sparse, dense = r(L, C), r(B, 2, C), r(1, C) * 0.3
f0, f1 = r(B * 16 * L, 32).to(ops.OP16), r(B * 4 * L, 64).to(ops.OP16)
target = (r(B, 4, 4 * E, 4 * E) > 0.5).float()
om, od = T.DecoderAdam(mod, lr=1e-6), T.DecoderAdam(dec, lr=1e-4)
args = (curr, pos, mem, mpos, 0, pe, sparse, f0, f1, B, E, E, target)
step = lambda sync=True: T.memory_decoder_finetune_step(mod, dec, om, od, *args, dense_tokens=dense, sync=sync)
l0 = step(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): l1 = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
print(f"memory attention + decoder fine-tuning step, eager: {dt * 1e3:.2f} ms ({B / dt:.1f} slices/s); loss {l0:.5f} -> {l1:.5f}")
st = torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(st):
    step(False)
torch.cuda.current_stream().wait_stream(st); torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    lt = step(False)
graph.replay(); torch.cuda.synchronize(); la = lt.item()
t0 = time.perf_counter()
for _ in range(10): graph.replay()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 10
print(f"hipGraph replay: {dt * 1e3:.2f} ms ({B / dt:.1f} slices/s); loss {la:.5f} -> {lt.item():.5f} after 10 more steps")
