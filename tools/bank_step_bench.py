"""Memory-bank fine-tuning step (one level of BPTT: decoder + memory attention + memory encoder + object-pointer projection) at 4 slices
of 1024^2 with one bank entry per slice.  Eager and as a hipGraph replay."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import medical_sam2_amd.build_sam as bs, medical_sam2_amd.weights as wts, medical_sam2_amd.training as T, medical_sam2_amd.ops as ops
torch.set_grad_enabled(False)
m = bs.build_sam2("sam2_hiera_s", device="cpu", hydra_overrides_extra=["++model.image_size=1024"])
m.load_state_dict(wts.init_weights("hiera_s", 0), strict=True)
m = m.cuda().eval()
B, E, C = 4, 64, 256
L = E * E
g = torch.Generator().manual_seed(0)
r = lambda *s: torch.randn(*s, generator=g).cuda()
curr, pos = r(L, B, C), r(L, B, C)
prev_pix, prev_mask, mpos = r(B * L, C), r(B, 1, 16 * E, 16 * E) * 4, r(L, B, 64)
pe, sparse, dense, tok = r(L, C), r(B, 2, C), r(1, C) * 0.3, r(B, C)
f0, f1 = r(B * 16 * L, 32).to(ops.OP16), r(B * 4 * L, 64).to(ops.OP16)
target = (r(B, 4, 4 * E, 4 * E) > 0.5).float()
opts = {"decoder": T.DecoderAdam(m.sam_mask_decoder, lr=1e-4), "memory_attention": T.DecoderAdam(m.memory_attention, lr=1e-6),
        "memory_encoder": T.DecoderAdam(m.memory_encoder, lr=1e-6), "obj_ptr_proj": T.DecoderAdam(m.obj_ptr_proj, lr=1e-6)}
args = (curr, pos, prev_pix, prev_mask, False, mpos, pe, sparse, f0, f1, B, E, E, target)
step = lambda sync=True: T.memory_bank_finetune_step(m, opts, *args, dense_tokens=dense, prev_sam_token=tok, sync=sync)
l0 = step(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): l1 = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
print(f"memory-bank fine-tuning step, eager: {dt * 1e3:.2f} ms ({B / dt:.1f} slices/s); loss {l0:.5f} -> {l1:.5f}")
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
