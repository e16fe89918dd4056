"""Timing of the memory-attention backward at the benchmark's shape (4 slices, 4096 tokens, 4 x 4096 memory tokens), eager launches."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import medical_sam2_amd.build_sam as bs, medical_sam2_amd.weights as wts, medical_sam2_amd.backward as B
torch.set_grad_enabled(False)
m = bs.build_sam2("sam2_hiera_s", device="cpu", hydra_overrides_extra=["++model.image_size=1024"])
m.load_state_dict(wts.init_weights("hiera_s", 0), strict=True)
mod = m.memory_attention.cuda().eval()
Bn, L, Nk = 4, 4096, 16384
g = torch.Generator().manual_seed(0)
curr, pos = torch.randn(L, Bn, 256, generator=g).cuda(), torch.randn(L, Bn, 256, generator=g).cuda()
mem, mpos = (torch.randn(Nk, Bn, 64, generator=g) * 0.5).cuda(), torch.randn(Nk, Bn, 64, generator=g).cuda()
dy = torch.randn(L, Bn, 256, generator=g).cuda()
for name, fn in (("forward", lambda: mod(curr=[curr], curr_pos=[pos], memory=mem, memory_pos=mpos, num_obj_ptr_tokens=0)),
                 ("forward+backward (recomputing)", lambda: B.memory_attention_backward(mod, curr, pos, mem, mpos, 0, dy))):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): fn()
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms")
