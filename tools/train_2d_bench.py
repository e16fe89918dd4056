"""Training-iteration throughput of the 2-D flow at the benchmark's configuration (BASELINE.json configs[1]: hiera_s, b = 4 x 1024^2,
4 sampled memories per slice): frozen image / prompt encoders, forward + backward + Adam of the memory attention and the mask decoder,
memory encoding of the new masks.  Eager and as a hipGraph replay."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import medical_sam2_amd.training as T
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
m = bench.build_model(dev)
imgs, pts, labels, bank_feats, sampled = bench.make_inputs(dev, 4, 0)
memory, memory_pos = bench.assemble_memory(m, bank_feats, sampled)
g = torch.Generator().manual_seed(3)
target = (torch.randn(4, 4, 256, 256, generator=g) > 0.5).float().to(dev)
om, od = T.DecoderAdam(m.memory_attention, lr=1e-6), T.DecoderAdam(m.sam_mask_decoder, lr=1e-4)
step = lambda sync=True: T.train_step_2d(m, om, od, imgs, pts, labels, memory, memory_pos, target, sync=sync)
l0, mem = step(); torch.cuda.synchronize()
assert mem.shape == (4, 64, 64, 64) and torch.isfinite(mem).all()
t0 = time.perf_counter()
for _ in range(5): l1, _ = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
print(f"2-D training iteration (frozen encoders), eager: {dt * 1e3:.2f} ms ({4 / dt:.1f} slices/s); loss {l0:.5f} -> {l1:.5f}")
st = torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(st):
    step(False)
torch.cuda.current_stream().wait_stream(st); torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    lt, _ = step(False)
graph.replay(); torch.cuda.synchronize(); la = lt.item()
t0 = time.perf_counter()
for _ in range(10): graph.replay()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 10
print(f"hipGraph replay: {dt * 1e3:.2f} ms ({4 / dt:.1f} slices/s); loss {la:.5f} -> {lt.item():.5f} after 10 more iterations")
