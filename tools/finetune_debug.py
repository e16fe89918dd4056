"""Loss trajectory of the decoder fine-tuning step, eager vs hipGraph replay (debug aid)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import medical_sam2_amd.build_sam as bs, medical_sam2_amd.weights as wts, medical_sam2_amd.training as T, medical_sam2_amd.ops as ops
torch.set_grad_enabled(False)
def setup():
    m = bs.build_sam2("sam2_hiera_s", device="cpu", hydra_overrides_extra=["++model.image_size=1024"])
    m.load_state_dict(wts.init_weights("hiera_s", 0), strict=True)
    return m.sam_mask_decoder.cuda().eval()
B, E, C = 4, 64, 256
g = torch.Generator().manual_seed(0)
r = lambda *s: torch.randn(*s, generator=g).cuda()
src, pe, sparse = r(B * E * E, C), r(E * E, C), r(B, 2, C)
f0, f1 = r(B * 16 * E * E, 32).to(ops.OP16), r(B * 4 * E * E, 64).to(ops.OP16)
target = (r(B, 4, 4 * E, 4 * E) > 0.5).float()
dec = setup(); opt = T.DecoderAdam(dec, lr=1e-4)
print("eager:", [round(T.decoder_finetune_step(dec, opt, src, pe, sparse, f0, f1, B, E, E, target), 5) for _ in range(16)])
dec = setup(); opt = T.DecoderAdam(dec, lr=1e-4)
st = torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(st):
    l = T.decoder_finetune_step(dec, opt, src, pe, sparse, f0, f1, B, E, E, target, sync=False)
torch.cuda.current_stream().wait_stream(st); torch.cuda.synchronize()
print("warm:", l.item())
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    loss_t = T.decoder_finetune_step(dec, opt, src, pe, sparse, f0, f1, B, E, E, target, sync=False)
out = []
for _ in range(15):
    graph.replay(); torch.cuda.synchronize(); out.append(round(loss_t.item(), 5))
print("graph:", out)
bad = [n for n, p in dec.named_parameters() if not torch.isfinite(p).all()]
print("non-finite params:", bad[:10], len(bad))

# eager with the bias correction frozen at t = 2 (what a graph captured at the second step replays) vs the graph, step by step
decA = setup(); optA = T.DecoderAdam(decA, lr=1e-4)
la = []
for i in range(8):
    if i >= 2:
        optA.t = 1
    la.append(round(T.decoder_finetune_step(decA, optA, src, pe, sparse, f0, f1, B, E, E, target), 5))
print("eager, frozen t=2:", la)
decB = setup(); optB = T.DecoderAdam(decB, lr=1e-4)
T.decoder_finetune_step(decB, optB, src, pe, sparse, f0, f1, B, E, E, target)
torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    loss_t = T.decoder_finetune_step(decB, optB, src, pe, sparse, f0, f1, B, E, E, target, sync=False)
decC = setup(); optC = T.DecoderAdam(decC, lr=1e-4)
T.decoder_finetune_step(decC, optC, src, pe, sparse, f0, f1, B, E, E, target)
for i in range(4):
    graph.replay(); torch.cuda.synchronize()
    optC.t = 1
    lc = T.decoder_finetune_step(decC, optC, src, pe, sparse, f0, f1, B, E, E, target)
    worst = max(((pb - pc).abs().max().item(), n) for (n, pb), (_, pc) in zip(decB.named_parameters(), decC.named_parameters()))
    print(f"replay {i}: graph loss {loss_t.item():.5f} eager loss {lc:.5f}  worst param diff {worst}")
