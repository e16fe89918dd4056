"""Which Python call sites of the FULL training iteration (train_step_2d with the image encoder) launch torch's own copy / fill /
elementwise kernels, weighted by element count: one eager step under torch.profiler with stacks."""
import os, sys, collections, copy, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import medical_sam2_amd.training as T
from torch.profiler import profile, ProfilerActivity
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
m = bench.build_model(dev)
imgs, pts, labels, bank_feats, sampled = bench.make_inputs(dev, 4, 0)
memory, memory_pos = bench.assemble_memory(m, bank_feats, sampled)
mt = copy.deepcopy(m)
target = (torch.randn(4, 4, 256, 256, generator=torch.Generator().manual_seed(3)) > 0.5).float().to(dev)
om, od, oe = T.DecoderAdam(mt.memory_attention, lr=1e-6), T.DecoderAdam(mt.sam_mask_decoder, lr=1e-4), T.DecoderAdam(mt.image_encoder, lr=1e-6)
run = lambda: T.train_step_2d(mt, om, od, imgs, pts, labels, memory, memory_pos, target, sync=False, opt_enc=oe)
run(); run(); torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
    run()
torch.cuda.synchronize()
sites = collections.Counter(); elems = collections.Counter()
names = ("aten::copy_", "aten::fill_", "aten::cat", "aten::add", "aten::mul", "aten::zero_", "aten::index", "aten::index_put_", "aten::_to_copy", "aten::div", "aten::neg", "aten::sub", "aten::clone", "aten::contiguous")
for ev in prof.events():
    if ev.name in names:
        st = [f for f in ev.stack if "medical-sam2_amd" in f or "medical_sam2_amd" in f]
        key = (ev.name, st[0][-100:] if st else "?")
        n = 1
        try:
            for d in (ev.input_shapes[0] or []): n *= d
        except Exception: pass
        sites[key] += 1; elems[key] += n
for key, e in elems.most_common(45):
    print(f"{e/1e6:9.1f} Melem {sites[key]:4d} calls {key[0]:16s} {key[1]}")
