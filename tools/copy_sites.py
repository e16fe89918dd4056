"""Which Python call sites of the benchmark step launch torch's own copy / fill / elementwise kernels (everything that is not a
libmsam2_hip.so launch): one eager step under torch.profiler with stacks."""
import os, sys, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from torch.profiler import profile, ProfilerActivity
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
m = bench.build_model(dev)
imgs, pts, labels, bank_feats, sampled = bench.make_inputs(dev, 4, 0)
memory, memory_pos = bench.assemble_memory(m, bank_feats, sampled)
run = lambda: bench.step_2d(m, imgs, pts, labels, memory, memory_pos)
run(); run(); torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
    run()
torch.cuda.synchronize()
sites = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::fill_", "aten::cat", "aten::add", "aten::mul", "aten::zero_", "aten::index", "aten::index_put_", "aten::_to_copy"):
        st = [f for f in ev.stack if "medical-sam2_amd" in f or "medical_sam2_amd" in f or "bench.py" in f]
        sites[(ev.name, st[0] if st else "?", str(ev.input_shapes)[:80])] += 1
for (name, site, shp), n in sites.most_common(60):
    print(f"{n:3d} {name:18s} {site[-90:]:90s} {shp}")
