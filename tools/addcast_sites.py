"""Which call sites launch the add/cast kernel in one benchmark step, with shapes (GPU)."""
import os, sys, collections, traceback, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import medical_sam2_amd.ops as ops
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
m = bench.build_model(dev)
imgs, pts, labels, bank_feats, sampled = bench.make_inputs(dev, 4, 0)
memory, memory_pos = bench.assemble_memory(m, bank_feats, sampled)
run = lambda: bench.step_2d(m, imgs, pts, labels, memory, memory_pos)
run(); torch.cuda.synchronize()
sites = collections.Counter()
for name in ("add_cast", "add_cast_into", "layernorm"):
    real = getattr(ops, name)
    def spy(*a, _real=real, _name=name, **k):
        st = [f for f in traceback.extract_stack()[:-1] if "medical-sam2_amd" in f.filename or "bench.py" in f.filename]
        f = st[-1]
        sites[(_name, f"{os.path.basename(f.filename)}:{f.lineno}", tuple(a[0].shape), str(a[0].dtype)[6:], str(k.get("out_dtype", a[3] if len(a) > 3 and _name == "add_cast" else ""))[-8:])] += 1
        return _real(*a, **k)
    setattr(ops, name, spy)
run()
for k, n in sorted(sites.items(), key=lambda kv: (kv[0][0], -kv[1])):
    print(n, k)
