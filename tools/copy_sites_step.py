"""Which Python call sites of the INFERENCE step (bench.step_2d: 4 slices, configs[1]) launch torch's own copy / fill / elementwise kernels:
one eager step under a TorchDispatchMode (call counts and element counts per site)."""
import os, sys, collections, traceback, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from torch.utils._python_dispatch import TorchDispatchMode
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
m = bench.build_model(dev)
imgs, pts, labels, bank_feats, sampled = bench.make_inputs(dev, 4, 0)
memory, memory_pos = bench.assemble_memory(m, bank_feats, sampled)
run = lambda: bench.step_2d(m, imgs, pts, labels, memory, memory_pos)
run(); run(); torch.cuda.synchronize()
sites = collections.Counter(); elems = collections.Counter()
SKIP = ("empty", "empty_like", "view", "reshape", "permute", "transpose", "as_strided", "slice", "select", "unsqueeze", "squeeze", "expand", "t", "detach",
        "alias", "_unsafe_view", "empty_strided", "_reshape_alias", "split", "unbind", "new_empty", "stride", "size", "is_contiguous", "sym_size", "sym_stride")


class Sites(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__.split(".")[0]
        if name not in SKIP:
            st = [f for f in traceback.extract_stack() if "sam2_amd" in f.filename]
            key = (name, " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in st[-2:][::-1]) if st else "?")
            n = 1
            for a in args:
                if isinstance(a, torch.Tensor):
                    n = a.numel(); break
                if isinstance(a, (list, tuple)) and a and isinstance(a[0], torch.Tensor):
                    n = sum(t.numel() for t in a); break
            sites[key] += 1; elems[key] += n
        return func(*args, **(kwargs or {}))


with Sites():
    run()
torch.cuda.synchronize()
print("torch kernels per step:", sum(sites.values()))
for key, e in elems.most_common(30):
    print(f"{sites[key]:4d} calls {e/1e6:9.2f} Melem {key[0]:14s} {key[1]}")
