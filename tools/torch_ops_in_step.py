"""Which Python call sites of the benchmark step (bench.step_2d, eager) launch torch's own kernels (fill / copy / cat / elementwise) and where
the library's add / cast kernel is called: every one is a launch of >= 4.5 us inside the replayed graph.  One eager step under a dispatch mode."""
import collections, os, sys, traceback
import torch
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
SKIP = ("view", "reshape", "expand", "permute", "transpose", "slice", "select", "unsqueeze", "squeeze", "detach", "alias", "as_strided", "t", "empty",
        "empty_like", "empty_strided", "_unsafe_view", "unbind", "split", "split_with_sizes", "flatten", "unflatten", "_reshape_alias", "sym_size",
        "sym_stride", "sym_numel", "is_contiguous", "stride", "size", "dim", "numel", "contiguous", "lift_fresh", "_local_scalar_dense", "item",
        "is_same_size", "sym_storage_offset", "narrow", "chunk", "view_as", "expand_as", "broadcast_to", "movedim", "swapaxes", "data_ptr", "resolve_conj",
        "resolve_neg", "is_pinned", "record_stream", "set_", "_to_copy_nop")
sites = collections.Counter()


class Sites(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__.split(".")[0]
        if name not in SKIP:
            st = [f for f in traceback.extract_stack() if "sam2_amd" in f.filename or f.filename.endswith("bench.py")]
            sites[(name, " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in st[-3:][::-1]) if st else "?")] += 1
        return func(*args, **(kwargs or {}))


with Sites():
    run()
torch.cuda.synchronize()
print("torch-native ops of one step (views excluded):", sum(sites.values()))
for (name, where), c in sorted(sites.items(), key=lambda kv: kv[0][1]):
    print(f"{c:3d} x {name:22s} {where}")
