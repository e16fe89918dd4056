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
import traceback
from torch.utils._python_dispatch import TorchDispatchMode
sites = collections.Counter(); elems = collections.Counter()
WATCH = ("copy_", "fill_", "cat", "add", "mul", "zero_", "index", "index_put_", "_to_copy", "div", "neg", "sub", "clone", "abs", "sum", "amax", "aminmax", "max", "zeros", "maximum")


class Sites(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__.split(".")[0]
        if name in WATCH:
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
for key, c in sites.most_common(40):
    print(f"{c:4d} calls {elems[key]/1e6:9.1f} Melem {key[0]:12s} {key[1]}")

# the library's own cast / add kernel (msam2_add_cast) by call site
import medical_sam2_amd.ops as _ops
_orig = _ops.add_cast
ac_sites = collections.Counter(); ac_bytes = collections.Counter()


def _spy(a, b=None, alpha=1.0, out_dtype=None, out=None, *args, **kw):
    st = [f for f in traceback.extract_stack()[:-1] if "sam2_amd" in f.filename]
    key = " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in st[-3:][::-1])
    r = _orig(a, b, alpha, out_dtype, out, *args, **kw) if out is not None or out_dtype is not None else _orig(a, b, alpha)
    ac_sites[key] += 1
    ac_bytes[key] += a.numel() * a.element_size() + (b.numel() * b.element_size() if b is not None else 0) + r.numel() * r.element_size()
    return r


_ops.add_cast = _spy
import medical_sam2_amd.backward as _B, medical_sam2_amd.backward_encoder as _BE, medical_sam2_amd.modeling.common as _C
for mod in (_B, _BE, _C):
    if hasattr(mod, "ops"):
        mod.ops.add_cast = _spy
run()
torch.cuda.synchronize()
print("add_cast calls per iteration:", sum(ac_sites.values()), " MB moved:", round(sum(ac_bytes.values()) / 1e6))
for key, by in ac_bytes.most_common(16):
    print(f"{ac_sites[key]:4d} calls {by/1e6:8.1f} MB  {key}")
