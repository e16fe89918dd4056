"""Which host thread count gives the best CPU-oracle time on this box (bench.py's cpu_baseline uses the result's rule)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
    if os.path.exists(p): print(p, open(p).read().strip())
from oracle import sam2_oracle as O
import medical_sam2_amd.synthetic as syn, medical_sam2_amd.weights as wts
torch.set_grad_enabled(False)
P = wts.init_weights("hiera_s", 0); cfg = O.model_config("hiera_s", 1024)
img, pts, labels = syn.image_batch([0], 1024)
for n in [int(a) for a in sys.argv[1:]] or [16, 32, 64, 128]:
    torch.set_num_threads(n)
    O.forward_image(P, cfg, img)
    t0 = time.perf_counter(); O.forward_image(P, cfg, img); dt = time.perf_counter() - t0
    print(f"threads {n}: forward_image {dt:.2f} s", flush=True)
