"""LayerNorm backward at the trunk's shapes (fp32 x, fp32 upstream gradient, fp32 by-pass gradient), graph-replayed launches."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import medical_sam2_amd.backward as B
for rows, C in ((16384, 384), (16384, 256), (65536, 192), (262144, 96)):
    x = torch.randn(rows, C, device="cuda"); g = torch.ones(C, device="cuda"); dy = torch.randn(rows, C, device="cuda"); add = torch.randn(rows, C, device="cuda")
    for _ in range(3): B.layernorm_backward(x, g, dy, 1e-6, add=add)
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        gph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gph, stream=s):
            for _ in range(10): out = B.layernorm_backward(x, g, dy, 1e-6, add=add)
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); gph.replay(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10 * 1e3)
    t = sorted(ts)[2]
    print(f"rows {rows:6d} C {C:4d}: {t:6.1f} us per call (incl. the zero fill of dgamma / dbeta)  {rows * C * 16 / t * 1e-6:5.2f} TB/s", flush=True)
