"""The fp32-output + fp32-residual GEMMs of the step (fc2 / proj of Hiera stage 3, the memory attention's output projections and linear2,
stage-1/2 and stage-4 shapes): stand-alone times, 30 launches replayed as one hipGraph.  A/B between library builds through MSAM2_LIB_PATH."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import medical_sam2_amd.ops as ops
SHAPES = [(16384, 384, 1536), (16384, 384, 384), (16384, 256, 2048), (16384, 256, 256), (16384, 256, 64), (262144, 96, 160), (262144, 96, 96),
          (65536, 192, 192), (4096, 768, 3072), (4096, 768, 768), (16384, 256, 1024)]
if os.environ.get("SHAPES"):
    SHAPES = [tuple(int(v) for v in t.split("x")) for t in os.environ["SHAPES"].split(",")]
g = torch.Generator().manual_seed(0)
tot = 0.0
for M, N, K in SHAPES:
    a = torch.randn(M, K, generator=g).to(ops.OP16).cuda()
    w = (torch.randn(N, K, generator=g) * 0.05).to(ops.OP16).cuda()
    b = torch.randn(N, generator=g).cuda()
    res = torch.randn(M, N, generator=g).cuda()
    out = torch.empty(M, N, dtype=torch.float32, device="cuda")
    fn = lambda: ops.gemm(a, w, b, residual=res, out=out)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(30):
            fn()
    gr.replay()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        gr.replay()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 30 * 1e3)
    t = sorted(ts)[2]
    tot += t
    print(f"  {M}x{N}x{K}: {t:6.1f} us  {2.0 * M * N * K / t * 1e-6:6.0f} TFLOP/s", flush=True)
print(f"  sum {tot:.1f} us")
