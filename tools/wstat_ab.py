"""A/B of the W-stationary GEMM kernels on the step's short-reduction 16-bit projections: gemm_wstat256_kernel (256 rows per step, MSAM2_GEMM_WSTAT256=1)
against gemm_wstat_kernel (the default) and the tiled kernels (MSAM2_GEMM_WSTAT=0); one child process per variant (the switches
are read once per process), alternating, HIP-event timing of 30 back-to-back launches, median of 5.
usage: wstat_ab.py [rounds]"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SHAPES = [(16384, 1152, 384, 0, "qkv"), (16384, 1536, 384, 1, "fc1+GELU"), (16384, 2048, 256, 2, "linear1+ReLU"), (16384, 768, 256, 0, "k|v 256"),
          (16384, 1024, 256, 1, "1024x256+GELU"), (32768, 1152, 384, 0, "qkv B=8"), (8192, 1152, 384, 0, "qkv B=2")]


def run():
    import torch
    import medical_sam2_amd.ops as ops
    g = torch.Generator().manual_seed(0)
    for M, N, K, act, name in SHAPES:
        a = torch.randn(M, K, generator=g).to(ops.OP16).cuda()
        w = (torch.randn(N, K, generator=g) * 0.05).to(ops.OP16).cuda()
        b = torch.randn(N, generator=g).cuda()
        out = torch.empty(M, N, dtype=ops.OP16, device="cuda")
        for _ in range(5):
            ops.gemm(a, w, b, act=act, out=out)
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30):
                ops.gemm(a, w, b, act=act, out=out)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 30 * 1e3)
        t = sorted(ts)[2]
        print(f"  {name:16s} {M}x{N}x{K}: {t:6.1f} us  {2.0 * M * N * K / t * 1e-6:6.0f} TFLOP/s  checksum {float(out.float().sum()):.4e}", flush=True)


if len(sys.argv) > 1 and sys.argv[1] == "child":
    run()
else:
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    for rnd in range(rounds):
        for label, env in (("wstat256", {"MSAM2_GEMM_WSTAT256": "1"}), ("wstat128", {}), ("tiled", {"MSAM2_GEMM_WSTAT": "0"})):
            print(f"== {label} (round {rnd})", flush=True)
            subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, **env), check=True)
