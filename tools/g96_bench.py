"""Hiera global attention (B=4, H=4, Lq=Lk=4096, D=96; the layout the encoder passes: q/k/v column slices of one [B, L, 3*384] qkv
buffer): new 64-queries-per-wave kernel against attn_glds_kernel<96,128,4,3> (MSAM2_G96_V1=1 selects the old one per process).
usage: g96_bench.py [reps]"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run():
    import torch
    import medical_sam2_amd.ops as ops
    B, H, L, D = 4, 4, 4096, 96
    g = torch.Generator().manual_seed(0)
    qkv = (torch.randn(B, L, 3, H, D, generator=g) * 1.0).to(ops.OP16).cuda()
    q, k, v = (qkv[:, :, i].permute(0, 2, 1, 3) for i in range(3))
    out = torch.empty(B, L, H, D, dtype=ops.OP16, device="cuda").permute(0, 2, 1, 3)
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    for _ in range(3):
        o = ops.attention(q, k, v, out=out)
    ref = torch.nn.functional.scaled_dot_product_attention(q.float(), k.float(), v.float())
    err = (o.float() - ref).abs().max().item()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            ops.attention(q, k, v, out=out)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps * 1e3)
    fl = 4.0 * B * H * L * L * D
    t = sorted(ts)[len(ts) // 2]
    print(f"{os.environ.get('MSAM2_G96_V1', '0')}: {t:.1f} us (min {min(ts):.1f})  {fl / t * 1e-6:.0f} TFLOP/s = {fl / t * 1e-6 / 2500:.3f} of peak; max err vs fp32 {err:.2e}", flush=True)


if len(sys.argv) > 1 and sys.argv[1] == "child":
    run()
else:
    for rnd in range(2):
        for v1 in ("1", "0"):
            env = dict(os.environ)
            if v1 == "1":
                env["MSAM2_G96_V1"] = "1"
            else:
                env.pop("MSAM2_G96_V1", None)
            subprocess.run([sys.executable, os.path.abspath(__file__), "child"] + sys.argv[1:2], env=env, check=True)
