"""Attention kernels timed with two builds of the library (child process per build; MSAM2_LIB_PATH selects it):
  global : Hiera global block  B=4 H=4 Lq=Lk=4096 D=96
  memory : memory cross-attention B=4 H=1 Lq=4096 Lk=16384, 256-wide keys / 64-wide rows, 4 splits (the dominant kernel)
  self   : memory-attention self-attention B=4 H=1 Lq=Lk=4096 D=256
usage: attn_ab.py libA.so [libB.so ...]     (env passed through, e.g. MSAM2_G96_V1=1)"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def timeit(fn, reps=30, rounds=5):
    import torch
    for _ in range(3):
        fn()
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps * 1e3)
    return sorted(ts)[len(ts) // 2], min(ts)


def run():
    import torch
    import medical_sam2_amd.ops as ops
    g = torch.Generator().manual_seed(0)
    r = lambda *s: torch.randn(*s, generator=g).to(ops.OP16).cuda()
    out = []
    B, H, L, D = 4, 4, 4096, 96
    qkv = r(B, L, 3, H, D)
    q, k, v = (qkv[:, :, i].permute(0, 2, 1, 3) for i in range(3))
    o = ops.attention(q, k, v)
    ref = torch.nn.functional.scaled_dot_product_attention(q.float(), k.float(), v.float())
    e1 = (o.float() - ref).abs().max().item()
    t, mn = timeit(lambda: ops.attention(q, k, v))
    out.append(f"global {t:6.1f} us (min {mn:6.1f}) {4.0 * B * H * L * L * D / t * 1e-6 / 2500:.3f} of peak, err {e1:.1e}")
    q, k, v = r(4, 1, 4096, 256), r(4, 1, 16384, 256), r(4, 1, 16384, 64)
    o = ops.attention_kv64(q, k, v, splits=4)
    ref = torch.nn.functional.scaled_dot_product_attention(q.float(), k.float(), v.float())
    e2 = (o.float() - ref).abs().max().item()
    t, mn = timeit(lambda: ops.attention_kv64(q, k, v, splits=4), reps=20)
    out.append(f"memory {t:6.1f} us (min {mn:6.1f}) {4.0 * 4 * 4096 * 16384 * 256 / t * 1e-6 / 2500:.3f} of peak (algorithmic), err {e2:.1e}")
    q, k, v = r(4, 1, 4096, 256), r(4, 1, 4096, 256), r(4, 1, 4096, 256)
    o = ops.attention(q, k, v)
    ref = torch.nn.functional.scaled_dot_product_attention(q.float(), k.float(), v.float())
    e3 = (o.float() - ref).abs().max().item()
    t, mn = timeit(lambda: ops.attention(q, k, v))
    out.append(f"self   {t:6.1f} us (min {mn:6.1f}) {4.0 * 4 * 4096 * 4096 * 256 / t * 1e-6 / 2500:.3f} of peak, err {e3:.1e}")
    print(os.path.basename(os.environ.get("MSAM2_LIB_PATH", "default")) + ":\n  " + "\n  ".join(out), flush=True)


if len(sys.argv) > 1 and sys.argv[1] == "child":
    run()
else:
    for rnd in range(2):
        for lib in sys.argv[1:]:
            env = dict(os.environ)
            env["MSAM2_LIB_PATH"] = os.path.abspath(lib)
            subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, check=True)
