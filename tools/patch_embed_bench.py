"""PatchEmbed at the benchmark's shape (4 x 3 x 1024^2 fp32 -> 262144 x 96 fp32 tokens + position table): the one-kernel form
(msam2_patch_embed7x7s4) against im2col + GEMM.  usage: patch_embed_bench.py [B S E]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import medical_sam2_amd.ops as ops
B, S, E = [int(x) for x in sys.argv[1:4]] if len(sys.argv) > 3 else (4, 1024, 96)
g = torch.Generator().manual_seed(0)
img = torch.randn(B, 3, S, S, generator=g).cuda()
w = torch.randn(E, 3, 7, 7, generator=g) * 0.1
bias = torch.randn(E, generator=g).cuda()
pos = torch.randn((S // 4) ** 2, E, generator=g).cuda()
wp = torch.zeros((E + 31) // 32 * 32, 22, 8)
wp[:E, :21, 1:] = w.reshape(E, 21, 7)
wp = wp.reshape(-1, 176).to(ops.OP16).cuda().contiguous()
w160 = torch.zeros(E, 160)
w160[:, :147] = w.reshape(E, 147)
w160 = w160.to(ops.OP16).cuda()


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps * 1e3)
    return sorted(ts)[2]


a = ops.patch_embed(img, wp, bias, pos)
b = ops.gemm(ops.im2col_patch(img), w160, bias, residual=pos, res_mod=pos.shape[0], out_dtype=torch.float32)
print("max |one kernel - two launches|", float((a - b).abs().max()))
t1 = timed(lambda: ops.patch_embed(img, wp, bias, pos))
t2 = timed(lambda: ops.gemm(ops.im2col_patch(img), w160, bias, residual=pos, res_mod=pos.shape[0], out_dtype=torch.float32))
mb = (img.numel() * 4 + a.numel() * 4) / 1e6
print(f"one kernel {t1:.1f} us ({mb / t1 * 1e-6 * 1e6 / 1e6:.2f} TB/s on {mb:.0f} MB of image + tokens); im2col + GEMM {t2:.1f} us")
t3 = timed(lambda: ops.patch_embed(img, wp, bias, None))
print(f"one kernel without the position table {t3:.1f} us")
