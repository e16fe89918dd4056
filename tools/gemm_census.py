#!/usr/bin/env python3
"""Census of the GEMM calls of one benchmark step (configs[1]): every distinct (M, N, K, epilogue) with its call count, its
stand-alone time and the share of the step's GEMM time (GPU only).  Usage: python tools/gemm_census.py"""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import medical_sam2_amd.ops as ops  # noqa: E402
from tools.kernel_bench import timeit  # noqa: E402


def main():
    torch.set_grad_enabled(False)
    dev = torch.device("cuda", 0)
    m = bench.build_model(dev)
    imgs, pts, labels, bank, sampled = bench.make_inputs(dev, 4, 0)
    memory, memory_pos = bench.assemble_memory(m, bank, sampled)
    bench.step_2d(m, imgs, pts, labels, memory, memory_pos)
    calls = collections.Counter()
    real = ops.gemm

    def spy(a, w, bias=None, *, act=0, colscale=None, residual=None, res_mod=0, out_dtype=ops.OP16, out=None):
        od = out.dtype if out is not None else out_dtype
        calls[(a.shape[0], w.shape[0], a.shape[1], od == torch.float32, residual is not None, act, colscale is not None)] += 1
        return real(a, w, bias, act=act, colscale=colscale, residual=residual, res_mod=res_mod, out_dtype=out_dtype, out=out)

    ops.gemm = spy
    bench.step_2d(m, imgs, pts, labels, memory, memory_pos)
    ops.gemm = real
    g = torch.Generator().manual_seed(0)
    rows = []
    for (M, N, K, f32, res, act, cs), n in calls.items():
        a = torch.randn(M, K, generator=g).to(ops.OP16).to(dev)
        w = (torch.randn(N, K, generator=g) * 0.05).to(ops.OP16).to(dev)
        bias = torch.randn(N, generator=g).to(dev)
        out = torch.empty(M, N, dtype=torch.float32 if f32 else ops.OP16, device=dev)
        r = torch.randn(M, N, generator=g).to(dev) if res else None
        c = torch.randn(N, generator=g).to(dev) if cs else None
        t = timeit(lambda: ops.gemm(a, w, bias, act=act, colscale=c, residual=r, out=out), n=30)
        by = 2 * M * K + 2 * N * K + M * N * (4 if f32 else 2) + (M * N * 4 if res else 0)
        rows.append((n * t, n, M, N, K, f32, res, act, t, 2 * M * N * K / t / 1e12, by / t / 1e9))
    rows.sort(reverse=True)
    tot = sum(r[0] for r in rows)
    print(f"{'n':>3s} {'M':>7s} {'N':>5s} {'K':>5s} out res act {'us':>8s} {'TF/s':>7s} {'GB/s':>6s} {'n*us':>8s} {'share':>6s}")
    for nt, n, M, N, K, f32, res, act, t, tf, gb in rows:
        print(f"{n:3d} {M:7d} {N:5d} {K:5d} {'f32' if f32 else 'h16'} {int(res):3d} {act:3d} {t * 1e6:8.1f} {tf:7.1f} {gb:6.0f} {nt * 1e6:8.1f} "
              f"{100 * nt / tot:5.1f}%")
    print(f"total {tot * 1e3:.3f} ms in {sum(r[1] for r in rows)} calls, {sum(2 * r[1] * r[2] * r[3] * r[4] for r in rows) / 1e12:.2f} TFLOP")


if __name__ == "__main__":
    main()
