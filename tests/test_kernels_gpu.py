"""Kernel-level parity (GPU): every C-ABI op against the CPU oracle's primitive on the same seeded inputs.

16-bit operands (fp16 by default) are rounded once on the host so both sides see identical inputs; accumulation is fp32 on the GPU and
fp32/fp64 in the oracle, so tolerances only cover accumulation order and the bf16 rounding of outputs / of the
softmax probabilities:  integer-valued GEMM data is compared bit-exactly (validates the MFMA fragment maps).
"""
import math

import numpy as np
import pytest
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
from helpers import btol, op16_is_fp16  # noqa: E402
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import sam2_oracle as O  # noqa: E402
from oracle import cc as cc_oracle  # noqa: E402


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import medical_sam2_amd.ops as ops_mod
    return ops_mod


DEV = "cuda"


def OP16():
    import medical_sam2_amd.ops as _o
    return _o.OP16


def bf(x):
    """round to the library's 16-bit MFMA operand dtype (fp16 by default, bf16 with -DMSAM2_OPERAND_BF16)"""
    import medical_sam2_amd.ops as _o
    return x.to(_o.OP16)


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def close(a, b, atol, rtol=0.0, what=""):
    a = a.detach().float().cpu()
    b = b.detach().float().cpu()
    err = (a - b).abs()
    lim = atol + rtol * b.abs()
    bad = (err > lim)
    assert not bad.any(), f"{what}: max err {err.max().item():.4g} (limit {atol}+{rtol}*|ref|), {bad.sum().item()} bad of {bad.numel()}"


def max_abs_t(a, b):
    return (a.float() - b.float()).abs().max().item()


# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (5, 288, 96), (300, 96, 384), (70, 32, 64), (257, 576, 160), (33, 64, 16),
                                   (1000, 2048, 256), (8, 4, 256), (4096, 768, 256), (300, 128, 64), (257, 640, 192), (4099, 1152, 384),
                                   (513, 576, 576), (256, 1536, 3072), (4, 256, 256), (32, 2048, 256), (7, 256, 2048), (5, 1, 256),
                                   (31, 33, 48), (32, 128, 24), (16384, 256, 2048), (2500, 192, 768), (700, 1536, 384)])
def test_gemm_exact_integers(ops, M, N, K):
    g = torch.Generator().manual_seed(M * 7 + N)
    a = torch.randint(-3, 4, (M, K), generator=g).float()
    w = torch.randint(-3, 4, (N, K), generator=g).float()
    # asymmetric structure so that a row/col swap or a k-permutation cannot pass
    a[:, 0] += torch.arange(M).float() % 5
    w[:, -1] += torch.arange(N).float() % 7
    ref = a.double() @ w.double().t()
    out = ops.gemm(bf(a).to(DEV), bf(w).to(DEV), out_dtype=torch.float32)
    assert torch.equal(out.cpu().double(), ref), (out.cpu().double() - ref).abs().max()


@pytest.mark.parametrize("M,N,K,act", [(8192, 1152, 384, 0), (8320, 256, 256, 0), (16384, 1536, 384, 2), (12800, 2048, 256, 2), (8192, 128, 384, 0),
                                       (9088, 768, 256, 0)])
def test_gemm_w_stationary_kernels_exact(ops, M, N, K, act):
    """The W-stationary persistent kernels (gemm_wstat256_kernel: 256 rows per step, round 4; gemm_wstat_kernel): 16-bit output, K = 256 /
    384, M >= 8192 -- the qkv / fc1 projections of Hiera stage 3 and the memory attention's linear1.  Sparse small-integer operands keep
    every output an integer below 256 (exact in fp16 AND bf16), so the comparison is bit-exact whatever the operand type: a wrong
    row / column / k mapping, a lost k-step, a unit paired with the wrong twin or a dropped tail unit cannot pass.  Shapes: even and odd
    numbers of 128-row units per workgroup (a left-over unit is paired with itself), one unit per workgroup, one panel, 16 panels."""
    g = torch.Generator().manual_seed(M + 3 * N + K)
    a = torch.zeros(M, K)
    w = torch.zeros(N, K)
    cols = torch.randint(0, K, (M, 6), generator=g)
    a.scatter_(1, cols, torch.randint(-3, 4, (M, 6), generator=g).float())
    colsw = torch.randint(0, K, (N, 6), generator=g)
    w.scatter_(1, colsw, torch.randint(-3, 4, (N, 6), generator=g).float())
    a[:, 0] = (torch.arange(M) % 3).float()                   # structure a row / column swap or a k-permutation cannot preserve
    w[:, 0] = (torch.arange(N) % 4).float() - 1
    a[:, K - 1] = ((torch.arange(M) // 128) % 5).float() - 2  # differs from one 128-row unit to the next
    w[:, K - 1] = 1.0
    bias = torch.randint(-4, 5, (N,), generator=g).float()
    ref = a.double() @ w.double().t() + bias.double()
    if act == 2:
        ref = ref.clamp_min(0)
    assert ref.abs().max() < 256
    out = ops.gemm(bf(a).to(DEV), bf(w).to(DEV), bias.to(DEV), act=act, out_dtype=OP16())
    assert out.dtype == OP16() and torch.equal(out.cpu().double(), ref), (out.cpu().double() - ref).abs().max()
    # strided output (a column block of a wider buffer), untouched outside
    buf = torch.full((M, N + 16), 5.0, dtype=OP16(), device=DEV)
    ops.gemm(bf(a).to(DEV), bf(w).to(DEV), bias.to(DEV), act=act, out=buf[:, 8:N + 8])
    assert torch.equal(buf[:, 8:N + 8].cpu().double(), ref) and (buf[:, :8] == 5).all() and (buf[:, N + 8:] == 5).all()


def test_gemm_w_stationary_256_row_kernel_in_a_child_process():
    """gemm_wstat256_kernel is opt-in (MSAM2_GEMM_WSTAT256=1, read once per process: as fast as the 128-row kernel at K = 256, slower at
    K = 384; kept for its diagnostic ladder): the exact cases above through it, in a child process."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import subprocess, sys
    env = dict(_os.environ, MSAM2_GEMM_WSTAT256="1")
    r = subprocess.run([sys.executable, "-m", "pytest", _os.path.abspath(__file__), "-q", "-m", "gpu", "-k", "test_gemm_w_stationary_kernels_exact",
                        "-p", "no:cacheprovider"], env=env, capture_output=True, text=True, timeout=900,
                       cwd=_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
    assert r.returncode == 0 and " passed" in r.stdout, (r.stdout + r.stderr)[-1500:]


@pytest.mark.parametrize("B,S,E", [(2, 256, 96), (1, 1024, 96), (2, 128, 112), (1, 512, 32), (3, 256, 128)])
def test_patch_embed_one_kernel(ops, B, S, E):
    """msam2_patch_embed7x7s4 (round 4): Conv2d(3, E, k7, s4, p3) + bias + position table in ONE kernel, the reduction re-ordered to
    (c, ky, kx padded to 8) so that MFMA fragments are contiguous pixels of the staged image rows.  Against F.conv2d on the 16-bit-rounded
    operands in float64 (exact products, fp32 accumulation: 1e-5) and against the two-launch path it replaces (im2col + GEMM).  Sizes:
    64 / 256 / 32 / 128 tokens per row (one or two 128-token segments, partial workgroups), E = 96 (hiera_t / s), 112 (hiera_b+: the
    fourth 32-column tile half used), 32, 128; borders (the 3-pixel padding) on every side."""
    g = torch.Generator().manual_seed(S + E)
    img = torch.randn(B, 3, S, S, generator=g)
    img[:, :, :5, :] *= 3.0                                    # structure at the borders: a padding or row / column offset error shows
    img[:, :, :, -5:] -= 2.0
    w = torch.randn(E, 3, 7, 7, generator=g) * 0.1
    bias = torch.randn(E, generator=g)
    So = S // 4
    pos = torch.randn(So * So, E, generator=g)
    assert ops.patch_embed_supported(S, E)
    wp = torch.zeros((E + 31) // 32 * 32, 22, 8)
    wp[:E, :21, 1:] = w.reshape(E, 21, 7)
    wp = bf(wp.reshape(-1, 176)).contiguous()
    ref = torch.nn.functional.conv2d(bf(img).double(), bf(w).double(), bias.double(), stride=4, padding=3).permute(0, 2, 3, 1).reshape(-1, E)
    out = ops.patch_embed(img.to(DEV), wp.to(DEV), bias.to(DEV), pos.to(DEV))
    want = ref + pos.double().repeat(B, 1)
    close(out, want.float(), 2e-5, 2e-5, "patch_embed + pos")
    out2 = ops.patch_embed(img.to(DEV), wp.to(DEV), bias.to(DEV))
    close(out2, ref.float(), 2e-5, 2e-5, "patch_embed")
    # the two-launch path (column order (c, ky, kx) padded to 160)
    w160 = torch.zeros(E, 160)
    w160[:, :147] = w.reshape(E, 147)
    old = ops.gemm(ops.im2col_patch(img.to(DEV)), bf(w160).to(DEV), bias.to(DEV), out_dtype=torch.float32)
    close(out2, old.cpu(), 2e-5, 2e-5, "patch_embed vs im2col + gemm")
    assert not ops.patch_embed_supported(64, 96) and not ops.patch_embed_supported(1024, 160)


def test_gemm_epilogues(ops):
    M, N, K = 200, 192, 96
    a, w = bf(rnd(M, K, seed=1)), bf(rnd(N, K, seed=2, scale=0.2))
    bias, cs = rnd(N, seed=3), rnd(N, seed=4)
    res = rnd(50, N, seed=5)
    lin = a.float() @ w.float().t() + bias
    for act, f in ((0, lambda x: x), (1, O.gelu), (2, torch.relu)):
        ref = f(lin) * cs + res.repeat(4, 1)
        out = ops.gemm(a.to(DEV), w.to(DEV), bias.to(DEV), act=act, colscale=cs.to(DEV), residual=res.to(DEV), res_mod=50,
                       out_dtype=torch.float32)
        close(out, ref, 2e-4, 1e-5, f"gemm act={act}")
    out = ops.gemm(a.to(DEV), w.to(DEV), bias.to(DEV), residual=bf(res.repeat(4, 1)).to(DEV), out_dtype=OP16())
    close(out, lin + bf(res.repeat(4, 1)).float(), 1e-3, 8e-3, "gemm bf16 out")
    # strided A (a column slice of a wider buffer) and strided output
    wide = bf(rnd(M, 3 * K, seed=6)).to(DEV)
    buf = torch.zeros(M, 2 * N, dtype=torch.float32, device=DEV)
    ops.gemm(wide[:, K:2 * K], w.to(DEV), out=buf[:, N:])
    close(buf[:, N:], wide[:, K:2 * K].float().cpu() @ w.float().t(), 2e-4, 1e-5, "gemm strided")
    assert buf[:, :N].abs().sum().item() == 0


def test_gelu_epilogue_accuracy(ops):
    """The GELU of the epilogues (nn.GELU, exact erf form; hieradet.py:158, sam2_utils.py:120) against erf in float64: the
    pre-activations are the bias row (A = 0), so every column sees one x of a dense grid over [-12, 12] incl. the clamp points +-4.5.
    Stated bound of csrc/common.h: 5e-5 absolute (the polynomial form) -- fp32 output; 16-bit output: + one rounding."""
    N, M, K = 4096, 128, 64
    x = torch.linspace(-12.0, 12.0, N, dtype=torch.float64)
    x[N // 2] = 0.0
    x[:4] = torch.tensor([-4.5, 4.5, -4.4999, 4.4999], dtype=torch.float64)
    x32 = x.float()
    ref = 0.5 * x32.double() * (1.0 + torch.erf(x32.double() / math.sqrt(2.0)))
    a = torch.zeros(M, K, dtype=OP16(), device=DEV)
    w = bf(rnd(N, K, seed=1)).to(DEV)
    out = ops.gemm(a, w, x32.to(DEV), act=1, out_dtype=torch.float32).double().cpu()
    err = (out - ref[None]).abs().max().item()
    assert err <= 6e-5, err
    assert (out[:, x32 < -4.6].abs() <= 6e-5).all() and (out[:, x32 > 4.6] - x32[x32 > 4.6].double()).abs().max() <= 6e-5
    out16 = ops.gemm(a, w, x32.to(DEV), act=1, out_dtype=OP16()).double().cpu()
    ulp = 2.0 ** (-11 if op16_is_fp16() else -8)
    assert ((out16 - ref[None]).abs() <= 6e-5 + ulp * ref.abs()[None] + 1e-7).all()


@pytest.mark.parametrize("M,N,K", [(256, 256, 16384), (128, 64, 65536), (300, 200, 4104), (37, 256, 8192)])
def test_gemm_split_k(ops, M, N, K):
    """Weight-gradient shapes (small output, K = tokens): the split-K path (fp32 atomics into the zeroed output).  Integer operands
    make every partial sum exact, so the result must not depend on the order of the atomics; strided output untouched outside."""
    g = torch.Generator().manual_seed(K + M)
    a = torch.randint(-2, 3, (M, K), generator=g).float()
    w = torch.randint(-2, 3, (N, K), generator=g).float()
    a[:, 0] += torch.arange(M).float() % 5
    w[:, -1] += torch.arange(N).float() % 3
    bias = torch.randint(-4, 5, (N,), generator=g).float()
    ref = a.double() @ w.double().t() + bias.double()
    buf = torch.full((M, N + 8), 7.0, dtype=torch.float32, device=DEV)
    ops.gemm(bf(a).to(DEV), bf(w).to(DEV), bias.to(DEV), out=buf[:, 4:N + 4])
    assert torch.equal(buf[:, 4:N + 4].cpu().double(), ref), (buf[:, 4:N + 4].cpu().double() - ref).abs().max()
    assert (buf[:, :4] == 7).all() and (buf[:, N + 4:] == 7).all()
    out = ops.gemm(bf(a).to(DEV), bf(w).to(DEV), out_dtype=torch.float32)
    assert torch.equal(out.cpu().double(), ref - bias.double())


@pytest.mark.parametrize("M,N,K", [(300, 256, 96), (1000, 384, 384), (8, 256, 2048), (260, 192, 192)])
def test_gemm_epilogue_modes(ops, M, N, K):
    """Every specialised epilogue (direct stores from the accumulator layout: linear -> 16-bit / fp32, + fp32 residual, GELU /
    ReLU -> 16-bit), the LDS path (row-modulo residual) and the generic fallback (sigmoid, 16-bit residual, unaligned views)."""
    a, w = bf(rnd(M, K, seed=11)), bf(rnd(N, K, seed=12, scale=0.2))
    bias, cs, res = rnd(N, seed=13), rnd(N, seed=14), rnd(M, N, seed=15)
    lin = a.float() @ w.float().t() + bias
    d = lambda t: t.to(DEV)
    close(ops.gemm(d(a), d(w), d(bias), out_dtype=OP16()), lin, 1e-3, 8e-3, "linear->16")
    close(ops.gemm(d(a), d(w), d(bias), out_dtype=torch.float32), lin, 2e-4, 1e-5, "linear->f32")
    close(ops.gemm(d(a), d(w), d(bias), residual=d(res), out_dtype=torch.float32), lin + res, 2e-4, 1e-5, "+res->f32")
    close(ops.gemm(d(a), d(w), d(bias), colscale=d(cs), residual=d(res), out_dtype=torch.float32), lin * cs + res, 2e-4, 1e-5, "scale+res")
    close(ops.gemm(d(a), d(w), d(bias), act=ops.ACT_GELU, out_dtype=OP16()), O.gelu(lin), 1e-3, 8e-3, "gelu->16")
    close(ops.gemm(d(a), d(w), d(bias), act=ops.ACT_RELU, out_dtype=OP16()), torch.relu(lin), 1e-3, 8e-3, "relu->16")
    close(ops.gemm(d(a), d(w), d(bias), act=ops.ACT_SIGMOID, out_dtype=torch.float32), torch.sigmoid(lin), 2e-4, 1e-5, "sigmoid")
    close(ops.gemm(d(a), d(w), d(bias), act=ops.ACT_GELU, out_dtype=torch.float32), O.gelu(lin), 2e-4, 1e-5, "gelu->f32 (generic)")
    close(ops.gemm(d(a), d(w), d(bias), residual=d(bf(res)), out_dtype=torch.float32), lin + bf(res).float(), 2e-4, 1e-5, "16-bit residual")
    # unaligned output view (column offset 1 of a wider fp32 buffer) and N not a multiple of 4
    buf = torch.zeros(M, N + 3, dtype=torch.float32, device=DEV)
    ops.gemm(d(a), d(w), d(bias), out=buf[:, 1:N + 1])
    close(buf[:, 1:N + 1], lin, 2e-4, 1e-5, "unaligned view")
    assert buf[:, 0].abs().sum().item() == 0 and buf[:, N + 1:].abs().sum().item() == 0
    close(ops.gemm(d(a), d(w[:N - 2]), d(bias[:N - 2]), out_dtype=OP16()), lin[:, :N - 2], 1e-3, 8e-3, "N % 4 != 0")


@pytest.mark.parametrize("B,L,n_excl,N,cols,D,K", [(2, 256, 0, 256, 256, 256, 64), (3, 260, 4, 256, 256, 256, 64), (2, 1024, 0, 768, 512, 256, 256),
                                                   (1, 4100, 4, 128, 128, 64, 96)])
def test_gemm_rope_fused(ops, B, L, n_excl, N, cols, D, K):
    """RoPE fused into the projection's store == projection (fp32) followed by the oracle's rotation; rows past n_rope of every
    batch (object-pointer tokens) and columns past rope_cols (the v third of a fused qkv) stay unrotated; keys tile the table."""
    side = 16
    cos, sin = ops.rope_table(side, D, 10000.0, DEV)
    a, w, bias = bf(rnd(B * L, K, seed=21)), bf(rnd(N, K, seed=22, scale=0.2)), rnd(N, seed=23)
    lin = (a.float() @ w.float().t() + bias).view(B, L, N)
    oc, osn = O.axial_rope_table(D, side, side, 10000.0)
    assert max_abs_t(cos.cpu(), oc) < 1e-5 and max_abs_t(sin.cpu(), osn) < 1e-5
    n_rope = L - n_excl
    ref = lin.clone()
    reps = -(-n_rope // (side * side))
    c, s_ = oc.repeat(reps, 1)[:n_rope], osn.repeat(reps, 1)[:n_rope]
    for h0 in range(0, cols, D):
        ref[:, :n_rope, h0:h0 + D] = O.rope_rotate(lin[:, :n_rope, h0:h0 + D], c, s_)
    out = ops.gemm_rope(a.to(DEV), w.to(DEV), bias.to(DEV), (cos, sin), rope_cols=cols, head_dim=D, rows_per_batch=L, n_rope=n_rope)
    close(out.view(B, L, N), ref, 2e-3, 8e-3, "gemm_rope")
    with pytest.raises(RuntimeError):
        ops.gemm_rope(a.to(DEV), w.to(DEV), bias.to(DEV), (cos, sin), rope_cols=cols + 4, head_dim=D, rows_per_batch=L, n_rope=n_rope)


@pytest.mark.parametrize("B,H,W,N,K", [(2, 16, 16, 192, 96), (1, 32, 24, 384, 192), (3, 18, 14, 768, 384), (1, 64, 64, 200, 64)])
def test_gemm_pool2x2_fused(ops, B, H, W, N, K):
    """projection + 2x2 max-pool in one GEMM == max_pool2d(linear(x)) (hieradet.py:23-34, 141-145), every kernel family"""
    a, w, bias = bf(rnd(B * H * W, K, seed=31)), bf(rnd(N, K, seed=32, scale=0.2)), rnd(N, seed=33)
    lin = (a.float() @ w.float().t() + bias).view(B, H, W, N).permute(0, 3, 1, 2)
    ref = F.max_pool2d(lin, 2, 2).permute(0, 2, 3, 1).reshape(-1, N)
    out = ops.gemm_pool2x2(a.to(DEV), w.to(DEV), bias.to(DEV), B, H, W)
    assert out.shape == ref.shape
    close(out, ref, 2e-4, 1e-5, "gemm_pool2x2")


@pytest.mark.parametrize("B,H,W,width,K", [(2, 16, 16, 192, 96), (1, 32, 24, 384, 192), (3, 18, 14, 768, 384), (1, 64, 64, 64, 64)])
def test_gemm_qkv_pool2x2_fused(ops, B, H, W, width, K):
    """k | v columns bit-identical to the plain projection (image order), pooled q == maxpool2x2 of the plain q columns"""
    N = 3 * width
    a, w, bias = bf(rnd(B * H * W, K, seed=41)), bf(rnd(N, K, seed=42, scale=0.2)), rnd(N, seed=43)
    plain = ops.gemm(a.to(DEV), w.to(DEV), bias.to(DEV))
    qkv, q2 = ops.gemm_qkv_pool2x2(a.to(DEV), w.to(DEV), bias.to(DEV), B, H, W, width)
    assert torch.equal(qkv[:, width:], plain[:, width:])
    ref_q = ops.maxpool2x2(plain[:, :width], B, H, W)
    assert torch.equal(q2, ref_q)


def test_fp16_operands_saturate_instead_of_overflowing(ops):
    """fp16 build: a 16-bit GEMM / LayerNorm output beyond +-65504 saturates (common.h f2op) instead of becoming inf -- with a real
    checkpoint a large MLP / qkv activation must not poison softmax, LayerNorm or the optimiser state.  bf16 build: in range anyway."""
    a = torch.full((128, 512), 300.0).to(OP16()).to(DEV)
    w = torch.ones(64, 512).to(OP16()).to(DEV)
    w[1::2] = -1.0
    out = ops.gemm(a, w, None)                                   # |a w^T| = 153600
    assert torch.isfinite(out.float()).all()
    if OP16() == torch.float16:
        assert out[:, 0].float().min().item() == 65504.0 and out[:, 1].float().max().item() == -65504.0
    else:
        assert abs(out[0, 0].float().item() - 153600.0) < 1200.0
    big = torch.zeros(4, 256)
    big[:, 0] = 3.0e6                                            # LayerNorm output ~ 16 * gamma: gamma 1e4 -> 1.6e5 > fp16 max
    y = ops.layernorm(big.to(DEV), torch.full((256,), 1.0e4, device=DEV), torch.zeros(256, device=DEV), 1e-6)
    assert torch.isfinite(y.float()).all() and y.float().abs().max().item() >= 65504.0 * (1.0 if OP16() == torch.float16 else 2.0)
    fine = ops.gemm(a, (w * 0.001).to(OP16()), None, out_dtype=torch.float32)     # fp32 outputs are never clamped
    assert abs(fine[0, 0].item() - 153.6) < 1.0


def test_gemm_rejects_bad_shapes(ops):
    with pytest.raises(Exception):
        ops.gemm(bf(rnd(8, 12)).to(DEV), bf(rnd(8, 12)).to(DEV))  # K % 8 != 0


@pytest.mark.parametrize("rows,C,eps,act", [(100, 96, 1e-6, 0), (37, 768, 1e-6, 0), (64, 256, 1e-5, 0), (10, 64, 1e-6, 1),
                                            (5, 4, 1e-6, 1)])
def test_layernorm(ops, rows, C, eps, act):
    x, w, b = rnd(rows, C, seed=1, scale=2.0) + 0.5, 1 + 0.1 * rnd(C, seed=2), 0.1 * rnd(C, seed=3)
    P = {"n.weight": w, "n.bias": b}
    ref = O.lnorm(P, "n", x, eps)
    if act:
        ref = O.gelu(ref)
    out = ops.layernorm(x.to(DEV), w.to(DEV), b.to(DEV), eps, act=act, out_dtype=torch.float32)
    close(out, ref, 2e-5, 1e-5, "layernorm f32")
    out = ops.layernorm(bf(x).to(DEV), w.to(DEV), b.to(DEV), eps, act=act, out_dtype=OP16())
    refb = O.lnorm(P, "n", bf(x).float(), eps)
    close(out, O.gelu(refb) if act else refb, 2e-3, 8e-3, "layernorm bf16")
    if not act and C % 4 == 0:
        # two outputs from one launch: the fp32 rows bit-equal to the one-output call, the 16-bit rows = their rounding
        one = ops.layernorm(x.to(DEV), w.to(DEV), b.to(DEV), eps, out_dtype=torch.float32)
        y32, y16 = ops.layernorm_dual(x.to(DEV), w.to(DEV), b.to(DEV), eps)
        assert torch.equal(y32, one) and y16.dtype == OP16() and torch.equal(y16, one.to(OP16()))


# ------------------------------------------------------------------------------------------------------------------
def _attn_ref(q, k, v):
    return O.softmax_attention(q.double(), k.double(), v.double()).float()


@pytest.mark.parametrize("B,H,Lq,Lk,D,splits", [(1, 1, 32, 32, 96, 1), (2, 4, 70, 100, 96, 1), (1, 2, 256, 256, 96, 1),
                                                (2, 1, 128, 520, 256, 1), (1, 1, 200, 2100, 256, 4), (1, 1, 64, 4096 + 8, 256, 8),
                                                (1, 2, 40, 33, 64, 1), (1, 1, 130, 64, 128, 2),
                                                # D = 96 with >= 256 queries: the 64-queries-per-wave kernel with 64-key stages (Hiera's
                                                # global blocks) -- one full stage, one short sub-tile, an odd sub-tile count, a ragged query
                                                # tile, a short last stage after full ones, splits with even / odd stage counts
                                                (1, 1, 256, 64, 96, 1), (1, 1, 256, 31, 96, 1), (1, 2, 512, 96, 96, 1), (2, 1, 300, 200, 96, 1),
                                                (1, 1, 260, 1000, 96, 1), (1, 2, 384, 1100, 96, 3), (1, 1, 1024, 4100, 96, 7)])
def test_attention_vs_oracle(ops, B, H, Lq, Lk, D, splits):
    q, k, v = bf(rnd(B, H, Lq, D, seed=1)), bf(rnd(B, H, Lk, D, seed=2)), bf(rnd(B, H, Lk, D, seed=3))
    # row-distinct V so that a key permutation inside a tile would show
    v = bf(v.float() + torch.arange(Lk).float()[None, None, :, None] / Lk)
    ref = _attn_ref(q.float(), k.float(), v.float())
    out = ops.attention(q.to(DEV), k.to(DEV), v.to(DEV), splits=splits)
    close(out, ref, 0.02, 0.01, "attention")


def test_attention_d96_four_wave_shape_in_a_child_process():
    """attn_g96x2_kernel<2, 4> (64 queries per wave, one wave per SIMD: MSAM2_G96_X2=1, read once per process) on the same cases as the
    default 8-wave shape."""
    import subprocess, sys
    env = dict(_os.environ)
    env["MSAM2_G96_X2"] = "1"
    r = subprocess.run([sys.executable, "-m", "pytest", _os.path.abspath(__file__), "-q", "-m", "gpu", "-k", "test_attention_vs_oracle and 96"],
                       env=env, capture_output=True, text=True, cwd=_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("B,H,Lq,Lk,splits", [(1, 1, 32, 32, 1), (2, 1, 200, 520, 1), (1, 1, 130, 2100, 4), (2, 1, 64, 4096 + 8, 8),
                                              (1, 2, 256, 1000, 3), (1, 1, 1024, 33, 1),
                                              # 64-queries-per-wave kernel (Lq >= 256): ragged query tile, a single full tile, a single
                                              # partial tile, splits with odd / even tile counts per split
                                              (2, 1, 300, 520, 1), (1, 1, 4096, 32, 1), (1, 1, 512, 31, 1), (1, 1, 260, 4100, 7), (1, 1, 384, 96, 3)])
def test_attention_kv64_vs_oracle(ops, B, H, Lq, Lk, splits):
    """memory cross-attention with 64-wide value rows (msam2_attention_kv64_fwd): partial tiles, ragged query tiles, splits"""
    q, k, v = bf(rnd(B, H, Lq, 256, seed=1)), bf(rnd(B, H, Lk, 256, seed=2)), bf(rnd(B, H, Lk, 64, seed=3))
    v = bf(v.float() + torch.arange(Lk).float()[None, None, :, None] / Lk)       # row-distinct values: a key permutation would show
    ref = _attn_ref(q.float(), k.float(), v.float())
    out = ops.attention_kv64(q.to(DEV), k.to(DEV), v.to(DEV), splits=splits)
    assert out.shape == (B, H, Lq, 64) and out.permute(0, 2, 1, 3).is_contiguous()
    close(out, ref, 0.02, 0.01, "attention kv64")


@pytest.mark.parametrize("Lq", [96, 320])
def test_attention_kv64_strided_and_rescale(ops, Lq):
    """k / v as strided views (the [B, Nk, C] layouts of the memory bank), a peaked key in a late tile (running-max rescale: in the
    64-queries-per-wave kernel, Lq = 320, the AGPR read / write path of the O accumulators), and the deferred merge == the fused call
    bit for bit"""
    B, Lk = 2, 700
    q = bf(rnd(B, 1, Lq, 256, seed=1))
    kbig, vbig = bf(rnd(B, Lk, 320, seed=2) * 0.2), bf(rnd(B, Lk, 128, seed=3))
    kbig[0, 610, 64:] = bf(q[0, 0, 7].float() * 3)
    k = kbig[:, :, 64:].unsqueeze(1)            # [B,1,Lk,256], row pitch 320
    v = vbig[:, :, 64:].unsqueeze(1)            # [B,1,Lk,64], row pitch 128
    ref = _attn_ref(q.float(), k.float(), v.float())
    dq, dk, dv = q.to(DEV), kbig.to(DEV)[:, :, 64:].unsqueeze(1), vbig.to(DEV)[:, :, 64:].unsqueeze(1)
    one = ops.attention_kv64(dq, dk, dv)
    close(one, ref, 0.02, 0.01, "kv64 strided")
    fused = ops.attention_kv64(dq, dk, dv, splits=5)
    close(fused, ref, 0.02, 0.01, "kv64 strided split")
    ws = ops.attention_workspace(B, 1, Lq, 64, 5, DEV)
    part = ops.attention_kv64(dq, dk, dv, splits=5, workspace=ws, defer_merge=True)
    ops.attention_merge(part, Lk, 5, ws)
    assert torch.equal(part, fused)


def test_attention_long_key_range_vs_fp64(ops):
    """Lk = 155 648 + 64 (BASELINE config 3's steady state: 38 memories of 4096 tokens + pointer tokens): both D = 256 kernels against
    an fp64 softmax on 256 query rows, through the split counts the model uses."""
    from medical_sam2_amd.modeling.common import attn_splits
    Lq, Lk = 256, 38 * 4096 + 64
    q, k = bf(rnd(1, 1, Lq, 256, seed=1) * 0.5), bf(rnd(1, 1, Lk, 256, seed=2) * 0.5)
    v, m = bf(rnd(1, 1, Lk, 256, seed=3)), bf(rnd(1, 1, Lk, 64, seed=4))
    sp = max(attn_splits(1, 1, 4096, Lk), 2)
    for name, vals, fn in (("d256", v, ops.attention), ("kv64", m, ops.attention_kv64)):
        ref = _attn_ref(q.float(), k.float(), vals.float())
        out = fn(q.to(DEV), k.to(DEV), vals.to(DEV), splits=sp)
        # 155 k-term convex combinations of N(0,1) values: |ref| ~ 0.02; the bound is the 16-bit rounding of p and of the partials
        close(out, ref, 2e-3, 0.01, f"long key range {name} (splits {sp})")


def test_attention_deferred_merge(ops):
    """split pass alone (negative split count through the C ABI) + msam2_attention_merge == the fused call, bit for bit"""
    B, H, Lq, Lk, D, splits = 2, 1, 200, 1000, 256, 4
    q, k, v = (bf(rnd(B, H, L, D, seed=s)).to(DEV) for L, s in ((Lq, 1), (Lk, 2), (Lk, 3)))
    ref = ops.attention(q, k, v, splits=splits)
    ws = ops.attention_workspace(B, H, Lq, D, splits, DEV)
    out = torch.zeros(B, Lq, H, D, dtype=OP16(), device=DEV).permute(0, 2, 1, 3)
    ops.attention(q, k, v, splits=splits, out=out, workspace=ws, defer_merge=True)
    assert out.abs().sum().item() == 0          # nothing written before the merge
    ops.attention_merge(out, Lk, splits, ws)
    assert torch.equal(out, ref)
    with pytest.raises(ValueError):
        ops.attention(q, k, v, splits=1, out=out, workspace=ws, defer_merge=True)


def test_attention_packed_strided_qkv(ops):
    """q/k/v as strided views of one fused [B, L, 3, H, D] projection buffer (how the callers hold them)."""
    B, H, L, D = 2, 4, 96, 96
    qkv = bf(rnd(B, L, 3, H, D, seed=5))
    ref = _attn_ref(*[qkv[:, :, i].permute(0, 2, 1, 3).float() for i in range(3)])
    dq = qkv.to(DEV)
    out = ops.attention(*[dq[:, :, i].permute(0, 2, 1, 3) for i in range(3)])
    close(out, ref, 0.02, 0.01, "attention packed")
    assert out.permute(0, 2, 1, 3).is_contiguous()  # heads recombined [B, L, H, D]


def test_attention_peaked_softmax_rescale(ops):
    """Forces the running-max rescale branch: one key far above the rest, placed in a late tile."""
    B, H, Lq, Lk, D = 1, 1, 64, 300, 256
    q, k, v = bf(rnd(B, H, Lq, D, seed=1)), bf(rnd(B, H, Lk, D, seed=2) * 0.1), bf(rnd(B, H, Lk, D, seed=3))
    k[0, 0, 257] = bf(q[0, 0, 5].float() * 4)
    k[0, 0, 40] = bf(q[0, 0, 9].float() * 2)
    ref = _attn_ref(q.float(), k.float(), v.float())
    close(ops.attention(q.to(DEV), k.to(DEV), v.to(DEV)), ref, 0.02, 0.01, "attention peaked")
    close(ops.attention(q.to(DEV), k.to(DEV), v.to(DEV), splits=3), ref, 0.02, 0.01, "attention peaked split")


@pytest.mark.parametrize("B,Hh,Ww,heads,ws,pool", [(2, 16, 16, 1, 8, False), (1, 16, 16, 2, 8, True), (1, 16, 16, 2, 4, False),
                                                    (1, 8, 8, 4, 4, True), (2, 16, 16, 4, 14, False), (1, 16, 16, 8, 14, True),
                                                    (1, 8, 8, 8, 7, False), (1, 64, 64, 4, 14, False)])
def test_window_attention_vs_oracle(ops, B, Hh, Ww, heads, ws, pool):
    D = 96
    dim_out = heads * D
    qkv = bf(rnd(B * Hh * Ww, 3 * dim_out, seed=7))
    bias = rnd(3 * dim_out, seed=8)
    # oracle: windows of the *token image*; padded tokens carry the bias (qkv of a zero row), as hieradet.py:138-148 yields
    img = qkv.float().reshape(B, Hh, Ww, 3 * dim_out)
    win, pad_hw = O.to_windows(img - bf(bias).float(), ws)  # zero pad, then add bias back -> pad rows == bias
    win = win + bf(bias).float()
    Bw = win.shape[0]
    t = win.reshape(Bw, ws * ws, 3, heads, D)
    q, k, v = t[:, :, 0], t[:, :, 1], t[:, :, 2]
    wq = ws
    if pool:
        q = O.maxpool2x2_nhwc(q.reshape(Bw, ws, ws, dim_out))
        wq = ws // 2
        q = q.reshape(Bw, wq * wq, heads, D)
    o = _attn_ref(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2)).transpose(1, 2).reshape(Bw, wq, wq, dim_out)
    Hq, Wq = (Hh // 2, Ww // 2) if pool else (Hh, Ww)
    padq = (Hq + (wq - Hq % wq) % wq, Wq + (wq - Wq % wq) % wq)
    ref = O.from_windows(o, wq, padq, (Hq, Wq)).reshape(B * Hq * Wq, dim_out)
    dq = qkv.to(DEV)
    qp = None
    if pool:
        qp = ops.maxpool2x2(dq[:, :dim_out], B, Hh, Ww)
    out = ops.window_attention(dq, B, Hh, Ww, heads, ws, bias.to(DEV), q_pooled=qp)
    close(out, ref, 0.02, 0.01, "window attention")


@pytest.mark.parametrize("B,Lq,Lk,heads,C", [(2, 8, 256, 8, 128), (2, 256, 8, 8, 128), (3, 7, 7, 8, 256), (1, 9, 4096, 8, 128),
                                            (2, 8, 1500, 8, 256), (4, 9, 4096, 8, 128), (2, 5, 3000, 8, 128), (1, 32, 1024, 4, 64)])
def test_attention_small(ops, B, Lq, Lk, heads, C):
    q, k, v = bf(rnd(B, Lq, C, seed=1)), bf(rnd(B, Lk, C, seed=2)), bf(rnd(B, Lk, C, seed=3))
    D = C // heads
    sp = lambda t: t.float().reshape(B, t.shape[1], heads, D).transpose(1, 2)
    ref = _attn_ref(sp(q), sp(k), sp(v)).transpose(1, 2).reshape(B, Lq, C)
    close(ops.attention_small(q.to(DEV), k.to(DEV), v.to(DEV), heads), ref, 0.01, 0.01, "attention_small")


# ------------------------------------------------------------------------------------------------------------------
def test_add_cast_strided_broadcast(ops):
    a, b = rnd(6, 5, 64, seed=1), rnd(6, 5, 64, seed=2)
    close(ops.add_cast(a.to(DEV), b.to(DEV), 0.1, torch.float32), a + 0.1 * b, 1e-6, 0, "add_cast")
    # seq-first -> batch-first move with a broadcast vector
    at = a.to(DEV).transpose(0, 1)  # [5, 6, 64] strided view
    vec = rnd(1, 1, 64, seed=3)
    out = ops.add_cast(at, vec.to(DEV), 1.0, OP16())
    close(out, (a.transpose(0, 1) + vec), 1e-2, 8e-3, "add_cast bf16 strided")
    close(ops.add_cast(bf(a).to(DEV), None, 1.0, torch.float32), bf(a).float(), 0, 0, "cast")


def test_maxpool_upsample(ops):
    B, H, W, C = 2, 8, 12, 48
    x = rnd(B * H * W, C, seed=1)
    ref = O.maxpool2x2_nhwc(x.reshape(B, H, W, C)).reshape(-1, C)
    close(ops.maxpool2x2(x.to(DEV), B, H, W), ref, 0, 0, "maxpool f32")
    wide = bf(rnd(B * H * W, 3 * C, seed=2)).to(DEV)
    close(ops.maxpool2x2(wide[:, C:2 * C], B, H, W), O.maxpool2x2_nhwc(wide[:, C:2 * C].float().cpu().reshape(B, H, W, C)).reshape(-1, C),
          0, 0, "maxpool strided bf16")
    y, top = rnd(B, H, W, C, seed=3), rnd(B, H // 2, W // 2, C, seed=4)
    ref = y + F.interpolate(top.permute(0, 3, 1, 2), scale_factor=2.0, mode="nearest").permute(0, 2, 3, 1)
    close(ops.upsample2x_add_(y.to(DEV).contiguous(), top.to(DEV).contiguous(), B, H, W), ref, 1e-6, 0, "upsample2x_add")


def test_rope(ops):
    side, D = 8, 256
    cs, sn = ops.rope_table(side, D, 10000.0, DEV)
    rc, rs = O.axial_rope_table(D, side, side)
    close(cs, rc, 2e-6, 0, "rope cos")
    close(sn, rs, 2e-6, 0, "rope sin")
    B, L = 2, 3 * side * side + 8
    x = bf(rnd(B, L, D, seed=1))
    n_rope = 3 * side * side
    ref = torch.cat([O.rope_rotate(x[:, :n_rope].float(), rc.repeat(3, 1), rs.repeat(3, 1)), x[:, n_rope:].float()], dim=1)
    wide = torch.zeros(B, L, 3 * D, dtype=OP16(), device=DEV)
    wide[:, :, D:2 * D] = x.to(DEV)
    ops.rope_(wide[:, :, D:2 * D], n_rope, (cs, sn))
    close(wide[:, :, D:2 * D], ref, 2e-2, 8e-3, "rope")
    assert wide[:, :, :D].abs().sum().item() == 0 and wide[:, :, 2 * D:].abs().sum().item() == 0


def test_bilinear_and_tables(ops):
    x = rnd(3, 2, 16, 16, seed=1)
    close(ops.bilinear_upsample(x.to(DEV), 64, 64), F.interpolate(x, size=(64, 64), mode="bilinear", align_corners=False), 1e-5, 0,
          "bilinear")
    for (h, w, C) in ((16, 16, 256), (64, 64, 64), (8, 12, 256)):
        close(ops.sine_pos_2d(h, w, C, DEV), O.sine_pos_2d(h, w, C).permute(1, 2, 0).reshape(h * w, C), 2e-5, 0, "sine pos")
    gauss = rnd(2, 128, seed=2)
    P = {"sam_prompt_encoder.pe_layer.positional_encoding_gaussian_matrix": gauss}
    close(ops.fourier_pe_grid(gauss.to(DEV), 16, 16), O.dense_pe(P, 16, 16)[0].permute(1, 2, 0).reshape(256, 256), 5e-5, 0, "dense pe")
    pe, pw = rnd(1, 96, 7, 7, seed=3), rnd(1, 96, 8, 8, seed=4)
    Pp = {"t.pos_embed": pe, "t.pos_embed_window": pw}
    close(ops.hiera_pos_embed(pe.to(DEV), pw.to(DEV), 64, 64), O.hiera_pos_embed(Pp, "t", 64, 64).reshape(64 * 64, 96), 2e-5, 1e-5,
          "hiera pos embed")
    pe14 = rnd(1, 112, 14, 14, seed=5)
    Pp = {"t.pos_embed": pe14, "t.pos_embed_window": rnd(1, 112, 8, 8, seed=6)}
    close(ops.hiera_pos_embed(pe14.to(DEV), Pp["t.pos_embed_window"].to(DEV), 32, 32), O.hiera_pos_embed(Pp, "t", 32, 32).reshape(1024, 112),
          2e-5, 1e-5, "hiera pos embed 14")


def test_patch_embed_via_im2col(ops):
    B, S, E = 2, 64, 96
    img, w, b = rnd(B, 3, S, S, seed=1), rnd(E, 3, 7, 7, seed=2, scale=0.1), rnd(E, seed=3)
    wb = bf(w)
    ref = F.conv2d(img, wb.float(), b, stride=4, padding=3).permute(0, 2, 3, 1).reshape(-1, E)
    cols = ops.im2col_patch(img.to(DEV))
    wmat = torch.zeros(E, 160, dtype=OP16())
    wmat[:, :147] = wb.reshape(E, 147)
    out = ops.gemm(cols, wmat.to(DEV), b.to(DEV), out_dtype=torch.float32)
    # the im2col rounds the image to bf16: compare against the conv of the rounded image
    ref_b = F.conv2d(bf(img).float(), wb.float(), b, stride=4, padding=3).permute(0, 2, 3, 1).reshape(-1, E)
    close(out, ref_b, 2e-4, 1e-5, "patch embed")
    close(out, ref, 0.05, 0.02, "patch embed vs unrounded image")


@pytest.mark.parametrize("cin,mode", [(1, 0), (1, 1), (1, 2), (4, 0), (16, 0)])
def test_conv3x3s2_ln_gelu(ops, cin, mode):
    B, H, W = 2, 16, 24
    cout = 4 * cin
    w, b = rnd(cout, cin, 3, 3, seed=1, scale=0.3), rnd(cout, seed=2)
    lw, lb = 1 + 0.1 * rnd(cout, seed=3), 0.1 * rnd(cout, seed=4)
    if cin == 1:
        x = rnd(B, 1, H, W, seed=5, scale=3.0)
        xin = x
        if mode == 1:
            xin = torch.sigmoid(x) * 20.0 - 10.0
        elif mode == 2:
            xin = (x > 0).float() * 20.0 - 10.0
        dx = x.reshape(B * H * W, 1).to(DEV)
    else:
        x = bf(rnd(B, cin, H, W, seed=5))
        xin = x.float()
        dx = x.permute(0, 2, 3, 1).reshape(-1, cin).contiguous().to(DEV)
    P = {"n.weight": lw, "n.bias": lb}
    ref = O.gelu(O.lnorm2d(P, "n", F.conv2d(xin, w, b, stride=2, padding=1))).permute(0, 2, 3, 1).reshape(-1, cout)
    out = ops.conv3x3s2_ln_gelu(dx, B, H, W, w.to(DEV), b.to(DEV), lw.to(DEV), lb.to(DEV), mode, 20.0, -10.0)
    close(out, ref, 6e-3, 8e-3, "conv3x3s2_ln_gelu")


def test_im2col3x3s2_gemm_matches_conv(ops):
    B, H, W, cin, cout = 1, 16, 16, 64, 256
    x, w = bf(rnd(B, cin, H, W, seed=1)), bf(rnd(cout, cin, 3, 3, seed=2, scale=0.05))
    ref = F.conv2d(x.float(), w.float(), None, stride=2, padding=1).permute(0, 2, 3, 1).reshape(-1, cout)
    cols = ops.im2col3x3s2(x.permute(0, 2, 3, 1).contiguous().to(DEV), B, H, W)
    wm = w.permute(0, 2, 3, 1).reshape(cout, 9 * cin).contiguous()
    close(ops.gemm(cols, wm.to(DEV), out_dtype=torch.float32), ref, 2e-4, 1e-5, "im2col3x3s2")


def test_dwconv7x7_ln(ops):
    B, H, W, C = 2, 16, 16, 256
    x, w, b = rnd(B, C, H, W, seed=1), rnd(C, 1, 7, 7, seed=2, scale=0.2), rnd(C, seed=3)
    lw, lb = 1 + 0.1 * rnd(C, seed=4), 0.1 * rnd(C, seed=5)
    P = {"n.weight": lw, "n.bias": lb}
    ref = O.lnorm2d(P, "n", F.conv2d(x, w, b, padding=3, groups=C)).permute(0, 2, 3, 1).reshape(-1, C)
    out = ops.dwconv7x7_ln(x.permute(0, 2, 3, 1).contiguous().to(DEV), B, H, W, w.reshape(C, 49).t().contiguous().to(DEV), b.to(DEV),
                           lw.to(DEV), lb.to(DEV))
    close(out, ref, 6e-3, 8e-3, "dwconv7x7_ln")


@pytest.mark.parametrize("cin,cout,ln", [(256, 64, True), (64, 32, False)])
def test_convtranspose_via_gemm_shuffle(ops, cin, cout, ln):
    B, h, w = 2, 8, 8
    x, wt, b = bf(rnd(B, cin, h, w, seed=1)), bf(rnd(cin, cout, 2, 2, seed=2, scale=0.1)), rnd(cout, seed=3)
    skip = bf(rnd(B, cout, 2 * h, 2 * w, seed=4))
    lw, lb = 1 + 0.1 * rnd(cout, seed=5), 0.1 * rnd(cout, seed=6)
    u = F.conv_transpose2d(x.float(), wt.float(), b, stride=2) + skip.float()
    if ln:
        u = O.lnorm2d({"n.weight": lw, "n.bias": lb}, "n", u)
    ref = O.gelu(u).permute(0, 2, 3, 1).reshape(-1, cout)
    wm = wt.permute(2, 3, 1, 0).reshape(4 * cout, cin).contiguous()  # [(ky,kx,co), ci]
    g = ops.gemm(x.permute(0, 2, 3, 1).reshape(-1, cin).contiguous().to(DEV), wm.to(DEV))
    out = ops.convt2x2_shuffle(g, b.to(DEV), skip.permute(0, 2, 3, 1).contiguous().to(DEV), lw.to(DEV) if ln else None,
                               lb.to(DEV) if ln else None, B, h, w)
    close(out, ref, 0.03, 0.02, "convT shuffle")
    # the skip features in fp32 (as the FPN returns them): same result as from their 16-bit copy when the values are 16-bit representable
    out32 = ops.convt2x2_shuffle(g, b.to(DEV), skip.float().permute(0, 2, 3, 1).contiguous().to(DEV), lw.to(DEV) if ln else None,
                                 lb.to(DEV) if ln else None, B, h, w)
    assert torch.equal(out32, out)


def test_hyper_masks_prompt_select(ops):
    n, K, P_, C = 2, 4, 64 * 64, 32
    hyper, up = rnd(n, K, C, seed=1), bf(rnd(n, P_, C, seed=2))
    close(ops.hyper_masks(hyper.to(DEV), up.to(DEV), n, P_), hyper @ up.float().transpose(1, 2), 2e-4, 1e-5, "hyper masks")
    import medical_sam2_amd.weights as wts
    W = wts.init_weights("hiera_s", 0)
    cfg = O.model_config("hiera_s", 1024)
    xy = torch.tensor([[[100.5, 200.0], [0.0, 0.0]], [[1000.0, 3.0], [512.0, 512.0]]])
    lab = torch.tensor([[1, -1], [0, 3]], dtype=torch.int32)
    pe = "sam_prompt_encoder"
    emb = torch.cat([W[f"{pe}.point_embeddings.{i}.weight"] for i in range(4)])
    out = ops.prompt_points(xy.to(DEV), lab.to(DEV), W[pe + ".pe_layer.positional_encoding_gaussian_matrix"].to(DEV), emb.to(DEV),
                            W[pe + ".not_a_point_embed.weight"].to(DEV), 1024.0)
    # oracle path: points given WITH an explicit pad point already, so call its internals through a box-free prompt
    ref, _ = O.prompt_encoder(W, cfg, (xy[:, :1], lab[:, :1]), None, None)  # appends (0,0,-1)
    close(out[0], ref[0], 5e-4, 0, "prompt points row0")
    e = O.random_fourier_pe(W, (xy[1] + 0.5) / 1024.0)
    e = e + torch.stack([emb[0], emb[3]])
    close(out[1], e, 5e-4, 0, "prompt points row1")
    # the padding point appended inside the kernel (prompt_encoder.py:87-91) == the oracle's prompt encoder on the un-padded points
    padded = ops.prompt_points(xy[:, :1].contiguous().to(DEV), lab[:, :1].contiguous().to(DEV), W[pe + ".pe_layer.positional_encoding_gaussian_matrix"].to(DEV),
                               emb.to(DEV), W[pe + ".not_a_point_embed.weight"].to(DEV), 1024.0, n_pad=1)
    assert padded.shape == (2, 2, emb.shape[1])
    close(padded, ref, 5e-4, 0, "prompt points + in-kernel padding point")
    # selection
    masks, ious, obj = rnd(3, 4, 16, 16, seed=3), torch.rand(3, 4, generator=torch.Generator().manual_seed(4)), torch.tensor([1.0, -1.0, 2.0])
    masks[2, 0] = masks[2, 0] * 0.01  # unstable single mask -> dynamic fallback
    for mm in (True, False):
        low, sel, iou_sel = ops.select_mask(masks.to(DEV), ious.to(DEV), obj.to(DEV), mm, True, 0.05, 0.98)
        if mm:
            rm, ri = masks[:, 1:], ious[:, 1:]
            best = ri.argmax(-1)
            rl, rsel = rm[torch.arange(3), best][:, None], best + 1
        else:
            rl, ri2 = O.dynamic_multimask_via_stability(cfg, masks, ious)
            rsel = None
        rl = torch.where((obj > 0)[:, None, None, None], rl, torch.full_like(rl, -1024.0))
        close(low, rl, 0, 0, f"select mm={mm}")
        if rsel is not None:
            assert torch.equal(sel.cpu().long(), rsel)


@pytest.mark.parametrize("shape,p", [((3, 1, 64, 64), 0.5), ((4, 1, 256, 256), 0.42), ((1, 1, 30, 70), 0.6), ((2, 1, 128, 128), 0.93),
                                      ((2, 1, 256, 256), 0.03), ((1, 1, 1024, 1024), 0.55)])
def test_connected_components_bit_exact(ops, shape, p):
    g = torch.Generator().manual_seed(int(p * 1000) + shape[2])
    m = (torch.rand(shape, generator=g) < p).to(torch.uint8)
    rl, rc = cc_oracle.connected_components(m)
    lab, cnt = ops.connected_components(m.to(DEV))
    assert torch.equal(lab.cpu(), rl)
    assert torch.equal(cnt.cpu(), rc)


def test_connected_components_edge_cases(ops):
    for m in (torch.zeros(2, 1, 8, 6, dtype=torch.uint8), torch.ones(1, 1, 8, 6, dtype=torch.uint8)):
        rl, rc = cc_oracle.connected_components(m)
        lab, cnt = ops.connected_components(m.to(DEV))
        assert torch.equal(lab.cpu(), rl) and torch.equal(cnt.cpu(), rc)
    with pytest.raises(RuntimeError, match="even"):
        ops.connected_components(torch.zeros(1, 1, 5, 4, dtype=torch.uint8, device=DEV))
    with pytest.raises(RuntimeError, match="uint8"):
        ops.connected_components(torch.zeros(1, 1, 4, 4, dtype=torch.int32, device=DEV))
    with pytest.raises(RuntimeError, match="CUDA"):
        ops.connected_components(torch.zeros(1, 1, 4, 4, dtype=torch.uint8))


def test_fill_holes(ops):
    g = torch.Generator().manual_seed(3)
    m = torch.randn(3, 1, 64, 64, generator=g) + 0.8
    ref = O.fill_holes_in_mask_scores(m, 8, cc_oracle.connected_components)
    out = ops.fill_holes_(m.clone().to(DEV), 8)
    assert torch.equal(out.cpu(), ref)
    assert (ref != m).any()


@pytest.mark.parametrize("M,N,K,add_cols", [(32, 768, 256, 512), (9, 128, 256, 128), (4, 2048, 256, 0), (28, 256, 2048, 0), (1, 96, 64, 64)])
def test_gemm_tokens(ops, M, N, K, add_cols):
    """fp32 token rows + optional addend on the leading columns, converted in the operand load (two-way decoder projections)"""
    a, a2 = rnd(M, K, seed=61), rnd(M, K, seed=62)
    w, bias, res = bf(rnd(N, K, seed=63, scale=0.1)), rnd(N, seed=64), rnd(M, N, seed=65)
    wf = w.float()
    ref = torch.empty(M, N)
    ref[:, :add_cols] = bf(a + a2).float() @ wf[:add_cols].t()
    ref[:, add_cols:] = bf(a).float() @ wf[add_cols:].t()
    ref = ref + bias
    d = lambda t: t.to(DEV)
    out = ops.gemm_tokens(d(a), d(w), d(bias), addend=d(a2) if add_cols else None, add_cols=add_cols, out_dtype=torch.float32)
    close(out, ref, 3e-4, 1e-5, "gemm_tokens f32")
    out = ops.gemm_tokens(d(a), d(w), d(bias), addend=d(a2) if add_cols else None, add_cols=add_cols, act=ops.ACT_RELU)
    close(out, torch.relu(ref), 2e-3, 8e-3, "gemm_tokens relu->16")
    out = ops.gemm_tokens(d(a), d(w), d(bias), addend=d(a2), add_cols=N + 5, residual=d(res), out_dtype=torch.float32)
    close(out, bf(a + a2).float() @ wf.t() + bias + res, 3e-4, 1e-5, "gemm_tokens all columns + residual")
    with pytest.raises(RuntimeError):
        ops.gemm_tokens(d(a), d(w), d(bias), addend=d(a2), add_cols=8)


def test_token_mlp3(ops):
    """six 3-layer ReLU MLPs of width 256 in one launch == the per-head fp32 MLPs (16-bit weights, fp32 activations)"""
    G, B, T, C = 6, 3, 9, 256
    hs = rnd(B, T, C, seed=51)
    tok = torch.tensor([2, 3, 4, 5, 1, 0], dtype=torch.int32)
    od = torch.tensor([32, 32, 32, 32, 4, 1], dtype=torch.int32)
    sg = torch.tensor([0, 0, 0, 0, 1, 0], dtype=torch.int32)
    w1, w2 = bf(rnd(G, C, C, seed=52, scale=0.08)), bf(rnd(G, C, C, seed=53, scale=0.08))
    w3 = bf(rnd(G, C, C, seed=54, scale=0.08))
    b1, b2, b3 = rnd(G, C, seed=55), rnd(G, C, seed=56), rnd(G, C, seed=57)
    for g in range(G):
        w3[g, od[g]:] = 0
        b3[g, od[g]:] = 0
    y = ops.token_mlp3(hs.to(DEV), tok.to(DEV), w1.to(DEV), b1.to(DEV), w2.to(DEV), b2.to(DEV), w3.to(DEV), b3.to(DEV), od.to(DEV), sg.to(DEV))
    assert y.shape == (B, G, C)
    for g in range(G):
        x = hs[:, tok[g]]
        h = torch.relu(x @ w1[g].float().t() + b1[g])
        h = torch.relu(h @ w2[g].float().t() + b2[g])
        o = (h @ w3[g].float().t() + b3[g])[:, : od[g]]
        if sg[g]:
            o = torch.sigmoid(o)
        close(y[:, g, : od[g]], o, 2e-4, 2e-4, f"token_mlp3 head {g}")
    # packed output: hyper [B, 4, 32] | iou [B, 4] | obj [B, 1] as contiguous tensors, bit-equal to the slices of the [B, G, 256] form
    off = torch.tensor([0, 32, 64, 96, B * 128, B * 128 + B * 4], dtype=torch.int32)
    ld = torch.tensor([128, 128, 128, 128, 4, 1], dtype=torch.int32)
    flat = ops.token_mlp3(hs.to(DEV), tok.to(DEV), w1.to(DEV), b1.to(DEV), w2.to(DEV), b2.to(DEV), w3.to(DEV), b3.to(DEV), od.to(DEV), sg.to(DEV),
                          packed=(off.to(DEV), ld.to(DEV), B * 133))
    assert torch.equal(flat[: B * 128].view(B, 4, 32), y[:, :4, :32]) and torch.equal(flat[B * 128: B * 132].view(B, 4), y[:, 4, :4])
    assert torch.equal(flat[B * 132:].view(B, 1), y[:, 5, :1])


def test_hip_graph_replay(ops):
    a, w = bf(rnd(256, 128, seed=1)).to(DEV), bf(rnd(64, 128, seed=2)).to(DEV)
    out = torch.zeros(256, 64, dtype=torch.float32, device=DEV)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        ops.gemm(a, w, out=out)  # warm (module load) outside capture
        s.synchronize()
        out.zero_()
        with ops.HipGraph() as gr:
            ops.gemm(a, w, out=out)
        s.synchronize()
        assert out.abs().sum().item() == 0  # captured, not executed
        gr.replay()
        s.synchronize()
    close(out, a.float().cpu() @ w.float().cpu().t(), 2e-4, 1e-5, "graph replay")


@pytest.mark.parametrize("dim,T", [(96, 512), (96, 4096 + 77), (192, 256), (192, 3000), (384, 64), (384, 4096 + 37)])
def test_ln_mlp_residual_fused(ops, dim, T):
    """msam2_ln_mlp_residual_fwd (LayerNorm + fc1 + exact-erf GELU + fc2 + residual in one kernel, hieradet.py:166-167) against the
    oracle's norm2 / MLP on the same 16-bit weights: ragged token counts (partial passes, clamped rows), several passes per workgroup."""
    x = rnd(T, dim, seed=1, scale=1.5) + 0.3
    P = {"n.weight": 1 + 0.1 * rnd(dim, seed=2), "n.bias": 0.1 * rnd(dim, seed=3),
         "m.layers.0.weight": bf(rnd(4 * dim, dim, seed=4) / dim ** 0.5).float(), "m.layers.0.bias": 0.1 * rnd(4 * dim, seed=5),
         "m.layers.1.weight": bf(rnd(dim, 4 * dim, seed=6) / (4 * dim) ** 0.5).float(), "m.layers.1.bias": 0.1 * rnd(dim, seed=7)}
    h = O.lnorm(P, "n", x, 1e-6)
    ref = x + O.lin(P, "m.layers.1", O.gelu(O.lin(P, "m.layers.0", h)))
    d = lambda t: t.to(DEV)
    w2p = ops.mlp_fused_permute_w2(bf(P["m.layers.1.weight"]).to(DEV))
    out = ops.ln_mlp_residual(d(x), d(P["n.weight"]), d(P["n.bias"]), 1e-6, bf(P["m.layers.0.weight"]).to(DEV), d(P["m.layers.0.bias"]), w2p,
                              d(P["m.layers.1.bias"]))
    assert out.shape == x.shape and out.dtype == torch.float32
    # two 16-bit roundings (normalised input, hidden activation) as in the three-launch path
    close(out, ref, btol(6e-3), btol(4e-3), "fused LN+MLP")
    # against the three-launch path of the same library: same operand roundings, different summation order only
    xn = ops.layernorm(d(x), d(P["n.weight"]), d(P["n.bias"]), 1e-6)
    hid = ops.gemm(xn, bf(P["m.layers.0.weight"]).to(DEV), d(P["m.layers.0.bias"]), act=ops.ACT_GELU)
    three = ops.gemm(hid, bf(P["m.layers.1.weight"]).to(DEV), d(P["m.layers.1.bias"]), residual=d(x), out_dtype=torch.float32)
    close(out, three, btol(4e-3), btol(2e-3), "fused vs three launches")
    # second, 16-bit output of the same store (msam2_ln_mlp_residual_fwd_dual): fp32 rows unchanged, 16-bit rows = their rounding
    o32, o16 = ops.ln_mlp_residual(d(x), d(P["n.weight"]), d(P["n.bias"]), 1e-6, bf(P["m.layers.0.weight"]).to(DEV), d(P["m.layers.0.bias"]), w2p,
                                   d(P["m.layers.1.bias"]), also16=True)
    assert torch.equal(o32, out) and o16.dtype == OP16() and torch.equal(o16, out.to(OP16()))
