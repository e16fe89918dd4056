"""Size-independent properties at the benchmark's FULL sizes (the oracle is too slow there): the kernels are checked against
identities of the operation itself rather than against a second implementation."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import medical_sam2_amd.ops as ops
    return ops


def rnd(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


@pytest.mark.parametrize("B,H,Lq,Lk,D,splits", [(4, 1, 4096, 16384, 256, 4), (4, 1, 4096, 4096, 256, 4), (4, 4, 4096, 4096, 96, 1),
                                                 (1, 1, 4096, 28704, 256, 16)])
def test_attention_rows_are_convex_combinations(ops, B, H, Lq, Lk, D, splits):
    """softmax weights sum to 1: with V = const the output is that constant; with V = one-hot over a key subset the output is the
    subset's probability mass in [0, 1]; and the output is linear in V."""
    q = rnd(B, H, Lq, D, seed=1).to(ops.OP16).to(DEV)
    k = rnd(B, H, Lk, D, seed=2).to(ops.OP16).to(DEV)
    ones = torch.full((B, H, Lk, D), 0.75, dtype=ops.OP16, device=DEV)
    o = ops.attention(q, k, ones, splits=splits).float()
    assert (o - 0.75).abs().max().item() < 2e-3
    v1 = rnd(B, H, Lk, D, seed=3).to(ops.OP16).to(DEV)
    v2 = rnd(B, H, Lk, D, seed=4).to(ops.OP16).to(DEV)
    o1, o2 = ops.attention(q, k, v1, splits=splits).float(), ops.attention(q, k, v2, splits=splits).float()
    o12 = ops.attention(q, k, (v1.float() + v2.float()).to(ops.OP16), splits=splits).float()
    assert (o12 - (o1 + o2)).abs().max().item() < 2e-2          # 16-bit rounding of v1+v2 and of the three outputs
    # split count must not change the result beyond 16-bit rounding of the partials
    if splits > 1:
        o_one = ops.attention(q, k, v1, splits=1).float()
        assert (o_one - o1).abs().max().item() < 4e-3


@pytest.mark.parametrize("M,N,K", [(16384, 1536, 384), (16384, 384, 1536), (262144, 576, 96), (65536, 1152, 192), (16384, 256, 2048)])
def test_gemm_linearity_and_identity(ops, M, N, K):
    """C(A, W1 + W2) = C(A, W1) + C(A, W2) for weights whose sum is exact in 16 bits; A times a 0/1 selection matrix copies columns
    of A exactly (every kernel variant the step uses, at the step's shapes)."""
    g = torch.Generator().manual_seed(M + N)
    a = (torch.randint(-8, 9, (M, K), generator=g).float() / 8).to(ops.OP16).to(DEV)
    w1 = (torch.randint(-4, 5, (N, K), generator=g).float() / 4).to(ops.OP16).to(DEV)
    w2 = (torch.randint(-4, 5, (N, K), generator=g).float() / 4).to(ops.OP16).to(DEV)
    c1 = ops.gemm(a, w1, out_dtype=torch.float32)
    c2 = ops.gemm(a, w2, out_dtype=torch.float32)
    c12 = ops.gemm(a, (w1.float() + w2.float()).to(ops.OP16), out_dtype=torch.float32)
    assert torch.equal(c12, c1 + c2)                              # all partial sums are exactly representable in fp32
    sel = torch.zeros(N, K)
    cols = torch.arange(N) % K
    sel[torch.arange(N), cols] = 1.0
    out = ops.gemm(a, sel.to(ops.OP16).to(DEV))
    assert torch.equal(out, a[:, cols.to(DEV)])


def test_connected_components_invariants_full_size(ops):
    """areas of the labelled components add up to the foreground count; labels are constant on 8-neighbours; relabelling the
    label>0 mask reproduces the labels; hole filling is idempotent (13 objects x 256 x 256, the 3-D config's low-res masks)."""
    g = torch.Generator().manual_seed(9)
    m = (torch.rand(13, 1, 256, 256, generator=g) > 0.62).to(torch.uint8).to(DEV)
    labels, counts = ops.connected_components(m)
    fg = m.bool()
    assert ((labels > 0) == fg).all()
    for o in range(13):
        lab, cnt = labels[o, 0], counts[o, 0]
        ids, first = torch.unique(lab[lab > 0], return_inverse=False), None
        # every component's stored area equals its pixel count, and the areas add up to the foreground
        sizes = torch.bincount(lab[lab > 0].long())
        assert int(sizes.sum()) == int(fg[o].sum())
        px = lab > 0
        assert torch.equal(cnt[px].long(), sizes[lab[px].long()])
    for dy, dx in ((0, 1), (1, 0), (1, 1), (1, -1)):
        a = labels[:, :, : 256 - dy, max(0, -dx): 256 - max(0, dx)]
        b = labels[:, :, dy:, max(0, dx): 256 + min(0, dx)]
        both = (a > 0) & (b > 0)
        assert torch.equal(a[both], b[both])
    l2, c2 = ops.connected_components((labels > 0).to(torch.uint8))
    assert torch.equal(l2, labels) and torch.equal(c2, counts)
    scores = rnd(13, 1, 256, 256, seed=10).to(DEV) + 0.6
    once = ops.fill_holes_(scores.clone(), 8)
    twice = ops.fill_holes_(once.clone(), 8)
    assert torch.equal(once, twice) and (once >= scores).all()


def test_layernorm_statistics_full_size(ops):
    """rows of the normalised output have mean 0 / variance 1 (gamma = 1, beta = 0) at the stage-1 token count"""
    x = (rnd(262144, 96, seed=11) * 3 + 1.5).to(DEV)
    y = ops.layernorm(x, torch.ones(96, device=DEV), torch.zeros(96, device=DEV), 1e-6, out_dtype=torch.float32)
    assert y.mean(1).abs().max().item() < 1e-4
    assert (y.var(1, unbiased=False) - 1).abs().max().item() < 1e-3


@pytest.mark.parametrize("B,H,Lq,Lk,D", [(4, 1, 4096, 16384, 256), (4, 1, 4096, 4096, 256), (1, 1, 4096, 28704, 256), (2, 4, 1024, 1024, 96),
                                         (3, 2, 777, 2051, 128), (1, 1, 130, 9000, 256), (2, 2, 4100, 96, 64), (1, 3, 33, 65, 128)])
def test_attention_backward_identities_at_full_size(ops, B, H, Lq, Lk, D, monkeypatch):
    """Flash-style backward at the benchmark's sizes (where the CPU oracle is too slow) through identities of the operation:
      * the rows of dP are orthogonal to the constant vector: sum_k dS_k = 0  =>  dQ = 0 when every key is the same vector,
      * V-linearity: dV = P^T dO  =>  sum over keys of dV equals sum over queries of dO (the softmax rows sum to 1),
      * and against the materialised (GEMM-composed) path on the same inputs."""
    import medical_sam2_amd.backward as bwd
    q = rnd(B, H, Lq, D, seed=11).to(ops.OP16).to(DEV)
    k = rnd(B, H, Lk, D, seed=12).to(ops.OP16).to(DEV)
    v = rnd(B, H, Lk, D, seed=13).to(ops.OP16).to(DEV)
    do = rnd(B, H, Lq, D, seed=14).to(ops.OP16).float().to(DEV)
    dq, dk, dv = bwd.attention_backward(q, k, v, do)
    assert torch.isfinite(dq).all() and torch.isfinite(dk).all() and torch.isfinite(dv).all()
    # column sums of dV: sum_k dV[k] = sum_q (sum_k P[q,k]) dO[q] = sum_q dO[q]
    lhs, rhs = dv.sum(dim=2), do.sum(dim=2)
    assert (lhs - rhs).abs().max().item() < 2e-2 * rhs.abs().max().item() + 1e-3 * (Lq ** 0.5)
    # identical keys: the scores of a row are constant, softmax is flat, dS = P (dP - sum P dP) sums to zero against identical K rows
    k_same = k[:, :, :1].expand(B, H, Lk, D).contiguous()
    dq0, _, _ = bwd.attention_backward(q, k_same, v, do)
    assert dq0.abs().max().item() < 5e-3 * dq.abs().max().item() + 1e-4
    if Lq * Lk <= 4096 * 16384:
        monkeypatch.setenv("MSAM2_MATERIALISED_BWD", "1")
        rq, rk, rv = bwd.attention_backward(q, k, v, do)
        monkeypatch.delenv("MSAM2_MATERIALISED_BWD")
        rel = lambda a, b: ((a - b).norm() / b.norm()).item()
        assert rel(dq, rq) < 8e-3 and rel(dk, rk) < 8e-3 and rel(dv, rv) < 8e-3, (rel(dq, rq), rel(dk, rk), rel(dv, rv))
