"""Pins oracle/cc_oracle.c: hand-written known answers + partition/area equivalence with scipy's 8-connected
labelling + the reference's label rule (1 + smallest 2x2-block top-left index of the component)."""
import numpy as np
import pytest
import torch
from scipy import ndimage

from oracle import cc as cc_oracle


def test_known_answer_small():
    m = torch.tensor([[1, 0, 0, 1],
                      [0, 1, 0, 0],
                      [0, 0, 0, 1],
                      [1, 0, 0, 1]], dtype=torch.uint8)[None, None]
    lab, cnt = cc_oracle.connected_components(m)
    # component A = {(0,0),(1,1)} root block (0,0) -> label 1; B = {(0,3)} block (0,2) -> label 3;
    # C = {(2,3),(3,3)} block (2,2) -> label 11; D = {(3,0)} block (2,0) -> label 9
    exp_lab = torch.tensor([[1, 0, 0, 3], [0, 1, 0, 0], [0, 0, 0, 11], [9, 0, 0, 11]], dtype=torch.int32)
    exp_cnt = torch.tensor([[2, 0, 0, 1], [0, 2, 0, 0], [0, 0, 0, 2], [1, 0, 0, 2]], dtype=torch.int32)
    assert torch.equal(lab[0, 0], exp_lab)
    assert torch.equal(cnt[0, 0], exp_cnt)


def test_empty_and_full():
    z = torch.zeros(2, 1, 8, 6, dtype=torch.uint8)
    lab, cnt = cc_oracle.connected_components(z)
    assert lab.abs().sum() == 0 and cnt.abs().sum() == 0
    o = torch.ones(1, 1, 8, 6, dtype=torch.uint8)
    lab, cnt = cc_oracle.connected_components(o)
    assert (lab == 1).all() and (cnt == 48).all()


def test_odd_size_rejected():
    with pytest.raises(RuntimeError):
        cc_oracle.connected_components(torch.zeros(1, 1, 5, 4, dtype=torch.uint8))


@pytest.mark.parametrize("shape,p", [((3, 1, 64, 64), 0.5), ((2, 1, 256, 256), 0.42), ((1, 1, 30, 70), 0.6),
                                      ((2, 1, 128, 128), 0.9), ((2, 1, 128, 128), 0.08)])
def test_against_scipy(shape, p):
    g = torch.Generator().manual_seed(int(p * 1000) + shape[2])
    m = (torch.rand(shape, generator=g) < p).to(torch.uint8)
    lab, cnt = cc_oracle.connected_components(m)
    for n in range(shape[0]):
        ref, k = ndimage.label(m[n, 0].numpy(), structure=np.ones((3, 3)))
        mine = lab[n, 0].numpy()
        assert ((mine > 0) == (ref > 0)).all()
        W = shape[3]
        for c in range(1, k + 1):
            sel = ref == c
            vals = np.unique(mine[sel])
            assert len(vals) == 1                      # one label per scipy component
            assert (mine == vals[0]).sum() == sel.sum()  # and no other pixels carry it
            assert (cnt[n, 0].numpy()[sel] == sel.sum()).all()
            ys, xs = np.nonzero(sel)
            root = ((ys // 2) * 2 * W + (xs // 2) * 2).min()
            assert vals[0] == root + 1
